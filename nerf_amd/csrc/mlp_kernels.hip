// Fused NeRF MLP kernels for MI355X (gfx950): ProposalNetwork (63->256x4->1) and MipNeRF
// (63->256x4, skip 319->256x3, sigma head, 256 bottleneck, 283->128->3) evaluated per sample without
// ever writing an activation to HBM.
//
// Design:
//   * one workgroup = NW wavefronts; each wavefront owns 32 samples (the N dimension of a 32x32 MFMA
//     tile) and computes ALL output features of every layer for them:  D[feature][sample] = W . X.
//   * the products are computed "transposed" (A operand = weights, B operand = activations) so that
//     the C/D register layout of layer l (lane = sample, registers = features) IS the B-operand layout
//     of layer l+1 once the weight K-order is permuted at pack time: activations stay in VGPRs for
//     the whole network, there is no LDS/shuffle traffic between layers.
//   * weights are pre-packed in MFMA-fragment order (pack_kernels.hip) and streamed
//     L2 -> LDS with global_load_lds (16 B/lane, lane-linear = conflict-free) through a 4 x 16 KiB ring
//     shared by all wavefronts of the workgroup, one raw s_barrier per 16 KiB chunk, counted vmcnt so
//     that two chunks stay in flight across every barrier.
//   * positional encoding is computed in-register straight into B-operand layout (lane half 0
//     evaluates the sin terms, half 1 the cos terms -- same instruction stream).
//   * workgroups are persistent: grid = #CUs, each loops over sample tiles; the weight stream
//     wraps around without a drain.
//
// Reference semantics: addtional.py:88-96 (proposal), mip_model.py:41-60 (fine).
#include "device_common.h"
#include "mlp_layout.h"

extern __shared__ __attribute__((aligned(16))) char smem[];

namespace {

// ------------------------------------------------------------------------------------------------
// precision policies
// ------------------------------------------------------------------------------------------------
struct PBF16 {
    using BReg = bf16x8;                       // one 16-feature K group of the B operand (4 VGPRs)
    static constexpr int PREC = NERF_AMD_BF16;
    static constexpr int NW = MLP_NW_BF16;     // wavefronts per workgroup
    static constexpr int FRAG_BYTES = 1024;    // one A fragment: 32 rows x 16 k, bf16
    static constexpr int FPC = MLP_CHUNK_BYTES / FRAG_BYTES;
    static DEVINL f32x16 mma(uint32_t frag_addr, const BReg& b, f32x16 acc) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(smem + frag_addr);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    static DEVINL void set(BReg& r, int e, float v) { r[e] = (__bf16)v; }
};

struct PF32 {
    using BReg = f32x8;                        // 8 VGPRs per 16-feature K group
    static constexpr int PREC = NERF_AMD_F32;
    static constexpr int NW = MLP_NW_F32;
    static constexpr int FRAG_BYTES = 2048;    // [2 halves][64 lanes][4 floats]
    static constexpr int FPC = MLP_CHUNK_BYTES / FRAG_BYTES;
    static DEVINL f32x16 mma(uint32_t frag_addr, const BReg& b, f32x16 acc) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(smem + frag_addr);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(smem + frag_addr + 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b[e], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b[4 + e], acc, 0, 0, 0);
        return acc;
    }
    static DEVINL void set(BReg& r, int e, float v) { r[e] = v; }
};

// ------------------------------------------------------------------------------------------------
// weight stream: L2 -> LDS ring, consumed in lock step by all wavefronts of the workgroup
// ------------------------------------------------------------------------------------------------
template <class P>
struct WeightStream {
    static constexpr int LPW = (MLP_CHUNK_BYTES / 1024) / P::NW;     // 1 KiB glds pieces per wave per chunk
    const char* src;        // packed stream + this lane's offset inside a chunk
    uint32_t n_chunks;      // chunks in one pass over the network
    uint32_t load_idx;      // next chunk of the stream to fetch (wraps)
    uint32_t load_slot;     // ring slot it goes to
    uint32_t cur;           // LDS byte offset of the chunk being consumed (+ lane*16)
    uint32_t cur_slot;
    uint32_t wave_lds;      // wave-uniform LDS offset of this wave's pieces inside a slot

    DEVINL void issue() {
        const char* g = src + (size_t)load_idx * MLP_CHUNK_BYTES;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(load_slot * MLP_CHUNK_BYTES + wave_lds);
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(g + i * 1024),
                (__attribute__((address_space(3))) void*)(smem + dst + i * 1024), 16, 0, 0);
        load_idx = (load_idx + 1 == n_chunks) ? 0u : load_idx + 1;
        load_slot = (load_slot + 1) & (MLP_NSLOT - 1);
    }
    DEVINL void init(const void* packed, uint32_t nchunks) {
        const int lane = lane_id();
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        src = reinterpret_cast<const char*>(packed) + (size_t)wave * LPW * 1024 + lane * 16;
        wave_lds = wave * LPW * 1024;
        n_chunks = nchunks;
        load_idx = 0; load_slot = 0;
        cur_slot = MLP_NSLOT - 1;                  // first boundary() advances to slot 0
        cur = 0;
#pragma unroll
        for (int i = 0; i < MLP_NSLOT - 1; ++i) issue();
    }
    // Called by every wavefront right before it reads the first fragment of the next chunk.
    DEVINL void boundary() {
        // my pieces of the next chunk have landed (two younger chunks may stay in flight) ...
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
        // ... and so have everybody else's; also: everybody is done reading the previous chunk
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();                                   // refill the slot that was just retired
        cur_slot = (cur_slot + 1) & (MLP_NSLOT - 1);
        cur = cur_slot * MLP_CHUNK_BYTES + lane_id() * 16;
    }
    DEVINL void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// ------------------------------------------------------------------------------------------------
// One dense layer for this wavefront's 32 samples:  out[fb] = act(W[fb] . in + bias[fb]).
//   NKG    K groups (16 input features each) consumed;  in(kg) returns the B registers of group kg
//   NFB    32-row output feature blocks;  out(fb, acc) receives the 16 accumulators of block fb
//   START  fragment index of the layer inside the stream modulo FPC (chunk phase)
// ------------------------------------------------------------------------------------------------
template <class P, int NKG, int NFB, int START, class InF, class OutF>
DEVINL void dense(WeightStream<P>& ws, uint32_t bias_lds, InF&& in, OutF&& out) {
    const int h = lane_id() >> 5;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + bias_lds + (32 * fb + 8 * q + 4 * h) * 4);
            acc[4 * q + 0] = b4[0]; acc[4 * q + 1] = b4[1]; acc[4 * q + 2] = b4[2]; acc[4 * q + 3] = b4[3];
        }
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) {
            constexpr int dummy = 0; (void)dummy;
            const int f = START + fb * NKG + kg;                   // compile-time after unrolling
            if (f % P::FPC == 0) ws.boundary();
            acc = P::mma(ws.cur + (f % P::FPC) * P::FRAG_BYTES, in(kg), acc);
        }
        out(fb, acc);
    }
}

// accumulators of feature block fb -> B registers of K groups 2fb, 2fb+1 of the next layer
template <class P, bool RELU>
DEVINL void to_breg(const f32x16& acc, typename P::BReg& lo, typename P::BReg& hi) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a = acc[e], b = acc[8 + e];
        if (RELU) { a = fmaxf(a, 0.0f); b = fmaxf(b, 0.0f); }
        P::set(lo, e, a);
        P::set(hi, e, b);
    }
}

// ------------------------------------------------------------------------------------------------
// positional encoding straight into B-operand layout (slot map: mlp_layout.h pe_slot_feature()).
// ------------------------------------------------------------------------------------------------
template <class P, int L, int NKG>
DEVINL void encode(float x, float y, float z, int h, typename P::BReg (&B)[NKG]) {
#pragma unroll
    for (int q = 0; q < 8 * NKG; ++q) {
        float v;
        if (q < 3 * L) {
            const int c = q % 3;
            const float comp = (c == 0) ? x : ((c == 1) ? y : z);
            v = sin_quadrant(comp * (float)(1 << (q / 3)), h);
        } else if (q == 3 * L) {
            v = h ? z : x;
        } else if (q == 3 * L + 1) {
            v = h ? 0.0f : y;
        } else {
            v = 0.0f;
        }
        P::set(B[q >> 3], q & 7, v);
    }
}

// ------------------------------------------------------------------------------------------------
// sample fetch: position (and raw direction) of sample m
// ------------------------------------------------------------------------------------------------
struct Sample { float x, y, z, dx, dy, dz; };

DEVINL Sample fetch_sample(const nerf_amd_samples& s, int64_t m, bool want_dir) {
    Sample r;
    if (s.mode == 0) {
        const float* p = s.pts + m * s.pts_stride;
        r.x = p[0]; r.y = p[1]; r.z = p[2];
        if (want_dir) { r.dx = p[3]; r.dy = p[4]; r.dz = p[5]; } else { r.dx = r.dy = r.dz = 0.0f; }
        return r;
    }
    const int64_t n = m / s.S;
    const int si = (int)(m - n * s.S);
    float ox, oy, oz;
    if (s.mode == 1) {
        const float* ry = s.rays + n * 6;
        ox = ry[0]; oy = ry[1]; oz = ry[2]; r.dx = ry[3]; r.dy = ry[4]; r.dz = ry[5];
    } else {
        const int row = (int)(n / s.W), col = (int)(n - (int64_t)row * s.W);
        const float cx = (((float)col - (float)s.W * 0.5f) + 0.5f) / s.fx;       // procedures.py:44-47
        const float cy = (((float)s.H * 0.5f - (float)row) + 0.5f) / s.fy;
        r.dx = (cx * s.pose[0] + cy * s.pose[1]) + (-1.0f) * s.pose[2];          // procedures.py:51
        r.dy = (cx * s.pose[4] + cy * s.pose[5]) + (-1.0f) * s.pose[6];
        r.dz = (cx * s.pose[8] + cy * s.pose[9]) + (-1.0f) * s.pose[10];
        ox = s.pose[3]; oy = s.pose[7]; oz = s.pose[11];
    }
    float zv;
    if (s.z) zv = s.z[n * s.z_stride + si];
    else     zv = s.z_base[si] + s.u[n * s.S + si] * s.z_jitter;                 // procedures.py:65
    r.x = ox + zv * r.dx; r.y = oy + zv * r.dy; r.z = oz + zv * r.dz;           // procedures.py:66
    return r;
}

DEVINL void load_biases(const void* packed, size_t stream_bytes, int n_bias) {
    const float* b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + stream_bytes);
    float* dst = reinterpret_cast<float*>(smem + MLP_RING_BYTES);
    for (int i = threadIdx.x; i < n_bias; i += blockDim.x) dst[i] = b[i];
    __syncthreads();
}

// ================================================================================================
// ProposalNetwork
// ================================================================================================
template <class P>
__global__ __launch_bounds__(P::NW * 64) void proposal_kernel(const void* __restrict__ packed, nerf_amd_samples s,
                                                              float* __restrict__ density) {
    using L = PropLayout;
    using BReg = typename P::BReg;
    constexpr int FPC = P::FPC;
    load_biases(packed, L::stream_bytes(P::PREC), L::N_BIAS);
    WeightStream<P> ws;
    ws.init(packed, L::N_FRAGS / FPC);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = threadIdx.x >> 6;
    constexpr int TS = P::NW * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    const uint32_t bias0 = MLP_RING_BYTES;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t m = tile * TS + wave * 32 + j;
        const Sample sm = fetch_sample(s, m < s.M ? m : s.M - 1, false);
        BReg enc[4];
        encode<P, 10, 4>(sm.x, sm.y, sm.z, h, enc);
        BReg a[16], b[16];
        dense<P, 4, 8, L::START[0] % FPC>(ws, bias0 + L::BIAS_OFF[0] * 4,
            [&](int kg) -> const BReg& { return enc[kg]; },
            [&](int fb, const f32x16& acc) { to_breg<P, true>(acc, a[2 * fb], a[2 * fb + 1]); });
#pragma unroll 1
        for (int l = 1; l <= 3; ++l) {
            dense<P, 16, 8, L::START[1] % FPC>(ws, bias0 + (L::BIAS_OFF[1] + (l - 1) * 256) * 4,
                [&](int kg) -> const BReg& { return a[kg]; },
                [&](int fb, const f32x16& acc) { to_breg<P, true>(acc, b[2 * fb], b[2 * fb + 1]); });
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = b[k];
        }
        float dens = 0.0f;
        dense<P, 16, 1, L::START[4] % FPC>(ws, bias0 + L::BIAS_OFF[4] * 4,
            [&](int kg) -> const BReg& { return a[kg]; },
            [&](int, const f32x16& acc) { dens = acc[0]; });
        if (h == 0 && m < s.M) density[m] = dens;
    }
    ws.drain();
}

// ================================================================================================
// MipNeRF
// ================================================================================================
template <class P>
__global__ __launch_bounds__(P::NW * 64) void mip_kernel(const void* __restrict__ packed, nerf_amd_samples s,
                                                         float* __restrict__ rgbo) {
    using L = MipLayout;
    using BReg = typename P::BReg;
    constexpr int FPC = P::FPC;
    load_biases(packed, L::stream_bytes(P::PREC), L::N_BIAS);
    WeightStream<P> ws;
    ws.init(packed, L::N_FRAGS / FPC);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = threadIdx.x >> 6;
    constexpr int TS = P::NW * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    const uint32_t bias0 = MLP_RING_BYTES;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t m = tile * TS + wave * 32 + j;
        const Sample sm = fetch_sample(s, m < s.M ? m : s.M - 1, true);
        BReg enc[4];
        encode<P, 10, 4>(sm.x, sm.y, sm.z, h, enc);
        BReg a[16], b[16];
        // lin_block1.0 : 63 -> 256
        dense<P, 4, 8, L::START[0] % FPC>(ws, bias0 + L::BIAS_OFF[0] * 4,
            [&](int kg) -> const BReg& { return enc[kg]; },
            [&](int fb, const f32x16& acc) { to_breg<P, true>(acc, a[2 * fb], a[2 * fb + 1]); });
        // lin_block1.{2,4,6} : 256 -> 256
#pragma unroll 1
        for (int l = 1; l <= 3; ++l) {
            dense<P, 16, 8, L::START[1] % FPC>(ws, bias0 + (L::BIAS_OFF[1] + (l - 1) * 256) * 4,
                [&](int kg) -> const BReg& { return a[kg]; },
                [&](int fb, const f32x16& acc) { to_breg<P, true>(acc, b[2 * fb], b[2 * fb + 1]); });
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = b[k];
        }
        // lin_block2.0 : cat(enc 63, h 256) -> 256
        dense<P, 20, 8, L::START[4] % FPC>(ws, bias0 + L::BIAS_OFF[4] * 4,
            [&](int kg) -> const BReg& { if (kg < 4) return enc[kg < 4 ? kg : 0]; return a[kg >= 4 ? kg - 4 : 0]; },
            [&](int fb, const f32x16& acc) { to_breg<P, true>(acc, b[2 * fb], b[2 * fb + 1]); });
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = b[k];
        // lin_block2.{2,4}
#pragma unroll 1
        for (int l = 5; l <= 6; ++l) {
            dense<P, 16, 8, L::START[5] % FPC>(ws, bias0 + (L::BIAS_OFF[5] + (l - 5) * 256) * 4,
                [&](int kg) -> const BReg& { return a[kg]; },
                [&](int fb, const f32x16& acc) { to_breg<P, true>(acc, b[2 * fb], b[2 * fb + 1]); });
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = b[k];
        }
        // bottle_neck.0 (rows 0..255, no activation) + opacity_head.0 (row 256)
        float sigma = 0.0f;
        dense<P, 16, 9, L::START[7] % FPC>(ws, bias0 + L::BIAS_OFF[7] * 4,
            [&](int kg) -> const BReg& { return a[kg]; },
            [&](int fb, const f32x16& acc) {
                if (fb < 8) to_breg<P, false>(acc, b[2 * (fb < 8 ? fb : 0)], b[2 * (fb < 8 ? fb : 0) + 1]);
                else sigma = acc[0];
            });
        // direction: d/|d| and PE4 (mip_model.py:43-46,51)
        BReg denc[2];
        {
            const float nrm = norm3(sm.dx, sm.dy, sm.dz);
            encode<P, 4, 2>(sm.dx / nrm, sm.dy / nrm, sm.dz / nrm, h, denc);
        }
        // rgb_layer.0 : cat(bottleneck 256, dir 27) -> 128, ReLU
        BReg c[8];
        dense<P, 18, 4, L::START[8] % FPC>(ws, bias0 + L::BIAS_OFF[8] * 4,
            [&](int kg) -> const BReg& { if (kg < 16) return b[kg < 16 ? kg : 0]; return denc[kg >= 16 ? kg - 16 : 0]; },
            [&](int fb, const f32x16& acc) { to_breg<P, true>(acc, c[2 * fb], c[2 * fb + 1]); });
        // rgb_layer.2 : 128 -> 3, sigmoid
        float r = 0.0f, g = 0.0f, bl = 0.0f;
        dense<P, 8, 1, L::START[9] % FPC>(ws, bias0 + L::BIAS_OFF[9] * 4,
            [&](int kg) -> const BReg& { return c[kg]; },
            [&](int, const f32x16& acc) { r = acc[0]; g = acc[1]; bl = acc[2]; });
        if (h == 0 && m < s.M) {
            f32x4 o;
            o[0] = 1.0f / (1.0f + expf(-r));
            o[1] = 1.0f / (1.0f + expf(-g));
            o[2] = 1.0f / (1.0f + expf(-bl));
            o[3] = sigma;
            *reinterpret_cast<f32x4*>(rgbo + m * 4) = o;
        }
    }
    ws.drain();
}

int grid_for(int64_t n_tiles) {
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) n_cu = 256;
        else n_cu = p.multiProcessorCount;
    }
    return (int)(n_tiles < n_cu ? n_tiles : n_cu);
}

template <class P, class Lay, class K>
int launch(K kernel, const void* packed, const nerf_amd_samples& s, float* out, hipStream_t st) {
    constexpr int TS = P::NW * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    if (n_tiles == 0) return 0;
    const size_t lds = MLP_RING_BYTES + Lay::N_BIAS * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kernel, dim3(grid_for(n_tiles)), dim3(P::NW * 64), lds, st, packed, s, out);
    return (int)hipGetLastError();
}

}  // namespace

// host-visible launchers (capi.hip)
int mlp_launch_proposal(const void* packed, int precision, const nerf_amd_samples& s, float* density, hipStream_t st) {
    if (precision == NERF_AMD_BF16) return launch<PBF16, PropLayout>(proposal_kernel<PBF16>, packed, s, density, st);
    return launch<PF32, PropLayout>(proposal_kernel<PF32>, packed, s, density, st);
}
int mlp_launch_mip(const void* packed, int precision, const nerf_amd_samples& s, float* rgbo, hipStream_t st) {
    if (precision == NERF_AMD_BF16) return launch<PBF16, MipLayout>(mip_kernel<PBF16>, packed, s, rgbo, st);
    return launch<PF32, MipLayout>(mip_kernel<PF32>, packed, s, rgbo, st);
}
