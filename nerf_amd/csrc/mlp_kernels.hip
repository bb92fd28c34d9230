// Fused NeRF MLP kernels for MI355X (gfx950): ProposalNetwork (63->256x4->1) and MipNeRF
// (63->256x4, skip 319->256x3, sigma head, 256 bottleneck, 283->128->3) evaluated per sample without
// ever writing an activation to HBM.
//
// Design:
//   * one workgroup = NW wavefronts; each wavefront owns NT x 32 samples (NT column tiles, each the N dimension of a
//     32x32 MFMA tile) and computes ALL output features of every layer for them:  D[feature][sample] = W . X.
//     Shipped policies: bf16 4 waves x 64 samples (PBF16W), fp32 4 waves x 32 samples (PF32).
//   * the products are computed "transposed" (A operand = weights, B operand = activations) so that
//     the C/D register layout of layer l (lane = sample, registers = features) IS the B-operand layout
//     of layer l+1 once the weight K-order is permuted at pack time: activations stay in VGPRs for
//     the whole network, there is no LDS/shuffle traffic between layers.
//   * weights are pre-packed in MFMA-fragment order (pack_kernels.hip) and streamed
//     L2 -> LDS with global_load_lds (16 B/lane, lane-linear = conflict-free) through an 8 x 8 KiB ring
//     shared by all wavefronts of the workgroup, one raw s_barrier per two chunks, counted vmcnt so
//     that three chunks stay in flight across every barrier.
//   * epilogues (fp32 -> bf16, ReLU) are deferred: sliced into the first K steps of the next block pair / layer.
//   * positional encoding is computed in-register straight into B-operand layout (lane half 0
//     evaluates the sin terms, half 1 the cos terms -- same instruction stream).
//   * workgroups are persistent: grid = #CUs, each loops over sample tiles; the weight stream
//     wraps around without a drain.
//
// Reference semantics: addtional.py:88-96 (proposal), mip_model.py:41-60 (fine).
#include "mlp_core.h"

namespace {

// ------------------------------------------------------------------------------------------------
// positional encoding straight into B-operand layout (slot map: mlp_layout.h pe_slot_feature()).
// ------------------------------------------------------------------------------------------------
template <class P, int L, int NKG>
DEVINL void encode(float x, float y, float z, int h, typename P::BReg (&B)[NKG]) {
    if constexpr (P::FAST_PE) {
        // bf16 mode: octave 0 through the exact reduction, octaves 1..L-1 by angle doubling
        // (sin 2a = 2 s c, cos 2a = 1 - 2 s^2).  The error doubles per octave: <= 2^9 * 1e-7 = 5e-5 << bf16 ulp (4e-3).
        float sv[3], cv[3];                                   // one range reduction per coordinate (sincos_quadrant == the two sin_quadrant calls, bit for bit)
        sincos_quadrant(x, sv[0], cv[0]); sincos_quadrant(y, sv[1], cv[1]); sincos_quadrant(z, sv[2], cv[2]);
#pragma unroll
        for (int f = 0; f < L; ++f) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int q = 3 * f + c;
                P::set(B[q >> 3], q & 7, h ? cv[c] : sv[c]);
                if (f + 1 < L) {
                    const float s2 = 2.0f * sv[c];
                    const float ns = s2 * cv[c];
                    const float nc = __builtin_fmaf(-s2, sv[c], 1.0f);
                    sv[c] = ns; cv[c] = nc;
                }
            }
        }
#pragma unroll
        for (int q = 3 * L; q < 8 * NKG; ++q) {
            const float v = (q == 3 * L) ? (h ? z : x) : ((q == 3 * L + 1) ? (h ? 0.0f : y) : 0.0f);
            P::set(B[q >> 3], q & 7, v);
        }
        return;
    }
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
// Paired tile prologue (round 4).  A wave's lanes are (sample j, half h); half 0 of a B register group holds the sin-type slots of
// sample j, half 1 the cos-type slots -- so in the plain prologue BOTH halves fetch the same sample, draw the same Philox block and run
// the same range reductions and angle-doubling chains, and each keeps half of what it computed.  With two column tiles per wave the
// halves can split the SAMPLES instead: lane (j, h) does the whole scalar prologue for sample j of column tile h only -- both the
// sin-type group S and the cos-type group C of that one sample come out of the one sincos chain it runs anyway -- and one
// v_permlane32_swap per packed dword (upper half of S <-> lower half of C) leaves S = tile 0's registers and C = tile 1's registers in
// B-operand layout.  Same values bit for bit; half the prologue's fetch / Philox / reduction / doubling instructions per tile.
// (A/B against the plain prologue: profiles/r04_paired_prologue_ab.log.)
// ------------------------------------------------------------------------------------------------
// a = [a.lo | b.lo], b = [a.hi | b.hi]   (lo / hi = lanes 0..31 / 32..63)
DEVINL void half_swap(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
template <int NKG>
DEVINL void half_swap_groups(bf16x8 (&S)[NKG], bf16x8 (&C)[NKG]) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
#pragma unroll
    for (int k = 0; k < NKG; ++k) {
        u32x4_ a = __builtin_bit_cast(u32x4_, S[k]), b = __builtin_bit_cast(u32x4_, C[k]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { uint32_t x = a[i], y = b[i]; half_swap(x, y); a[i] = x; b[i] = y; }
        S[k] = __builtin_bit_cast(bf16x8, a); C[k] = __builtin_bit_cast(bf16x8, b);
    }
}
// encode<> for BOTH lane halves of one sample: S = what half 0 holds (sin slots, x, y), C = what half 1 holds (cos slots, z, 0)
template <class P, int L, int NKG>
DEVINL void encode_pair(float x, float y, float z, typename P::BReg (&S)[NKG], typename P::BReg (&C)[NKG]) {
    static_assert(P::FAST_PE, "the paired prologue is the bf16 form (octaves by angle doubling)");
    float sv[3], cv[3];
    sincos_quadrant(x, sv[0], cv[0]); sincos_quadrant(y, sv[1], cv[1]); sincos_quadrant(z, sv[2], cv[2]);
#pragma unroll
    for (int f = 0; f < L; ++f) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int q = 3 * f + c;
            P::set(S[q >> 3], q & 7, sv[c]);
            P::set(C[q >> 3], q & 7, cv[c]);
            if (f + 1 < L) {
                const float s2 = 2.0f * sv[c];
                const float ns = s2 * cv[c];
                const float nc = __builtin_fmaf(-s2, sv[c], 1.0f);
                sv[c] = ns; cv[c] = nc;
            }
        }
    }
#pragma unroll
    for (int q = 3 * L; q < 8 * NKG; ++q) {
        P::set(S[q >> 3], q & 7, (q == 3 * L) ? x : ((q == 3 * L + 1) ? y : 0.0f));
        P::set(C[q >> 3], q & 7, (q == 3 * L) ? z : 0.0f);
    }
}

// Integrated positional encoding (mip_methods.py:36-58) into the same B-operand slots: slot (l, c) = sin|cos(2^l mu_c) * exp(-0.5 * 4^l var_c),
// raw slots = mu (the cat_origin prefix).  bf16 mode: octaves by angle doubling, attenuation by att_{l+1} = att_l^4.
template <class P, int L, int NKG>
DEVINL void encode_ipe(const float (&mu)[3], const float (&var)[3], int h, typename P::BReg (&B)[NKG]) {
    if constexpr (P::FAST_PE) {
        float sv[3], cv[3], at[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { sincos_quadrant(mu[c], sv[c], cv[c]); at[c] = expf(-0.5f * var[c]); }
#pragma unroll
        for (int f = 0; f < L; ++f) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int q = 3 * f + c;
                P::set(B[q >> 3], q & 7, (h ? cv[c] : sv[c]) * at[c]);
                if (f + 1 < L) {
                    const float s2 = 2.0f * sv[c];
                    const float ns = s2 * cv[c];
                    const float nc = __builtin_fmaf(-s2, sv[c], 1.0f);
                    sv[c] = ns; cv[c] = nc;
                    const float a2 = at[c] * at[c];
                    at[c] = a2 * a2;
                }
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 3 * L; ++q) {
            const int c = q % 3, f = q / 3;
            const float v = sin_quadrant(mu[c] * (float)(1 << f), h) * expf(-0.5f * (var[c] * (float)(1u << (2 * f))));
            P::set(B[q >> 3], q & 7, v);
        }
    }
#pragma unroll
    for (int q = 3 * L; q < 8 * NKG; ++q) {
        const float v = (q == 3 * L) ? (h ? mu[2] : mu[0]) : ((q == 3 * L + 1) ? (h ? 0.0f : mu[1]) : 0.0f);
        P::set(B[q >> 3], q & 7, v);
    }
}

// ------------------------------------------------------------------------------------------------
// sample fetch: position (and raw direction) of sample m
// ------------------------------------------------------------------------------------------------
struct Sample { float x, y, z, dx, dy, dz; };

// Mip-NeRF 360 scene contraction (Barron et al. 2022, eq. 10) of a position; the direction is left alone
DEVINL void contract_position(Sample& r) {
    const float n = norm3(r.x, r.y, r.z);
    if (n > 1.0f) {
        const float k = (2.0f - 1.0f / n) / n;
        r.x *= k; r.y *= k; r.z *= k;
    }
}

// sample index -> (ray, sample of the ray).  Sample counts below 2^31 (every practical launch: the bench's fine pass has 8.2e7) take a
// 32-bit division (~20 VALU instructions instead of ~80 for the emulated 64-bit one; the branch is wave-uniform).
DEVINL void split_sample_index(const nerf_amd_samples& s, int64_t m, int64_t& n, int& si) {
    if (s.M <= 0x7fffffffll) {
        const uint32_t q = (uint32_t)m / (uint32_t)s.S;
        n = q; si = (int)((uint32_t)m - q * (uint32_t)s.S);
    } else {
        n = m / s.S; si = (int)(m - n * s.S);
    }
}
DEVINL Sample fetch_sample(const nerf_amd_samples& s, int64_t m, bool want_dir) {
    Sample r;
    if (s.mode == 0) {
        const float* p = s.pts + m * s.pts_stride;
        r.x = p[0]; r.y = p[1]; r.z = p[2];
        if (want_dir) { r.dx = p[3]; r.dy = p[4]; r.dz = p[5]; } else { r.dx = r.dy = r.dz = 0.0f; }
        if (s.contract) contract_position(r);
        return r;
    }
    int64_t n; int si;
    split_sample_index(s, m, n, si);
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
    else {                                                                       // procedures.py:65; the uniform from memory, or drawn here
        const float uu = s.u ? s.u[n * s.S + si] : philox_u_strat(s.rng_seed, n + s.rng_ray_offset, si);
        zv = s.z_base[si] + uu * s.z_jitter;
    }
    r.x = ox + zv * r.dx; r.y = oy + zv * r.dy; r.z = oz + zv * r.dz;           // procedures.py:66
    if (s.contract) contract_position(r);
    return r;
}

// IPE sample (nerf_amd_samples.ipe; mode 1 with explicit depths): frustum si of ray n spans z[si] .. z[si+1]
struct IpeSample { float mu[3], var[3], dx, dy, dz; };
DEVINL IpeSample fetch_sample_ipe(const nerf_amd_samples& s, int64_t m) {
    IpeSample r;
    int64_t n; int si;
    split_sample_index(s, m, n, si);
    const float* ry = s.rays + n * 6;
    const float* zz = s.z + n * s.z_stride + si;
    const float rr = s.ipe_radius;
    const ConeMoments c = cone_moments(zz[0], zz[1], (float)((double)rr * (double)rr));
    const float dn = s.ipe_dir_norm[0];
#pragma unroll
    for (int k = 0; k < 3; ++k) cone_mean_cov(c, ry[k], ry[3 + k], dn, r.mu[k], r.var[k]);
    r.dx = ry[3]; r.dy = ry[4]; r.dz = ry[5];
    if (s.contract) {                                      // (mean only; the covariance is left in metric space)
        Sample p{r.mu[0], r.mu[1], r.mu[2], 0.0f, 0.0f, 0.0f};
        contract_position(p);
        r.mu[0] = p.x; r.mu[1] = p.y; r.mu[2] = p.z;
    }
    return r;
}

constexpr uint32_t LDS_BIAS = MLP_RING_BYTES;
constexpr uint32_t LDS_STASH = MLP_RING_BYTES + 9216;                 // bias table: <= 2240 floats
template <class P> constexpr uint32_t lds_dir() { return LDS_STASH + P::NW * P::NT * 4 * P::BREG_LDS; }
template <class P> constexpr uint32_t lds_tile() { return lds_dir<P>() + P::NW * P::NT * 1024; }            // fused compositing: (tile samples) x 2 float4, segment totals, tickets
template <class P> constexpr uint32_t lds_total() { return lds_tile<P>() + P::NW * P::NT * 32 * 32 + P::NW * P::NT * 8 + 64; }   // records, segment totals, tickets

DEVINL void load_biases(const void* packed, size_t stream_bytes, int n_bias, uint32_t lds_off = MLP_RING_BYTES) {
    const float* b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + stream_bytes);
    float* dst = reinterpret_cast<float*>(smem + lds_off);
    for (int i = threadIdx.x; i < n_bias; i += blockDim.x) dst[i] = b[i];
    __syncthreads();
}

// Training forward: the TRAIN instantiations also write every hidden layer's post-ReLU activations to
// HBM, in FRAGMENT order -- block (layer, subtile, K group) = the 64 lanes' B register group, 1 KiB (bf16) / 2 KiB (fp32), i.e.
// one fully coalesced 16-byte store per lane and register group.  The backward reads them back as row-major matrices through
// frag_to_rows_kernel.  Layer slots are `layer_stride` bytes apart, every slot holds 16 K groups per subtile.
struct ActDump {
    char* base;
    unsigned long long layer_stride;
    char* mask_base;                       // ReLU bit masks (proposal / MipNeRF): slot l, subtile s -> mask_base + l*mask_layer_stride + s*1024
    unsigned long long mask_layer_stride;  //   one 16-byte record per lane = 128 bits: K group kg -> dword kg>>2, nibble pair 4*(kg&3)
};
template <class P>
DEVINL void dump_breg(const ActDump& d, int layer, int64_t subtile, int kg, int lane, const typename P::BReg& r) {
#ifndef ABL_NODUMPST   // cost probe (scripts/gpu_ref_fwd_probe.sh): the training forwards WITHOUT their activation stores -- wrong results, right cost
    P::store_global(d.base + (size_t)layer * d.layer_stride + ((size_t)subtile * 16 + kg) * (size_t)P::BREG_LDS, lane, r);
#endif
}

// The backward's ReLU adjoint only needs [y > 0]: next to the activations (the weight gradients' operand) the training forwards
// leave ONE BIT per activation -- 32 B per sample and layer instead of the 512 B the dgrad chain would otherwise re-read.
// Bit layout of a B register group: element e -> bit (e >> 1) + 16 (e & 1) (i.e. the low / high halves of its four packed dwords),
// shifted by 4 (kg & 3) inside dword kg >> 2 of the lane's 16-byte record.  The bits are OR-ed into a per-wave LDS record
// (ds_or_b32: no registers held across the layer) and written out once per layer and subtile.
template <class P> constexpr uint32_t lds_maskacc() { return lds_total<P>(); }
template <class P> constexpr uint32_t lds_total_train() { return lds_total<P>() + P::NW * 2 * P::NT * 1024; }
// fp8 dumps: per wave two (layer parity) x NT records of scale exponents behind the mask records
template <class P> constexpr uint32_t lds_scaleacc() { return lds_total_train<P>(); }
template <class P> constexpr uint32_t lds_total_train_f8() { return lds_total_train<P>() + P::NW * 2 * P::NT * 1024; }
// one hidden-layer K group of the training dump: bf16 fragment block, or (F8) scaled e4m3 + its exponent into the wave's LDS record
template <class P, bool F8>
DEVINL void dump_hidden(const ActDump& d, uint32_t sacc_wave, int layer, int64_t subtile, int t, int kg, int lane, const typename P::BReg& r) {
    if constexpr (F8) {
        const uint32_t E = f8_group_exponent<false>(r);
        f8_store_group(d.base + (size_t)layer * d.layer_stride + (size_t)subtile * F8_SUB_BYTES, kg, lane, f8_encode_group(r, E), E,
                       sacc_wave + ((layer & 1) * P::NT + t) * 1024);
    } else {
        dump_breg<P>(d, layer, subtile, kg, lane, r);
    }
}
DEVINL uint32_t breg_bits(const bf16x8& v) {
    // per dword: min(element, 1) on both 16-bit halves (a post-ReLU bf16 is +0 or a positive pattern), shifted into place.  The min is
    // written as asm: from __builtin_elementwise_min on a 2 x u16 vector hipcc builds two 16-bit compares, two selects and a v_perm per
    // dword (3 100 VALU instructions per tile in the fine training forward, against 1 088 MFMAs) instead of one v_pk_min_u16.
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4 d = __builtin_bit_cast(u32x4, v);
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = d[i];
        uint32_t on;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(on) : "v"(x), "v"(0x00010001u));
        m |= on << i;
    }
    return m;
}
DEVINL uint32_t breg_bits(const f32x8& v) {
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) m |= (v[e] > 0.0f ? 1u : 0u) << ((e >> 1) + 16 * (e & 1));
    return m;
}
template <class P>
DEVINL void mask_or(uint32_t acc_wave, int layer, int t, int kg, int lane, const typename P::BReg& v) {
    unsigned* w = reinterpret_cast<unsigned*>(smem + acc_wave + ((layer & 1) * P::NT + t) * 1024 + lane * 16 + (kg >> 2) * 4);
    __hip_atomic_fetch_or(w, breg_bits(v) << (4 * (kg & 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the layer's records are complete (its last feature-block pair has been converted): write them out and clear the LDS copy
template <class P, bool F8 = false>
DEVINL void mask_flush(const ActDump& d, uint32_t acc_wave, int layer, int64_t sub0, int lane, uint32_t sacc_wave = 0) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < P::NT; ++t) {
        f32x4* rec = reinterpret_cast<f32x4*>(smem + acc_wave + ((layer & 1) * P::NT + t) * 1024 + lane * 16);
        const f32x4 v = *rec;
        *reinterpret_cast<f32x4*>(d.mask_base + (size_t)layer * d.mask_layer_stride + (size_t)(sub0 + t) * 1024 + lane * 16) = v;
        *rec = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (F8)                                    // the slot's scale exponents (every byte is rewritten by the next layer of this parity)
            f8_flush_scales(d.base + (size_t)layer * d.layer_stride + (size_t)(sub0 + t) * F8_SUB_BYTES, lane, sacc_wave + ((layer & 1) * P::NT + t) * 1024);
    }
    asm volatile("" ::: "memory");
}
template <class P>
DEVINL void mask_acc_init(uint32_t acc_wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2 * P::NT; ++i) *reinterpret_cast<f32x4*>(smem + acc_wave + i * 1024 + lane * 16) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// ================================================================================================
// ProposalNetwork
// ================================================================================================
template <class P, bool TRAIN, bool F8 = false>
__global__ __launch_bounds__(P::NW * 64) void proposal_kernel(const void* __restrict__ packed, nerf_amd_samples s,
                                                              float* __restrict__ density, ActDump dump) {
    using L = PropLayout;
    using BReg = typename P::BReg;
    constexpr int FPC = P::FPC;
    load_biases(packed, L::stream_bytes(P::PREC), L::N_BIAS);
    WeightStream<P, MLP_NSLOT> ws;
    ws.init(packed, L::N_FRAGS / FPC);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform: lets address arithmetic on it run on the scalar unit)
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    const uint32_t bias0 = MLP_RING_BYTES;
    const uint32_t macc = lds_maskacc<P>() + wave * 2 * NT * 1024;      // TRAIN: this wave's ReLU bit-mask records
    const uint32_t sacc = lds_scaleacc<P>() + wave * 2 * NT * 1024;     // F8: this wave's scale-exponent records
    if constexpr (TRAIN) mask_acc_init<P>(macc, lane);

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int64_t m[NT];
        BReg enc[NT][4];
        if constexpr (NT == 2 && P::FAST_PE) {
            // paired prologue: lane (j, h) fetches and encodes sample j of column tile h for BOTH halves, then the halves trade
            m[0] = tile * TS + (wave * NT) * 32 + j;
            m[1] = m[0] + 32;
            const int64_t mo = h ? m[1] : m[0];
            const Sample sm = fetch_sample(s, mo < s.M ? mo : s.M - 1, false);
            encode_pair<P, 10, 4>(sm.x, sm.y, sm.z, enc[0], enc[1]);
            half_swap_groups<4>(enc[0], enc[1]);
            if constexpr (TRAIN) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) dump_breg<P>(dump, 4, tile * (TS / 32) + wave * NT + t, k, lane, enc[t][k]);
            }
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            m[t] = tile * TS + (wave * NT + t) * 32 + j;
            const Sample sm = fetch_sample(s, m[t] < s.M ? m[t] : s.M - 1, false);
            encode<P, 10, 4>(sm.x, sm.y, sm.z, h, enc[t]);
            if constexpr (TRAIN) {                           // the first-layer weight gradient's operand: slot 4, K groups 0..3
#pragma unroll
                for (int k = 0; k < 4; ++k) dump_breg<P>(dump, 4, tile * (TS / 32) + wave * NT + t, k, lane, enc[t][k]);
            }
        }
        }
        BReg a[NT][16], b[NT][16];
        // (layer indices for the training dump: `lay` = the layer whose block pairs are being computed, `lay_pend` = the layer
        // that owns the deferred pair -- the deferred slices run while the NEXT layer is computed)
        int lay = 0, lay_pend = 0;
        const int64_t sub0 = tile * (TS / 32) + wave * NT;
        auto put = [&](BReg (&buf)[NT][16], int layer, int fb, int t, const f32x16& acc, int half) {
            buf[t][2 * fb + half] = to_breg_half<P, true>(acc, half);
            if constexpr (TRAIN) {
                dump_hidden<P, F8>(dump, sacc, layer, sub0 + t, t, 2 * fb + half, lane, buf[t][2 * fb + half]);
                mask_or<P>(macc, layer, t, 2 * fb + half, lane, buf[t][2 * fb + half]);
            }
        };
        auto OA = [&](int fb, int t, const f32x16& acc, int half) { put(a, lay, fb, t, acc, half); };
        auto OB = [&](int fb, int t, const f32x16& acc, int half) { put(b, lay, fb, t, acc, half); };
        auto OA_pend = [&](int fb, int t, const f32x16& acc, int half) { put(a, lay_pend, fb, t, acc, half); };
        auto OB_pend = [&](int fb, int t, const f32x16& acc, int half) { put(b, lay_pend, fb, t, acc, half); };
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        // d = the last feature-block pair of a layer (features 192..255 = K groups 12..15 of the next one), converted
        // into `a` during the first K steps of whatever runs next
        Deferred<P, 6, 2> d = dense<P, 4, 8, L::START[0]>(ws, bias0 + L::BIAS_OFF[0] * 4,
            [&](int kg, int t) -> BReg { return enc[t][kg]; }, OA, NoPrev{});
        // layers.2/4/6 ping-pong between the two register buffers (a -> b -> a -> b): no copies; the two a->b layers
        // share one code instance through the loop (same chunk parity, asserted)
        static_assert(L::START[1] % (2 * FPC) == L::START[3] % (2 * FPC), "chunk parity");
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            lay_pend = lay; lay = 1 + 2 * r;
            d = dense<P, 16, 8, L::START[1]>(ws, bias0 + (L::BIAS_OFF[1] + r * 512) * 4, IN_A, OB, prev_of(d, OA_pend));
            if constexpr (TRAIN) mask_flush<P, F8>(dump, macc, lay_pend, sub0, lane, sacc);        // (its last pair was converted during this layer)
            if (r == 0) {
                lay_pend = lay; lay = 2;
                d = dense<P, 16, 8, L::START[2]>(ws, bias0 + L::BIAS_OFF[2] * 4, IN_B, OA, prev_of(d, OB_pend));
                if constexpr (TRAIN) mask_flush<P, F8>(dump, macc, lay_pend, sub0, lane, sacc);
            }
        }
        lay_pend = lay;
        float dens[NT];
        auto OH = [&](int, int t, const f32x16& acc, int half) { if (half == 0) dens[t] = acc[0]; };
        dense<P, 16, 1, L::START[4]>(ws, bias0 + L::BIAS_OFF[4] * 4, IN_B, OH, prev_of(d, OB_pend)).flush(OH);
        if constexpr (TRAIN) mask_flush<P, F8>(dump, macc, lay_pend, sub0, lane, sacc);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (h == 0 && m[t] < s.M) density[m[t]] = dens[t];
    }
    ws.drain();
}

// ================================================================================================
// ProposalNetwork, hidden width 128 (PropLayout128; the reference's class default, addtional.py:61, and --prop_net_width 128)
// ================================================================================================
// Inference only (the training path of a narrow network runs the 256-wide kernels on zero-padded tensors: exact, nerf_amd/_packed.py).
// Same machinery as proposal_kernel at half the K groups and feature blocks per hidden layer; NT column tiles per wave (bf16: 4).
constexpr uint32_t lds_total_narrow() { return MLP_RING_BYTES + 9216; }
template <class P>
__global__ __launch_bounds__(P::NW * 64) void proposal128_kernel(const void* __restrict__ packed, nerf_amd_samples s, float* __restrict__ density) {
    using L = PropLayout128;
    using BReg = typename P::BReg;
    constexpr int HK = L::HK, HFB = L::HFB;
    load_biases(packed, L::stream_bytes(P::PREC), L::N_BIAS);
    WeightStream<P, MLP_NSLOT> ws;
    ws.init(packed, L::N_FRAGS / P::FPC);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    const uint32_t bias0 = MLP_RING_BYTES;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int64_t m[NT];
        BReg a[NT][HK], b[NT][HK];
        auto OA = [&](int fb, int t, const f32x16& acc, int half) { a[t][2 * fb + half] = to_breg_half<P, true>(acc, half); };
        auto OB = [&](int fb, int t, const f32x16& acc, int half) { b[t][2 * fb + half] = to_breg_half<P, true>(acc, half); };
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
        Deferred<P, HFB - 2, 2> d;
        {
            BReg enc[NT][4];
            constexpr int PAIRED = (NT >= 2 && P::FAST_PE) ? 2 : 0;       // column tiles 0 and 1 through the paired prologue
#pragma unroll
            for (int t = 0; t < NT; ++t) m[t] = tile * TS + (wave * NT + t) * 32 + j;
            if constexpr (PAIRED) {
                const int64_t mo = h ? m[1] : m[0];
                const Sample sm = fetch_sample(s, mo < s.M ? mo : s.M - 1, false);
                encode_pair<P, 10, 4>(sm.x, sm.y, sm.z, enc[0], enc[1]);
                half_swap_groups<4>(enc[0], enc[1]);
            }
#pragma unroll
            for (int t = PAIRED; t < NT; ++t) {
                const Sample sm = fetch_sample(s, m[t] < s.M ? m[t] : s.M - 1, false);
                encode<P, 10, 4>(sm.x, sm.y, sm.z, h, enc[t]);
            }
            d = dense<P, 4, HFB, L::START[0]>(ws, bias0 + L::BIAS_OFF[0] * 4, [&](int kg, int t) -> BReg { return enc[t][kg]; }, OA, NoPrev{});
        }
        // layers.2/4/6 ping-pong between the two register buffers (a -> b -> a -> b); d = a layer's last feature-block pair, converted
        // during the first K steps of the next layer (which reads those features -- K groups HK-4 .. HK-1 -- in its second half only)
        d = dense<P, HK, HFB, L::START[1]>(ws, bias0 + L::BIAS_OFF[1] * 4, IN_A, OB, prev_of(d, OA));
        d = dense<P, HK, HFB, L::START[2]>(ws, bias0 + L::BIAS_OFF[2] * 4, IN_B, OA, prev_of(d, OB));
        d = dense<P, HK, HFB, L::START[3]>(ws, bias0 + L::BIAS_OFF[3] * 4, IN_A, OB, prev_of(d, OA));
        float dens[NT];
        auto OH = [&](int, int t, const f32x16& acc, int half) { if (half == 0) dens[t] = acc[0]; };
        dense<P, HK, 1, L::START[4]>(ws, bias0 + L::BIAS_OFF[4] * 4, IN_B, OH, prev_of(d, OB)).flush(OH);
        skip_frags<P, L::USED_FRAGS, L::N_FRAGS - L::USED_FRAGS>(ws);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (h == 0 && m[t] < s.M) density[m[t]] = dens[t];
    }
    ws.drain();
}

// ================================================================================================
// MipNeRF
// ================================================================================================
// Optional fused compositing epilogue (nerf_base.py:91-113): when `rgb` is set, the (rgb, sigma) of a tile are parked in
// LDS and the LAST wavefront of each ray to arrive (LDS atomic ticket) composites the ray: sigma -> alpha, wave
// prefix-product transmittance, weighted sums -- the (N,S,4) network output never goes to HBM.
struct FusedComposite {
    float* rgb;        // (N,3) or nullptr = epilogue off
    float* depth;      // (N) or nullptr
    float* weights;    // (N,S) or nullptr
    int white_bkg;
    float near, far;
};

// FUSED: the compositing epilogue below (nerf_amd_mip_forward_composite) -- a template flag since round 4, so that the shipped two-launch
// instantiations carry neither its code nor its branches and can use the paired tile prologue.
template <class P, bool TRAIN, bool IPE = false, bool F8 = false, bool FUSED = false>
__global__ __launch_bounds__(P::NW * 64) void mip_kernel(const void* __restrict__ packed, nerf_amd_samples s,
                                                         float* __restrict__ rgbo, FusedComposite fc, ActDump dump) {
    using L = MipLayout;
    using BReg = typename P::BReg;
    constexpr int FPC = P::FPC;
#ifdef MLP_CLOCKPROBE
    const uint64_t probe_c0 = __builtin_readcyclecounter(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (FUSED) { if (threadIdx.x < 16) reinterpret_cast<unsigned*>(smem + lds_tile<P>() + P::NW * P::NT * 32 * 32 + P::NW * P::NT * 8)[threadIdx.x] = 0u; }  // ray tickets
    load_biases(packed, L::stream_bytes(P::PREC), L::N_BIAS);
    WeightStream<P, MLP_NSLOT> ws;
    ws.init(packed, L::N_FRAGS / FPC);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform: lets address arithmetic on it run on the scalar unit)
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    const uint32_t bias0 = LDS_BIAS;
    const uint32_t macc = lds_maskacc<P>() + wave * 2 * NT * 1024;      // TRAIN: this wave's ReLU bit-mask records
    const uint32_t sacc = lds_scaleacc<P>() + wave * 2 * NT * 1024;     // F8: this wave's scale-exponent records
    if constexpr (TRAIN) mask_acc_init<P>(macc, lane);
    // per 32-sample column tile ("subtile" sub = wave*NT + t) LDS slots
    const uint32_t enc_lds0 = LDS_STASH + wave * NT * 4 * P::BREG_LDS + lane * 16;
    const uint32_t dir_lds0 = lds_dir<P>() + wave * NT * 1024 + lane * 16;
    auto enc_lds = [&](int t) -> uint32_t { return enc_lds0 + t * 4 * P::BREG_LDS; };
    auto dir_lds = [&](int t) -> uint32_t { return dir_lds0 + t * 1024; };

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int64_t m[NT];
        BReg a[NT][16], b[NT][16];
        int lay = 0, lay_pend = 0;                                  // training dump: see proposal_kernel
        const int64_t sub0 = tile * (TS / 32) + wave * NT;
        auto put = [&](BReg (&buf)[NT][16], int layer, int fb, int t, const f32x16& acc, int half) {
            buf[t][2 * fb + half] = to_breg_half<P, true>(acc, half);
            if constexpr (TRAIN) {
                dump_hidden<P, F8>(dump, sacc, layer, sub0 + t, t, 2 * fb + half, lane, buf[t][2 * fb + half]);
                mask_or<P>(macc, layer, t, 2 * fb + half, lane, buf[t][2 * fb + half]);
            }
        };
        auto OA = [&](int fb, int t, const f32x16& acc, int half) { put(a, lay, fb, t, acc, half); };
        auto OB = [&](int fb, int t, const f32x16& acc, int half) { put(b, lay, fb, t, acc, half); };
        auto OA_pend = [&](int fb, int t, const f32x16& acc, int half) { put(a, lay_pend, fb, t, acc, half); };
        auto OB_pend = [&](int fb, int t, const f32x16& acc, int half) { put(b, lay_pend, fb, t, acc, half); };
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        // d = the last feature-block pair of a layer (features 192..255 = K groups 12..15 of the next one), converted
        // into `a` during the first K steps of whatever runs next
        Deferred<P, 6, 2> d;
        {
            BReg enc[NT][4];
            if constexpr (NT == 2 && P::FAST_PE && !IPE && !FUSED) {
                // paired prologue (see encode_pair): lane (j, h) fetches and encodes sample j of column tile h, the halves then trade
                m[0] = tile * TS + (wave * NT) * 32 + j;
                m[1] = m[0] + 32;
                const int64_t mo = h ? m[1] : m[0];
                const Sample sm = fetch_sample(s, mo < s.M ? mo : s.M - 1, true);
                encode_pair<P, 10, 4>(sm.x, sm.y, sm.z, enc[0], enc[1]);
                half_swap_groups<4>(enc[0], enc[1]);
                // the raw direction of BOTH tiles' samples for both halves: swap(d, copy of d) -> [d.lo | d.lo], [d.hi | d.hi]
                uint32_t d0[3] = {__builtin_bit_cast(uint32_t, sm.dx), __builtin_bit_cast(uint32_t, sm.dy), __builtin_bit_cast(uint32_t, sm.dz)};
                uint32_t d1[3] = {d0[0], d0[1], d0[2]};
#pragma unroll
                for (int c = 0; c < 3; ++c) half_swap(d0[c], d1[c]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) P::stash(enc_lds(t) + k * P::BREG_LDS, enc[t][k]);
                    if constexpr (TRAIN) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) dump_breg<P>(dump, 8, sub0 + t, k, lane, enc[t][k]);
                    }
                    const uint32_t* dd = t ? d1 : d0;
                    const f32x4 dv = {__builtin_bit_cast(float, dd[0]), __builtin_bit_cast(float, dd[1]), __builtin_bit_cast(float, dd[2]), 0.0f};
                    *reinterpret_cast<f32x4*>(smem + dir_lds(t)) = dv;
                }
            } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                m[t] = tile * TS + (wave * NT + t) * 32 + j;
                Sample sm;
                if constexpr (IPE) {                        // integrated PE of the frustum (row 12) instead of the point PE (row 3)
                    const IpeSample is = fetch_sample_ipe(s, m[t] < s.M ? m[t] : s.M - 1);
                    encode_ipe<P, 10, 4>(is.mu, is.var, h, enc[t]);
                    sm.dx = is.dx; sm.dy = is.dy; sm.dz = is.dz;
                } else {
                    sm = fetch_sample(s, m[t] < s.M ? m[t] : s.M - 1, true);
                    encode<P, 10, 4>(sm.x, sm.y, sm.z, h, enc[t]);
                }
                // the encoding is needed again by the skip layer and the direction by the colour head: park them in
                // this wavefront's private LDS stash instead of holding 20+ VGPRs through six layers
#pragma unroll
                for (int k = 0; k < 4; ++k) P::stash(enc_lds(t) + k * P::BREG_LDS, enc[t][k]);
                if constexpr (TRAIN) {                       // operand of the first- and skip-layer weight gradients: slot 8, K groups 0..3
#pragma unroll
                    for (int k = 0; k < 4; ++k) dump_breg<P>(dump, 8, sub0 + t, k, lane, enc[t][k]);
                }
                f32x4 dv = {sm.dx, sm.dy, sm.dz, 0.0f};
                if constexpr (FUSED) {
                    // fused compositing needs z|d| and the distance to the next sample at the END of the tile; fetch them now,
                    // while the tile's other global loads are in flight (a load at the tile end would drain the weight DMA queue)
                    const int64_t mm_ = m[t] < s.M ? m[t] : s.M - 1;
                    const int64_t n_ = mm_ / s.S;
                    const int si_ = (int)(mm_ - n_ * s.S);
                    const float nrm_ = norm3(sm.dx, sm.dy, sm.dz);
                    const float* zz_ = s.z + n_ * s.z_stride;
                    const float zn0 = zz_[si_] * nrm_;
                    const float dl = (si_ + 1 < s.S) ? (zz_[si_ + 1] * nrm_ - zn0) : 1e10f;
                    dv[3] = h ? dl : zn0;
                }
                *reinterpret_cast<f32x4*>(smem + dir_lds(t)) = dv;
            }
            }
            // lin_block1.0 : 63 -> 256
            d = dense<P, 4, 8, L::START[0]>(ws, bias0 + L::BIAS_OFF[0] * 4,
                [&](int kg, int t) -> BReg { return enc[t][kg]; }, OA, NoPrev{});
        }
        // lin_block1.{2,4,6}, lin_block2.{0,2,4}: the layers ping-pong between the two register buffers
        //   l1.2 a->b, l1.4 b->a, l1.6 a->b, l2.0 (skip: cat(enc 63, h 256)) b->a, l2.2 a->b, l2.4 b->a
        // so nothing is ever copied; three loop rounds of (a->b layer, then the skip layer or a b->a layer) keep it at three code
        // instances.  A loop may only reuse an instance for layers whose first chunk has the same parity (the early/late
        // barrier assignment is compiled in) and whose biases are evenly spaced -- asserted.
        static_assert(L::START[1] % (2 * FPC) == L::START[3] % (2 * FPC) && L::START[1] % (2 * FPC) == L::START[5] % (2 * FPC) &&
                      L::START[2] % (2 * FPC) == L::START[6] % (2 * FPC) && L::BIAS_OFF[6] == 6 * 256, "uniform layer loop");
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
#pragma unroll 1
        for (int r = 0; r < 3; ++r) {
            lay_pend = lay; lay = 1 + 2 * r;
            d = dense<P, 16, 8, L::START[1]>(ws, bias0 + (1 + 2 * r) * 256 * 4, IN_A, OB, prev_of(d, OA_pend));
            if constexpr (TRAIN) mask_flush<P, F8>(dump, macc, lay_pend, sub0, lane, sacc);        // (its last pair was converted during this layer)
            lay_pend = lay; lay = 2 + 2 * r;
            if (r == 1) {
                d = dense<P, 20, 8, L::START[4]>(ws, bias0 + L::BIAS_OFF[4] * 4,
                    [&](int kg, int t) -> BReg { if (kg < 4) return P::unstash(enc_lds(t) + kg * P::BREG_LDS); return b[t][kg >= 4 ? kg - 4 : 0]; },
                    OA, prev_of(d, OB_pend));
            } else {
                d = dense<P, 16, 8, L::START[2]>(ws, bias0 + (2 + 2 * r) * 256 * 4, IN_B, OA, prev_of(d, OB_pend));
            }
            if constexpr (TRAIN) mask_flush<P, F8>(dump, macc, lay_pend, sub0, lane, sacc);
        }
        lay_pend = lay;
        // opacity_head.0 : 256 -> 1 (raw sigma)
        float sigma[NT];
        auto OSIG = [&](int, int t, const f32x16& acc, int half) { if (half == 0) sigma[t] = acc[0]; };
        const auto dsig = dense<P, 16, 1, L::START[7]>(ws, bias0 + L::BIAS_OFF[7] * 4, IN_A, OSIG, prev_of(d, OA_pend));
        if constexpr (TRAIN) mask_flush<P, F8>(dump, macc, 6, sub0, lane, sacc);
        // direction: d/|d| and PE4 (mip_model.py:43-46,51)
        BReg denc[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 dv = *reinterpret_cast<const f32x4*>(smem + dir_lds(t));
            const float nrm = norm3(dv[0], dv[1], dv[2]);
            encode<P, 4, 2>(dv[0] / nrm, dv[1] / nrm, dv[2] / nrm, h, denc[t]);
            if constexpr (TRAIN) {                           // operand of rgb_layer.0's direction columns: slot 8, K groups 4..5
                dump_breg<P>(dump, 8, sub0 + t, 4, lane, denc[t][0]);
                dump_breg<P>(dump, 8, sub0 + t, 5, lane, denc[t][1]);
            }
        }
        // rgb_layer.0 with bottle_neck.0 folded in (mlp_layout.h): cat(g 256, dir 27) -> 128, ReLU
        BReg c[NT][8];
        auto OC = [&](int fb, int t, const f32x16& acc, int half) {
            c[t][2 * fb + half] = to_breg_half<P, true>(acc, half);
            if constexpr (TRAIN) {                                                                       // slot 7: rgb_layer.0 output
                dump_hidden<P, F8>(dump, sacc, 7, sub0 + t, t, 2 * fb + half, lane, c[t][2 * fb + half]);
                mask_or<P>(macc, 7, t, 2 * fb + half, lane, c[t][2 * fb + half]);
            }
        };
        const auto dc = dense<P, 18, 4, L::START[8]>(ws, bias0 + L::BIAS_OFF[8] * 4,
            [&](int kg, int t) -> BReg { if (kg < 16) return a[t][kg < 16 ? kg : 0]; return denc[t][kg >= 16 ? kg - 16 : 0]; },
            OC, prev_of(dsig, OSIG));
        // rgb_layer.2 : 128 -> 3, sigmoid
        float r[NT], g[NT], bl[NT];
        auto ORGB = [&](int, int t, const f32x16& acc, int half) { if (half == 0) { r[t] = acc[0]; g[t] = acc[1]; bl[t] = acc[2]; } };
        dense<P, 8, 1, L::START[9]>(ws, bias0 + L::BIAS_OFF[9] * 4,
            [&](int kg, int t) -> BReg { return c[t][kg]; }, ORGB, prev_of(dc, OC)).flush(ORGB);
        if constexpr (TRAIN) mask_flush<P, F8>(dump, macc, 7, sub0, lane, sacc);
        f32x4 o[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            o[t][0] = 1.0f / (1.0f + expf(-r[t]));
            o[t][1] = 1.0f / (1.0f + expf(-g[t]));
            o[t][2] = 1.0f / (1.0f + expf(-bl[t]));
            o[t][3] = sigma[t];
            if constexpr (!FUSED) {
                if (h == 0 && m[t] < s.M) *reinterpret_cast<f32x4*>(rgbo + m[t] * 4) = o[t];
            }
        }
        if constexpr (FUSED) {
            // ---- fused compositing epilogue (nerf_base.py:91-113; S in {32, 64, 128}: a ray = S / 32 consecutive 32-sample SEGMENTS of the tile) ----
            // Every wave does the expensive part for ITS OWN segments: sigma -> alpha and the exclusive prefix product of the transmittance
            // INSIDE each segment (fp64, one 32-segmented DPP scan for both column tiles), parks per sample (colour, alpha, z|d|, that
            // product) and per segment its total in LDS, and takes an LDS ticket per segment.  The wave whose ticket completes a ray only
            // COMBINES: segment products chained exactly as the 64-lane scan of the two-launch composite kernel chains them (its last DPP
            // step multiplies lanes 32..63 by lane 31's prefix -- here: by the first segment's total), weights, weighted sums.  The
            // (N,S,4) network output never reaches HBM, nobody waits for anybody, and the serial work of the last arriver is a fraction of
            // compositing the whole ray alone (round 2's form).  Bit-compatible with nerf_amd_mip_forward + nerf_amd_composite.
            const int S = s.S;
            const int spr = S >> 5;                                        // segments per ray
            char* tb = smem + lds_tile<P>();
            f32x4* t_ca = reinterpret_cast<f32x4*>(tb);                                  // [TS] r, g, b, alpha
            u32x4* t_pz = reinterpret_cast<u32x4*>(tb + TS * 16);                        // [TS] exclusive product inside the segment (double bits), z|d|, -
            double* t_seg = reinterpret_cast<double*>(tb + TS * 32);                     // [TS / 32] product over the segment
            unsigned* tickets = reinterpret_cast<unsigned*>(tb + TS * 32 + (TS / 32) * 8);
            {
                // segment-lane layout: lane L <-> column tile L >> 5 (NT = 2; lanes 32..63 idle for NT = 1), sample L & 31
                const int tt = (NT > 1) ? h : 0;
                const bool live = (NT > 1) || h == 0;
                const float k0 = (*reinterpret_cast<const f32x4*>(smem + dir_lds(0)))[3];              // half 0: z|d|, half 1: delta (tile start)
                const float k1 = (NT > 1) ? (*reinterpret_cast<const f32x4*>(smem + dir_lds(NT - 1)))[3] : 0.0f;
                float cr, cg, cb, sg, zn, dl;
                if (NT > 1) {
                    // (the lane moves run on ALL lanes, then the select: a shuffle under a lane condition reads inactive source lanes as 0)
                    const float s0 = __shfl(o[NT - 1][0], j, 64), s1 = __shfl(o[NT - 1][1], j, 64), s2 = __shfl(o[NT - 1][2], j, 64),
                                s3 = __shfl(o[NT - 1][3], j, 64), sz = __shfl(k1, j, 64), d0 = __shfl(k0, j + 32, 64);
                    cr = h ? s0 : o[0][0]; cg = h ? s1 : o[0][1]; cb = h ? s2 : o[0][2]; sg = h ? s3 : o[0][3];
                    zn = h ? sz : k0;
                    dl = h ? k1 : d0;
                } else {
                    cr = o[0][0]; cg = o[0][1]; cb = o[0][2]; sg = o[0][3]; zn = k0; dl = __shfl(k0, j + 32, 64);
                }
                const float mm = expf(-fmaxf(sg, 0.0f) * dl);
                const float al = live ? 1.0f - mm : 0.0f;
                const double p = live ? (double)(mm + 1e-10f) : 1.0;
                const double incl = seg32_incl_scan_mul(p);
                double excl = wave_shift_up1(incl, 1.0);
                if (lane == 32) excl = 1.0;                                  // (the second segment starts over)
                const int sub = wave * NT + tt;
                if (live) {
                    const int ts = sub * 32 + j;
                    t_ca[ts] = f32x4{cr, cg, cb, al};
                    const u32x2 eb = __builtin_bit_cast(u32x2, excl);
                    t_pz[ts] = u32x4{eb[0], eb[1], __builtin_bit_cast(uint32_t, zn), 0u};
                }
                if (j == 31 && live) t_seg[sub] = incl;                      // lanes 31 and 63 hold their segment's total
            }
            // LDS executes one wavefront's DS instructions in order, so the tickets are ordered behind the record writes without a fence
            // (a workgroup-scope fence would also wait vmcnt(0) and drain the weight DMA queue); the asm statements only stop the compiler
            // from moving LDS accesses across the tickets.
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int sub = wave * NT + t;
                const int ray_in_tile = sub / spr;
                const int64_t n = (tile * TS) / S + ray_in_tile;
                unsigned old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(&tickets[ray_in_tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                asm volatile("" ::: "memory");
                if (((old + 1) % (unsigned)spr) != 0 || n * S >= s.M) continue;      // not the last segment of this ray (tickets only ever grow)
                const int seg0 = ray_in_tile * spr;
                float ar = 0.0f, ag = 0.0f, abl = 0.0f, aw = 0.0f, ad = 0.0f;
                float* wout = fc.weights ? fc.weights + n * S : nullptr;
                double carry = 1.0;
                for (int base = 0; base < S; base += 64) {                   // 64 samples = two segments per pass, like the composite kernel's chunks
                    const int q = base + lane;
                    const int sg0 = seg0 + (base >> 5);
                    const bool two = base + 32 < S;
                    const double T0 = t_seg[sg0], T1 = two ? t_seg[sg0 + (two ? 1 : 0)] : 1.0;
                    float w = 0.0f;
                    f32x4 ca = {0.0f, 0.0f, 0.0f, 0.0f};
                    float zn = 0.0f;
                    if (q < S) {
                        const int ts = seg0 * 32 + q;
                        ca = t_ca[ts];
                        const u32x4 pz = t_pz[ts];
                        const u32x2 eb = {pz[0], pz[1]};
                        double excl = __builtin_bit_cast(double, eb);
                        if (h) excl = excl * T0;                             // what row_bcast:31 does in the 64-lane scan
                        zn = __builtin_bit_cast(float, (uint32_t)pz[2]);
                        w = ca[3] * (float)(carry * excl);
                    }
                    carry *= (T1 * T0);
                    if (q < S) {
                        ar += w * ca[0]; ag += w * ca[1]; abl += w * ca[2]; aw += w; ad += w * zn;
                        if (wout) wout[q] = w;
                    }
                }
                ar = wave_sum(ar); ag = wave_sum(ag); abl = wave_sum(abl); aw = wave_sum(aw); ad = wave_sum(ad);
                if (lane == 0) {
                    if (fc.white_bkg) { const float bg = 1.0f - aw; ar += bg; ag += bg; abl += bg; }
                    fc.rgb[n * 3] = ar; fc.rgb[n * 3 + 1] = ag; fc.rgb[n * 3 + 2] = abl;
                    if (fc.depth) fc.depth[n] = (ad - fc.near) / (fc.far - fc.near);
                }
            }
        }
    }
    ws.drain();
#ifdef MLP_CLOCKPROBE
    // diagnostic build: shader cycles vs 100 MHz real-time ticks of workgroup 0 overwrite the first output record
    if (blockIdx.x == 0 && threadIdx.x == 0 && rgbo != nullptr) {
        reinterpret_cast<uint64_t*>(rgbo)[0] = __builtin_readcyclecounter() - probe_c0;
        reinterpret_cast<uint64_t*>(rgbo)[1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
    }
#endif
}


// ================================================================================================
// MipNeRF, hidden width 128 (MipLayout128; --nerf_net_width 128).  Inference only, point PE; same structure as mip_kernel with half the
// K groups / feature blocks in the 128-wide layers (lin_block2.4 widens to 256 for the heads, mip_model.py:28).
// ================================================================================================
template <class P>
__global__ __launch_bounds__(P::NW * 64) void mip128_kernel(const void* __restrict__ packed, nerf_amd_samples s, float* __restrict__ rgbo) {
    using L = MipLayout128;
    using BReg = typename P::BReg;
    load_biases(packed, L::stream_bytes(P::PREC), L::N_BIAS);
    WeightStream<P, MLP_NSLOT> ws;
    ws.init(packed, L::N_FRAGS / P::FPC);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    const uint32_t bias0 = LDS_BIAS;
    const uint32_t enc_lds0 = LDS_STASH + wave * NT * 4 * P::BREG_LDS + lane * 16;
    const uint32_t dir_lds0 = lds_dir<P>() + wave * NT * 1024 + lane * 16;
    auto enc_lds = [&](int t) -> uint32_t { return enc_lds0 + t * 4 * P::BREG_LDS; };
    auto dir_lds = [&](int t) -> uint32_t { return dir_lds0 + t * 1024; };
    auto bias = [&](int l) -> uint32_t { return bias0 + L::BIAS_OFF[l] * 4; };
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int64_t m[NT];
        BReg a[NT][8], b[NT][8], g[NT][16];
        auto OA = [&](int fb, int t, const f32x16& acc, int half) { a[t][2 * fb + half] = to_breg_half<P, true>(acc, half); };
        auto OB = [&](int fb, int t, const f32x16& acc, int half) { b[t][2 * fb + half] = to_breg_half<P, true>(acc, half); };
        auto OG = [&](int fb, int t, const f32x16& acc, int half) { g[t][2 * fb + half] = to_breg_half<P, true>(acc, half); };
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
        auto IN_G = [&](int kg, int t) -> BReg { return g[t][kg]; };
        Deferred<P, 2, 2> d;
        {
            BReg enc[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t) m[t] = tile * TS + (wave * NT + t) * 32 + j;
            if constexpr (NT == 2 && P::FAST_PE) {
                const int64_t mo = h ? m[1] : m[0];
                const Sample sm = fetch_sample(s, mo < s.M ? mo : s.M - 1, true);
                encode_pair<P, 10, 4>(sm.x, sm.y, sm.z, enc[0], enc[1]);
                half_swap_groups<4>(enc[0], enc[1]);
                uint32_t d0[3] = {__builtin_bit_cast(uint32_t, sm.dx), __builtin_bit_cast(uint32_t, sm.dy), __builtin_bit_cast(uint32_t, sm.dz)};
                uint32_t d1[3] = {d0[0], d0[1], d0[2]};
#pragma unroll
                for (int c = 0; c < 3; ++c) half_swap(d0[c], d1[c]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) P::stash(enc_lds(t) + k * P::BREG_LDS, enc[t][k]);
                    const uint32_t* dd = t ? d1 : d0;
                    const f32x4 dv = {__builtin_bit_cast(float, dd[0]), __builtin_bit_cast(float, dd[1]), __builtin_bit_cast(float, dd[2]), 0.0f};
                    *reinterpret_cast<f32x4*>(smem + dir_lds(t)) = dv;
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const Sample sm = fetch_sample(s, m[t] < s.M ? m[t] : s.M - 1, true);
                    encode<P, 10, 4>(sm.x, sm.y, sm.z, h, enc[t]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) P::stash(enc_lds(t) + k * P::BREG_LDS, enc[t][k]);
                    const f32x4 dv = {sm.dx, sm.dy, sm.dz, 0.0f};
                    *reinterpret_cast<f32x4*>(smem + dir_lds(t)) = dv;
                }
            }
            d = dense<P, 4, 4, L::START[0]>(ws, bias(0), [&](int kg, int t) -> BReg { return enc[t][kg]; }, OA, NoPrev{});     // lin_block1.0
        }
        // lin_block1.{2,4,6}: a -> b -> a -> b;  lin_block2.0 (skip: cat(enc 63, h 128)): b -> a;  lin_block2.2: a -> b;  lin_block2.4 (128 -> 256): b -> g.
        // A layer's last feature-block pair is converted during the first K steps of the next layer (which reads it in its second half only).
        // (the two a -> b layers share one code instance through the loop -- same chunk parity, asserted; short loops also keep hipcc's
        //  register allocation from losing track in one giant basic block)
        static_assert(L::START[1] % (2 * P::FPC) == L::START[3] % (2 * P::FPC) && L::BIAS_OFF[3] == L::BIAS_OFF[1] + 256, "shared a -> b instance");
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            d = dense<P, 8, 4, L::START[1]>(ws, bias0 + (L::BIAS_OFF[1] + r * 256) * 4, IN_A, OB, prev_of(d, OA));
            if (r == 0) d = dense<P, 8, 4, L::START[2]>(ws, bias(2), IN_B, OA, prev_of(d, OB));
        }
        d = dense<P, 12, 4, L::START[4]>(ws, bias(4),
            [&](int kg, int t) -> BReg { if (kg < 4) return P::unstash(enc_lds(t) + kg * P::BREG_LDS); return b[t][kg >= 4 ? kg - 4 : 0]; }, OA, prev_of(d, OB));
        d = dense<P, 8, 4, L::START[5]>(ws, bias(5), IN_A, OB, prev_of(d, OA));
        const auto dg = dense<P, 8, 8, L::START[6]>(ws, bias(6), IN_B, OG, prev_of(d, OB));
        // opacity_head.0 : 256 -> 1 (raw sigma)
        float sigma[NT];
        auto OSIG = [&](int, int t, const f32x16& acc, int half) { if (half == 0) sigma[t] = acc[0]; };
        const auto dsig = dense<P, 16, 1, L::START[7]>(ws, bias(7), IN_G, OSIG, prev_of(dg, OG));
        BReg denc[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 dv = *reinterpret_cast<const f32x4*>(smem + dir_lds(t));
            const float nrm = norm3(dv[0], dv[1], dv[2]);
            encode<P, 4, 2>(dv[0] / nrm, dv[1] / nrm, dv[2] / nrm, h, denc[t]);
        }
        // rgb_layer.0 with bottle_neck.0 folded in: cat(g 256, dir 27) -> 128, ReLU;  rgb_layer.2 : 128 -> 3, sigmoid
        BReg c[NT][8];
        auto OC = [&](int fb, int t, const f32x16& acc, int half) { c[t][2 * fb + half] = to_breg_half<P, true>(acc, half); };
        const auto dc = dense<P, 18, 4, L::START[8]>(ws, bias(8),
            [&](int kg, int t) -> BReg { if (kg < 16) return g[t][kg < 16 ? kg : 0]; return denc[t][kg >= 16 ? kg - 16 : 0]; }, OC, prev_of(dsig, OSIG));
        float r[NT], gg[NT], bl[NT];
        auto ORGB = [&](int, int t, const f32x16& acc, int half) { if (half == 0) { r[t] = acc[0]; gg[t] = acc[1]; bl[t] = acc[2]; } };
        dense<P, 8, 1, L::START[9]>(ws, bias(9), [&](int kg, int t) -> BReg { return c[t][kg]; }, ORGB, prev_of(dc, OC)).flush(ORGB);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 o;
            o[0] = 1.0f / (1.0f + expf(-r[t]));
            o[1] = 1.0f / (1.0f + expf(-gg[t]));
            o[2] = 1.0f / (1.0f + expf(-bl[t]));
            o[3] = sigma[t];
            if (h == 0 && m[t] < s.M) *reinterpret_cast<f32x4*>(rgbo + m[t] * 4) = o;
        }
    }
    ws.drain();
}

// ================================================================================================
// RefNeRF (ref_model.py:68-106, eval mode, use_srgb = False)
// ================================================================================================
constexpr uint32_t REF_LDS_BIAS = MLP_CHUNK_BYTES * MLP_NSLOT_REF;
constexpr uint32_t REF_LDS_STASH = REF_LDS_BIAS + RefLayout::N_BIAS * 4;          // 16-byte aligned (4288 floats)
template <class P> constexpr uint32_t ref_lds_total() { return REF_LDS_STASH + P::NW * P::NT * 11 * P::BREG_LDS; }

// Integrated directional encoding (ref_func.py:76-108) of the reflected direction, straight into B-operand slots:
// lane half 0 produces the real parts, half 1 the imaginary parts of the 19 (m,l) terms; slot 19 of half 0 = n.d
template <class P>
DEVINL void ide_encode(float x, float y, float z, float kappa_inv, float nv_dot, int h, const float* __restrict__ mat,
                       typename P::BReg (&out)[3]) {
    constexpr int TM[19] = {0, 1, 0, 1, 2, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5, 6, 7, 8};
    constexpr int TL[19] = {1, 1, 2, 2, 2, 4, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8, 8, 8, 8};
    float zp[9], re[9], im[9];
    zp[0] = 1.0f; re[0] = 1.0f; im[0] = 0.0f;
#pragma unroll
    for (int k = 1; k < 9; ++k) {
        zp[k] = zp[k - 1] * z;
        re[k] = re[k - 1] * x - im[k - 1] * y;
        im[k] = re[k - 1] * y + im[k - 1] * x;
    }
    const float att1 = expf(-1.0f * kappa_inv), att2 = expf(-3.0f * kappa_inv), att4 = expf(-10.0f * kappa_inv),
                att8 = expf(-36.0f * kappa_inv);                                   // sigma_l = l(l+1)/2
#pragma unroll
    for (int t = 0; t < 19; ++t) {
        const int m = TM[t], l = TL[t];
        float poly = 0.0f;
#pragma unroll
        for (int k = 0; k <= l - m; ++k) poly = __builtin_fmaf(mat[k * 19 + t], zp[k], poly);
        const float att = (l == 1) ? att1 : ((l == 2) ? att2 : ((l == 4) ? att4 : att8));
        const float v = ((h ? im[m] : re[m]) * poly) * att;
        P::set(out[t >> 3], t & 7, v);
    }
    P::set(out[2], 3, h ? 0.0f : nv_dot);
#pragma unroll
    for (int e = 4; e < 8; ++e) P::set(out[2], e, 0.0f);
}

// bn_noise (training forward, ref_model.py:84-85): (M, 128) row-major perturbation added to the bottle-neck vector, or nullptr
// TRAIN: additionally dumps, in fragment order, every hidden layer's activations (slots 0..7 spatial, 9..16
// directional), the directional network's input vector and the position encoding (slot 8: K groups 0..7 bottle-neck incl. the
// noise, 8..10 integrated directional encoding + n.d, 11..14 PE10 of the position), and per sample the 14 pre-activation head values
// the backward's element-wise stage needs (aux (M,16) fp32: normal 0..2, roughness 3, diffuse 4..6, density 7, tint 8..10, spec 11..13).
template <class P, bool TRAIN>
__global__ __launch_bounds__(P::NW * 64) void ref_kernel(const void* __restrict__ packed, nerf_amd_samples s,
                                                         float* __restrict__ rgbo, float* __restrict__ normal_out,
                                                         const float* __restrict__ bn_noise, ActDump dump, float* __restrict__ aux, int flags,
                                                         unsigned long long noise_seed, const unsigned long long* __restrict__ noise_seed_dev, float noise_std) {
    using L = RefLayout;
    // bn_noise == nullptr and noise_std > 0: the bottle-neck perturbation is drawn HERE (device_common.h philox_normal8), keyed by
    // (seed, sample index) -- `noise_seed_dev` (a device scalar, read at run time: a captured hipGraph replays fresh noise) or `noise_seed`
    if (noise_seed_dev != nullptr) noise_seed = noise_seed_dev[0];
    using BReg = typename P::BReg;
    constexpr int FPC = P::FPC;
    load_biases(packed, L::stream_bytes(P::PREC), L::N_BIAS, REF_LDS_BIAS);
    const float* ide_mat = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + L::stream_bytes(P::PREC)) + L::N_BIAS;
    WeightStream<P, MLP_NSLOT_REF> ws;
    ws.init(packed, L::N_FRAGS / FPC);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform: lets address arithmetic on it run on the scalar unit)
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    const uint32_t bias0 = REF_LDS_BIAS;
    const uint32_t stash0 = REF_LDS_STASH + wave * NT * 11 * P::BREG_LDS + lane * 16;  // 11 per-subtile blocks of one K group each
    auto stash = [&](int t) -> uint32_t { return stash0 + t * 11 * P::BREG_LDS; };
    auto dir_lds = [&](int t) -> uint32_t { return stash(t) + 10 * P::BREG_LDS; };         // block 10 doubles as the direction slot

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int64_t m[NT];
        BReg a[NT][16], b[NT][16];
        int lay = 0, lay_pend = 0;                                  // training dump slots of the layer being computed / of the pending pair
        const int64_t sub0 = tile * (TS / 32) + wave * NT;
        // (`lay` = the layer whose block pairs are being computed, `lay_pend` = the layer that owns the deferred pair, see proposal_kernel)
        // TRAIN (round 4): next to the activations, ONE ReLU bit per activation for the dgrad chains (32 B per sample and layer instead of
        // the 512 B of activations they re-read as masks until round 3: -38 GB of the 156 GB a 2^14-ray step moved).  Same record layout as
        // the proposal / MipNeRF forwards' (mask_or), but accumulated in ONE REGISTER per column tile instead of an LDS record -- this
        // kernel's LDS is full (ring 48 + biases 16.75 + stash 88 KiB): the four K groups of a feature-block pair are converted back to back
        // (pair p = K groups 4p .. 4p+3 = dword p of the lane's 16-byte record), so the dword is complete, stored and cleared at kg & 3 == 3.
        uint32_t mreg[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) mreg[t] = 0u;
        uint32_t mrec[NT][4];
        auto put = [&](BReg (&buf)[NT][16], int layer, int fb, int t, const f32x16& acc, int half) {
            const int kg = 2 * fb + half;
            buf[t][kg] = to_breg_half<P, true>(acc, half);
            if constexpr (TRAIN) {
                dump_breg<P>(dump, layer, sub0 + t, kg, lane, buf[t][kg]);
                mreg[t] |= breg_bits(buf[t][kg]) << (4 * (kg & 3));
                if ((kg & 3) == 3) {                               // the dword of K groups kg - 3 .. kg is complete
                    mrec[t][kg >> 2] = mreg[t];
                    mreg[t] = 0u;
                    if (kg == 15) {                                // ... and so is the lane's 16-byte record: ONE store per layer and subtile (it was four)
                        const u32x4 rec = {mrec[t][0], mrec[t][1], mrec[t][2], mrec[t][3]};
                        *reinterpret_cast<u32x4*>(dump.mask_base + (size_t)layer * dump.mask_layer_stride + (size_t)(sub0 + t) * 1024 + lane * 16) = rec;
                    }
                }
            }
        };
        auto OA = [&](int fb, int t, const f32x16& acc, int half) { put(a, lay, fb, t, acc, half); };
        auto OB = [&](int fb, int t, const f32x16& acc, int half) { put(b, lay, fb, t, acc, half); };
        auto OA_pend = [&](int fb, int t, const f32x16& acc, int half) { put(a, lay_pend, fb, t, acc, half); };
        auto OB_pend = [&](int fb, int t, const f32x16& acc, int half) { put(b, lay_pend, fb, t, acc, half); };
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
        Deferred<P, 6, 2> d;                                        // see mip_kernel
        {
            BReg enc[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t) m[t] = tile * TS + (wave * NT + t) * 32 + j;
            if constexpr (NT == 2 && P::FAST_PE) {        // paired prologue (see encode_pair)
                const int64_t mo = h ? m[1] : m[0];
                const Sample sm = fetch_sample(s, mo < s.M ? mo : s.M - 1, true);
                encode_pair<P, 10, 4>(sm.x, sm.y, sm.z, enc[0], enc[1]);
                half_swap_groups<4>(enc[0], enc[1]);
                uint32_t d0[3] = {__builtin_bit_cast(uint32_t, sm.dx), __builtin_bit_cast(uint32_t, sm.dy), __builtin_bit_cast(uint32_t, sm.dz)};
                uint32_t d1[3] = {d0[0], d0[1], d0[2]};
#pragma unroll
                for (int c = 0; c < 3; ++c) half_swap(d0[c], d1[c]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) P::stash(stash(t) + k * P::BREG_LDS, enc[t][k]);
                    if constexpr (TRAIN) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) dump_breg<P>(dump, 8, sub0 + t, 11 + k, lane, enc[t][k]);
                    }
                    const uint32_t* dd = t ? d1 : d0;
                    const f32x4 dv = {__builtin_bit_cast(float, dd[0]), __builtin_bit_cast(float, dd[1]), __builtin_bit_cast(float, dd[2]), 0.0f};
                    *reinterpret_cast<f32x4*>(smem + dir_lds(t)) = dv;
                }
            } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const Sample sm = fetch_sample(s, m[t] < s.M ? m[t] : s.M - 1, true);
                encode<P, 10, 4>(sm.x, sm.y, sm.z, h, enc[t]);
#pragma unroll
                for (int k = 0; k < 4; ++k) P::stash(stash(t) + k * P::BREG_LDS, enc[t][k]);
                if constexpr (TRAIN) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) dump_breg<P>(dump, 8, sub0 + t, 11 + k, lane, enc[t][k]);
                }
                f32x4 dv = {sm.dx, sm.dy, sm.dz, 0.0f};
                *reinterpret_cast<f32x4*>(smem + dir_lds(t)) = dv;
            }
            }
            d = dense<P, 4, 8, L::START[0]>(ws, bias0 + L::BIAS_OFF[0] * 4,                  // spa_block1.0 (lay = lay_pend = 0)
                [&](int kg, int t) -> BReg { return enc[t][kg]; }, OA, NoPrev{});
        }
        static_assert(L::START[1] % (2 * FPC) == L::START[3] % (2 * FPC) && L::START[5] % (2 * FPC) == L::START[7] % (2 * FPC) &&
                      L::BIAS_OFF[3] == L::BIAS_OFF[1] + 512 && L::BIAS_OFF[7] == L::BIAS_OFF[5] + 512, "chunk parity / bias spacing of the shared a -> b instance");
        // The layers ping-pong between the two register buffers (a -> b -> a -> b; the skip layer b -> a; a -> b -> a -> b), so nothing is
        // copied between layers; the two a -> b layers of a block share one code instance through the loop (same chunk parity, asserted).
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {                                                         // spa_block1.{2,4,6}
            lay_pend = lay; lay = 1 + 2 * r;
            d = dense<P, 16, 8, L::START[1]>(ws, bias0 + (L::BIAS_OFF[1] + 2 * r * 256) * 4, IN_A, OB, prev_of(d, OA_pend));
            if (r == 0) {
                lay_pend = lay; lay = 2;
                d = dense<P, 16, 8, L::START[2]>(ws, bias0 + L::BIAS_OFF[2] * 4, IN_B, OA, prev_of(d, OB_pend));
            }
        }
        lay_pend = lay; lay = 4;
        d = dense<P, 20, 8, L::START[4]>(ws, bias0 + L::BIAS_OFF[4] * 4,                      // spa_block2.0 (skip): cat(enc, b) -> a
            [&](int kg, int t) -> BReg { if (kg < 4) return P::unstash(stash(t) + kg * P::BREG_LDS); return b[t][kg >= 4 ? kg - 4 : 0]; },
            OA, prev_of(d, OB_pend));
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {                                                         // spa_block2.{2,4,6}
            lay_pend = lay; lay = 5 + 2 * r;
            d = dense<P, 16, 8, L::START[5]>(ws, bias0 + (L::BIAS_OFF[5] + 2 * r * 256) * 4, IN_A, OB, prev_of(d, OA_pend));
            if (r == 0) {
                lay_pend = lay; lay = 6;
                d = dense<P, 16, 8, L::START[6]>(ws, bias0 + L::BIAS_OFF[6] * 4, IN_B, OA, prev_of(d, OB_pend));
            }
        }
        lay_pend = lay;                                                                       // (the spatial features are in b)
        // heads: bottle_neck (4 blocks, no activation) + [normal | roughness || diffuse | density || tint]
        BReg bn[NT][8];
        f32x16 hd[NT];
        auto OHD = [&](int fb, int t, const f32x16& acc, int half) {
            if (fb < 4) {
                f32x16 v = acc;
                if (bn_noise != nullptr) {
                    // accumulators 8*half + e of block fb are features 32 fb + 16 half + 8 (e >> 2) + 4 h + (e & 3): two runs of four
                    const int64_t mm = m[t] < s.M ? m[t] : s.M - 1;
                    const float* np_ = bn_noise + mm * 128 + 32 * (fb < 4 ? fb : 0) + 16 * half + 4 * h;
                    const f32x4 n0 = *reinterpret_cast<const f32x4*>(np_), n1 = *reinterpret_cast<const f32x4*>(np_ + 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[8 * half + e] += n0[e]; v[8 * half + 4 + e] += n1[e]; }
                } else if (noise_std > 0.0f) {
                    float z[8];
                    philox_normal8(noise_seed, m[t] < s.M ? m[t] : s.M - 1, 2 * (2 * (fb < 4 ? fb : 0) + half) + h, noise_std, z);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[8 * half + e] += z[e];
                }
                bn[t][2 * (fb < 4 ? fb : 0) + half] = to_breg_half<P, false>(v, half);
                if constexpr (TRAIN) dump_breg<P>(dump, 8, sub0 + t, 2 * (fb < 4 ? fb : 0) + half, lane, bn[t][2 * (fb < 4 ? fb : 0) + half]);
            } else if (half == 0) hd[t] = acc;
        };
        dense<P, 16, 5, L::START[8]>(ws, bias0 + L::BIAS_OFF[8] * 4, IN_B, OHD, prev_of(d, OB_pend)).flush(OHD);
        // half 0 holds rows 0-3 (normal, roughness) in hd[0..3] and rows 8-10 (tint) in hd[4..6]; half 1 rows 4-7 (diffuse, density) in hd[0..3]
        float keep[NT][4];
        BReg ide[NT][3];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float nx0 = __shfl(hd[t][0], j, 64), ny0 = __shfl(hd[t][1], j, 64), nz0 = __shfl(hd[t][2], j, 64);
            const float rough = softplus_f(__shfl(hd[t][3], j, 64) - 1.0f);                   // ref_model.py:82
            const float nn = norm3(nx0, ny0, nz0) + 1e-7f;                                    // ref_model.py:87
            const float nx = -nx0 / nn, ny = -ny0 / nn, nz = -nz0 / nn;
            const f32x4 dv = *reinterpret_cast<const f32x4*>(smem + dir_lds(t));
            const float dot = (dv[0] * nx + dv[1] * ny) + dv[2] * nz;
            const float t2 = 2.0f * dot;
            const float rx = dv[0] - t2 * nx, ry = dv[1] - t2 * ny, rz = dv[2] - t2 * nz;    // ref_model.py:90
            ide_encode<P>(rx, ry, rz, rough, dot, h, ide_mat, ide[t]);
            if (normal_out && h == 0 && m[t] < s.M) { normal_out[m[t] * 3] = nx; normal_out[m[t] * 3 + 1] = ny; normal_out[m[t] * 3 + 2] = nz; }
            keep[t][0] = h ? hd[t][0] : hd[t][4]; keep[t][1] = h ? hd[t][1] : hd[t][5]; keep[t][2] = h ? hd[t][2] : hd[t][6]; keep[t][3] = hd[t][3];
            if constexpr (TRAIN) {                           // pre-activation heads for the backward: lane half 0 holds rows 0-3 and 8-10, half 1 rows 4-7
#pragma unroll
                for (int k = 0; k < 3; ++k) dump_breg<P>(dump, 8, sub0 + t, 8 + k, lane, ide[t][k]);
                if (m[t] < s.M) {
                    float* ax = aux + m[t] * 16;
                    if (h == 0) {
                        *reinterpret_cast<f32x4*>(ax) = f32x4{hd[t][0], hd[t][1], hd[t][2], hd[t][3]};
                        ax[8] = hd[t][4]; ax[9] = hd[t][5]; ax[10] = hd[t][6];
                    } else {
                        *reinterpret_cast<f32x4*>(ax + 4) = f32x4{hd[t][0], hd[t][1], hd[t][2], hd[t][3]};
                    }
                }
            }
            // all_inputs = [bottle_neck 128 | ide 38 | n.d]: needed again by dir_block2.0 -> park in the stash (blocks 0..10)
#pragma unroll
            for (int k = 0; k < 8; ++k) P::stash(stash(t) + k * P::BREG_LDS, bn[t][k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) P::stash(stash(t) + (8 + k) * P::BREG_LDS, ide[t][k]);
        }
        lay = lay_pend = 9;
        d = dense<P, 11, 8, L::START[9]>(ws, bias0 + L::BIAS_OFF[9] * 4,                      // dir_block1.0
            [&](int kg, int t) -> BReg { if (kg < 8) return bn[t][kg < 8 ? kg : 0]; return ide[t][kg >= 8 ? kg - 8 : 0]; },
            OA, NoPrev{});
        static_assert(L::START[10] % (2 * FPC) == L::START[12] % (2 * FPC) && L::START[14] % (2 * FPC) == L::START[16] % (2 * FPC) &&
                      L::BIAS_OFF[12] == L::BIAS_OFF[10] + 512 && L::BIAS_OFF[16] == L::BIAS_OFF[14] + 512, "chunk parity / bias spacing of the shared a -> b instance");
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {                                                         // dir_block1.{2,4,6}
            lay_pend = lay; lay = 10 + 2 * r;
            d = dense<P, 16, 8, L::START[10]>(ws, bias0 + (L::BIAS_OFF[10] + 2 * r * 256) * 4, IN_A, OB, prev_of(d, OA_pend));
            if (r == 0) {
                lay_pend = lay; lay = 11;
                d = dense<P, 16, 8, L::START[11]>(ws, bias0 + L::BIAS_OFF[11] * 4, IN_B, OA, prev_of(d, OB_pend));
            }
        }
        lay_pend = lay; lay = 13;
        d = dense<P, 27, 8, L::START[13]>(ws, bias0 + L::BIAS_OFF[13] * 4,                    // dir_block2.0 (skip): cat(all_inputs, b) -> a
            [&](int kg, int t) -> BReg { if (kg < 11) return P::unstash(stash(t) + kg * P::BREG_LDS); return b[t][kg >= 11 ? kg - 11 : 0]; },
            OA, prev_of(d, OB_pend));
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {                                                         // dir_block2.{2,4,6}
            lay_pend = lay; lay = 14 + 2 * r;
            d = dense<P, 16, 8, L::START[14]>(ws, bias0 + (L::BIAS_OFF[14] + 2 * r * 256) * 4, IN_A, OB, prev_of(d, OA_pend));
            if (r == 0) {
                lay_pend = lay; lay = 15;
                d = dense<P, 16, 8, L::START[15]>(ws, bias0 + L::BIAS_OFF[15] * 4, IN_B, OA, prev_of(d, OB_pend));
            }
        }
        lay_pend = lay;
        float sr[NT], sg[NT], sb[NT];
        auto OSPEC = [&](int, int t, const f32x16& acc, int half) { if (half == 0) { sr[t] = acc[0]; sg[t] = acc[1]; sb[t] = acc[2]; } };
        dense<P, 16, 1, L::START[17]>(ws, bias0 + L::BIAS_OFF[17] * 4, IN_B, OSPEC, prev_of(d, OB_pend)).flush(OSPEC);     // spec_rgb_head.0
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // half 1's keep[] = (diffuse, density); fetch it into half 0, which holds spec and tint (ref_model.py:98-105)
            const float d0 = __shfl(keep[t][0], j + 32, 64), d1 = __shfl(keep[t][1], j + 32, 64), d2 = __shfl(keep[t][2], j + 32, 64),
                        dens = __shfl(keep[t][3], j + 32, 64);
            if (h == 0 && m[t] < s.M) {
                auto sig = [](float v) { return 1.0f / (1.0f + expf(-v)); };
                f32x4 o;
                if (flags & NERF_AMD_REF_SRGB) {                                   // ref_model.py:100-102
                    o[0] = srgb_from_linear(sig(sr[t]) * sig(keep[t][0]) + sig(d0 - SRGB_LOG3));
                    o[1] = srgb_from_linear(sig(sg[t]) * sig(keep[t][1]) + sig(d1 - SRGB_LOG3));
                    o[2] = srgb_from_linear(sig(sb[t]) * sig(keep[t][2]) + sig(d2 - SRGB_LOG3));
                } else {                                                            // ref_model.py:104-105
                    o[0] = sig(sr[t]) * sig(keep[t][0]) + sig(d0);
                    o[1] = sig(sg[t]) * sig(keep[t][1]) + sig(d1);
                    o[2] = sig(sb[t]) * sig(keep[t][2]) + sig(d2);
                }
                o[3] = dens;
                *reinterpret_cast<f32x4*>(rgbo + m[t] * 4) = o;
                if constexpr (TRAIN) { float* ax = aux + m[t] * 16; ax[11] = sr[t]; ax[12] = sg[t]; ax[13] = sb[t]; ax[14] = 0.0f; ax[15] = 0.0f; }
            }
        }
    }
    ws.drain();
}

int grid_for(int64_t n_tiles) {
    const int n_cu = nerf_host::cu_count();
    return (int)(n_tiles < n_cu ? n_tiles : n_cu);
}

// Dynamic LDS above 64 KiB is an opt-in per KERNEL FUNCTION and device (host_common.h)
int allow_dynamic_lds(const void* fn, size_t lds) { return nerf_host::allow_dynamic_lds(fn, lds); }

template <class P, class Lay, bool TRAIN = false, bool F8 = false, class K, class... Extra>
int launch(K kernel, const void* packed, const nerf_amd_samples& s, float* out, hipStream_t st, Extra... extra) {
    constexpr int TS = P::NW * P::NT * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    if (n_tiles == 0) return 0;
    const size_t lds = F8 ? lds_total_train_f8<P>() : (TRAIN ? lds_total_train<P>() : lds_total<P>());
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(kernel), lds)) return e;
    hipLaunchKernelGGL(kernel, dim3(grid_for(n_tiles)), dim3(P::NW * 64), lds, st, packed, s, out, extra...);
    return (int)hipGetLastError();
}
const ActDump NO_DUMP{nullptr, 0ull, nullptr, 0ull};

// bf16 policy of the shipped library: the wide tile (measured 2.5 % faster end to end);
// -DMLP_BF16_NARROW selects the 8-wave x 32-sample tile for A/B runs
using PB16 = PBF16W;

}  // namespace

// host-visible launchers (capi.hip).  The file is compiled three times (Makefile: -DMLP_TU=1 proposal, 2 MipNeRF, 3 Ref-NeRF; undefined
// = everything) so that the three kernel families build in parallel -- one instance of the fine kernel alone takes a minute of hipcc.
#ifndef MLP_TU
#define MLP_TU 0
#endif
#if MLP_TU == 0 || MLP_TU == 1
int mlp_launch_proposal(const void* packed, int precision, const nerf_amd_samples& s, float* density, hipStream_t st) {
    if (precision == NERF_AMD_BF16) return launch<PB16, PropLayout>(proposal_kernel<PB16, false>, packed, s, density, st, NO_DUMP);
    return launch<PF32, PropLayout>(proposal_kernel<PF32, false>, packed, s, density, st, NO_DUMP);
}
#endif
#if MLP_TU == 0 || MLP_TU == 2
int mlp_launch_mip(const void* packed, int precision, const nerf_amd_samples& s, float* rgbo, hipStream_t st) {
    const FusedComposite off{nullptr, nullptr, nullptr, 0, 0.0f, 1.0f};
    if (s.ipe) {                                            // integrated positional encoding (validated by the C-ABI: mode 1, z, dir norm)
        if (precision == NERF_AMD_BF16) return launch<PB16, MipLayout>(mip_kernel<PB16, false, true>, packed, s, rgbo, st, off, NO_DUMP);
        return launch<PF32, MipLayout>(mip_kernel<PF32, false, true>, packed, s, rgbo, st, off, NO_DUMP);
    }
    if (precision == NERF_AMD_BF16) return launch<PB16, MipLayout>(mip_kernel<PB16, false>, packed, s, rgbo, st, off, NO_DUMP);
    return launch<PF32, MipLayout>(mip_kernel<PF32, false>, packed, s, rgbo, st, off, NO_DUMP);
}
// packed = nerf_amd_pack_weights(NERF_AMD_NET_MIP_128, ...); point PE only (the C-ABI refuses s.ipe with this layout)
int mlp_launch_mip128(const void* packed, int precision, const nerf_amd_samples& s, float* rgbo, hipStream_t st) {
    if (precision == NERF_AMD_BF16) return launch<PB16, MipLayout128>(mip128_kernel<PB16>, packed, s, rgbo, st);
    return launch<PF32, MipLayout128>(mip128_kernel<PF32>, packed, s, rgbo, st);
}
// fine MLP + compositing in one launch; requires mode 1 (rays + z) and S in {32, 64, 128}
int mlp_launch_mip_composite(const void* packed, int precision, const nerf_amd_samples& s, float* rgb, float* depth, float* weights,
                             int white_bkg, float near, float far, hipStream_t st) {
    const FusedComposite fc{rgb, depth, weights, white_bkg, near, far};
    if (precision == NERF_AMD_BF16) return launch<PB16, MipLayout>(mip_kernel<PB16, false, false, false, true>, packed, s, (float*)nullptr, st, fc, NO_DUMP);
    return launch<PF32, MipLayout>(mip_kernel<PF32, false, false, false, true>, packed, s, (float*)nullptr, st, fc, NO_DUMP);
}
#endif
#if MLP_TU == 0 || MLP_TU == 1
template <class P>
static int launch_proposal128(const void* packed, const nerf_amd_samples& s, float* density, hipStream_t st) {
    constexpr int TS = P::NW * P::NT * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    if (n_tiles == 0) return 0;
    const size_t lds = lds_total_narrow();
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(proposal128_kernel<P>), lds)) return e;
    hipLaunchKernelGGL(proposal128_kernel<P>, dim3(grid_for(n_tiles)), dim3(P::NW * 64), lds, st, packed, s, density);
    return (int)hipGetLastError();
}
// packed = nerf_amd_pack_weights(NERF_AMD_NET_PROPOSAL_128, ...)
int mlp_launch_proposal128(const void* packed, int precision, const nerf_amd_samples& s, float* density, hipStream_t st) {
    if (precision == NERF_AMD_BF16) return launch_proposal128<PBF16N>(packed, s, density, st);
    return launch_proposal128<PF32>(packed, s, density, st);
}
#endif
// training forwards: the same kernels, also dumping the hidden activations (ActDump) for the backward
#if MLP_TU == 0 || MLP_TU == 1
size_t mlp_train_layer_stride(int precision, int64_t M) {
    const int64_t ts = (precision != NERF_AMD_F32) ? (int64_t)PB16::NW * PB16::NT * 32 : (int64_t)PF32::NW * PF32::NT * 32;   // (BF16_F8 slots keep the bf16 footprint)
    const int64_t n_sub = ((M + ts - 1) / ts) * (ts / 32);
    return (size_t)n_sub * 16 * (precision != NERF_AMD_F32 ? 1024 : 2048);
}
// the ReLU bit masks sit behind the `slots` activation slots of a dump: 1 KiB per slot and subtile
size_t mlp_train_mask_stride(int precision, int64_t M) { return mlp_train_layer_stride(precision, M) / (16 * (precision != NERF_AMD_F32 ? 1024 : 2048)) * 1024; }
#else
size_t mlp_train_layer_stride(int precision, int64_t M);
size_t mlp_train_mask_stride(int precision, int64_t M);
#endif
static ActDump make_dump(void* dump, int precision, int64_t M, int slots) {
    const unsigned long long ls = mlp_train_layer_stride(precision, M);
    return ActDump{reinterpret_cast<char*>(dump), ls, reinterpret_cast<char*>(dump) + (size_t)slots * ls, (unsigned long long)mlp_train_mask_stride(precision, M)};
}
#if MLP_TU == 0 || MLP_TU == 1
int mlp_launch_proposal_train(const void* packed, int precision, const nerf_amd_samples& s, float* density, void* dump, hipStream_t st) {
    if (precision == NERF_AMD_BF16_F8) {                    // bf16 arithmetic, hidden slots of the dump in scaled e4m3 (mlp_layout.h)
        const ActDump d8 = make_dump(dump, NERF_AMD_BF16, s.M, PROP_DUMP_SLOTS);
        return launch<PB16, PropLayout, true, true>(proposal_kernel<PB16, true, true>, packed, s, density, st, d8);
    }
    const ActDump d = make_dump(dump, precision, s.M, PROP_DUMP_SLOTS);
    // bf16 training forward of the proposal network: the 8-wave x 32-sample tile (two waves per SIMD hide the dump's stores and mask
    // arithmetic, and the 64-sample tile spills ~90 registers here) -- 0.83 -> 0.73 ms at 16 384 rays in a same-box A/B; the dump layout
    // does not depend on the tile policy (32-sample subtiles either way).  The MipNeRF forward and both dgrad chains measured no gain.
    if (precision == NERF_AMD_BF16) return launch<PBF16, PropLayout, true>(proposal_kernel<PBF16, true>, packed, s, density, st, d);
    return launch<PF32, PropLayout, true>(proposal_kernel<PF32, true>, packed, s, density, st, d);
}
#endif
#if MLP_TU == 0 || MLP_TU == 2
int mlp_launch_mip_train(const void* packed, int precision, const nerf_amd_samples& s, float* rgbo, void* dump, hipStream_t st) {
    const FusedComposite off{nullptr, nullptr, nullptr, 0, 0.0f, 1.0f};
    if (precision == NERF_AMD_BF16_F8) {                    // bf16 arithmetic, hidden slots of the dump in scaled e4m3 (mlp_layout.h)
        const ActDump d8 = make_dump(dump, NERF_AMD_BF16, s.M, MIP_DUMP_SLOTS);
        if (s.ipe) return launch<PB16, MipLayout, true, true>(mip_kernel<PB16, true, true, true>, packed, s, rgbo, st, off, d8);
        return launch<PB16, MipLayout, true, true>(mip_kernel<PB16, true, false, true>, packed, s, rgbo, st, off, d8);
    }
    const ActDump d = make_dump(dump, precision, s.M, MIP_DUMP_SLOTS);
    if (s.ipe) {                                            // integrated PE on the training path (BASELINE configs[2]): the dumped encoding slot
        // holds the IPE features in the PE10 slot map, so the dgrad chain and the weight-gradient kernels run unchanged
        if (precision == NERF_AMD_BF16) return launch<PB16, MipLayout, true>(mip_kernel<PB16, true, true>, packed, s, rgbo, st, off, d);
        return launch<PF32, MipLayout, true>(mip_kernel<PF32, true, true>, packed, s, rgbo, st, off, d);
    }
    if (precision == NERF_AMD_BF16) return launch<PB16, MipLayout, true>(mip_kernel<PB16, true>, packed, s, rgbo, st, off, d);
    return launch<PF32, MipLayout, true>(mip_kernel<PF32, true>, packed, s, rgbo, st, off, d);
}

#endif
#if MLP_TU == 0 || MLP_TU == 3
template <class P, bool TRAIN>
static int launch_ref(const void* packed, const nerf_amd_samples& s, float* rgbo, float* normal, const float* bn_noise, ActDump dump, float* aux,
                      int flags, hipStream_t st, unsigned long long seed = 0, const unsigned long long* seed_dev = nullptr, float noise_std = 0.0f) {
    constexpr int TS = P::NW * P::NT * 32;
    const int64_t n_tiles = (s.M + TS - 1) / TS;
    if (n_tiles == 0) return 0;
    const size_t lds = ref_lds_total<P>();
    if (int e = allow_dynamic_lds(reinterpret_cast<const void*>(ref_kernel<P, TRAIN>), lds)) return e;
    hipLaunchKernelGGL((ref_kernel<P, TRAIN>), dim3(grid_for(n_tiles)), dim3(P::NW * 64), lds, st, packed, s, rgbo, normal, bn_noise, dump, aux, flags, seed, seed_dev, noise_std);
    return (int)hipGetLastError();
}
int mlp_launch_ref(const void* packed, int precision, const nerf_amd_samples& s, float* rgbo, float* normal, const float* bn_noise,
                   int flags, hipStream_t st) {
    if (precision == NERF_AMD_BF16) return launch_ref<PB16, false>(packed, s, rgbo, normal, bn_noise, NO_DUMP, nullptr, flags, st);
    return launch_ref<PF32, false>(packed, s, rgbo, normal, bn_noise, NO_DUMP, nullptr, flags, st);
}
// training forward of Ref-NeRF: activation dump (REF_DUMP_SLOTS slots of mlp_train_layer_stride bytes) + aux (M,16)
int mlp_launch_ref_train(const void* packed, int precision, const nerf_amd_samples& s, float* rgbo, float* normal, const float* bn_noise,
                         void* dump, float* aux, int flags, hipStream_t st, unsigned long long seed, const unsigned long long* seed_dev, float noise_std) {
    const ActDump d = make_dump(dump, precision, s.M, REF_DUMP_SLOTS);     // 17 activation slots + (round 4) their ReLU bit-mask records
    // bf16 training forward: the 8-wave x 32-sample tile, like the proposal network's -- its 512 dump stores per wave and tile cost their
    // ISSUE slots (~80 cycles of the vector-memory path each, during which a lone wave per SIMD issues no MFMA; not a wait: the store-aware
    // ring wait changed nothing, mlp_core.h), which a second wave per SIMD fills: 9.01 -> 8.38 ms per 2^14-ray step, same box, alternated
    // twice (profiles/r04_ref_train_fwd_8wave_ab.log); the dump layout does not depend on the tile policy.  -DREF_TRAIN_WIDE = the A side.
#ifndef REF_TRAIN_WIDE
    if (precision == NERF_AMD_BF16) return launch_ref<PBF16, true>(packed, s, rgbo, normal, bn_noise, d, aux, flags, st, seed, seed_dev, noise_std);
#endif
    if (precision == NERF_AMD_BF16) return launch_ref<PB16, true>(packed, s, rgbo, normal, bn_noise, d, aux, flags, st, seed, seed_dev, noise_std);
    return launch_ref<PF32, true>(packed, s, rgbo, normal, bn_noise, d, aux, flags, st, seed, seed_dev, noise_std);
}
#endif
