// Layer products on bf16 ROWS (round 5): the inference route of networks larger than the shapes the fused MLP kernels are compiled for
// (`--nerf_net_width` / `--prop_net_width` above 256, procedures.py:176-177; `--ide_level 5`, ref_model.py:33-36) under bf16 precision.
// generic_kernels.hip's `gemm_kernel` carries the activations of such a network through HBM as fp32 rows and stages them through registers
// (210 TFLOP/s forward at 262 144 x 512 x 512: its HBM floor alone is 640 TFLOP/s); it stays the path of the fp32 parity mode and of the
// backward (whose weight-gradient products want fp32 rows).  Here one nn.Linear (+ activation) of mip_model.py:41-60 / addtional.py:88-96 /
// ref_model.py:68-106 is
//
//     C[m, n] = act( sum_k X[m, k] * W[n, k] + bias[n] )          X: bf16 rows (M, K), W: bf16 rows (N, K) -- both K-contiguous
//
// with C written as bf16 rows for the next layer (or fp32 rows for the element-wise stage / the caller that follows a head).
//
// Tile (RowsCfg): BM samples x BN features per workgroup, WM x WN wavefronts of MB x NB MFMA blocks of 32 x 32 each
// (v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand, like the fused kernels: an accumulator register then holds four consecutive
// features of one sample).  Contraction in stages of BK elements: a ring of [BM x BK activations | BN x BK weights] images filled
// SLOTS - 1 stages ahead by global_load_lds_dwordx4 -- no staging registers, no ds_write -- ONE s_barrier per stage.  The LDS image of a
// 1 KiB piece is lane-linear (16 rows x 64 B at BK = 32, 8 rows x 128 B at BK = 64); the bank swizzle is a permutation of the SOURCE
// chunk a lane fetches (RowsCfg::swz), which puts the 16 lanes of every ds_read_b128 service group on 16 different 16-byte slots of the
// 256-byte bank row (MI355X_MICROARCH.md, LDS; SQ_LDS_BANK_CONFLICT = 0, profiles/r05_rows_gemm_pmc.txt).  Out-of-range rows / chunks are
// fetched from a 16-byte page of zeros; a partly valid last chunk multiplies the row's own padding (finite by contract) with packed zero
// weights.  Epilogue: bias + activation in registers, bf16, transposed through the (then idle) ring into full row pieces.
//
// Where it stands (262 144 x 512 x 512, profiles/r05_rows_gemm_*): 520-570 TFLOP/s sustained (598 in a five-launch profile), against
// 215 of nerf_amd_gemm and 713 of hipBLASLt's plain product (256 x 256 x 32 tiles on four wavefronts of 128 x 128, hand-scheduled).
// Probes on one box: products + epilogue without the operand loads 0.163 ms, loads + epilogue without the products 0.205 ms, the
// epilogue alone (268 MB of output rows) 0.093 ms, all three 0.266 ms -- the launch waits for the memory system (operand delivery
// L2 -> LDS at ~6 TB/s chip-wide, rows at 2-3 TB/s of HBM; TCP_PENDING_STALL half of every CU's cycles), the matrix cores are 26 % busy.
// Measured and not kept: 64 x / 128 x 128-wide and whole-row tiles, the library's tile shape, staggering the two resident workgroups,
// nt loads / stores, the activations through registers (global_load_dwordx4 -> ds_write_b128) instead of the DMA, the same four
// stages ahead of the weights: all within +- 5 % or slower.  The step beyond is the fused kernels' design (activations never leave the CU) or the library's hand-scheduled loop.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "host_common.h"

namespace {

// tile configuration: BM samples x BN features per workgroup, BK contraction elements per stage, WM x WN wavefronts, MINB workgroups per CU
template <int BM_, int BN_, int BK_, int SLOTS_, int WM_, int WN_, int MINB_>
struct RowsCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, SLOTS = SLOTS_, WM = WM_, WN = WN_, MINB = MINB_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int MB = BM / WM / 32, NB = BN / WN / 32;     // 32 x 32 blocks of a wave: MB sample blocks x NB feature blocks
    static constexpr int ROWB = BK * 2;                             // bytes of a row piece: 64 or 128 (a full cache line)
    static constexpr int CH = BK / 8;                               // its 16-byte chunks: 4 or 8
    static constexpr int RPP = 64 / CH;                             // rows of a 1 KiB piece (one global_load_lds_dwordx4): 16 or 8
    static constexpr int XIMG = BM * ROWB, WIMG = BN * ROWB, SLOT = XIMG + WIMG;      // a stage's images: [activations | weights]
    static constexpr int CT_LD = BN * 2 + 8;                        // row stride of the transposed output image (/ 4 = 2 mod 32: ds_write_b64 of 16 rows hit 32 banks once)
    static constexpr int BIAS_OFF = BM * CT_LD > SLOTS * SLOT ? BM * CT_LD : SLOTS * SLOT;      // the tile's BN bias values, behind the ring / output image
    static constexpr int LDS = BIAS_OFF + BN * 4;
    static constexpr int XLOADS = XIMG / (THREADS * 16), WLOADS = WIMG / (THREADS * 16), LOADS = XLOADS + WLOADS;
    static constexpr int KSTEPS = BK / 16;
    static_assert((BK == 32 || BK == 64) && XLOADS >= 1 && WLOADS >= 1 && MB >= 1 && NB >= 1 && SLOTS >= 2 && SLOTS <= 4 && THREADS >= BN, "RowsCfg");
    // the bank swizzle: row r keeps its chunk c at position c ^ swz(r) -- the 16 lanes of a ds_read_b128 service group read 16 different rows
    // (mod 16) at one chunk index, and land on 16 different 16-byte slots of the 256-byte bank row
    static DEVINL int swz(int r) { return CH == 4 ? (r >> 2) & 3 : (r >> 1) & 7; }
};

extern __shared__ __attribute__((aligned(16))) char rg_smem[];
__device__ __attribute__((aligned(16))) uint32_t rg_zero_page[4] = {0u, 0u, 0u, 0u};

struct RowsGemmArgs {
    int64_t M, N, K;                // N, K: the layer's own sizes (the packed weights are padded)
    const uint16_t* X; int64_t ldx; // elements; multiple of 8, base 16-byte aligned
    const uint16_t* W; int64_t ldw; // packed weights (Npad, Kpad): Npad % 256 == 0, Kpad = ldw % 64 == 0, zeros outside (N, K)
    const float* bias;              // Npad floats (zeros beyond N)
    void* C; int64_t ldc;           // bf16 rows (ldc % 4 == 0, base 8-byte aligned, N % 4 == 0) or fp32 rows (any)
    int act;
    int tiles_m, tiles_n;
};

// workgroup -> output tile, XCD-aware: the dispatcher places block b on XCD b % 8; every XCD gets a contiguous range of logical ids, and the
// logical order keeps the feature tiles of one block of samples adjacent (they share its activations through one L2)
DEVINL void rows_tile_of(const RowsGemmArgs& g, int64_t& ti, int64_t& tj) {
    const int64_t nwg = gridDim.x, b = blockIdx.x;
    const int64_t q = nwg / 8, r = nwg % 8, xcd = b % 8;
    const int64_t id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    ti = id / g.tiles_n;
    tj = id - ti * g.tiles_n;
}

DEVINL void glds16(const char* src, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}

template <int N> DEVINL void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

DEVINL uint32_t pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

DEVINL float activate(float v, int act) {
    if (act == 1) return v > 0.0f ? v : 0.0f;
    if (act == 2) return 1.0f / (1.0f + expf(-v));
    return v;
}

// two 16-steps of a stage for this wave's MB x NB blocks, all feature blocks live: the fragment reads of both steps are issued together (the
// compiler's own schedule re-used four fragment registers and waited for LDS four times per stage), the products follow as they land
template <class C>
DEVINL void rows_two_steps(f32x16 (&acc)[C::NB][C::MB], uint32_t x0, uint32_t w0, uint32_t x1, uint32_t w1) {
    bf16x8 xf[2][C::MB], wf[2][C::NB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) xf[0][mb] = *reinterpret_cast<const bf16x8*>(rg_smem + x0 + mb * 32 * C::ROWB);
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) wf[0][nb] = *reinterpret_cast<const bf16x8*>(rg_smem + w0 + nb * 32 * C::ROWB);
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) xf[1][mb] = *reinterpret_cast<const bf16x8*>(rg_smem + x1 + mb * 32 * C::ROWB);
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) wf[1][nb] = *reinterpret_cast<const bf16x8*>(rg_smem + w1 + nb * 32 * C::ROWB);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < C::MB; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][nb], xf[ks][mb], acc[nb][mb], 0, 0, 0);
}

// FULL: every wave of every tile holds only live feature blocks (N % BN == 0: the hidden layers at widths 256 / 512 / ...) -- no guards in
// the contraction loop; otherwise (a ragged last feature tile, the heads) the blocks beyond N are skipped block by block.
template <class C, bool OUT_BF16, bool FULL>
__global__ __launch_bounds__(C::THREADS, C::MINB) void rows_gemm_kernel(RowsGemmArgs g) {
    constexpr int MB = C::MB, NB = C::NB, ROWB = C::ROWB;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave / C::WN, wn = wave % C::WN;               // this wave: samples 32 MB wm .. , features 32 NB wn .. of the tile
    int64_t ti, tj;
    rows_tile_of(g, ti, tj);
    const int64_t m0 = ti * C::BM, n0 = tj * C::BN;
    const int S = (int)((g.K + C::BK - 1) / C::BK);
    // the tile's bias values go to LDS once (first version: sixteen dependent f32x4 loads from L2 per lane in the epilogue, each behind
    // its own vmcnt(0) -- 11.6 us of the 32 us a workgroup lived)
    float bias_v = 0.0f;
    if (threadIdx.x < C::BN) bias_v = g.bias[n0 + threadIdx.x];

    // ---- the loader: 1 KiB pieces (RPP rows x ROWB bytes) of the operand images; lane -> row lane / CH of the piece, position lane % CH
    const char* xsrc[C::XLOADS];
    const char* wsrc[C::WLOADS];
    bool xrow_ok[C::XLOADS];
    int xchunk[C::XLOADS];                                        // the logical 16-byte chunk (8 elements) this lane fetches of its row
#pragma unroll
    for (int j = 0; j < C::XLOADS; ++j) {
        const int r = (C::XLOADS * wave + j) * C::RPP + lane / C::CH;
        const int c = (lane % C::CH) ^ C::swz(r);
        xchunk[j] = c;
        xrow_ok[j] = m0 + r < g.M;
        xsrc[j] = reinterpret_cast<const char*>(g.X) + ((m0 + r) * g.ldx + 8 * c) * 2;
    }
#pragma unroll
    for (int j = 0; j < C::WLOADS; ++j) {
        const int r = (C::WLOADS * wave + j) * C::RPP + lane / C::CH;
        const int c = (lane % C::CH) ^ C::swz(r);
        wsrc[j] = reinterpret_cast<const char*>(g.W) + ((n0 + r) * g.ldw + 8 * c) * 2;
    }
    const char* zero = reinterpret_cast<const char*>(rg_zero_page);
    const int64_t k_chunks = (g.K + 7) / 8;                       // 16-byte chunks of a row that hold valid elements
    const uint32_t xpiece = (uint32_t)(C::XLOADS * wave) * 1024u, wpiece = C::XIMG + (uint32_t)(C::WLOADS * wave) * 1024u;
    uint32_t load_slot = 0;
    auto issue = [&](int s) {
        const uint32_t slot = load_slot * C::SLOT;
#pragma unroll
        for (int j = 0; j < C::XLOADS; ++j) {
            const bool ok = xrow_ok[j] && ((int64_t)C::CH * s + xchunk[j] < k_chunks);
            glds16(ok ? xsrc[j] + (size_t)s * ROWB : zero, slot + xpiece + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < C::WLOADS; ++j) glds16(wsrc[j] + (size_t)s * ROWB, slot + wpiece + j * 1024);
        load_slot = load_slot + 1 == C::SLOTS ? 0u : load_slot + 1;
    };

    // ---- fragment reads: lane l holds row (l & 31) of a 32-row block, elements 8 (l >> 5) .. +7 of 16-step ks = chunk 2 ks + (l >> 5)
    const int row = lane & 31, kh = lane >> 5;
    uint32_t fo[C::KSTEPS];
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks) fo[ks] = (uint32_t)row * ROWB + (uint32_t)(((2 * ks + kh) ^ C::swz(row)) * 16);
    const uint32_t x_frag = (uint32_t)(32 * MB * wm) * ROWB, w_frag = C::XIMG + (uint32_t)(32 * NB * wn) * ROWB;
    // feature blocks of this wave that hold a feature of the layer (heads: N = 1 .. 11 of a tile)
    const int64_t n_left = g.N - n0 - 32 * NB * wn;
    const int nb_live = n_left <= 0 ? 0 : (n_left >= 32 * NB ? NB : (int)((n_left + 31) / 32));

    f32x16 acc[NB][MB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.0f;

#pragma unroll
    for (int s = 0; s < C::SLOTS - 1; ++s)
        if (s < S) issue(s);
    if (threadIdx.x < C::BN) reinterpret_cast<float*>(rg_smem + C::BIAS_OFF)[threadIdx.x] = bias_v;      // (visible after the first barrier below)
    uint32_t slot = 0;
    for (int s = 0; s < S; ++s) {
        // stages <= s + SLOTS - 2 are issued: stage s must have landed -- my pieces here, everybody's after the barrier
        const int ahead = (s + C::SLOTS - 2 < S ? s + C::SLOTS - 2 : S - 1) - s;
        if (ahead >= 2) wait_vm<2 * C::LOADS>();
        else if (ahead == 1) wait_vm<C::LOADS>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + C::SLOTS - 1 < S) issue(s + C::SLOTS - 1);        // into the slot of stage s - 1: every wave finished reading it before this barrier
        const uint32_t so = slot * C::SLOT;
        if constexpr (FULL) {
#pragma unroll
            for (int k2 = 0; k2 < C::KSTEPS; k2 += 2)
                rows_two_steps<C>(acc, so + fo[k2] + x_frag, so + fo[k2] + w_frag, so + fo[k2 + 1] + x_frag, so + fo[k2 + 1] + w_frag);
        } else if (nb_live) {
#pragma unroll
            for (int ks = 0; ks < C::KSTEPS; ++ks) {
                bf16x8 xf[MB], wf[NB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) xf[mb] = *reinterpret_cast<const bf16x8*>(rg_smem + so + fo[ks] + x_frag + mb * 32 * ROWB);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    if (nb < nb_live) wf[nb] = *reinterpret_cast<const bf16x8*>(rg_smem + so + fo[ks] + w_frag + nb * 32 * ROWB);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    if (nb < nb_live) {
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[nb][mb], 0, 0, 0);
                    }
            }
        }
        slot = slot + 1 == C::SLOTS ? 0u : slot + 1;
    }

    // ---- epilogue.  Accumulator register r of lane l in block (nb, mb): feature 32 nb + (r & 3) + 8 (r >> 2) + 4 (l >> 5), sample 32 mb + (l & 31)
    if constexpr (OUT_BF16) {
        __builtin_amdgcn_s_barrier();                             // the ring is idle: it becomes the transposed image [BM samples][BN features] bf16
        asm volatile("" ::: "memory");
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (nb >= nb_live) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = 32 * NB * wn + 32 * nb + 8 * q + 4 * kh;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(rg_smem + C::BIAS_OFF + nl * 4);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
                    const int ml = 32 * MB * wm + 32 * mb + row;
                    u32x2 o;
                    o[0] = pack_bf16(activate(acc[nb][mb][4 * q + 0] + bv[0], g.act), activate(acc[nb][mb][4 * q + 1] + bv[1], g.act));
                    o[1] = pack_bf16(activate(acc[nb][mb][4 * q + 2] + bv[2], g.act), activate(acc[nb][mb][4 * q + 3] + bv[3], g.act));
                    *reinterpret_cast<u32x2*>(rg_smem + ml * C::CT_LD + nl * 2) = o;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the image as 8-byte units (four features), LPR = BN / 4 per sample row: thread t stores units t, t + THREADS, ... -- a wave
        // instruction writes 512 contiguous bytes of a row (or several complete rows)
        constexpr int LPR = C::BN / 4;
        uint16_t* Cb = reinterpret_cast<uint16_t*>(g.C);
#pragma unroll 8
        for (int i = 0; i < C::BM * LPR / C::THREADS; ++i) {
            typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
            const int u = (int)threadIdx.x + i * C::THREADS;
            const int ml = u / LPR, c4 = u % LPR;
            const int64_t n = n0 + 4 * c4;
            if (m0 + ml < g.M && n < g.N) {                       // (N % 4 == 0: a unit's four features are inside or outside together)
                const u32x2 v = *reinterpret_cast<const u32x2*>(rg_smem + ml * C::CT_LD + c4 * 8);
                *reinterpret_cast<u32x2*>(Cb + (m0 + ml) * g.ldc + n) = v;
            }
        }
    } else {
        float* Cf = reinterpret_cast<float*>(g.C);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (nb >= nb_live) continue;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int64_t m = m0 + 32 * MB * wm + 32 * mb + row;
                if (m >= g.M) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nl = 32 * NB * wn + 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (n0 + nl < g.N) Cf[m * g.ldc + n0 + nl] = activate(acc[nb][mb][r] + reinterpret_cast<const float*>(rg_smem + C::BIAS_OFF)[nl], g.act);
                }
            }
        }
    }
}

template <class C>
int rows_launch(RowsGemmArgs g, int out_bf16, hipStream_t st) {
    g.tiles_m = (int)((g.M + C::BM - 1) / C::BM);
    g.tiles_n = (int)((g.N + C::BN - 1) / C::BN);
    const dim3 grid((unsigned)((int64_t)g.tiles_m * g.tiles_n));
    const bool full = (g.N % C::BN) == 0;
    const void* fn = out_bf16 ? (full ? reinterpret_cast<const void*>(&rows_gemm_kernel<C, true, true>) : reinterpret_cast<const void*>(&rows_gemm_kernel<C, true, false>))
                              : (full ? reinterpret_cast<const void*>(&rows_gemm_kernel<C, false, true>) : reinterpret_cast<const void*>(&rows_gemm_kernel<C, false, false>));
    if (int e = nerf_host::allow_dynamic_lds(fn, C::LDS)) return e;
    if (out_bf16) {
        if (full) hipLaunchKernelGGL((rows_gemm_kernel<C, true, true>), grid, dim3(C::THREADS), C::LDS, st, g);
        else hipLaunchKernelGGL((rows_gemm_kernel<C, true, false>), grid, dim3(C::THREADS), C::LDS, st, g);
    } else {
        if (full) hipLaunchKernelGGL((rows_gemm_kernel<C, false, true>), grid, dim3(C::THREADS), C::LDS, st, g);
        else hipLaunchKernelGGL((rows_gemm_kernel<C, false, false>), grid, dim3(C::THREADS), C::LDS, st, g);
    }
    return (int)hipGetLastError();
}

// dst[m, c0 + j] = bf16(src[m, j]) (RNE) for j < cols, 0 for cols <= j < fill: fp32 rows (encodings, packed weights, an element-wise stage's
// output) into a column range of bf16 rows, with the zero padding the product's last chunk and tile rows rely on
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ src, int64_t rows_src, int64_t lds, int64_t rows, int cols, int fill,
                                                           uint16_t* __restrict__ dst, int64_t ldd) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * fill) return;
    const int64_t m = idx / fill;
    const int j = (int)(idx - m * fill);
    const float v = (m < rows_src && j < cols) ? src[m * lds + j] : 0.0f;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const uint32_t p = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v, 0.0f}, bf16x2));
    dst[m * ldd + j] = (uint16_t)(p & 0xffffu);
}

}  // namespace

int rg_rows_gemm(int64_t M, int64_t N, int64_t K, const void* X, int64_t ldx, const void* W, int64_t ldw, int64_t n_pad, const float* bias, int act, void* C,
                 int64_t ldc, int out_bf16, hipStream_t st) {
    if (M == 0 || N == 0) return 0;
    if (K < 1 || (ldx & 7) || (reinterpret_cast<uintptr_t>(X) & 15u) || (ldw & 63) || ldw < K || (reinterpret_cast<uintptr_t>(W) & 15u) || (n_pad & 255) || n_pad < N ||
        (reinterpret_cast<uintptr_t>(bias) & 15u))
        return (int)hipErrorInvalidValue;
    if (out_bf16 && ((N & 3) || (ldc & 3) || (reinterpret_cast<uintptr_t>(C) & 7u))) return (int)hipErrorInvalidValue;
    RowsGemmArgs g{M, N, K, reinterpret_cast<const uint16_t*>(X), ldx, reinterpret_cast<const uint16_t*>(W), ldw, bias, C, ldc, act, 0, 0};
    // Two tile configurations (profiles/r05_rows_gemm_tile_config_ab.log, five candidates on one box): wide layers whose feature count fills
    // 256-wide tiles run 256 x 256 tiles with full-cache-line stages (8 wavefronts, one workgroup per CU: fewest operand bytes per product,
    // +5-8 % at 512 / 1024); everything else -- ragged feature tiles, the heads, the 256-wide layers of Ref-NeRF -- 128 x 256 tiles with
    // 64-byte row pieces and two workgroups per CU.  (128 x 128 x 64 in 2 or 3 slots and 128 x 256 x 64 with 8 wavefronts measured 4-25 % slower; whole-row tiles 128 x 512 x 32 -- the
    // activations read once, the output written in whole rows -- 5-8 % slower than 256 x 256 x 64 at 512 / 1024; hipBLASLt's own shape for these sizes, 256 x 256 x 32 on FOUR wavefronts of
    // 128 x 128 (accumulators in 256 AGPRs), 25-40 % slower with this file's per-stage schedule.)
    if (N >= 512 && N % 256 == 0) return rows_launch<RowsCfg<256, 256, 64, 2, 4, 2, 1>>(g, out_bf16, st);
    return rows_launch<RowsCfg<128, 256, 32, 3, 2, 2, 2>>(g, out_bf16, st);
}

int rg_rows_to_bf16(const float* src, int64_t rows_src, int64_t lds, int64_t rows, int cols, int fill, void* dst, int64_t ldd, hipStream_t st) {
    if (rows * fill == 0) return 0;
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)((rows * fill + 255) / 256)), dim3(256), 0, st, src, rows_src, lds, rows, cols, fill,
                       reinterpret_cast<uint16_t*>(dst), ldd);
    return (int)hipGetLastError();
}
