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
// Tile: 128 samples x 256 features per 256-thread workgroup (4 wavefronts as 2 x 2, 64 samples x 128 features each = 2 x 4 MFMA blocks of
// 32 x 32, v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand, like the fused kernels: an accumulator register then holds four
// consecutive features of one sample), TWO workgroups per CU: a layer's tiles all take the same time, so with one workgroup per CU the
// whole chip loads, multiplies and stores in phases (first version, 256 x 256 tiles of 8 wavefronts: 15 us of every 34 us tile were its
// unoverlapped prologue / epilogue, 504 TFLOP/s at 262 144 x 512 x 512); two independent workgroups drift apart and one's epilogue
// runs under the other's products.  Contraction in stages of 32 (64-byte row pieces): a 3-slot ring of [128 x 32 activations | 256 x 32
// weights] images (3 x 24 KiB) filled two stages ahead by global_load_lds_dwordx4 -- no staging registers, no ds_write -- ONE s_barrier
// per stage.  The LDS image of a 1 KiB piece is lane-linear (16 rows x 64 B); the bank swizzle is a permutation of the SOURCE chunk a
// lane fetches: row r keeps its 16-byte chunk c at position c ^ ((r >> 2) & 3), which puts the 16 lanes of every ds_read_b128 service
// group on 16 different 16-byte slots of the 256-byte bank row (MI355X_MICROARCH.md, LDS).  Out-of-range rows / chunks are fetched from a
// 16-byte page of zeros; a partly valid last chunk multiplies the row's own padding (finite by contract) with packed zero weights.
// Epilogue: bias + activation in registers, bf16, transposed through the (then idle) ring into full 512-byte row pieces.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "host_common.h"

namespace {

constexpr int RBM = 128, RBN = 256, RBK = 32, RSLOTS = 3, RTHREADS = 256;
constexpr int R_XIMG = RBM * RBK * 2;                // 8 KiB: the activations' image of a stage
constexpr int R_WIMG = RBN * RBK * 2;                // 16 KiB: the weights'
constexpr int R_SLOT = R_XIMG + R_WIMG;              // [activations | weights]
constexpr int R_CT_LD = 520;                         // bytes per sample row of the transposed output image (520 / 4 = 130: ds_write_b64 of 16 rows hit 32 banks once)
constexpr int R_LDS = RBM * R_CT_LD > RSLOTS * R_SLOT ? RBM * R_CT_LD : RSLOTS * R_SLOT;       // 73 728 B: two workgroups per CU
constexpr int R_XLOADS = R_XIMG / (RTHREADS * 16), R_WLOADS = R_WIMG / (RTHREADS * 16);         // global_load_lds per thread and stage: 2 + 4
constexpr int R_LOADS = R_XLOADS + R_WLOADS;

extern __shared__ __attribute__((aligned(16))) char rg_smem[];
__device__ __attribute__((aligned(16))) uint32_t rg_zero_page[4] = {0u, 0u, 0u, 0u};

struct RowsGemmArgs {
    int64_t M, N, K;                // N, K: the layer's own sizes (the packed weights are padded)
    const uint16_t* X; int64_t ldx; // elements; multiple of 8, base 16-byte aligned
    const uint16_t* W; int64_t ldw; // packed weights (Npad, Kpad): Npad % 256 == 0, Kpad = ldw % 32 == 0, zeros outside (N, K)
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

// one stage (two 16-steps) for this wave's 2 x 4 blocks, all four feature blocks live: the twelve fragment reads are issued together (the
// compiler's own schedule re-used four fragment registers and waited for LDS four times per stage), the products follow as they land
DEVINL void rows_stage_full(f32x16 (&acc)[4][2], uint32_t x0, uint32_t w0, uint32_t x1, uint32_t w1) {
    bf16x8 xf[2][2], wf[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) xf[0][mb] = *reinterpret_cast<const bf16x8*>(rg_smem + x0 + mb * 2048);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) wf[0][nb] = *reinterpret_cast<const bf16x8*>(rg_smem + w0 + nb * 2048);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) xf[1][mb] = *reinterpret_cast<const bf16x8*>(rg_smem + x1 + mb * 2048);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) wf[1][nb] = *reinterpret_cast<const bf16x8*>(rg_smem + w1 + nb * 2048);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][nb], xf[ks][mb], acc[nb][mb], 0, 0, 0);
}

// FULL: every wave of every tile holds four live feature blocks (N % 256 == 0: the hidden layers at widths 256 / 512 / ...) -- no guards
// in the contraction loop; otherwise (a ragged last feature tile, the heads) the blocks beyond N are skipped block by block.
template <bool OUT_BF16, bool FULL>
__global__ __launch_bounds__(RTHREADS, 2) void rows_gemm_kernel(RowsGemmArgs g) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave >> 1, wn = wave & 1;                      // this wave: samples 64 wm .. +63, features 128 wn .. +127 of the tile
    int64_t ti, tj;
    rows_tile_of(g, ti, tj);
    const int64_t m0 = ti * RBM, n0 = tj * RBN;
    const int S = (int)((g.K + RBK - 1) / RBK);

    // ---- the loader: 1 KiB pieces (16 rows x 64 B) of the operand images; this wave's pieces: activations rows 32 wave + 16 j + (lane >> 2),
    //      j < 2; weights rows 64 wave + 16 j + (lane >> 2), j < 4
    const char* xsrc[R_XLOADS];
    const char* wsrc[R_WLOADS];
    bool xrow_ok[R_XLOADS];
    int xchunk[R_XLOADS];                                         // the logical 16-byte chunk (8 elements) this lane fetches of its row
#pragma unroll
    for (int j = 0; j < R_XLOADS; ++j) {
        const int r = 16 * R_XLOADS * wave + 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        xchunk[j] = c;
        xrow_ok[j] = m0 + r < g.M;
        xsrc[j] = reinterpret_cast<const char*>(g.X) + ((m0 + r) * g.ldx + 8 * c) * 2;
    }
#pragma unroll
    for (int j = 0; j < R_WLOADS; ++j) {
        const int r = 16 * R_WLOADS * wave + 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        wsrc[j] = reinterpret_cast<const char*>(g.W) + ((n0 + r) * g.ldw + 8 * c) * 2;
    }
    const char* zero = reinterpret_cast<const char*>(rg_zero_page);
    const int64_t k_chunks = (g.K + 7) / 8;                       // 16-byte chunks of a row that hold valid elements
    const uint32_t xpiece = (uint32_t)(R_XLOADS * wave) * 1024u, wpiece = R_XIMG + (uint32_t)(R_WLOADS * wave) * 1024u;
    uint32_t load_slot = 0;
    auto issue = [&](int s) {
        const uint32_t slot = load_slot * R_SLOT;
#pragma unroll
        for (int j = 0; j < R_XLOADS; ++j) {
            const bool ok = xrow_ok[j] && (4 * (int64_t)s + xchunk[j] < k_chunks);
            glds16(ok ? xsrc[j] + (size_t)s * (RBK * 2) : zero, slot + xpiece + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < R_WLOADS; ++j) glds16(wsrc[j] + (size_t)s * (RBK * 2), slot + wpiece + j * 1024);
        load_slot = load_slot + 1 == RSLOTS ? 0u : load_slot + 1;
    };

    // ---- fragment reads: lane l holds row (l & 31) of a 32-row block, elements 8 (l >> 5) .. +7 of a 16-step = chunk 2 ks + (l >> 5)
    const int row = lane & 31, kh = lane >> 5;
    const int sw = (row >> 2) & 3;
    const uint32_t f0 = (uint32_t)row * 64u + (uint32_t)((kh ^ sw) * 16), f1 = (uint32_t)row * 64u + (uint32_t)(((kh ^ sw) ^ 2) * 16);
    const uint32_t x_frag = (uint32_t)(64 * wm) * 64u, w_frag = R_XIMG + (uint32_t)(128 * wn) * 64u;
    // feature blocks of this wave that hold a feature of the layer (heads: N = 1 .. 11 of a 256-wide tile)
    const int64_t n_left = g.N - n0 - 128 * wn;
    const int nb_live = n_left <= 0 ? 0 : (n_left >= 128 ? 4 : (int)((n_left + 31) / 32));

    f32x16 acc[4][2];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.0f;

#pragma unroll
    for (int s = 0; s < RSLOTS - 1; ++s)
        if (s < S) issue(s);
    uint32_t slot = 0;
    for (int s = 0; s < S; ++s) {
        // stages <= s + 1 are issued: stage s must have landed -- my pieces here, everybody's after the barrier
        if (s + 1 < S) wait_vm<R_LOADS>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + RSLOTS - 1 < S) issue(s + RSLOTS - 1);            // into the slot of stage s - 1: every wave finished reading it before this barrier
        const uint32_t so = slot * R_SLOT;
        if constexpr (FULL) {
            rows_stage_full(acc, so + f0 + x_frag, so + f0 + w_frag, so + f1 + x_frag, so + f1 + w_frag);
        } else if (nb_live) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t fo = so + (ks ? f1 : f0);
                bf16x8 xf[2], wf[4];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) xf[mb] = *reinterpret_cast<const bf16x8*>(rg_smem + fo + x_frag + mb * 2048);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    if (nb < nb_live) wf[nb] = *reinterpret_cast<const bf16x8*>(rg_smem + fo + w_frag + nb * 2048);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    if (nb < nb_live) {
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[nb][mb], 0, 0, 0);
                    }
            }
        }
        slot = slot + 1 == RSLOTS ? 0u : slot + 1;
    }

    // ---- epilogue.  Accumulator register r of lane l in block (nb, mb): feature 32 nb + (r & 3) + 8 (r >> 2) + 4 (l >> 5), sample 32 mb + (l & 31)
    if constexpr (OUT_BF16) {
        __builtin_amdgcn_s_barrier();                             // the ring is idle: it becomes the transposed image [128 samples][256 features] bf16
        asm volatile("" ::: "memory");
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb >= nb_live) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = 128 * wn + 32 * nb + 8 * q + 4 * kh;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n0 + nl);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
                    const int ml = 64 * wm + 32 * mb + row;
                    u32x2 o;
                    o[0] = pack_bf16(activate(acc[nb][mb][4 * q + 0] + bv[0], g.act), activate(acc[nb][mb][4 * q + 1] + bv[1], g.act));
                    o[1] = pack_bf16(activate(acc[nb][mb][4 * q + 2] + bv[2], g.act), activate(acc[nb][mb][4 * q + 3] + bv[3], g.act));
                    *reinterpret_cast<u32x2*>(rg_smem + ml * R_CT_LD + nl * 2) = o;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // wave w: sample rows 32 w .. +31, lane = features 4 lane .. +3 (8 bytes): one 512-byte row piece per instruction
        const int64_t n = n0 + 4 * lane;
        if (n < g.N) {                                            // (N % 4 == 0: a lane's four features are inside or outside together)
            uint16_t* Cb = reinterpret_cast<uint16_t*>(g.C);
#pragma unroll 8
            for (int i = 0; i < RBM / 4; ++i) {
                typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
                const int ml = (RBM / 4) * wave + i;
                if (m0 + ml >= g.M) break;
                const u32x2 v = *reinterpret_cast<const u32x2*>(rg_smem + ml * R_CT_LD + lane * 8);
                *reinterpret_cast<u32x2*>(Cb + (m0 + ml) * g.ldc + n) = v;
            }
        }
    } else {
        float* Cf = reinterpret_cast<float*>(g.C);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (nb >= nb_live) continue;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int64_t m = m0 + 64 * wm + 32 * mb + row;
                if (m >= g.M) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t n = n0 + 128 * wn + 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (n < g.N) Cf[m * g.ldc + n] = activate(acc[nb][mb][r] + g.bias[n], g.act);
                }
            }
        }
    }
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
    const int64_t tiles_n = (N + RBN - 1) / RBN;
    if (K < 1 || (ldx & 7) || (reinterpret_cast<uintptr_t>(X) & 15u) || (ldw % RBK) || ldw < (K + RBK - 1) / RBK * RBK || (reinterpret_cast<uintptr_t>(W) & 15u) ||
        n_pad < tiles_n * RBN || (reinterpret_cast<uintptr_t>(bias) & 15u))
        return (int)hipErrorInvalidValue;
    if (out_bf16 && ((N & 3) || (ldc & 3) || (reinterpret_cast<uintptr_t>(C) & 7u))) return (int)hipErrorInvalidValue;
    RowsGemmArgs g{M, N, K, reinterpret_cast<const uint16_t*>(X), ldx, reinterpret_cast<const uint16_t*>(W), ldw, bias, C, ldc, act, (int)((M + RBM - 1) / RBM), (int)tiles_n};
    const dim3 grid((unsigned)((int64_t)g.tiles_m * g.tiles_n));
    const bool full = (N % RBN) == 0;
    const void* fn = out_bf16 ? (full ? reinterpret_cast<const void*>(&rows_gemm_kernel<true, true>) : reinterpret_cast<const void*>(&rows_gemm_kernel<true, false>))
                              : (full ? reinterpret_cast<const void*>(&rows_gemm_kernel<false, true>) : reinterpret_cast<const void*>(&rows_gemm_kernel<false, false>));
    if (int e = nerf_host::allow_dynamic_lds(fn, R_LDS)) return e;
    if (out_bf16) {
        if (full) hipLaunchKernelGGL((rows_gemm_kernel<true, true>), grid, dim3(RTHREADS), R_LDS, st, g);
        else hipLaunchKernelGGL((rows_gemm_kernel<true, false>), grid, dim3(RTHREADS), R_LDS, st, g);
    } else {
        if (full) hipLaunchKernelGGL((rows_gemm_kernel<false, true>), grid, dim3(RTHREADS), R_LDS, st, g);
        else hipLaunchKernelGGL((rows_gemm_kernel<false, false>), grid, dim3(RTHREADS), R_LDS, st, g);
    }
    return (int)hipGetLastError();
}

int rg_rows_to_bf16(const float* src, int64_t rows_src, int64_t lds, int64_t rows, int cols, int fill, void* dst, int64_t ldd, hipStream_t st) {
    if (rows * fill == 0) return 0;
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)((rows * fill + 255) / 256)), dim3(256), 0, st, src, rows_src, lds, rows, cols, fill,
                       reinterpret_cast<uint16_t*>(dst), ldd);
    return (int)hipGetLastError();
}
