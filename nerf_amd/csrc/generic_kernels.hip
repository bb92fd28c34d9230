// Generic-shape layer products (round 4): the path of networks LARGER than the shapes the fused MLP kernels are compiled for
// (`--nerf_net_width` / `--prop_net_width` above 256, procedures.py:176-177; more than 10 encoding octaves, mip_model.py:15-18).
// Such a network is evaluated layer by layer with its activations as fp32 row-major matrices in HBM (the host side chunks the rays the
// way the reference's render_image tiles them, procedures.py:53-56, so a chunk's activations stay small) -- nn.Linear forward
// (nerf_helper.py:28-36 makeMLP), and what torch.autograd computes for it backward, as ONE hand-written MFMA GEMM with explicit strides:
//
//     C[i, j] = act( sum_p A(i, p) * B(p, j) + bias[j] ) * [mask(i, j) > 0]
//
//   forward            y  = act(x W^T + b):  A = x (M, K),  B(p, j) = W[j, p]                      (N, T)
//   input gradient     dx = (dy W) . [x > 0]:  A = dy (M, N),  B(p, j) = W[p, j],  mask = x        (N, N)
//   weight gradient    dW = dy^T x:  A(i, p) = dy[p, i],  B(p, j) = x[p, j],  contraction over the samples   (T, N)
//     -- split over workgroups, partial sums added in a fixed order (deterministic, no atomics); db = dy^T 1 is the same call with a
//        column of ones.
//
// Tile: 128 x 128 outputs per 256-thread workgroup (four wavefronts, 2 x 2 MFMA blocks of 32 x 32 each), 32 contraction elements per
// LDS stage; operands are staged through LDS as [output index][contraction index] whatever their memory order (the loader walks the
// unit-stride direction), fp32 mode = v_mfma_f32_32x32x2_f32 on the fp32 values (the parity mode, like the fused kernels'), bf16 mode =
// operands rounded to bf16 (RNE) on the way into LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulation.  This is the COMPATIBILITY path of the
// shape arguments: every compiled shape keeps its fused kernel (activations in registers, no HBM round trip per layer).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "host_common.h"

namespace {

// contraction elements per LDS stage: 32.  A 64-wide stage (two 32-wide sub-tiles, twice the loads in flight) was measured and retired
// (profiles/r04_generic_gemm_gbk64_relayout_ab.log): bf16 forward / input gradient unchanged (208 vs 210, 113 vs 113 TFLOP/s), weight gradient
// +3 %, the fp32 parity mode 12-20 % SLOWER (one workgroup less per CU: 66 KiB of LDS) -> not the stage length that bounds this kernel.
constexpr int GBM = 128, GBN = 128, GBK = 32;
static_assert(GBK == 32 || GBK == 64, "GBK");

struct GemmArgs {
    int64_t M, N, P;
    const float* A; int64_t a_si, a_sp;
    const float* B; int64_t b_sp, b_sj;
    float* C; int64_t ldc;
    const float* bias; int act;
    const float* mask; int64_t ldm;
    int64_t p_chunk;                        // contraction elements per slice (>= P: no split)
    int64_t part_stride;                    // split: slice z writes C + z * part_stride (dense ld = ldc), no bias / act / mask
    int tiles_m, tiles_n, slices;           // the 1-D grid = tiles_m * tiles_n * slices workgroups (tile_of())
};

// Workgroup -> (output tile, contraction slice), XCD-aware.  The dispatcher places block b on XCD b % 8 and every XCD has its own L2, so
// with the plain 3-D grid the workgroups that read the SAME operand tile -- the N tiles of one 128-row block of activations (forward /
// input gradient: 256 KiB of fp32 rows each), all output tiles of one contraction slice (weight gradient) -- sat on different XCDs and each
// fetched it through its own L2 (same box, alternated, profiles/r04_generic_gemm_xcd_ab.log: forward 182 -> 210 TFLOP/s bf16 at 262 144 x 512
// x 512, input gradient 100 -> 106, weight gradient 212 -> 220; render_image at width 512 +10 %).  Here every XCD gets a CONTIGUOUS range of logical
// ids (bijective for any grid size: the first nwg % 8 XCDs take one more) and the logical order keeps the sharers adjacent: [slice][M tile]
// [N tile] when the rows are the long dimension, [slice][N tile][M tile] otherwise.
struct TileId { int64_t ti, tj, tz; };
DEVINL TileId tile_of(const GemmArgs& g) {
    const int64_t nwg = gridDim.x, b = blockIdx.x;
    const int64_t q = nwg / 8, r = nwg % 8, xcd = b % 8;
    const int64_t per_slice = (int64_t)g.tiles_m * g.tiles_n;
    TileId t;
    const int64_t id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    t.tz = id / per_slice;
    const int64_t rem = id - t.tz * per_slice;
    if (g.tiles_m >= g.tiles_n) { t.ti = rem / g.tiles_n; t.tj = rem - t.ti * g.tiles_n; }
    else { t.tj = rem / g.tiles_m; t.ti = rem - t.tj * g.tiles_m; }
    return t;
}

template <bool BF16> struct Stage;
template <> struct Stage<true> {
    typedef unsigned short elem;
    static constexpr int LD = GBK + 8;      // 80-byte rows: 16-byte aligned fragment reads, 8-byte aligned staging stores
    // four fp32 -> four bf16 (RNE, v_cvt_pk_bf16_f32) as two dwords
    static DEVINL void store4(elem* dst, const f32x4& v) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
        const bf16x2 lo = __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2), hi = __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2);
        *reinterpret_cast<u32x2*>(dst) = u32x2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
    }
    static DEVINL void store2(elem* dst, float a, float b) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        *reinterpret_cast<uint32_t*>(dst) = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
    }
};
template <> struct Stage<false> {
    typedef float elem;
    static constexpr int LD = GBK + 1;      // odd stride: conflict-free column walks (so the staging stores are scalar)
    static DEVINL void store4(elem* dst, const f32x4& v) { dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; }
    static DEVINL void store2(elem* dst, float a, float b) { dst[0] = a; dst[1] = b; }
};

// four consecutive elements along an operand's unit-stride direction, `n_valid` (0..4) of them inside the matrix; one 16-byte load when aligned
DEVINL f32x4 load4(const float* __restrict__ p, bool vec_ok, int64_t n_valid) {
    if (vec_ok && n_valid >= 4) return *reinterpret_cast<const f32x4*>(p);
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (n_valid > 0) v[0] = p[0];
    if (n_valid > 1) v[1] = p[1];
    if (n_valid > 2) v[2] = p[2];
    if (n_valid > 3) v[3] = p[3];
    return v;
}

// One operand tile -- rows = output index (0..127), columns = contraction index (0..GBK-1, as 32-wide sub-tiles) -- in two steps so that the global loads of the NEXT
// tile are in flight while the matrix cores work on the current one: fetch() into 16 registers per thread, stage() into LDS.
//   contraction index contiguous in memory (s_p == 1):  thread t -> rows (t >> 3) + 32 r, columns 4 (t & 7) .. +3        (r = 0..3)
//   output index contiguous (s_out == 1):               thread t -> rows 8 (t & 15) .. +7, columns 2 (t >> 4), 2 (t >> 4) + 1
struct OperandTile {
    static constexpr int NH = GBK / 32;     // 32-wide sub-tiles of a stage
    f32x4 v[NH][4];
    bool p_contig;
    DEVINL void fetch(const float* __restrict__ src, int64_t s_out, int64_t s_p, bool vec_ok, int64_t out0, int64_t n_out, int64_t p0, int64_t p_end) {
        const int t = threadIdx.x;
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            const int64_t ps = p0 + 32 * s;
            if (p_contig) {
                const int64_t pp = ps + 4 * (t & 7);
                const int64_t np = p_end - pp;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t oo = out0 + (t >> 3) + 32 * r;
                    v[s][r] = load4(src + oo * s_out + pp, vec_ok, oo < n_out ? np : 0);
                }
            } else {
                const int64_t oo = out0 + 8 * (t & 15);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int64_t pp = ps + 2 * (t >> 4) + q;
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[s][2 * q + h] = load4(src + (oo + 4 * h) + pp * s_p, vec_ok, pp < p_end ? n_out - (oo + 4 * h) : 0);
                }
            }
        }
    }
    template <bool BF16>
    DEVINL void stage(typename Stage<BF16>::elem* dst) const {
        constexpr int LD = Stage<BF16>::LD;
        const int t = threadIdx.x;
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            if (p_contig) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Stage<BF16>::store4(dst + ((t >> 3) + 32 * r) * LD + 32 * s + 4 * (t & 7), v[s][r]);
            } else {
                const int o = 8 * (t & 15), p = 32 * s + 2 * (t >> 4);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Stage<BF16>::store2(dst + (o + 4 * h + e) * LD + p, v[s][h][e], v[s][2 + h][e]);
            }
        }
    }
};

DEVINL bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <bool BF16>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    typedef typename Stage<BF16>::elem elem;
    constexpr int LD = Stage<BF16>::LD;
    __shared__ __attribute__((aligned(16))) elem As[GBM * LD];
    __shared__ __attribute__((aligned(16))) elem Bs[GBN * LD];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;                  // this wave's 64 x 64 quadrant
    const TileId tid = tile_of(g);
    const int64_t i0 = tid.ti * GBM, j0 = tid.tj * GBN;
    const int64_t p_begin = tid.tz * g.p_chunk;
    const int64_t p_end = (p_begin + g.p_chunk < g.P) ? p_begin + g.p_chunk : g.P;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const int row = lane & 31, kh = lane >> 5;
    OperandTile ta, tb;
    ta.p_contig = g.a_sp == 1;
    tb.p_contig = g.b_sp == 1;
    // 16-byte loads: the stride of the non-contiguous index keeps every group of four aligned (tile origins are multiples of 8 / 32)
    const bool va = aligned16(g.A) && (((ta.p_contig ? g.a_si : g.a_sp) & 3) == 0);
    const bool vb = aligned16(g.B) && (((tb.p_contig ? g.b_sj : g.b_sp) & 3) == 0);
    if (p_begin < p_end) {
        ta.fetch(g.A, g.a_si, g.a_sp, va, i0, g.M, p_begin, p_end);
        tb.fetch(g.B, g.b_sj, g.b_sp, vb, j0, g.N, p_begin, p_end);
    }
    for (int64_t p0 = p_begin; p0 < p_end; p0 += GBK) {
        ta.template stage<BF16>(As);
        tb.template stage<BF16>(Bs);
        __syncthreads();
        if (p0 + GBK < p_end) {                               // the next tile's loads fly during this tile's MFMAs
            ta.fetch(g.A, g.a_si, g.a_sp, va, i0, g.M, p0 + GBK, p_end);
            tb.fetch(g.B, g.b_sj, g.b_sp, vb, j0, g.N, p0 + GBK, p_end);
        }
        if constexpr (BF16) {
#pragma unroll
            for (int kk = 0; kk < GBK; kk += 16) {
                bf16x8 af[2], bf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const bf16x8*>(&As[(wi * 64 + a * 32 + row) * LD + kk + kh * 8]);
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const bf16x8*>(&Bs[(wj * 64 + b * 32 + row) * LD + kk + kh * 8]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        } else {
#pragma unroll 4
            for (int kk = 0; kk < GBK; kk += 2) {
                float af[2], bf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a] = As[(wi * 64 + a * 32 + row) * LD + kk + kh];
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b] = Bs[(wj * 64 + b * 32 + row) * LD + kk + kh];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // accumulator register r of lane l: output row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 of the 32 x 32 block
    const bool split = g.part_stride != 0;
    float* C = g.C + (split ? tid.tz * g.part_stride : 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t j = j0 + wj * 64 + b * 32 + row;
            if (j >= g.N) continue;
            const float bj = (!split && g.bias) ? g.bias[j] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t i = i0 + wi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (i >= g.M) continue;
                float v = acc[a][b][r];
                if (!split) {
                    v += bj;
                    if (g.act == 1) v = v > 0.0f ? v : 0.0f;
                    else if (g.act == 2) v = 1.0f / (1.0f + expf(-v));
                    if (g.mask && !(g.mask[i * g.ldm + j] > 0.0f)) v = 0.0f;
                }
                C[i * g.ldc + j] = v;
            }
        }
}

// C[i, j] = sum_z part[z][i][j] in ascending z (fixed order), then the epilogue the split launch skipped
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ part, int64_t part_stride, int S, int64_t M, int64_t N, float* __restrict__ C,
                                                          int64_t ldc, const float* __restrict__ bias, int act, const float* __restrict__ mask, int64_t ldm) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * N) return;
    const int64_t i = idx / N, j = idx - i * N;
    float v = 0.0f;
    for (int z = 0; z < S; ++z) v += part[(int64_t)z * part_stride + i * N + j];
    if (bias) v += bias[j];
    if (act == 1) v = v > 0.0f ? v : 0.0f;
    else if (act == 2) v = 1.0f / (1.0f + expf(-v));
    if (mask && !(mask[i * ldm + j] > 0.0f)) v = 0.0f;
    C[i * ldc + j] = v;
}

// out[m, c] = g[m, c] * y[m, c] * (1 - y[m, c]): the adjoint of y = sigmoid(.) (rgb_layer.2, mip_model.py:35)
__global__ __launch_bounds__(256) void sigmoid_backward_kernel(const float* __restrict__ gr, int64_t gs, const float* __restrict__ y, int64_t ys, int64_t M, int cols,
                                                               float* __restrict__ out, int64_t os) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * cols) return;
    const int64_t m = idx / cols;
    const int c = (int)(idx - m * cols);
    const float yy = y[m * ys + c];
    out[m * os + c] = gr[m * gs + c] * (yy * (1.0f - yy));
}

// the split of the contraction: only when it is long and the output small (the weight-gradient shape), sized to ~4 workgroups per CU
int split_count(int64_t M, int64_t N, int64_t P) {
    const int64_t tiles = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
    if (P < 4096 || tiles >= 512) return 1;
    int64_t s = (4 * (int64_t)nerf_host::cu_count() + tiles - 1) / tiles;
    const int64_t max_s = (P + 4 * GBK - 1) / (4 * GBK);          // at least four LDS stages per slice
    if (s > max_s) s = max_s;
    if (s > 1024) s = 1024;
    return s < 1 ? 1 : (int)s;
}

}  // namespace

size_t gk_gemm_workspace_bytes(int64_t M, int64_t N, int64_t P) {
    const int s = split_count(M, N, P);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)N * sizeof(float) : 0;
}

int gk_gemm(int bf16, int64_t M, int64_t N, int64_t P, const float* A, int64_t a_si, int64_t a_sp, const float* B, int64_t b_sp, int64_t b_sj, float* C,
            int64_t ldc, const float* bias, int act, const float* mask, int64_t ldm, void* workspace, hipStream_t st) {
    if (M == 0 || N == 0) return 0;
    const int S = split_count(M, N, P);
    GemmArgs g{M, N, P, A, a_si, a_sp, B, b_sp, b_sj, C, ldc, bias, act, mask, ldm, P > 0 ? P : 1, 0, (int)((M + GBM - 1) / GBM), (int)((N + GBN - 1) / GBN), S};
    const dim3 grid((unsigned)((int64_t)g.tiles_m * g.tiles_n * S));
    if (S > 1) {
        g.p_chunk = ((P + S - 1) / S + GBK - 1) / GBK * GBK;
        g.C = reinterpret_cast<float*>(workspace);
        g.ldc = N;
        g.part_stride = M * N;
    }
    if (bf16) hipLaunchKernelGGL(gemm_kernel<true>, grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_kernel<false>, grid, dim3(256), 0, st, g);
    if (S > 1)
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float*>(workspace), M * N, S, M, N, C,
                           ldc, bias, act, mask, ldm);
    return (int)hipGetLastError();
}

int gk_sigmoid_backward(const float* g, int64_t gs, const float* y, int64_t ys, int64_t M, int cols, float* out, int64_t os, hipStream_t st) {
    if (M * cols == 0) return 0;
    hipLaunchKernelGGL(sigmoid_backward_kernel, dim3((unsigned)((M * cols + 255) / 256)), dim3(256), 0, st, g, gs, y, ys, M, cols, out, os);
    return (int)hipGetLastError();
}
