// Generic-shape Ref-NeRF (round 4): the element-wise stages between the layer products of a RefNeRF that the fused `ref_kernel` is not
// compiled for -- hidden width above 256, more than 10 position octaves, `--ide_level 5` (36 spherical-harmonic terms; procedures.py:211)
// -- evaluated layer by layer on `nerf_amd_gemm` (generic_kernels.hip) by nerf_amd/generic_path.py.  One thread per sample; all matrices
// fp32 row-major with explicit row strides (so column ranges of a concatenated buffer are views).  Head columns, in the order
// cat(norm_col_tint_head, rho_tau_head) produces them (ref_model.py:78-79):  [normal 0-2 | diffuse 3-5 | tint 6-8 | roughness 9 | density 10].
//
//   ref_dir_inputs            ref_model.py:80-92: roughness = softplus(rho - 1), n = -n / (|n| + 1e-7), w_r = d - 2 (d.n) n,
//                             IDE(w_r, roughness) (ref_func.py:76-108, any level 1..5), n.d  ->  [IDE real T | IDE imag T | n.d], the normal
//   ref_dir_inputs_backward   its adjoint w.r.t. the normal and roughness heads
//   ref_combine               ref_model.py:98-105: rgb = spec * sigmoid(tint) + sigmoid(diffuse) (use_srgb: linear_to_srgb(spec * sigmoid(tint)
//                             + sigmoid(diffuse - log 3))), density passed through
//   ref_combine_backward      its adjoint w.r.t. the spec head's pre-activation and the diffuse / tint / density heads
//   pe_backward               d [x | sin 2^f x | cos 2^f x] / d x applied to a row of encoding gradients (RefNeRF.get_grad, ref_model.py:119-125)
//   add_rows                  dst += src on a column range
// The arithmetic follows the fused kernels' (ref_heads_delta_kernel, bwd_kernels.hip), with the term tables generated for the level.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "host_common.h"

namespace {

DEVINL float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int DEG> struct Ide {
    static constexpr int LMAX = 1 << (DEG - 1);
    static constexpr int T = (1 << DEG) - 1 + DEG;
};

struct RefFwd {                                   // the per-sample forward quantities both directions need
    float n0x, n0y, n0z, len, nn, nx, ny, nz, dot, rx, ry, rz, kinv;
};
DEVINL RefFwd ref_fwd(const float* h, float dx, float dy, float dz) {
    RefFwd f;
    f.n0x = h[0]; f.n0y = h[1]; f.n0z = h[2];
    f.len = norm3(f.n0x, f.n0y, f.n0z); f.nn = f.len + 1e-7f;
    f.nx = -f.n0x / f.nn; f.ny = -f.n0y / f.nn; f.nz = -f.n0z / f.nn;
    f.dot = (dx * f.nx + dy * f.ny) + dz * f.nz;
    f.rx = dx - 2.0f * f.dot * f.nx; f.ry = dy - 2.0f * f.dot * f.ny; f.rz = dz - 2.0f * f.dot * f.nz;
    f.kinv = softplus_f(h[9] - 1.0f);
    return f;
}

template <int DEG>
__global__ __launch_bounds__(256) void ref_dir_inputs_kernel(const float* __restrict__ heads, int64_t ldh, const float* __restrict__ dirs, int64_t ds,
                                                             const float* __restrict__ mat, int64_t M, float* __restrict__ out, int64_t ldo,
                                                             float* __restrict__ normal) {
    constexpr int LMAX = Ide<DEG>::LMAX, T = Ide<DEG>::T;
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float dx = dirs[m * ds], dy = dirs[m * ds + 1], dz = dirs[m * ds + 2];
    const RefFwd f = ref_fwd(heads + m * ldh, dx, dy, dz);
    float zp[LMAX + 1], re[LMAX + 1], im[LMAX + 1];
    zp[0] = 1.0f; re[0] = 1.0f; im[0] = 0.0f;
#pragma unroll
    for (int k = 1; k <= LMAX; ++k) { zp[k] = zp[k - 1] * f.rz; re[k] = re[k - 1] * f.rx - im[k - 1] * f.ry; im[k] = re[k - 1] * f.ry + im[k - 1] * f.rx; }
    float* o = out + m * ldo;
    int t = 0;
#pragma unroll
    for (int i = 0; i < DEG; ++i) {
        const int l = 1 << i;
        const float att = expf(-(0.5f * (float)(l * (l + 1))) * f.kinv);
#pragma unroll
        for (int mm = 0; mm <= l; ++mm, ++t) {
            float poly = 0.0f;
#pragma unroll
            for (int k = 0; k <= l - mm; ++k) poly = __builtin_fmaf(mat[k * T + t], zp[k], poly);
            o[t] = (re[mm] * poly) * att;
            o[T + t] = (im[mm] * poly) * att;
        }
    }
    o[2 * T] = f.dot;
    normal[m * 3] = f.nx; normal[m * 3 + 1] = f.ny; normal[m * 3 + 2] = f.nz;
}

template <int DEG>
__global__ __launch_bounds__(256) void ref_dir_inputs_backward_kernel(const float* __restrict__ heads, int64_t ldh, const float* __restrict__ dirs, int64_t ds,
                                                                      const float* __restrict__ mat, int64_t M, const float* __restrict__ d_in, int64_t ldi,
                                                                      const float* __restrict__ g_normal, int64_t ldg, float* __restrict__ d_heads, int64_t ldd) {
    constexpr int LMAX = Ide<DEG>::LMAX, T = Ide<DEG>::T;
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float dx = dirs[m * ds], dy = dirs[m * ds + 1], dz = dirs[m * ds + 2];
    const float* h = heads + m * ldh;
    const RefFwd f = ref_fwd(h, dx, dy, dz);
    float zp[LMAX + 1], re[LMAX + 1], im[LMAX + 1], d_re[LMAX + 1], d_im[LMAX + 1];
    zp[0] = 1.0f; re[0] = 1.0f; im[0] = 0.0f;
#pragma unroll
    for (int k = 1; k <= LMAX; ++k) { zp[k] = zp[k - 1] * f.rz; re[k] = re[k - 1] * f.rx - im[k - 1] * f.ry; im[k] = re[k - 1] * f.ry + im[k - 1] * f.rx; }
#pragma unroll
    for (int k = 0; k <= LMAX; ++k) { d_re[k] = 0.0f; d_im[k] = 0.0f; }
    const float* da = d_in + m * ldi;
    float d_rz = 0.0f, d_kinv = 0.0f;
    int t = 0;
#pragma unroll
    for (int i = 0; i < DEG; ++i) {
        const int l = 1 << i;
        const float sig = 0.5f * (float)(l * (l + 1));
        const float att = expf(-sig * f.kinv);
#pragma unroll
        for (int mm = 0; mm <= l; ++mm, ++t) {
            float poly = 0.0f, dpoly = 0.0f;
#pragma unroll
            for (int k = 0; k <= l - mm; ++k) {
                poly = __builtin_fmaf(mat[k * T + t], zp[k], poly);
                if (k >= 1) dpoly = __builtin_fmaf((float)k * mat[k * T + t], zp[k - 1], dpoly);
            }
            const float gr = da[t], gi = da[T + t];
            const float A = gr * re[mm] + gi * im[mm];
            d_rz += A * att * dpoly;
            d_kinv -= A * poly * sig * att;
            d_re[mm] += gr * poly * att;
            d_im[mm] += gi * poly * att;
        }
    }
    float d_rx = 0.0f, d_ry = 0.0f;
#pragma unroll
    for (int k = 1; k <= LMAX; ++k) {                        // d (x + i y)^k / dx = k (x + i y)^(k-1),  d / dy = i k (x + i y)^(k-1)
        d_rx += (float)k * (d_re[k] * re[k - 1] + d_im[k] * im[k - 1]);
        d_ry += (float)k * (d_im[k] * re[k - 1] - d_re[k] * im[k - 1]);
    }
    // normal: gradient from the loss (the predicted-normal output), from n.d and from the reflection r = d - 2 (d.n) n
    const float g_nd = da[2 * T];
    const float rdotn = (d_rx * f.nx + d_ry * f.ny) + d_rz * f.nz;
    const float* gn = g_normal + m * ldg;
    const float dnx = gn[0] + g_nd * dx - 2.0f * (rdotn * dx + f.dot * d_rx);
    const float dny = gn[1] + g_nd * dy - 2.0f * (rdotn * dy + f.dot * d_ry);
    const float dnz = gn[2] + g_nd * dz - 2.0f * (rdotn * dz + f.dot * d_rz);
    // n = -n0 / (|n0| + eps):  d n0 = -( dn / nn - n0 (n0 . dn) / (|n0| nn^2) )
    const float n0dn = (f.n0x * dnx + f.n0y * dny) + f.n0z * dnz;
    const float cden = n0dn / (fmaxf(f.len, 1e-30f) * f.nn * f.nn);
    float* dh = d_heads + m * ldd;
    dh[0] = -(dnx / f.nn - f.n0x * cden); dh[1] = -(dny / f.nn - f.n0y * cden); dh[2] = -(dnz / f.nn - f.n0z * cden);
    dh[9] = d_kinv * sigm(h[9] - 1.0f);                                                       // softplus'(v) = sigmoid(v)
}

__global__ __launch_bounds__(256) void ref_combine_kernel(const float* __restrict__ heads, int64_t ldh, const float* __restrict__ spec, int64_t lds_, int64_t M,
                                                          int srgb, float* __restrict__ rgbo) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* h = heads + m * ldh;
    const float* sp = spec + m * lds_;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float lin = sp[c] * sigm(h[6 + c]) + sigm(srgb ? h[3 + c] - SRGB_LOG3 : h[3 + c]);
        rgbo[m * 4 + c] = srgb ? srgb_from_linear(lin) : lin;
    }
    rgbo[m * 4 + 3] = h[10];
}

__global__ __launch_bounds__(256) void ref_combine_backward_kernel(const float* __restrict__ g, int64_t ldg, const float* __restrict__ heads, int64_t ldh,
                                                                   const float* __restrict__ spec, int64_t lds_, int64_t M, int srgb, float* __restrict__ d_spec,
                                                                   int64_t ldsp, float* __restrict__ d_heads, int64_t ldd) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* h = heads + m * ldh;
    const float* sp = spec + m * lds_;
    float* dh = d_heads + m * ldd;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float st = sigm(h[6 + c]), sd = sigm(srgb ? h[3 + c] - SRGB_LOG3 : h[3 + c]), s = sp[c];
        const float gl = g[m * ldg + c] * (srgb ? srgb_slope(s * st + sd) : 1.0f);
        d_spec[m * ldsp + c] = (gl * st) * (s * (1.0f - s));              // through the spec head's sigmoid
        dh[3 + c] = gl * (sd * (1.0f - sd));
        dh[6 + c] = (gl * s) * (st * (1.0f - st));
    }
    dh[10] = g[m * ldg + 3];
}

// d_x[c] = d_enc[c] (cat_origin) + sum_f 2^f (cos(2^f x_c) d_sin[f, c] - sin(2^f x_c) d_cos[f, c]);  row layout of nerf_helper.py:38-48
__global__ __launch_bounds__(256) void pe_backward_kernel(const float* __restrict__ d_enc, int64_t ldd, const float* __restrict__ x, int64_t ldx, int64_t M, int L,
                                                          int cat_origin, float* __restrict__ d_x) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * 3) return;
    const int64_t m = idx / 3;
    const int c = (int)(idx - m * 3);
    const float* de = d_enc + m * ldd;
    const float xv = x[m * ldx + c];
    const int off = cat_origin ? 3 : 0;
    float acc = cat_origin ? de[c] : 0.0f;
    float scale = 1.0f;
    for (int f = 0; f < L; ++f, scale *= 2.0f) {
        const float a = scale * xv;
        acc += scale * (cosf(a) * de[off + 6 * f + c] - sinf(a) * de[off + 6 * f + 3 + c]);
    }
    d_x[idx] = acc;
}

__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ dst, int64_t ldd, const float* __restrict__ src, int64_t lds_, int64_t M, int cols) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * cols) return;
    const int64_t m = idx / cols;
    const int c = (int)(idx - m * cols);
    dst[m * ldd + c] += src[m * lds_ + c];
}

unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

// Mip-NeRF 360 scene contraction (Barron et al. 2022, eq. 10; mlp_kernels.hip contract_position, the same operations in the same order)
// for the layer-by-layer route, where it is a stage of its own in front of the encoder: out = contract(x) (g == nullptr), or the
// pull-back of a gradient g w.r.t. contract(x) through the contraction's (symmetric) Jacobian: J = I inside the unit ball,
// (2 - 1/r) / r (I - u u^T) + u u^T / r^2 outside (r = |x|, u = x / r).  One thread per position.
__global__ __launch_bounds__(256) void contract_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, const float* __restrict__ g, int64_t ldg,
                                                       float* __restrict__ out) {
    for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
        const float x0 = x[m * ldx], x1 = x[m * ldx + 1], x2 = x[m * ldx + 2];
        const float r = norm3(x0, x1, x2);
        const float k = r > 1.0f ? (2.0f - 1.0f / r) / r : 1.0f;
        if (g == nullptr) {
            float o0 = x0, o1 = x1, o2 = x2;
            if (r > 1.0f) { o0 *= k; o1 *= k; o2 *= k; }
            out[m * 3] = o0; out[m * 3 + 1] = o1; out[m * 3 + 2] = o2;
        } else {
            float g0 = g[m * ldg], g1 = g[m * ldg + 1], g2 = g[m * ldg + 2];
            if (r > 1.0f) {
                const float u0 = x0 / r, u1 = x1 / r, u2 = x2 / r;
                const float ug = (u0 * g0 + u1 * g1) + u2 * g2;
                const float t = ug * (1.0f / (r * r) - k);
                g0 = k * g0 + t * u0; g1 = k * g1 + t * u1; g2 = k * g2 + t * u2;
            }
            out[m * 3] = g0; out[m * 3 + 1] = g1; out[m * 3 + 2] = g2;
        }
    }
}

}  // namespace

#define GR_DEG_SWITCH(CALL)                                                       \
    switch (deg) {                                                                \
        case 1: CALL(1); break;                                                   \
        case 2: CALL(2); break;                                                   \
        case 3: CALL(3); break;                                                   \
        case 4: CALL(4); break;                                                   \
        case 5: CALL(5); break;                                                   \
        default: return -1;                                                       \
    }

int gr_dir_inputs(const float* heads, int64_t ldh, const float* dirs, int64_t ds, int64_t M, int deg, const float* mat, float* out, int64_t ldo, float* normal,
                  hipStream_t st) {
    if (M == 0) return 0;
#define GR_CALL(D) hipLaunchKernelGGL(ref_dir_inputs_kernel<D>, dim3(blocks(M)), dim3(256), 0, st, heads, ldh, dirs, ds, mat, M, out, ldo, normal)
    GR_DEG_SWITCH(GR_CALL)
#undef GR_CALL
    return (int)hipGetLastError();
}

int gr_dir_inputs_backward(const float* heads, int64_t ldh, const float* dirs, int64_t ds, int64_t M, int deg, const float* mat, const float* d_in, int64_t ldi,
                           const float* g_normal, int64_t ldg, float* d_heads, int64_t ldd, hipStream_t st) {
    if (M == 0) return 0;
#define GR_CALL(D) hipLaunchKernelGGL(ref_dir_inputs_backward_kernel<D>, dim3(blocks(M)), dim3(256), 0, st, heads, ldh, dirs, ds, mat, M, d_in, ldi, g_normal, ldg, d_heads, ldd)
    GR_DEG_SWITCH(GR_CALL)
#undef GR_CALL
    return (int)hipGetLastError();
}

int gr_combine(const float* heads, int64_t ldh, const float* spec, int64_t lds_, int64_t M, int srgb, float* rgbo, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(ref_combine_kernel, dim3(blocks(M)), dim3(256), 0, st, heads, ldh, spec, lds_, M, srgb, rgbo);
    return (int)hipGetLastError();
}

int gr_combine_backward(const float* g, int64_t ldg, const float* heads, int64_t ldh, const float* spec, int64_t lds_, int64_t M, int srgb, float* d_spec, int64_t ldsp,
                        float* d_heads, int64_t ldd, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(ref_combine_backward_kernel, dim3(blocks(M)), dim3(256), 0, st, g, ldg, heads, ldh, spec, lds_, M, srgb, d_spec, ldsp, d_heads, ldd);
    return (int)hipGetLastError();
}

int gr_pe_backward(const float* d_enc, int64_t ldd, const float* x, int64_t ldx, int64_t M, int L, int cat_origin, float* d_x, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(pe_backward_kernel, dim3(blocks(M * 3)), dim3(256), 0, st, d_enc, ldd, x, ldx, M, L, cat_origin, d_x);
    return (int)hipGetLastError();
}

int gr_contract(const float* x, int64_t ldx, int64_t M, const float* g, int64_t ldg, float* out, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(contract_kernel, dim3(blocks(M)), dim3(256), 0, st, x, ldx, M, g, ldg, out);
    return (int)hipGetLastError();
}

int gr_add_rows(float* dst, int64_t ldd, const float* src, int64_t lds_, int64_t M, int cols, hipStream_t st) {
    if (M * cols == 0) return 0;
    hipLaunchKernelGGL(add_rows_kernel, dim3(blocks(M * cols)), dim3(256), 0, st, dst, ldd, src, lds_, M, cols);
    return (int)hipGetLastError();
}
