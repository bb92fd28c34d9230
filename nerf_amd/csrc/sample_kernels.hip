// Sampling / compositing kernels: one 64-lane wavefront per
// ray, samples across lanes, wave-level prefix scans for transmittance and CDF.  All HBM-bound.
//
// Numerics follow the reference's CPU torch kernels so that results agree to ~1 ulp:
//   * cumprod / cumsum accumulate float inputs in fp64 and round every prefix to fp32;
//   * position and depth arithmetic is separate mul + add (the file is built with -ffp-contract=off);
//   * searchsorted(right=True) = number of CDF entries <= u.
#include "device_common.h"
#include "host_common.h"
#include "../../include/nerf_amd.h"

extern __shared__ __attribute__((aligned(16))) char smem[];

namespace {

constexpr int WAVES_PER_BLOCK = 4;

DEVINL int wave_in_block() { return (int)(threadIdx.x >> 6); }

// ---------------------------------------------------------------------------------------- rows 3 and 12: the stand-alone encoders
// Both are HBM-bound by their OUTPUT (24 L bytes per sample against 12-30 read): a workgroup takes ENC_TILE samples,
//   phase 1  one thread per sample parks the encoder's per-coordinate inputs in LDS (PE: x; IPE: Gaussian mean and diagonal covariance);
//   phase 2  one thread per (sample, frequency, coordinate) PAIR evaluates sin and cos from one range reduction (and, for the IPE, one
//            exp shared by both) and stages the two values in LDS in output order;
//   phase 3  the tile's 24 L ENC_TILE bytes leave as fully coalesced 16-byte stores.
// The first version computed every output element on its own (one range reduction per element, a 64-bit division per element, 4-byte
// stores): 0.2 TB/s (PE) / 2.0 TB/s (IPE) at 640 000 x 128 samples.
constexpr int ENC_TILE = 64;                          // dynamic LDS: ENC_TILE x (6 L staged outputs + 6 moments) floats = 16.5 KiB at L = 10
static inline size_t enc_lds_bytes(int L) { return (size_t)ENC_TILE * (6 * L + 6) * 4; }
template <bool IPE, class Moments>
DEVINL void encode_tile(float (*mom)[6], float* stage, int64_t base, int64_t total, int L, float* __restrict__ feat, Moments&& moments) {
    const int width = 6 * L, pairs_per = 3 * L;
    const int nsamp = (int)(total - base < ENC_TILE ? total - base : ENC_TILE);
    if ((int)threadIdx.x < nsamp) moments((int)threadIdx.x, base + threadIdx.x);
    __syncthreads();
    const int n_pairs = nsamp * pairs_per;
    for (int p = threadIdx.x; p < n_pairs; p += blockDim.x) {
        const int sl = p / pairs_per, rem = p - sl * pairs_per;
        const int l = rem / 3, c = rem - 3 * l;
        float sn, cs;
        sincos_quadrant(mom[sl][c] * (float)(1 << l), sn, cs);                        // exact scaling (nerf_helper.py:42-44, mip_methods.py:43)
        if (IPE) {
            const float e = expf(-0.5f * (mom[sl][3 + c] * (float)(1u << (2 * l))));  // diag_P * diag (mip_methods.py:42,55)
            sn *= e; cs *= e;
        }
        stage[sl * width + 6 * l + c] = sn;
        stage[sl * width + 6 * l + 3 + c] = cs;
    }
    __syncthreads();
    const int cnt = nsamp * width;
    float* o = feat + base * width;                                                     // (base * width * 4 is a multiple of 16)
    const f32x4* s4 = reinterpret_cast<const f32x4*>(stage);
    f32x4* o4 = reinterpret_cast<f32x4*>(o);
    for (int i = threadIdx.x; i < cnt / 4; i += blockDim.x) o4[i] = s4[i];
    for (int i = (cnt & ~3) + threadIdx.x; i < cnt; i += blockDim.x) o[i] = stage[i];
    __syncthreads();
}

__global__ __launch_bounds__(256) void pe_kernel(const float* __restrict__ x, int64_t M, int L, float* __restrict__ out) {
    float* stage = reinterpret_cast<float*>(smem);
    float (*mom)[6] = reinterpret_cast<float (*)[6]>(stage + ENC_TILE * 6 * L);
    for (int64_t base = blockIdx.x * (int64_t)ENC_TILE; base < M; base += (int64_t)gridDim.x * ENC_TILE)
        encode_tile<false>(mom, stage, base, M, L, out, [&](int sl, int64_t m) {
            mom[sl][0] = x[m * 3]; mom[sl][1] = x[m * 3 + 1]; mom[sl][2] = x[m * 3 + 2];
        });
}

// Integrated positional encoding (mip_methods.py:15-58): the Gaussian moments of the frustum (cone_moments / cone_mean_cov,
// device_common.h: the reference's operation order), then
//   out[.., 6 l + 3 t + c] = (t ? cos : sin)(2^l mu_c) * exp(-0.5 * (4^l diag_c))     (multFreq :36-45, ipe_feature :51-58)
__global__ __launch_bounds__(256) void ipe_feature_kernel(const float* __restrict__ z, const float* __restrict__ rays, int64_t N, int S, int L,
                                                           float r2, const float* __restrict__ dir_norm, float* __restrict__ feat,
                                                           float* __restrict__ mu_out, float* __restrict__ mu_t_out, int contract) {
    float* stage = reinterpret_cast<float*>(smem);
    float (*mom)[6] = reinterpret_cast<float (*)[6]>(stage + ENC_TILE * 6 * L);
    const int64_t total = N * S;
    const float dn = dir_norm[0];
    for (int64_t base = blockIdx.x * (int64_t)ENC_TILE; base < total; base += (int64_t)gridDim.x * ENC_TILE)
        encode_tile<true>(mom, stage, base, total, L, feat, [&](int sl, int64_t m) {
            const int64_t n = m / S;
            const int si = (int)(m - n * S);
            const float* zz = z + n * (S + 1) + si;
            const ConeMoments c = cone_moments(zz[0], zz[1], r2);
            const float* ry = rays + n * 6;
#pragma unroll
            for (int k = 0; k < 3; ++k) cone_mean_cov(c, ry[k], ry[3 + k], dn, mom[sl][k], mom[sl][3 + k]);
            if (contract) {                                  // Mip-NeRF 360 contraction of the frustum MEAN (the covariance stays metric): the fused
                const float nn = norm3(mom[sl][0], mom[sl][1], mom[sl][2]);   // kernels' sample fetch does the same (mlp_kernels.hip contract_position)
                if (nn > 1.0f) {
                    const float kk = (2.0f - 1.0f / nn) / nn;
                    mom[sl][0] *= kk; mom[sl][1] *= kk; mom[sl][2] *= kk;
                }
            }
            if (mu_out) { mu_out[m * 3] = mom[sl][0]; mu_out[m * 3 + 1] = mom[sl][1]; mu_out[m * 3 + 2] = mom[sl][2]; }
            if (mu_t_out) mu_t_out[m] = c.mu_t;
        });
}

// coneParameters alone (mip_methods.py:15-23): z (N, S+1) -> mu_t, sigma_t^2, sigma_r^2 (N, S)
__global__ void cone_parameters_kernel(const float* __restrict__ z, int64_t N, int S, float r2, float* __restrict__ mu_t, float* __restrict__ var_t,
                                       float* __restrict__ var_r) {
    const int64_t total = N * S;
    for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < total; m += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = m / S;
        const float* zz = z + n * (S + 1) + (m - n * S);
        const ConeMoments c = cone_moments(zz[0], zz[1], r2);
        mu_t[m] = c.mu_t; var_t[m] = c.var_t; var_r[m] = c.var_r;
    }
}

// norm of the whole direction tensor (mip_methods.py:31 `cam_rays[:, 3:].norm()`): one workgroup, fp64 partial sums, fixed order
__global__ __launch_bounds__(1024) void dirs_norm_kernel(const float* __restrict__ rays, int64_t N, float* __restrict__ out) {
    __shared__ double part[1024];
    double acc = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += 1024) {
        const float* d = rays + n * 6 + 3;
        acc += (double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2];
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sqrt(part[0]);
}

// the same over the whole chip when the caller has scratch for the workgroup partials (the render path: 0.36 ms -> a few microseconds per
// 640 000 rays): fp64 partial sums per workgroup, then one workgroup adds the partials in a fixed order
constexpr int DN_BLOCKS = 256;
__global__ __launch_bounds__(256) void dirs_norm_partial_kernel(const float* __restrict__ rays, int64_t N, double* __restrict__ partials) {
    __shared__ double part[256];
    double acc = 0.0;
    for (int64_t n = blockIdx.x * (int64_t)256 + threadIdx.x; n < N; n += (int64_t)DN_BLOCKS * 256) {
        const float* d = rays + n * 6 + 3;
        acc += (double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2];
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = part[0];
}
__global__ __launch_bounds__(256) void dirs_norm_final_kernel(const double* __restrict__ partials, float* __restrict__ out) {
    __shared__ double part[256];
    part[threadIdx.x] = partials[threadIdx.x];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sqrt(part[0]);
}

// ---------------------------------------------------------------------------------------- row 1
struct Cam { int H, W; float fx, fy; float pose[12]; };

__global__ void raygen_kernel(Cam cam, int64_t first, int64_t count, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = first + i;
        float* rays = out - first * 6;
        const int row = (int)(n / cam.W), col = (int)(n - (int64_t)row * cam.W);
        const float cx = (((float)col - (float)cam.W * 0.5f) + 0.5f) / cam.fx;
        const float cy = (((float)cam.H * 0.5f - (float)row) + 0.5f) / cam.fy;
        float* r = rays + n * 6;
        r[0] = cam.pose[3]; r[1] = cam.pose[7]; r[2] = cam.pose[11];
        r[3] = (cx * cam.pose[0] + cy * cam.pose[1]) + (-1.0f) * cam.pose[2];
        r[4] = (cx * cam.pose[4] + cy * cam.pose[5]) + (-1.0f) * cam.pose[6];
        r[5] = (cx * cam.pose[8] + cy * cam.pose[9]) + (-1.0f) * cam.pose[10];
    }
}

// training twin of row 1 (utils.py:78-85): integer (col - W//2, H//2 - row) coords -> rays
__global__ void pixel_rays_kernel(Cam cam, const int64_t* __restrict__ coords, int64_t N, float* __restrict__ rays) {
    for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const float cx = ((float)coords[n * 2] + 0.5f) / cam.fx;
        const float cy = ((float)coords[n * 2 + 1] + 0.5f) / cam.fy;
        float* r = rays + n * 6;
        r[0] = cam.pose[3]; r[1] = cam.pose[7]; r[2] = cam.pose[11];
        r[3] = (cx * cam.pose[0] + cy * cam.pose[1]) + (-1.0f) * cam.pose[2];
        r[4] = (cx * cam.pose[4] + cy * cam.pose[5]) + (-1.0f) * cam.pose[6];
        r[5] = (cx * cam.pose[8] + cy * cam.pose[9]) + (-1.0f) * cam.pose[10];
    }
}

// row 2 (utils.py:87-90 / procedures.py:65-66): z = z_base + u*jitter ; pts = o + d*z
__global__ void stratified_points_kernel(const float* __restrict__ rays, const float* __restrict__ z_base,
                                         const float* __restrict__ u, float jitter, int64_t N, int S,
                                         float* __restrict__ z_out, float* __restrict__ pts) {
    const int64_t total = N * S;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / S;
        const int s = (int)(i - n * S);
        const float zv = z_base[s] + u[i] * jitter;
        z_out[i] = zv;
        if (pts) {
            const float* r = rays + n * 6;
            pts[i * 3] = r[0] + r[3] * zv; pts[i * 3 + 1] = r[1] + r[4] * zv; pts[i * 3 + 2] = r[2] + r[5] * zv;
        }
    }
}

// ---------------------------------------------------------------------------------------- training sampler
// validSampler (utils.py:72-94) in ONE launch with every random number drawn in the kernel: pixel index of ray n = floor(P * u) from
// Philox stream 'IX', ground-truth colour gather, ray through the pixel (utils.py:78-85), stratified depths z = base_s + u * res from
// stream 'TS' (utils.py:87-89) and the sample positions o + d z (utils.py:90).  The reference draws both on the CPU generator and
// copies them to the device every iteration (train.py:153-157).  One workgroup = 4 rays x C samples.
constexpr uint32_t PHILOX_STREAM_INDEX = 0x4958u, PHILOX_STREAM_TRAIN = 0x5453u;
// `pose_dev` / `seed_dev` (both optional): the camera pose and the seed read from DEVICE memory instead of the launch arguments, so that
// a captured hipGraph of the training step sees a new image pose and fresh random numbers on every replay.
__global__ __launch_bounds__(256) void train_sampler_kernel(const float* __restrict__ rgbs, const int64_t* __restrict__ coords, int64_t P, Cam cam,
                                                            float near, float res, int64_t N, int C, uint64_t seed, float* __restrict__ pts,
                                                            float* __restrict__ lengths, float* __restrict__ rgb, float* __restrict__ rays,
                                                            const float* __restrict__ pose_dev, const uint64_t* __restrict__ seed_dev) {
    __shared__ float ray_s[4][6];
    if (pose_dev != nullptr) {
#pragma unroll
        for (int i = 0; i < 12; ++i) cam.pose[i] = pose_dev[i];
    }
    if (seed_dev != nullptr) seed = seed_dev[0];
    for (int64_t n0 = blockIdx.x * (int64_t)4; n0 < N; n0 += (int64_t)gridDim.x * 4) {
        __syncthreads();
        if (threadIdx.x < 4 && n0 + threadIdx.x < N) {
            const int64_t n = n0 + threadIdx.x;
            const Philox4 r = philox4x32_10((uint32_t)n, (uint32_t)((uint64_t)n >> 32), 0u, PHILOX_STREAM_INDEX, (uint32_t)seed, (uint32_t)(seed >> 32));
            const uint64_t x = ((uint64_t)r.w[0] << 32) | r.w[1];
            const int64_t idx = (int64_t)__umul64hi(x, (uint64_t)P);                     // uniform in [0, P)
            const float cx = ((float)coords[idx * 2] + 0.5f) / cam.fx;                   // utils.py:78-81
            const float cy = ((float)coords[idx * 2 + 1] + 0.5f) / cam.fy;
            float* q = ray_s[threadIdx.x];
            q[0] = cam.pose[3]; q[1] = cam.pose[7]; q[2] = cam.pose[11];
            q[3] = (cx * cam.pose[0] + cy * cam.pose[1]) + (-1.0f) * cam.pose[2];
            q[4] = (cx * cam.pose[4] + cy * cam.pose[5]) + (-1.0f) * cam.pose[6];
            q[5] = (cx * cam.pose[8] + cy * cam.pose[9]) + (-1.0f) * cam.pose[10];
#pragma unroll
            for (int k = 0; k < 6; ++k) rays[n * 6 + k] = q[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) rgb[n * 3 + k] = rgbs[idx * 3 + k];
        }
        __syncthreads();
        if (lengths == nullptr) continue;
        const int64_t cnt = ((N - n0 < 4) ? N - n0 : 4) * C;
        for (int64_t i = threadIdx.x; i < cnt; i += 256) {
            const int rl = (int)(i / C), sidx = (int)(i - (int64_t)rl * C);
            const int64_t n = n0 + rl;
            const Philox4 r = philox4x32_10((uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)(sidx >> 2), PHILOX_STREAM_TRAIN, (uint32_t)seed,
                                            (uint32_t)(seed >> 32));
            const int w = sidx & 3;
            const float u = u01_from_bits(w == 0 ? r.w[0] : (w == 1 ? r.w[1] : (w == 2 ? r.w[2] : r.w[3])));
            const float z = (near + (float)sidx * res) + u * res;                        // linspace(near, far - res, C)[s] + u res
            lengths[n * C + sidx] = z;
            const float* q = ray_s[rl];
            float* o = pts + (n * C + sidx) * 3;
            o[0] = q[0] + q[3] * z; o[1] = q[1] + q[4] * z; o[2] = q[2] + q[5] * z;
        }
    }
}

// ---------------------------------------------------------------------------------------- row 8
__global__ void length2pts_kernel(const float* __restrict__ rays, const float* __restrict__ z, int64_t N, int S,
                                  float* __restrict__ out) {
    const int64_t total = N * S * 6;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ns = i / 6;
        const int k = (int)(i - ns * 6);
        const int64_t n = ns / S;
        const float* r = rays + n * 6;
        out[i] = (k < 3) ? (r[k] + r[3 + k] * z[ns]) : r[k];
    }
}

__global__ __launch_bounds__(256) void sigma_to_weights_kernel(const float* __restrict__ sigma, const float* __restrict__ z,
                                                              const float* __restrict__ dirs, int64_t N, int S, int act,
                                                              float* __restrict__ w) {
    for (int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); n < N; n += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        float nrm = 1.0f;
        const bool scale = dirs != nullptr;
        if (scale) nrm = norm3(dirs[n * 3], dirs[n * 3 + 1], dirs[n * 3 + 2]);
        const float* sg = sigma + n * S;
        const float* zz = z + n * S;
        float* wo = w + n * S;
        wave_sigma_to_weights(S, act, [&](int s) { return sg[s]; },
                              [&](int s) { return scale ? zz[s] * nrm : zz[s]; },
                              [&](int s, float wv, float) { wo[s] = wv; });
    }
}

// ---------------------------------------------------------------------------------------- row 6
__global__ void max_blur_kernel(const float* __restrict__ w, int64_t N, int S, float alpha, float* __restrict__ out) {
    const int64_t total = N * S;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % S);
        const float c = w[i];
        const float front = (s == 0) ? c : fmaxf(w[i - 1], c);
        const float rear = (s == S - 1) ? c : fmaxf(c, w[i + 1]);
        out[i] = 0.5f * (front + rear) + alpha;
    }
}

// ------------------------------------------------------------------------------------------------
// Inverse-transform sampling for one ray by one wavefront (row 7, utils.py:108-133).
//   pw[nw]   raw pdf weights (1e-5 is added here), bins[nw+1] bin edges -- both in LDS, visible to the wave
//   cdf[nw+1], samp[K], bel[K]: LDS scratch
// ------------------------------------------------------------------------------------------------
DEVINL void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#ifndef SORT_NB_DEF
#define SORT_NB_DEF 256
#define SORT_CAP_DEF 6
#endif
constexpr int SORT_NB = SORT_NB_DEF, SORT_CAP = SORT_CAP_DEF, SORT_PER_LANE = SORT_NB / 64;
constexpr int SORT_LDS_FLOATS = 2 * SORT_NB + SORT_NB * SORT_CAP / 2;      // cnt, pre (int) + members (uint16)

// uf(i, k) = the uniform of sample k = lane + 64 i (a global pointer, or registers loaded one ray ahead)
template <class UF>
DEVINL void wave_inverse_sample(const float* pw, const float* bins, int nw, float* cdf, float* samp, int* bel, int* sortbuf,
                                UF&& uf, int K, int sort, float* __restrict__ z_out,
                                int64_t* __restrict__ below_out, int64_t* __restrict__ above_out) {
    const int lane = lane_id();
    const int nb = nw + 1;           // bins == cdf entries incl. the leading 0
    // pdf normaliser torch.sum(weights + 1e-5, -1) (utils.py:110-111) in the summation ORDER of torch's CPU kernel, so that the CDF --
    // and with it every searchsorted decision -- is bit-identical to the reference's.  ATen's cascade_sum over a contiguous row
    // (SumKernel.cpp vectorized_inner_sum; verified against torch 2.10 for row lengths 14..255, tests/test_oracle_golden.py): the row is
    // read as 8-float vectors, vector v goes to accumulator v % 4 while whole groups of four remain and to accumulator 0 afterwards,
    // accumulators 1..3 are added to 0 in order; the scalar result is 0 + the < 8 tail elements in order + the 8 vector slots in order.
    // (Rows of >= 512 elements add cascade levels; nw < 256 here.)
    float* part8 = reinterpret_cast<float*>(sortbuf);             // the sort scratch is free until the sort
    const int nv = nw >> 3;
    if (lane < 8) {
        const int n4 = nv >> 2;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int i = 0; i < n4; ++i) {
            a0 += pw[32 * i + lane] + 1e-5f;
            a1 += pw[32 * i + 8 + lane] + 1e-5f;
            a2 += pw[32 * i + 16 + lane] + 1e-5f;
            a3 += pw[32 * i + 24 + lane] + 1e-5f;
        }
        for (int v = 4 * n4; v < nv; ++v) a0 += pw[8 * v + lane] + 1e-5f;
        a0 += a1; a0 += a2; a0 += a3;
        part8[lane] = a0;
    }
    lds_wave_sync();
    float total = 0.0f;              // every lane forms the same scalar (LDS broadcast reads)
    if (nv > 0) {
        for (int j = 8 * nv; j < nw; ++j) total += pw[j] + 1e-5f;
#pragma unroll
        for (int c = 0; c < 8; ++c) total += part8[c];
    } else {                         // rows shorter than one vector: the scalar form (4 accumulators over single elements)
        float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        if (nw >= 4) { total += pw[0] + 1e-5f; a1 += pw[1] + 1e-5f; a2 += pw[2] + 1e-5f; a3 += pw[3] + 1e-5f; }
        for (int j = nw & ~3; j < nw; ++j) total += pw[j] + 1e-5f;
        total += a1; total += a2; total += a3;
    }
    // cdf: fp64 running sum of fp32 pdf values, rounded per element (torch CPU cumsum)
    double carry = 0.0;
    if (lane == 0) cdf[0] = 0.0f;
    for (int base = 0; base < nw; base += 64) {
        const int j = base + lane;
        double p = 0.0;
        if (j < nw) p = (double)((pw[j] + 1e-5f) / total);
        const double incl = wave_incl_scan_add(p);
        if (j < nw) cdf[1 + j] = (float)(carry + incl);
        carry += wave_last(incl);
    }
    lds_wave_sync();
    // searchsorted(right=True): count of cdf entries <= u.  Three samples per lane (K <= 192: all of the ray's) are searched in
    // lock step with a fixed trip count, so that their dependent LDS reads overlap instead of running back to back.
    const int halvings = 32 - __builtin_clz((unsigned)nb);       // an interval of nb + 1 candidates is empty after that many steps
    for (int k0 = 0, i0 = 0; k0 < K; k0 += 192, i0 += 3) {
        float uu[3]; int lo[3], hi[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = k0 + 64 * q + lane;
            uu[q] = (k < K) ? uf(i0 + q, k) : 0.0f;
            lo[q] = 0; hi[q] = (k < K) ? nb : 0;
        }
        for (int it = 0; it < halvings; ++it) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int mid = (lo[q] + hi[q]) >> 1;
                const bool open_ = lo[q] < hi[q];
                const float c = cdf[open_ ? mid : 0];
                if (open_) { if (c <= uu[q]) lo[q] = mid + 1; else hi[q] = mid; }
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = k0 + 64 * q + lane;
            if (k >= K) continue;
            const int below = lo[q] - 1 > 0 ? lo[q] - 1 : 0;
            const int above = lo[q] < nb - 1 ? lo[q] : nb - 1;
            const float c0 = cdf[below], c1 = cdf[above];
            float denom = c1 - c0;
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uu[q] - c0) / denom;
            const float b0 = bins[below], b1 = bins[above];
            const float v = b0 + t * (b1 - b0);
            if (sort) { samp[k] = v; bel[k] = below; }
            else {
                z_out[k] = v;
                if (below_out) below_out[k] = below;
                if (above_out) above_out[k] = above;
            }
        }
    }
    if (!sort) return;
    lds_wave_sync();
    // Sort.  The inverse CDF is monotone, so the order of the samples is the order of their uniforms: bucket the
    // elements by floor(u * 256) (LDS atomics hand out the slot inside a bucket), prefix-sum the bucket counts, and rank
    // every element against the <= SORT_CAP members of its own bucket only.  Adversarial inputs (a bucket holding more
    // than SORT_CAP elements) fall back to the O(K^2) stable rank sort; both give the same sorted values.
    int* cnt = sortbuf;
    int* pre = sortbuf + SORT_NB;
    unsigned short* mem = reinterpret_cast<unsigned short*>(sortbuf + 2 * SORT_NB);
    for (int b = lane; b < SORT_NB; b += 64) cnt[b] = 0;
    lds_wave_sync();
    // (the order-of-uniforms argument needs ascending bin edges; a ray whose edges are not -- unsorted depths passed to inverseSample,
    // or stratified depths whose jitter exceeds the bin spacing -- takes the rank sort below, which sorts the VALUES like torch.sort)
    int overflow = 0;
    for (int j = lane; j + 1 < nb; j += 64) overflow |= (bins[j + 1] < bins[j]) ? 1 : 0;
    for (int k = lane, i = 0; k < K; k += 64, ++i) {
        int b = (int)(uf(i, k) * (float)SORT_NB);
        b = b < 0 ? 0 : (b > SORT_NB - 1 ? SORT_NB - 1 : b);
        const int pos = atomicAdd(&cnt[b], 1);
        if (pos < SORT_CAP) mem[b * SORT_CAP + pos] = (unsigned short)k; else overflow = 1;
    }
    lds_wave_sync();
    if (__any(overflow)) {
        for (int k = lane; k < K; k += 64) {          // stable rank sort (LDS broadcast reads)
            const float v = samp[k];
            int rank = 0;
            for (int j = 0; j < K; ++j) {
                const float o = samp[j];
                rank += (o < v || (o == v && j < k)) ? 1 : 0;
            }
            z_out[rank] = v;
            if (below_out) below_out[rank] = bel[k];
        }
        return;
    }
    {   // exclusive prefix sum of the SORT_NB bucket counts: SORT_NB / 64 per lane (128 x 8, 128 x 6 and 64 x 12 layouts measured slower: more rank work per bucket than the extra occupancy pays)
        int c[SORT_PER_LANE], tot = 0;
#pragma unroll
        for (int i = 0; i < SORT_PER_LANE; ++i) { c[i] = cnt[lane * SORT_PER_LANE + i]; tot += c[i]; }
        int run = wave_incl_scan_add(tot) - tot;
#pragma unroll
        for (int i = 0; i < SORT_PER_LANE; ++i) { pre[lane * SORT_PER_LANE + i] = run; run += c[i]; }
    }
    lds_wave_sync();
    for (int k0 = 0, i0 = 0; k0 < K; k0 += 192, i0 += 3) {          // three elements per lane in lock step (dependent LDS reads overlap)
        float v[3]; int rank[3], cntb[3], bb[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = k0 + 64 * q + lane;
            const bool ok = k < K;
            int b = ok ? (int)(uf(i0 + q, k) * (float)SORT_NB) : 0;
            b = b < 0 ? 0 : (b > SORT_NB - 1 ? SORT_NB - 1 : b);
            bb[q] = b;
            v[q] = samp[ok ? k : 0];
            rank[q] = pre[b];
            cntb[q] = ok ? cnt[b] : 0;
        }
        int most = cntb[0] > cntb[1] ? cntb[0] : cntb[1];
        most = most > cntb[2] ? most : cntb[2];
        const int trips = wave_max_i(most);                       // wave-uniform: the fullest bucket any lane looks at (typically 3..4 of SORT_CAP)
        for (int p = 0; p < trips; ++p) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int k = k0 + 64 * q + lane;
                const bool live = p < cntb[q];
                const int jx = mem[bb[q] * SORT_CAP + p];
                const float o = samp[live ? jx : 0];
                rank[q] += (live && (o < v[q] || (o == v[q] && jx < k))) ? 1 : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = k0 + 64 * q + lane;
            if (k >= K) continue;
            z_out[rank[q]] = v[q];
            if (below_out) below_out[rank[q]] = bel[k];
        }
    }
}

// LDS floats per wave: pw[C] bins[C] cdf[C] samp[K] bel[K] sortbuf
DEVINL size_t inv_lds_floats(int C, int K) { return (size_t)3 * C + 2 * K + SORT_LDS_FLOATS; }

// mode 0: inverseSample(weights (N,C), depths (N,C)) -> bins = mid-points, pdf = weights[1:-1]   (utils.py:34-44)
// mode 1: sample_pdf(bins (N,C), weights (N,C-1))                                               (utils.py:108-133)
__global__ __launch_bounds__(256) void inverse_sample_kernel(const float* __restrict__ w, const float* __restrict__ z,
                                                            const float* __restrict__ u, int64_t N, int C, int K, int sort,
                                                            int mode, float* __restrict__ z_out, int64_t* __restrict__ below,
                                                            int64_t* __restrict__ above) {
    float* base = reinterpret_cast<float*>(smem) + wave_in_block() * inv_lds_floats(C, K);
    float* pw = base; float* bins = pw + C; float* cdf = bins + C; float* samp = cdf + C;
    int* bel = reinterpret_cast<int*>(samp + K);
    int* sortbuf = bel + K;
    const int lane = lane_id();
    const int nw = mode == 0 ? C - 2 : C - 1;
    const int wstride = mode == 0 ? C : C - 1;
    for (int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); n < N; n += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        lds_wave_sync();
        if (mode == 0) {
            for (int j = lane; j < nw; j += 64) pw[j] = w[n * wstride + 1 + j];
            for (int j = lane; j < nw + 1; j += 64) bins[j] = 0.5f * (z[n * C + j + 1] + z[n * C + j]);
        } else {
            for (int j = lane; j < nw; j += 64) pw[j] = w[n * wstride + j];
            for (int j = lane; j < nw + 1; j += 64) bins[j] = z[n * C + j];
        }
        lds_wave_sync();
        const float* un = u + n * K;
        wave_inverse_sample(pw, bins, nw, cdf, samp, bel, sortbuf, [&](int, int k) { return un[k]; }, K, sort, z_out + n * K,
                            below ? below + n * K : nullptr, above ? above + n * K : nullptr);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused proposal resampling: density -> weights -> max-blur -> inverse sampling (procedures.py:68-70).
// ------------------------------------------------------------------------------------------------
struct ResampleArgs {
    const float* density; const float* z; const float* z_base; const float* u_strat; float z_jitter;
    const float* dirs; int dirs_stride; const float* u_inv; int64_t N; int C; int K; int softplus; float alpha;
    float* z_fine; int64_t* below; float* w_prop; float* z_coarse;
    uint64_t rng_seed; int64_t rng_ray_offset;      // Philox source of u_strat / u_inv when the pointers are NULL (device_common.h)
};

#ifndef RS_BLOCKS_PER_CU
#define RS_BLOCKS_PER_CU 5      /* measured: 3 -> 1.54 ms, 4 -> 1.29 ms, 5 -> 1.20 ms per 640 000 rays (96 VGPRs, 13 spilled; LDS fits 5 at the render shapes) */
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RS_BLOCKS_PER_CU, RS_BLOCKS_PER_CU))) void resample_kernel(ResampleArgs a) {
    const int C = a.C, K = a.K;
    // per wave: pw[C] bins[C] cdf[C] samp[K] bel[K] | zl[C] wraw[C]
    float* base = reinterpret_cast<float*>(smem) + wave_in_block() * (inv_lds_floats(C, K) + 2 * C);
    float* pw = base; float* bins = pw + C; float* cdf = bins + C; float* samp = cdf + C;
    int* bel = reinterpret_cast<int*>(samp + K);
    int* sortbuf = bel + K;
    float* zl = reinterpret_cast<float*>(sortbuf + SORT_LDS_FLOATS);
    float* wraw = zl + C;
    const int lane = lane_id();
    // All global inputs of a ray (direction, depths or their uniforms, densities, the K inverse-CDF uniforms) are loaded ONE RAY
    // AHEAD into registers: the ray's own phases then never wait for HBM (three exposed round trips per ray before).
    struct RayIn { float dx, dy, dz, zv0, zv1, dens0, dens1, u0, u1, u2; };      // scalars only: indexed members end up in scratch
    const bool fits = C <= 128 && K <= 192;
    auto load_in = [&](int64_t n) -> RayIn {
        RayIn r;
        const float* dd = a.dirs + n * a.dirs_stride;
        r.dx = dd[0]; r.dy = dd[1]; r.dz = dd[2];
        // Philox mode: lane j's one block of the ray carries u_strat(n, j) and u_inv(n, j + {0, 64, 128})   (device_common.h)
        Philox4 blk = {{0u, 0u, 0u, 0u}};
        const bool draw = !a.u_inv || (!a.z && !a.u_strat);
        if (draw) blk = philox_ray_block(a.rng_seed, n + a.rng_ray_offset, lane);
        auto depth_of = [&](int j, bool first) {
            if (a.z) return a.z[n * C + j];
            const float uu = a.u_strat ? a.u_strat[n * C + j] : (first ? u01_from_bits(blk.w[0]) : philox_u_strat(a.rng_seed, n + a.rng_ray_offset, j));
            return a.z_base[j] + uu * a.z_jitter;
        };
        r.zv0 = r.zv1 = r.dens0 = r.dens1 = 0.0f;
        if (lane < C) { r.zv0 = depth_of(lane, true); r.dens0 = a.density[n * C + lane]; }
        if (lane + 64 < C) { r.zv1 = depth_of(lane + 64, false); r.dens1 = a.density[n * C + lane + 64]; }
        if (a.u_inv) {
            const float* un = a.u_inv + n * K;
            r.u0 = (lane < K) ? un[lane] : 0.0f;
            r.u1 = (lane + 64 < K) ? un[lane + 64] : 0.0f;
            r.u2 = (lane + 128 < K) ? un[lane + 128] : 0.0f;
        } else {
            r.u0 = u01_from_bits(blk.w[1]); r.u1 = u01_from_bits(blk.w[2]); r.u2 = u01_from_bits(blk.w[3]);
        }
        return r;
    };
    const int64_t stride = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block();
    if (n >= a.N) return;
    RayIn cur = {};
    if (fits) cur = load_in(n);
    for (; n < a.N; n += stride) {
        RayIn nxt = cur;
        if (fits && n + stride < a.N) nxt = load_in(n + stride);
        lds_wave_sync();
        float nrm;
        if (fits) {
            nrm = norm3(cur.dx, cur.dy, cur.dz);
            if (lane < C) { zl[lane] = cur.zv0; if (a.z_coarse) a.z_coarse[n * C + lane] = cur.zv0; }
            if (lane + 64 < C) { zl[lane + 64] = cur.zv1; if (a.z_coarse) a.z_coarse[n * C + lane + 64] = cur.zv1; }
        } else {
            const float* dd = a.dirs + n * a.dirs_stride;
            nrm = norm3(dd[0], dd[1], dd[2]);
            for (int j = lane; j < C; j += 64) {
                const float zv = a.z ? a.z[n * C + j]
                                     : (a.z_base[j] + (a.u_strat ? a.u_strat[n * C + j] : philox_u_strat(a.rng_seed, n + a.rng_ray_offset, j)) * a.z_jitter);
                zl[j] = zv;
                if (a.z_coarse) a.z_coarse[n * C + j] = zv;
            }
        }
        lds_wave_sync();
        const float* sg = a.density + n * C;
        const int soft = a.softplus;
        const float dens0 = cur.dens0, dens1 = cur.dens1, cu0 = cur.u0, cu1 = cur.u1, cu2 = cur.u2;   // by-value captures: a reference to `cur` pins it in scratch
        wave_sigma_to_weights(C, NERF_AMD_ACT_RELU,
                              [=](int s) { const float d = fits ? ((s >> 6) ? dens1 : dens0) : sg[s]; return soft ? softplus_f(d) : d; },
                              [=](int s) { return zl[s] * nrm; },
                              [=](int s, float wv, float) { wraw[s] = wv; });
        lds_wave_sync();
        for (int j = lane; j < C; j += 64) {                                   // max-blur (mip_methods.py:61-66)
            const float c = wraw[j];
            const float front = (j == 0) ? c : fmaxf(wraw[j - 1], c);
            const float rear = (j == C - 1) ? c : fmaxf(c, wraw[j + 1]);
            const float wb = 0.5f * (front + rear) + a.alpha;
            if (j >= 1 && j <= C - 2) pw[j - 1] = wb;                          // pdf over weights[1:-1]
            if (a.w_prop) a.w_prop[n * C + j] = wb;
        }
        for (int j = lane; j < C - 1; j += 64) bins[j] = 0.5f * (zl[j + 1] + zl[j]);   // mid-points of the RAW depths
        lds_wave_sync();
        const float* un = a.u_inv ? a.u_inv + n * K : nullptr;
        const uint64_t seed = a.rng_seed; const int64_t nn = n + a.rng_ray_offset;
        wave_inverse_sample(pw, bins, C - 2, cdf, samp, bel, sortbuf,
                            [=](int i, int k) {
                                if (fits) return i == 0 ? cu0 : (i == 1 ? cu1 : cu2);
                                if (un) return un[k];
                                return philox_u_inv(seed, nn, k);                                  // generic shapes
                            },
                            K, 1, a.z_fine + n * K, a.below ? a.below + n * K : nullptr, nullptr);
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------- row 10
struct CompositeArgs {
    const float* rgbo; const float* z; int z_stride; const float* dirs; int dirs_stride; int64_t N; int S;
    int flags; int act; float sigma_shift; float near, far; const float* normal; const float* cam_dir;
    float* rgb; float* weights; float* depth; float* normal_img;
};

// Fast path of the compositing kernel for S <= 64 NCH (NCH = 2 or 4) without normals (the render path): a ray is NCH 64-lane chunks;
// all of its loads (one 16-byte rgbo record and one z per lane and chunk) are issued up front -- the neighbour z of the transmittance
// step comes from a lane shuffle instead of a second load -- and the NEXT ray of this wavefront is loaded before the current
// one is reduced, so two rays' worth of HBM requests are in flight per wave.  Same arithmetic, same order as the generic path.
template <int NCH>
struct RayRecs { f32x4 c[NCH]; float z[NCH]; float dx, dy, dz; };
template <int NCH>
DEVINL RayRecs<NCH> load_ray(const CompositeArgs& a, int64_t n, int lane) {
    RayRecs<NCH> r;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const f32x4* px = reinterpret_cast<const f32x4*>(a.rgbo) + n * a.S;
    const float* zz = a.z + n * a.z_stride;
    const float* dd = a.dirs + n * a.dirs_stride;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int s = 64 * k + lane;
        r.c[k] = (s < a.S) ? px[s] : zero;
        r.z[k] = (s < a.S) ? zz[s] : 0.0f;
    }
    r.dx = dd[0]; r.dy = dd[1]; r.dz = dd[2];
    return r;
}
template <int NCH>
DEVINL void composite_ray_fast(const CompositeArgs& a, int64_t n, const RayRecs<NCH>& r, int lane) {
    const int S = a.S;
    const bool mul = (a.flags & 1) != 0;
    const float nrm = mul ? norm3(r.dx, r.dy, r.dz) : 1.0f;
    float zn[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) zn[k] = mul ? r.z[k] * nrm : r.z[k];
    float accr = 0.0f, accg = 0.0f, accb = 0.0f, accw = 0.0f, accd = 0.0f;
    double carry = 1.0;
    float* wout = a.weights ? a.weights + n * S : nullptr;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (k > 0 && 64 * k >= S) break;                        // wave-uniform
        const int s = 64 * k + lane;
        float nx = __shfl_down(zn[k], 1, 64);                   // z of sample s + 1: the next lane, or lane 0 of the next chunk
        if (k + 1 < NCH) { const float first = __shfl(zn[k + 1 < NCH ? k + 1 : k], 0, 64); if (lane == 63) nx = first; }
        float w = 0.0f;
        double p = 1.0;
        if (s < S) {
            const float delta = (s + 1 < S) ? (nx - zn[k]) : 1e10f;
            const float m = expf(-density_act(r.c[k][3] + a.sigma_shift, a.act) * delta);
            w = 1.0f - m; p = (double)(m + 1e-10f);
        }
        const double incl = wave_incl_scan_mul(p);
        const double excl = wave_shift_up1(incl, 1.0);
        w *= (float)(carry * excl);
        carry *= wave_last(incl);
        if (s < S) {
            accr += w * r.c[k][0]; accg += w * r.c[k][1]; accb += w * r.c[k][2]; accw += w; accd += w * zn[k];
            if (wout) wout[s] = w;
        }
    }
    accr = wave_sum(accr); accg = wave_sum(accg); accb = wave_sum(accb); accw = wave_sum(accw); accd = wave_sum(accd);
    if (lane == 0) {
        if (a.flags & 2) { const float bg = 1.0f - accw; accr += bg; accg += bg; accb += bg; }
        a.rgb[n * 3] = accr; a.rgb[n * 3 + 1] = accg; a.rgb[n * 3 + 2] = accb;
        if (a.depth) a.depth[n] = (accd - a.near) / (a.far - a.near);
    }
}
template <int NCH>
DEVINL void composite_rays_fast(const CompositeArgs& a, int lane) {
    const int64_t stride = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block();
    if (n >= a.N) return;
    RayRecs<NCH> cur = load_ray<NCH>(a, n, lane);
    for (; n < a.N; n += stride) {
        RayRecs<NCH> nxt = cur;
        if (n + stride < a.N) nxt = load_ray<NCH>(a, n + stride, lane);
        composite_ray_fast<NCH>(a, n, cur, lane);
        cur = nxt;
    }
}

__global__ __launch_bounds__(256) void composite_kernel(CompositeArgs a) {
    const int S = a.S;
    const int lane = lane_id();
    if (S <= 256 && a.normal == nullptr && a.normal_img == nullptr) {
        if (S <= 128) composite_rays_fast<2>(a, lane);
        else composite_rays_fast<4>(a, lane);
        return;
    }
    for (int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); n < a.N; n += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        const float* dd = a.dirs + n * a.dirs_stride;
        const bool mul = (a.flags & 1) != 0;
        const float nrm = mul ? norm3(dd[0], dd[1], dd[2]) : 1.0f;
        const float* zz = a.z + n * a.z_stride;
        const f32x4* px = reinterpret_cast<const f32x4*>(a.rgbo) + n * S;
        float accr = 0.0f, accg = 0.0f, accb = 0.0f, accw = 0.0f, accd = 0.0f, accn = 0.0f;
        float cx = 0.0f, cy = 0.0f, cz = 0.0f;
        if (a.normal_img) { cx = a.cam_dir[0]; cy = a.cam_dir[1]; cz = a.cam_dir[2]; }
        float* wout = a.weights ? a.weights + n * S : nullptr;
        const float* nm = a.normal ? a.normal + n * S * 3 : nullptr;
        const float shift = a.sigma_shift;
        wave_sigma_to_weights(S, a.act, [&](int s) { return px[s][3] + shift; },
                              [&](int s) { return mul ? zz[s] * nrm : zz[s]; },
                              [&](int s, float w, float zn) {
                                  const f32x4 c = px[s];
                                  accr += w * c[0]; accg += w * c[1]; accb += w * c[2];
                                  accw += w; accd += w * zn;
                                  if (nm) accn += w * ((nm[s * 3] * cx + nm[s * 3 + 1] * cy) + nm[s * 3 + 2] * cz);
                                  if (wout) wout[s] = w;
                              });
        accr = wave_sum(accr); accg = wave_sum(accg); accb = wave_sum(accb); accw = wave_sum(accw);
        accd = wave_sum(accd);
        if (a.normal_img) accn = wave_sum(accn);
        if (lane == 0) {
            if (a.flags & 2) { const float bg = 1.0f - accw; accr += bg; accg += bg; accb += bg; }
            a.rgb[n * 3] = accr; a.rgb[n * 3 + 1] = accg; a.rgb[n * 3 + 2] = accb;
            if (a.depth) a.depth[n] = (accd - a.near) / (a.far - a.near);
            if (a.normal_img) a.normal_img[n] = (accn + 1.0f) * 0.5f;
        }
    }
}

// ---------------------------------------------------------------------------------------- row 11
__global__ __launch_bounds__(256) void get_bounds_kernel(const float* __restrict__ w, const int64_t* __restrict__ below,
                                                        int64_t N, int C, int K, float* __restrict__ bounds) {
    float* sat = reinterpret_cast<float*>(smem) + wave_in_block() * (C + 1);
    const int lane = lane_id();
    for (int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); n < N; n += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        lds_wave_sync();
        double carry = 0.0;
        if (lane == 0) sat[0] = 0.0f;
        for (int base = 0; base < C; base += 64) {
            const int j = base + lane;
            const double p = (j < C) ? (double)w[n * C + j] : 0.0;
            const double incl = wave_incl_scan_add(p);
            if (j < C) sat[1 + j] = (float)(carry + incl);
            carry += wave_last(incl);
        }
        lds_wave_sync();
        const int64_t* bl = below + n * K;
        for (int k = lane; k < K - 1; k += 64) {
            const int st = (int)bl[k], en = (int)bl[k + 1] + 1;
            bounds[n * (K - 1) + k] = sat[en] - sat[st];
        }
    }
}

// ================================================================================================
// Backward kernels of the sampling / compositing rows: one wavefront per ray, the transmittance
// product's adjoint is a suffix sum.  Up to 1 024 samples per ray (4 / 8 / 16 register chunks).
//   forward:  m_i = exp(-act(sigma_i + shift) * delta_i),  q_i = m_i + 1e-10,  T_i = prod_{j<i} q_j,  w_i = (1 - m_i) T_i
//   given G_i = dL/dw_i:   dL/dm_j = -G_j T_j + (sum_{i>j} G_i w_i) / q_j,   dL/dsigma_j = dL/dm_j * (-delta_j m_j) * act'(sigma_j + shift)
// ================================================================================================
// Rows of any length up to 1 024 samples (round 5: `-t --fine_sample_pnum 256` merges 320 samples per ray, nerf_base.py:91-113 has no limit):
// the kernel is instantiated for 4 / 8 / 16 register chunks of 64 samples (7 registers per chunk) and the launcher picks by S.
constexpr int BWD_MAX_CHUNKS = 16;
struct WeightsBwdArgs {
    const float* sigma; int sigma_stride;      // sigma of sample s of ray n at sigma[(n*S + s) * sigma_stride + sigma_off]
    int sigma_off;
    const float* z; int z_stride; const float* dirs; int dirs_stride; int64_t N; int S; int mul_norm; int act; float sigma_shift;
    // composite extras (rgbo != nullptr): colours come from rgbo[(n*S+s)*4 + 0..2]
    const float* rgbo; const float* d_rgb; const float* d_weights; const float* d_depth; int white_bkg; float near, far;
    float* d_sigma; int d_sigma_stride; int d_sigma_off; float* d_rgbo;
};
DEVINL float density_act_grad(float x, int act) {
    if (act == 0) return x > 0.0f ? 1.0f : 0.0f;
    if (act == 2) return x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));
    return 1.0f;
}
template <int BWD_CHUNKS>
__global__ __launch_bounds__(256) void weights_backward_kernel(WeightsBwdArgs a) {
    const int S = a.S;
    const int lane = lane_id();
    for (int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); n < a.N; n += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        float nrm = 1.0f;
        if (a.mul_norm && a.dirs) { const float* dd = a.dirs + n * a.dirs_stride; nrm = norm3(dd[0], dd[1], dd[2]); }
        const float* zz = a.z + n * a.z_stride;
        float dr = 0.0f, dg = 0.0f, db = 0.0f, dd_ = 0.0f;
        if (a.d_rgb) { dr = a.d_rgb[n * 3]; dg = a.d_rgb[n * 3 + 1]; db = a.d_rgb[n * 3 + 2]; }
        if (a.d_depth) dd_ = a.d_depth[n] / (a.far - a.near);
        const float bg = (a.white_bkg && a.d_rgb) ? ((dr + dg) + db) : 0.0f;       // rgb += 1 - sum w
        float m[BWD_CHUNKS], T[BWD_CHUNKS], G[BWD_CHUNKS], delta[BWD_CHUNKS], x[BWD_CHUNKS];
        double carry = 1.0, gw_carry = 0.0;
        double P[BWD_CHUNKS];                                                       // inclusive prefix of G_i w_i
#pragma unroll
        for (int c = 0; c < BWD_CHUNKS; ++c) {
            const int s = c * 64 + lane;
            const bool ok = s < S;
            m[c] = 1.0f; T[c] = 0.0f; G[c] = 0.0f; delta[c] = 0.0f; x[c] = 0.0f;
            double p = 1.0;
            float zn0 = 0.0f;
            if (c * 64 < S) {
                if (ok) {
                    zn0 = zz[s] * nrm;
                    delta[c] = (s + 1 < S) ? (zz[s + 1] * nrm - zn0) : 1e10f;
                    x[c] = a.sigma[(n * S + s) * a.sigma_stride + a.sigma_off] + a.sigma_shift;
                    m[c] = expf(-density_act(x[c], a.act) * delta[c]);
                    p = (double)(m[c] + 1e-10f);
                }
                const double incl = wave_incl_scan_mul(p);
                const double excl = wave_shift_up1(incl, 1.0);
                T[c] = (float)(carry * excl);
                carry *= wave_last(incl);
                float g = 0.0f;
                if (ok) {
                    if (a.d_weights) g += a.d_weights[n * S + s];
                    if (a.rgbo) {
                        const float* cc = a.rgbo + (n * S + s) * 4;
                        g += ((dr * cc[0] + dg * cc[1]) + db * cc[2]) - bg;
                    }
                    g += dd_ * zn0;
                }
                G[c] = g;
                const float w = (1.0f - m[c]) * T[c];
                const double gi = wave_incl_scan_add(ok ? (double)(g * w) : 0.0);
                P[c] = gw_carry + gi;
                gw_carry += wave_last(gi);
                if (ok && a.d_rgbo) {                                               // dL/dc_i = w_i * d_rgb
                    float* o = a.d_rgbo + (n * S + s) * 4;
                    o[0] = w * dr; o[1] = w * dg; o[2] = w * db;
                }
            }
        }
        const double total = gw_carry;
#pragma unroll
        for (int c = 0; c < BWD_CHUNKS; ++c) {
            const int s = c * 64 + lane;
            if (s < S) {
                const float suffix = (float)(total - P[c]);                         // sum_{i>s} G_i w_i
                const float dm = -G[c] * T[c] + suffix / (m[c] + 1e-10f);
                const float ds = dm * (-delta[c] * m[c]) * density_act_grad(x[c], a.act);
                a.d_sigma[(n * S + s) * a.d_sigma_stride + a.d_sigma_off] = ds;
            }
        }
    }
}

// maxBlurFilter backward (mip_methods.py:61-66): torch.maximum sends the gradient to the larger argument, half to each on a tie
__global__ void max_blur_backward_kernel(const float* __restrict__ w, const float* __restrict__ g, int64_t N, int S, float* __restrict__ dw) {
    const int64_t total = N * S;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % S);
        const float c = w[i];
        float acc = 0.0f;
        if (s == 0) acc += 0.5f * g[i];
        if (s == S - 1) acc += 0.5f * g[i];
        if (s + 1 < S) {                                    // mx_s = max(w_s, w_{s+1}) feeds rear_s and front_{s+1}
            const float o = w[i + 1];
            const float share = c > o ? 1.0f : (c == o ? 0.5f : 0.0f);
            acc += share * (0.5f * g[i] + 0.5f * g[i + 1]);
        }
        if (s >= 1) {                                       // mx_{s-1} = max(w_{s-1}, w_s) feeds rear_{s-1} and front_s
            const float o = w[i - 1];
            const float share = c > o ? 1.0f : (c == o ? 0.5f : 0.0f);
            acc += share * (0.5f * g[i - 1] + 0.5f * g[i]);
        }
        dw[i] = acc;
    }
}

// getBounds backward (addtional.py:14-18): bounds_k = sat[below_{k+1} + 1] - sat[below_k] = +sum of w over [below_k, below_{k+1}] (or minus the
// sum over the gap when the indices are not ascending)  ->  dw_j = sum_k g_k * ([st_k <= j < en_k] - [en_k <= j < st_k]).
// Every lane gathers its own j in ascending k: no atomics, reproducible.
__global__ __launch_bounds__(256) void get_bounds_backward_kernel(const int64_t* __restrict__ below, const float* __restrict__ g, int64_t N, int C,
                                                                  int K, float* __restrict__ dw) {
    int* bl_lds = reinterpret_cast<int*>(smem) + wave_in_block() * (2 * K);
    float* g_lds = reinterpret_cast<float*>(bl_lds + K);
    const int lane = lane_id();
    for (int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); n < N; n += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        lds_wave_sync();
        for (int k = lane; k < K; k += 64) {
            int b = (int)below[n * K + k];
            bl_lds[k] = b < 0 ? 0 : (b > C ? C : b);
            g_lds[k] = (k < K - 1) ? g[n * (K - 1) + k] : 0.0f;
        }
        lds_wave_sync();
        for (int j = lane; j < C; j += 64) {
            float acc = 0.0f;
            for (int k = 0; k < K - 1; ++k) {
                const int st = bl_lds[k];
                int en = bl_lds[k + 1] + 1;
                en = en > C ? C : en;
                if (j >= st && j < en) acc += g_lds[k];
                else if (j >= en && j < st) acc -= g_lds[k];
            }
            dw[n * C + j] = acc;
        }
    }
}

// Training dump (mlp_kernels.hip ActDump) -> row-major activations.  Block (subtile, K group) of the dump holds, for lane
// (h = lane >> 5, j = lane & 31), the 8 features 16 kg + 8 (e >> 2) + 4 h + (e & 3), e = 0..7, of sample 32 subtile + j
// (mlp_layout.h dmap_feature): two runs of four consecutive features.  ELEM = 2 (bf16) or 4 (fp32, halves 1 KiB apart).
template <int ELEM>
__global__ __launch_bounds__(256) void frag_to_rows_kernel(const char* __restrict__ frag, int64_t n_sub, int n_kg, int64_t M,
                                                           char* __restrict__ out, int ld) {
    // One wavefront per subtile: the 32 x (16 n_kg) tile is transposed through LDS (row pitch padded by 16 bytes against bank
    // conflicts), so that the global side is 16-byte lane-linear on the way in AND on the way out (the subtile's 32 rows are one
    // contiguous block of the row-major matrix).
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    constexpr int BLOCK = 512 * ELEM;                           // bytes per (subtile, K group)
    const int row_bytes = n_kg * 16 * ELEM;
    const int pitch = row_bytes + 16;
    char* tile = smem + (size_t)wave_in_block() * 32 * pitch;
    for (int64_t sub = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); sub < n_sub; sub += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        lds_wave_sync();
        char* trow = tile + j * pitch;
        for (int kg = 0; kg < n_kg; ++kg) {
            const char* blk = frag + ((size_t)sub * 16 + kg) * BLOCK;
            if constexpr (ELEM == 2) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(blk + lane * 16);          // 8 bf16
                f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
                *reinterpret_cast<f32x2*>(trow + (16 * kg + 4 * h) * 2) = lo;
                *reinterpret_cast<f32x2*>(trow + (16 * kg + 8 + 4 * h) * 2) = hi;
            } else {
                *reinterpret_cast<f32x4*>(trow + (16 * kg + 4 * h) * 4) = *reinterpret_cast<const f32x4*>(blk + lane * 16);
                *reinterpret_cast<f32x4*>(trow + (16 * kg + 8 + 4 * h) * 4) = *reinterpret_cast<const f32x4*>(blk + 1024 + lane * 16);
            }
        }
        lds_wave_sync();
        const int chunks_per_row = row_bytes / 16;
        const int total = 32 * chunks_per_row;
        char* obase = out + (size_t)sub * 32 * ld * ELEM;
        for (int c = lane; c < total; c += 64) {
            const int r = c / chunks_per_row, q = c - r * chunks_per_row;
            if (sub * 32 + r < M)
                *reinterpret_cast<f32x4*>(obase + (size_t)r * ld * ELEM + q * 16) = *reinterpret_cast<const f32x4*>(tile + r * pitch + q * 16);
        }
    }
}

// frag_to_rows_kernel + relu_mask_bias_kernel in one pass over a layer: the subtile's activations go dump -> LDS -> row-major
// `act_out`, and while they sit in LDS the same 32 rows of `delta` (row-major, same shape) are masked in place by [act > 0] and
// summed per column.  Saves re-reading the activation matrix (4 instead of 5 matrix passes per layer).  Column sums: lane c owns
// 16-byte chunk c % cpr of rows c / cpr + (64 / cpr) i, i.e. a fixed set of columns; each wavefront writes ONE partial row of
// fp32 sums (deterministic, no atomics), `col_sum` has gridDim.x * WAVES_PER_BLOCK rows.
template <int ELEM>
__global__ __launch_bounds__(256) void frag_rows_mask_kernel(const char* __restrict__ frag, int64_t n_sub, int n_kg, int64_t M,
                                                             char* __restrict__ act_out, char* __restrict__ delta, float* __restrict__ col_sum) {
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    constexpr int BLOCK = 512 * ELEM;
    constexpr int NV = 16 / ELEM;                                // values per 16-byte chunk
    const int row_bytes = n_kg * 16 * ELEM;
    const int pitch = row_bytes + 16;
    const int ld = n_kg * 16;
    char* tile = smem + (size_t)wave_in_block() * 32 * pitch;
    const int cpr = row_bytes / 16;                              // 16, 32 or 64: divides the wave size
    const int q = lane % cpr, r0 = lane / cpr, rstep = 64 / cpr;
    float acc[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[e] = 0.0f;
    for (int64_t sub = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); sub < n_sub; sub += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        lds_wave_sync();
        char* trow = tile + j * pitch;
        for (int kg = 0; kg < n_kg; ++kg) {
            const char* blk = frag + ((size_t)sub * 16 + kg) * BLOCK;
            if constexpr (ELEM == 2) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(blk + lane * 16);
                f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
                *reinterpret_cast<f32x2*>(trow + (16 * kg + 4 * h) * 2) = lo;
                *reinterpret_cast<f32x2*>(trow + (16 * kg + 8 + 4 * h) * 2) = hi;
            } else {
                *reinterpret_cast<f32x4*>(trow + (16 * kg + 4 * h) * 4) = *reinterpret_cast<const f32x4*>(blk + lane * 16);
                *reinterpret_cast<f32x4*>(trow + (16 * kg + 8 + 4 * h) * 4) = *reinterpret_cast<const f32x4*>(blk + 1024 + lane * 16);
            }
        }
        lds_wave_sync();
        const size_t base = (size_t)sub * 32 * ld * ELEM;
        for (int r = r0; r < 32; r += rstep) {
            if (sub * 32 + r >= M) break;
            const size_t off = base + (size_t)r * ld * ELEM + q * 16;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            if constexpr (ELEM == 4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(tile + r * pitch + q * 16);
                f32x4 d = *reinterpret_cast<const f32x4*>(delta + off);
                *reinterpret_cast<f32x4*>(act_out + off) = a;
#pragma unroll
                for (int e = 0; e < 4; ++e) { if (!(a[e] > 0.0f)) d[e] = 0.0f; acc[e] += d[e]; }
                *reinterpret_cast<f32x4*>(delta + off) = d;
            } else {
                const u32x4 a = *reinterpret_cast<const u32x4*>(tile + r * pitch + q * 16);
                const u32x4 d = *reinterpret_cast<const u32x4*>(delta + off);
                *reinterpret_cast<u32x4*>(act_out + off) = a;
                uint32_t w[4] = {d[0], d[1], d[2], d[3]};
                const uint32_t aw[4] = {a[0], a[1], a[2], a[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = aw[e] & 0xFFFFu, hi = aw[e] >> 16;
                    if (!(lo != 0u && lo < 0x8000u)) w[e] &= 0xFFFF0000u;
                    if (!(hi != 0u && hi < 0x8000u)) w[e] &= 0x0000FFFFu;
                    acc[2 * e] += __builtin_bit_cast(float, w[e] << 16);
                    acc[2 * e + 1] += __builtin_bit_cast(float, w[e] & 0xFFFF0000u);
                }
                const u32x4 o = {w[0], w[1], w[2], w[3]};
                *reinterpret_cast<u32x4*>(delta + off) = o;
            }
        }
    }
    // lanes with equal q (lane bits >= log2 cpr) hold sums of the same columns: fold them, then lanes 0 .. cpr-1 write the partial row
    for (int m = cpr; m < 64; m <<= 1) {
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[e] += __shfl_xor(acc[e], m, 64);
    }
    if (lane < cpr) {
        float* prow = col_sum + ((size_t)blockIdx.x * WAVES_PER_BLOCK + wave_in_block()) * (size_t)ld + q * NV;
#pragma unroll
        for (int e = 0; e < NV; ++e) prow[e] = acc[e];
    }
}

// First-layer / skip-layer operand of the wgrad GEMMs: row m = [x | positional_encoding_L(x) | 0 pad] (nerf_helper.py:38-48 order:
// per octave k the three sines, then the three cosines), NCOL = 3 + 6 L rounded up to 8, as bf16 (ELEM 2) or fp32 (ELEM 4).
// `normalize` divides x by its norm first (the view direction of mip_model.py:52).  bf16 rows take octave 0 from sincosf and the
// higher octaves by angle doubling like the forward kernels (error <= 2^9 * 1e-7, far below a bf16 ulp); fp32 rows call sincosf
// per octave.
template <int L, int ELEM>
__global__ __launch_bounds__(256) void encode_rows_kernel(const float* __restrict__ x, int x_stride, int64_t M, int normalize,
                                                          char* __restrict__ out) {
    constexpr int NCOL = (3 + 6 * L + 7) / 8 * 8;
    for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
        const float* px = x + m * x_stride;
        float c[3] = {px[0], px[1], px[2]};
        if (normalize) { const float n = norm3(c[0], c[1], c[2]); c[0] /= n; c[1] /= n; c[2] /= n; }
        float v[NCOL];
#pragma unroll
        for (int q = 0; q < NCOL; ++q) v[q] = 0.0f;
        float sv[3], cv[3];
#pragma unroll
        for (int k = 0; k < L; ++k) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (ELEM == 4 || k == 0) sincosf(c[i] * (float)(1 << k), &sv[i], &cv[i]);
                else { const float s2 = 2.0f * sv[i]; const float ns = s2 * cv[i]; cv[i] = __builtin_fmaf(-s2, sv[i], 1.0f); sv[i] = ns; }
                v[3 + 6 * k + i] = sv[i];
                v[3 + 6 * k + 3 + i] = cv[i];
            }
        }
        v[0] = c[0]; v[1] = c[1]; v[2] = c[2];
        char* o = out + (size_t)m * NCOL * ELEM;
        if constexpr (ELEM == 4) {
#pragma unroll
            for (int q = 0; q < NCOL; q += 4) { f32x4 w = {v[q], v[q + 1], v[q + 2], v[q + 3]}; *reinterpret_cast<f32x4*>(o + q * 4) = w; }
        } else {
#pragma unroll
            for (int q = 0; q < NCOL; q += 8) {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = __builtin_bit_cast(uint32_t, v[q + 2 * e]), hi = __builtin_bit_cast(uint32_t, v[q + 2 * e + 1]);
                    const uint32_t rl = (lo + 0x7fffu + ((lo >> 16) & 1u)) >> 16, rh = (hi + 0x7fffu + ((hi >> 16) & 1u)) >> 16;   // RNE (finite inputs)
                    w[e] = rl | (rh << 16);
                }
                f32x4 pk = {__builtin_bit_cast(float, w[0]), __builtin_bit_cast(float, w[1]), __builtin_bit_cast(float, w[2]), __builtin_bit_cast(float, w[3])};
                *reinterpret_cast<f32x4*>(o + q * 2) = pk;
            }
        }
    }
}

// delta[i] = act[i] > 0 ? delta[i] : 0 -- the ReLU adjoint of the dgrad chain, in place (ELEM = 2: bf16 pairs, 4: fp32)
template <int ELEM>
__global__ void relu_mask_kernel(uint32_t* __restrict__ delta, const uint32_t* __restrict__ act, int64_t n_words) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t a = act[i];
        uint32_t d = delta[i];
        if constexpr (ELEM == 4) {
            if (!(__builtin_bit_cast(float, a) > 0.0f)) d = 0u;
        } else {                                             // two bf16: positive <=> sign bit clear and not (+)zero
            const uint32_t lo = a & 0xFFFFu, hi = a >> 16;
            if (!(lo != 0u && lo < 0x8000u)) d &= 0xFFFF0000u;
            if (!(hi != 0u && hi < 0x8000u)) d &= 0x0000FFFFu;
        }
        delta[i] = d;
    }
}

// The same with the bias gradient fused in: partial column sums of the MASKED delta, one fp32 row per (block, row group) -- no atomics,
// so that the result is reproducible; the caller adds the partial rows up.
// Thread t of a 256-thread block owns 32-bit word (t % W) of rows (t / W), (t / W) + 256 / W, ...; W = words per row <= 256.
template <int ELEM>
__global__ __launch_bounds__(256) void relu_mask_bias_kernel(uint32_t* __restrict__ delta, const uint32_t* __restrict__ act, int64_t rows, int W,
                                                             float* __restrict__ col_sum) {
    const int wc = threadIdx.x % W, r0 = threadIdx.x / W, rpb = 256 / W;
    float s0 = 0.0f, s1 = 0.0f;
    for (int64_t r = (int64_t)blockIdx.x * rpb + r0; r < rows; r += (int64_t)gridDim.x * rpb) {
        const int64_t i = r * W + wc;
        const uint32_t a = act[i];
        uint32_t d = delta[i];
        if constexpr (ELEM == 4) {
            if (!(__builtin_bit_cast(float, a) > 0.0f)) d = 0u;
            s0 += __builtin_bit_cast(float, d);
        } else {
            const uint32_t lo = a & 0xFFFFu, hi = a >> 16;
            if (!(lo != 0u && lo < 0x8000u)) d &= 0xFFFF0000u;
            if (!(hi != 0u && hi < 0x8000u)) d &= 0x0000FFFFu;
            s0 += __builtin_bit_cast(float, d << 16);           // bf16 -> fp32: the same bits in the upper half
            s1 += __builtin_bit_cast(float, d & 0xFFFF0000u);
        }
        delta[i] = d;
    }
    // deterministic: every (block, row group) writes its own partial row; the host sums the (gridDim.x * rpb, cols) partials
    if (threadIdx.x < rpb * W) {
        float* prow = col_sum + ((size_t)blockIdx.x * rpb + r0) * (size_t)(ELEM == 4 ? W : 2 * W);
        if constexpr (ELEM == 4) prow[wc] = s0;
        else { prow[2 * wc] = s0; prow[2 * wc + 1] = s1; }
    }
}

// ---------------------------------------------------------------------------------------- row 8 (Ref-NeRF render path)
// coarseFineMerge's sort (nerf_base.py:59-73) for the render path: the K fine depths come sorted out of the resampling and the C
// coarse ones are stratified, so the sort of their concatenation is normally a MERGE: element i of `a` goes to i + #(b < a_i),
// element j of `b` to j + #(a <= b_j) (binary searches in LDS).  The last merged depth is dropped.  Sorted VALUES are independent of
// how ties are ordered, so z_out equals torch.sort(cat(a, b))[0][:, :-1] bit for bit.  The stratified depths are only ascending
// while the jitter (far - near) / n_fine does not exceed the spacing of the 64 bins, i.e. for n_fine >= 63: a wave that finds one
// of its inputs out of order first sorts it in LDS (stable rank sort, O(n^2 / 64) per lane -- the rare path).
DEVINL void wave_sort_if_needed(float* v, float* tmp, int n, int lane) {
    int bad = 0;
    for (int i = lane; i + 1 < n; i += 64) bad |= (v[i] > v[i + 1]) ? 1 : 0;
    if (!__any(bad)) return;
    for (int i = lane; i < n; i += 64) {
        const float x = v[i];
        int r = 0;
        for (int k = 0; k < n; ++k) { const float y = v[k]; r += (y < x || (y == x && k < i)) ? 1 : 0; }
        tmp[r] = x;
    }
    lds_wave_sync();
    for (int i = lane; i < n; i += 64) v[i] = tmp[i];
    lds_wave_sync();
}
__global__ __launch_bounds__(256) void merge_sorted_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t N, int K, int C,
                                                           float* __restrict__ out) {
    float* la = reinterpret_cast<float*>(smem) + wave_in_block() * 2 * (K + C);
    float* lb = la + K;
    float* tmp = lb + C;                                        // K + C floats of scratch for the rare sorting path
    const int lane = lane_id();
    const int T = K + C - 1;
    for (int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); n < N; n += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        lds_wave_sync();
        for (int i = lane; i < K; i += 64) la[i] = a[n * K + i];
        for (int j = lane; j < C; j += 64) lb[j] = b[n * C + j];
        lds_wave_sync();
        wave_sort_if_needed(la, tmp, K, lane);
        wave_sort_if_needed(lb, tmp, C, lane);
        float* o = out + n * T;
        for (int i = lane; i < K; i += 64) {
            const float v = la[i];
            int lo = 0, hi = C;                                 // #(b < v)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (lb[mid] < v) lo = mid + 1; else hi = mid; }
            if (i + lo < T) o[i + lo] = v;
        }
        for (int j = lane; j < C; j += 64) {
            const float v = lb[j];
            int lo = 0, hi = K;                                 // #(a <= v)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (la[mid] <= v) lo = mid + 1; else hi = mid; }
            if (j + lo < T) o[j + lo] = v;
        }
    }
}

// The same merge with its sort order, for the training path (train.py:176: coarseFineMerge(..., below_idxs) -> all_inds for getBounds,
// sort_inds for RefNeRF.coarse_grad_select).  order (N, K + C) = the STABLE argsort of cat(a, b) -- fine index i before coarse index
// K + j among equal depths, which is what the device radix sort behind torch.sort produces --, all_inds = gather(cat(f_inds, 0..C-1), order).
DEVINL void wave_sort_idx_if_needed(float* v, int* idx, float* tmp, int* itmp, int n, int lane) {
    int bad = 0;
    for (int i = lane; i + 1 < n; i += 64) bad |= (v[i] > v[i + 1]) ? 1 : 0;
    if (!__any(bad)) return;
    for (int i = lane; i < n; i += 64) {
        const float x = v[i];
        int r = 0;
        for (int k = 0; k < n; ++k) { const float y = v[k]; r += (y < x || (y == x && k < i)) ? 1 : 0; }
        tmp[r] = x; itmp[r] = idx[i];
    }
    lds_wave_sync();
    for (int i = lane; i < n; i += 64) { v[i] = tmp[i]; idx[i] = itmp[i]; }
    lds_wave_sync();
}
__global__ __launch_bounds__(256) void merge_sorted_order_kernel(const float* __restrict__ a, const float* __restrict__ b, const int64_t* __restrict__ f_inds,
                                                                 int64_t N, int K, int C, float* __restrict__ out, int64_t* __restrict__ order,
                                                                 int64_t* __restrict__ all_inds) {
    float* la = reinterpret_cast<float*>(smem) + wave_in_block() * 4 * (K + C);
    float* lb = la + K;
    float* tmp = lb + C;
    int* ia = reinterpret_cast<int*>(tmp + K + C);
    int* ib = ia + K;
    int* itmp = ib + C;
    const int lane = lane_id();
    const int T = K + C - 1;
    for (int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); n < N; n += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        lds_wave_sync();
        for (int i = lane; i < K; i += 64) { la[i] = a[n * K + i]; ia[i] = i; }
        for (int j = lane; j < C; j += 64) { lb[j] = b[n * C + j]; ib[j] = K + j; }
        lds_wave_sync();
        wave_sort_idx_if_needed(la, ia, tmp, itmp, K, lane);
        wave_sort_idx_if_needed(lb, ib, tmp, itmp, C, lane);
        float* o = out + n * T;
        int64_t* ord = order + n * (K + C);
        int64_t* ai = all_inds ? all_inds + n * (K + C) : nullptr;
        for (int i = lane; i < K; i += 64) {
            const float v = la[i];
            int lo = 0, hi = C;                                 // #(b < v)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (lb[mid] < v) lo = mid + 1; else hi = mid; }
            const int p = i + lo, src = ia[i];
            if (p < T) o[p] = v;
            ord[p] = src;
            if (ai) ai[p] = f_inds[n * K + src];
        }
        for (int j = lane; j < C; j += 64) {
            const float v = lb[j];
            int lo = 0, hi = K;                                 // #(a <= v)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (la[mid] <= v) lo = mid + 1; else hi = mid; }
            const int p = j + lo, src = ib[j];
            if (p < T) o[p] = v;
            ord[p] = src;
            if (ai) ai[p] = src - K;
        }
    }
}

// RefNeRF.coarse_grad_select (ref_model.py:108-117): out[n, k] = grads[n, p_k], p_k the k-th position (ascending) whose sort index is
// flagged (>= T - c_pnum: the reference's mask cat(zeros(T - c), ones(c)) gathered by the sort order); should a row hold fewer than c_pnum
// flagged positions, the unflagged ones follow in order -- exactly what the reference's boolean mask does when the row counts agree and
// what a stable descending sort of the flags does otherwise.  One wavefront per ray, ranks by ballot.
__global__ __launch_bounds__(256) void coarse_grad_select_kernel(const float* __restrict__ grads, const int64_t* __restrict__ sort_inds, int64_t N, int T, int D,
                                                                 int c_pnum, float* __restrict__ out) {
    const int lane = lane_id();
    const int64_t thr = (int64_t)T - c_pnum;
    for (int64_t n = blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_in_block(); n < N; n += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        const int64_t* si = sort_inds + n * T;
        const float* g = grads + n * (int64_t)T * D;
        float* o = out + n * (int64_t)c_pnum * D;
        int base = 0;
        for (int want = 1; want >= 0; --want) {                // flagged positions first, then the others
            for (int p0 = 0; p0 < T && base < c_pnum; p0 += 64) {
                const int p = p0 + lane;
                const bool hit = p < T && ((si[p] >= thr) == (want == 1));
                const unsigned long long m = __ballot(hit);
                const int k = base + __popcll(m & ((1ull << lane) - 1ull));
                if (hit && k < c_pnum)
                    for (int d = 0; d < D; ++d) o[(int64_t)k * D + d] = g[(int64_t)p * D + d];
                base += __popcll(m);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- Ref-NeRF's normal losses (ref_model.py:127-143)
// WeightedNormalLoss: sum w (1 - <d_norm, p_norm>)  (mode 0);  BackFaceLoss: mean w relu(<normal, ray_d>)  (mode 1; the mean's 1 / M is `scale`).
// One streaming pass + a fixed-order two-stage sum (block partials, then one block) -- deterministic -- instead of the reference's
// five element-wise torch launches per loss; the backward is one pass too (a product rule per element).
constexpr int LOSS_BLOCKS = 256;
DEVINL float dot_loss_term(float x, int mode) { return mode == 0 ? 1.0f - x : (x > 0.0f ? x : 0.0f); }
__global__ __launch_bounds__(256) void weighted_dot_loss_kernel(const float* __restrict__ w, const float* __restrict__ a, const float* __restrict__ b, int64_t M,
                                                                int mode, float* __restrict__ partial) {
    double acc = 0.0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const float x = (a[3 * i] * b[3 * i] + a[3 * i + 1] * b[3 * i + 1]) + a[3 * i + 2] * b[3 * i + 2];
        acc += (double)(w[i] * dot_loss_term(x, mode));
    }
    acc = wave_sum_d(acc);
    double* red = reinterpret_cast<double*>(smem);
    if (lane_id() == 0) red[wave_in_block()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<double*>(partial)[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(256) void weighted_dot_loss_final_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += reinterpret_cast<const double*>(partial)[i];
    acc = wave_sum_d(acc);
    double* red = reinterpret_cast<double*>(smem);
    if (lane_id() == 0) red[wave_in_block()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((((red[0] + red[1]) + red[2]) + red[3]) * (double)scale);
}
__global__ void weighted_dot_loss_backward_kernel(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ a, const float* __restrict__ b,
                                                  int64_t M, int mode, float scale, float* __restrict__ d_w, float* __restrict__ d_a, float* __restrict__ d_b) {
    const float gs = g[0] * scale;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
        const float ax = a[3 * i], ay = a[3 * i + 1], az = a[3 * i + 2], bx = b[3 * i], by = b[3 * i + 1], bz = b[3 * i + 2];
        const float x = (ax * bx + ay * by) + az * bz;
        if (d_w) d_w[i] = gs * dot_loss_term(x, mode);
        const float k = gs * w[i] * (mode == 0 ? -1.0f : (x > 0.0f ? 1.0f : 0.0f));      // d term / d x, times the weight
        if (d_a) { d_a[3 * i] = k * bx; d_a[3 * i + 1] = k * by; d_a[3 * i + 2] = k * bz; }
        if (d_b) { d_b[3 * i] = k * ax; d_b[3 * i + 1] = k * ay; d_b[3 * i + 2] = k * az; }
    }
}

int blocks_for(int64_t work, int per_block) {
    int64_t b = (work + per_block - 1) / per_block;
    const int64_t cap = 256 * 8;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host launchers
int sk_positional_encoding(const float* x, int64_t M, int L, float* out, hipStream_t st) {
    if (M * L == 0) return 0;
    hipLaunchKernelGGL(pe_kernel, dim3(blocks_for(M, ENC_TILE)), dim3(256), enc_lds_bytes(L), st, x, M, L, out);
    return (int)hipGetLastError();
}
int sk_ipe_feature(const float* z, const float* rays, int64_t N, int Sn, int L, float r2, const float* dir_norm, float* feat, float* mu,
                   float* mu_t, int contract, hipStream_t st) {
    if (N * Sn == 0) return 0;
    hipLaunchKernelGGL(ipe_feature_kernel, dim3(blocks_for(N * Sn, ENC_TILE)), dim3(256), enc_lds_bytes(L), st, z, rays, N, Sn, L, r2, dir_norm, feat, mu, mu_t, contract);
    return (int)hipGetLastError();
}
int sk_cone_parameters(const float* z, int64_t N, int Sn, float r2, float* mu_t, float* var_t, float* var_r, hipStream_t st) {
    if (N * Sn == 0) return 0;
    hipLaunchKernelGGL(cone_parameters_kernel, dim3(blocks_for(N * Sn, 256)), dim3(256), 0, st, z, N, Sn, r2, mu_t, var_t, var_r);
    return (int)hipGetLastError();
}
int sk_dirs_norm(const float* rays, int64_t N, float* out, hipStream_t st) {
    hipLaunchKernelGGL(dirs_norm_kernel, dim3(1), dim3(1024), 0, st, rays, N, out);
    return (int)hipGetLastError();
}
// `partials`: DN_BLOCKS doubles of device scratch (8-byte aligned)
int sk_dirs_norm_scratch(const float* rays, int64_t N, float* out, void* partials, hipStream_t st) {
    hipLaunchKernelGGL(dirs_norm_partial_kernel, dim3(DN_BLOCKS), dim3(256), 0, st, rays, N, reinterpret_cast<double*>(partials));
    hipLaunchKernelGGL(dirs_norm_final_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const double*>(partials), out);
    return (int)hipGetLastError();
}
int sk_train_sampler(const float* rgbs, const int64_t* coords, int64_t P, const float* pose, const float* pose_dev, float fx, float fy, float near, float far,
                     int64_t N, int C, uint64_t seed, const uint64_t* seed_dev, float* pts, float* lengths, float* rgb, float* rays, hipStream_t st) {
    if (N == 0) return 0;
    Cam c; c.H = 0; c.W = 0; c.fx = fx; c.fy = fy;
    for (int i = 0; i < 12; ++i) c.pose[i] = pose ? pose[i] : 0.0f;
    hipLaunchKernelGGL(train_sampler_kernel, dim3(blocks_for(N, 4)), dim3(256), 0, st, rgbs, coords, P, c, near, C > 0 ? (far - near) / (float)C : 0.0f, N, C, seed,
                       pts, lengths, rgb, rays, pose_dev, seed_dev);
    return (int)hipGetLastError();
}

// Philox uniforms as a tensor, u (N,K) = the inverse-CDF stream of device_common.h (philox_u_inv) -- for callers that keep the
// reference's op-by-op structure (inverseSample(weights, depths, u)) but want the draw on the device and, with `seed_dev`, replayable
// from a captured graph.  One thread per element.
// (ABI 121) `ray_offset`: row n of the tensor is GLOBAL ray ray_offset + n (a chunk / shard of a larger ray list draws the bits the whole
// list would); `strat`: the stratified-jitter stream u_strat(n, s) (word 0 of the ray's blocks) instead of the inverse-CDF stream
__global__ void philox_uniforms_kernel(float* __restrict__ out, int64_t N, int K, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                       int64_t ray_offset, int strat) {
    if (seed_dev != nullptr) seed = seed_dev[0];
    const int64_t total = N * K;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / K;
        const int k = (int)(i - n * K);
        out[i] = strat ? philox_u_strat(seed, ray_offset + n, k) : philox_u_inv(seed, ray_offset + n, k);
    }
}
// The bottle-neck perturbation of Ref-NeRF's training forward as a tensor, out (M, 128) ~ N(0, std): bit-identical to what ref_kernel
// draws in place for the same key and sample index (device_common.h philox_normal8) -- for tests, the layer-by-layer route and callers
// that want the reference's explicit `spa_info_b + noise` (ref_model.py:84-85).  One thread per (sample, block of eight deviates).
__global__ void philox_normal_kernel(float* __restrict__ out, int64_t M, uint64_t seed, const uint64_t* __restrict__ seed_dev, float std, int64_t sample_offset) {
    if (seed_dev != nullptr) seed = seed_dev[0];
    const int64_t total = M * 16;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i >> 4;
        const int q = (int)(i & 15);
        float z[8];
        philox_normal8(seed, sample_offset + m, q, std, z);
        float* row = out + m * 128 + 16 * (q >> 1) + 4 * (q & 1);
        *reinterpret_cast<f32x4*>(row) = f32x4{z[0], z[1], z[2], z[3]};
        *reinterpret_cast<f32x4*>(row + 8) = f32x4{z[4], z[5], z[6], z[7]};
    }
}
int sk_philox_normal(float* out, int64_t M, uint64_t seed, const uint64_t* seed_dev, float std, int64_t sample_offset, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(philox_normal_kernel, dim3(blocks_for(M * 16, 256)), dim3(256), 0, st, out, M, seed, seed_dev, std, sample_offset);
    return (int)hipGetLastError();
}
// seed <- a new, unrelated key for the next step (golden-ratio increment + a xorshift-multiply mix); one thread
__global__ void advance_seed_kernel(uint64_t* __restrict__ seed_dev) {
    uint64_t x = seed_dev[0] + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    seed_dev[0] = x ^ (x >> 31);
}
int sk_philox_uniforms(float* out, int64_t N, int K, uint64_t seed, const uint64_t* seed_dev, int64_t ray_offset, int strat, hipStream_t st) {
    if (N * K == 0) return 0;
    hipLaunchKernelGGL(philox_uniforms_kernel, dim3(blocks_for(N * K, 256)), dim3(256), 0, st, out, N, K, seed, seed_dev, ray_offset, strat);
    return (int)hipGetLastError();
}
int sk_advance_seed(uint64_t* seed_dev, hipStream_t st) {
    hipLaunchKernelGGL(advance_seed_kernel, dim3(1), dim3(1), 0, st, seed_dev);
    return (int)hipGetLastError();
}
int sk_generate_rays(const float* pose, int H, int W, float fx, float fy, int64_t first, int64_t count, float* rays,
                     hipStream_t st) {
    Cam c; c.H = H; c.W = W; c.fx = fx; c.fy = fy;
    for (int i = 0; i < 12; ++i) c.pose[i] = pose[i];
    if (count == 0) return 0;
    hipLaunchKernelGGL(raygen_kernel, dim3(blocks_for(count, 256)), dim3(256), 0, st, c, first, count, rays);
    return (int)hipGetLastError();
}
int sk_pixel_rays(const float* pose, float fx, float fy, const int64_t* coords, int64_t N, float* rays, hipStream_t st) {
    Cam c; c.H = 0; c.W = 0; c.fx = fx; c.fy = fy;
    for (int i = 0; i < 12; ++i) c.pose[i] = pose[i];
    if (N == 0) return 0;
    hipLaunchKernelGGL(pixel_rays_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, st, c, coords, N, rays);
    return (int)hipGetLastError();
}
int sk_stratified_points(const float* rays, const float* z_base, const float* u, float jitter, int64_t N, int S, float* z_out,
                         float* pts, hipStream_t st) {
    if (N * S == 0) return 0;
    hipLaunchKernelGGL(stratified_points_kernel, dim3(blocks_for(N * S, 256)), dim3(256), 0, st, rays, z_base, u, jitter, N, S, z_out, pts);
    return (int)hipGetLastError();
}
int sk_length2pts(const float* rays, const float* z, int64_t N, int S, float* out, hipStream_t st) {
    if (N * S == 0) return 0;
    hipLaunchKernelGGL(length2pts_kernel, dim3(blocks_for(N * S * 6, 256)), dim3(256), 0, st, rays, z, N, S, out);
    return (int)hipGetLastError();
}
int sk_sigma_to_weights(const float* sigma, const float* z, const float* dirs, int64_t N, int S, int act, float* w, hipStream_t st) {
    if (N * S == 0) return 0;
    hipLaunchKernelGGL(sigma_to_weights_kernel, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), 0, st, sigma, z, dirs, N, S, act, w);
    return (int)hipGetLastError();
}
int sk_max_blur(const float* w, int64_t N, int S, float alpha, float* out, hipStream_t st) {
    if (N * S == 0) return 0;
    hipLaunchKernelGGL(max_blur_kernel, dim3(blocks_for(N * S, 256)), dim3(256), 0, st, w, N, S, alpha, out);
    return (int)hipGetLastError();
}
int sk_inverse_sample(const float* w, const float* z, const float* u, int64_t N, int C, int K, int sort, int mode, float* z_out,
                      int64_t* below, int64_t* above, hipStream_t st) {
    if (N == 0) return 0;
    const size_t lds = WAVES_PER_BLOCK * ((size_t)3 * C + 2 * K + SORT_LDS_FLOATS) * 4;
    hipLaunchKernelGGL(inverse_sample_kernel, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), lds, st, w, z, u, N, C, K, sort, mode,
                       z_out, below, above);
    return (int)hipGetLastError();
}
int sk_resample(const float* density, const float* z, const float* z_base, const float* u_strat, float z_jitter,
                const float* dirs, int dirs_stride, const float* u_inv, int64_t N, int C, int K, int softplus, float alpha,
                uint64_t rng_seed, int64_t rng_ray_offset, float* z_fine, int64_t* below, float* w_prop, float* z_coarse, hipStream_t st) {
    if (N == 0) return 0;
    ResampleArgs a{density, z, z_base, u_strat, z_jitter, dirs, dirs_stride, u_inv, N, C, K, softplus, alpha, z_fine, below, w_prop, z_coarse,
                   rng_seed, rng_ray_offset};
    const size_t lds = WAVES_PER_BLOCK * ((size_t)5 * C + 2 * K + SORT_LDS_FLOATS) * 4;
    // persistent over rays: exactly one resident round (RS_BLOCKS_PER_CU workgroups fit a CU by registers and LDS at the render shapes),
    // so that no partially filled last round trails behind
    int64_t blocks = (N + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const int64_t resident = (int64_t)nerf_host::cu_count() * RS_BLOCKS_PER_CU;
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return (int)hipGetLastError();
}
int sk_composite(const float* rgbo, const float* z, int z_stride, const float* dirs, int dirs_stride, int64_t N, int S, int flags,
                 int act, float sigma_shift, float near, float far, const float* normal, const float* cam_dir, float* rgb, float* weights,
                 float* depth, float* normal_img, hipStream_t st) {
    if (N == 0) return 0;
    CompositeArgs a{rgbo, z, z_stride, dirs, dirs_stride, N, S, flags, act, sigma_shift, near, far, normal, cam_dir, rgb, weights, depth, normal_img};
    hipLaunchKernelGGL(composite_kernel, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
int sk_get_bounds(const float* w, const int64_t* below, int64_t N, int C, int K, float* bounds, hipStream_t st) {
    if (N == 0 || K < 2) return 0;
    const size_t lds = WAVES_PER_BLOCK * ((size_t)C + 1) * 4;
    hipLaunchKernelGGL(get_bounds_kernel, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), lds, st, w, below, N, C, K, bounds);
    return (int)hipGetLastError();
}

// ---- backward launchers ----
int sk_weights_backward(const float* sigma, int sigma_stride, int sigma_off, const float* z, int z_stride, const float* dirs, int dirs_stride,
                        int64_t N, int S, int mul_norm, int act, float sigma_shift, const float* rgbo, const float* d_rgb,
                        const float* d_weights, const float* d_depth, int white_bkg, float near, float far, float* d_sigma,
                        int d_sigma_stride, int d_sigma_off, float* d_rgbo, hipStream_t st) {
    if (N * S == 0) return 0;
    if (S > BWD_MAX_CHUNKS * 64) return (int)hipErrorInvalidValue;
    WeightsBwdArgs a{sigma, sigma_stride, sigma_off, z, z_stride, dirs, dirs_stride, N, S, mul_norm, act, sigma_shift, rgbo, d_rgb, d_weights,
                     d_depth, white_bkg, near, far, d_sigma, d_sigma_stride, d_sigma_off, d_rgbo};
    if (S <= 256) hipLaunchKernelGGL(weights_backward_kernel<4>, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), 0, st, a);
    else if (S <= 512) hipLaunchKernelGGL(weights_backward_kernel<8>, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(weights_backward_kernel<16>, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
int sk_max_blur_backward(const float* w, const float* g, int64_t N, int S, float* dw, hipStream_t st) {
    if (N * S == 0) return 0;
    hipLaunchKernelGGL(max_blur_backward_kernel, dim3(blocks_for(N * S, 256)), dim3(256), 0, st, w, g, N, S, dw);
    return (int)hipGetLastError();
}
int sk_get_bounds_backward(const int64_t* below, const float* g, int64_t N, int C, int K, float* dw, hipStream_t st) {
    if (N == 0) return 0;
    if (K < 2) { return (int)hipMemsetAsync(dw, 0, (size_t)N * C * 4, st); }
    const size_t lds = WAVES_PER_BLOCK * (size_t)(2 * K) * 4;
    hipLaunchKernelGGL(get_bounds_backward_kernel, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), lds, st, below, g, N, C, K, dw);
    return (int)hipGetLastError();
}
int sk_frag_to_rows(const void* frag, int elem_bytes, int64_t n_sub, int n_kg, int64_t M, void* out, hipStream_t st) {
    if (n_sub == 0 || M == 0) return 0;
    const size_t lds = (size_t)WAVES_PER_BLOCK * 32 * ((size_t)n_kg * 16 * elem_bytes + 16);
    if (elem_bytes == 2) {
        hipLaunchKernelGGL(frag_to_rows_kernel<2>, dim3(blocks_for(n_sub, WAVES_PER_BLOCK)), dim3(256), lds, st, (const char*)frag, n_sub, n_kg, M, (char*)out, n_kg * 16);
    } else {
        if (int e = nerf_host::allow_dynamic_lds(reinterpret_cast<const void*>(frag_to_rows_kernel<4>), 140 * 1024)) return e;
        hipLaunchKernelGGL(frag_to_rows_kernel<4>, dim3(blocks_for(n_sub, WAVES_PER_BLOCK)), dim3(256), lds, st, (const char*)frag, n_sub, n_kg, M, (char*)out, n_kg * 16);
    }
    return (int)hipGetLastError();
}
int sk_relu_mask(void* delta, const void* act, int elem_bytes, int64_t n, hipStream_t st) {
    if (n == 0) return 0;
    const int64_t words = elem_bytes == 2 ? n / 2 : n;
    if (elem_bytes == 2) hipLaunchKernelGGL(relu_mask_kernel<2>, dim3(blocks_for(words, 256)), dim3(256), 0, st, (uint32_t*)delta, (const uint32_t*)act, words);
    else hipLaunchKernelGGL(relu_mask_kernel<4>, dim3(blocks_for(words, 256)), dim3(256), 0, st, (uint32_t*)delta, (const uint32_t*)act, words);
    return (int)hipGetLastError();
}
int sk_relu_mask_bias(void* delta, const void* act, int elem_bytes, int64_t rows, int cols, float* col_sum, hipStream_t st) {
    if (rows == 0) return 0;
    const int W = cols * elem_bytes / 4;
    if (W < 1 || W > 256 || (256 % W) != 0) return (int)hipErrorInvalidValue;
    const int rpb = 256 / W;
    int64_t blocks = (rows + rpb - 1) / rpb;
    if (blocks > 1024) blocks = 1024;                            // = nerf_amd_relu_mask_bias_partials() / rpb rows of partial sums
    if (elem_bytes == 2) hipLaunchKernelGGL(relu_mask_bias_kernel<2>, dim3((int)blocks), dim3(256), 0, st, (uint32_t*)delta, (const uint32_t*)act, rows, W, col_sum);
    else hipLaunchKernelGGL(relu_mask_bias_kernel<4>, dim3((int)blocks), dim3(256), 0, st, (uint32_t*)delta, (const uint32_t*)act, rows, W, col_sum);
    return (int)hipGetLastError();
}
int sk_merge_sorted(const float* a, const float* b, int64_t N, int K, int C, float* out, hipStream_t st) {
    if (N == 0) return 0;
    const size_t lds = WAVES_PER_BLOCK * (size_t)(K + C) * 2 * 4;
    hipLaunchKernelGGL(merge_sorted_kernel, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), lds, st, a, b, N, K, C, out);
    return (int)hipGetLastError();
}
int sk_merge_sorted_order(const float* a, const float* b, const int64_t* f_inds, int64_t N, int K, int C, float* out, int64_t* order, int64_t* all_inds,
                          hipStream_t st) {
    if (N == 0) return 0;
    const size_t lds = WAVES_PER_BLOCK * (size_t)(K + C) * 4 * 4;
    hipLaunchKernelGGL(merge_sorted_order_kernel, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), lds, st, a, b, f_inds, N, K, C, out, order, all_inds);
    return (int)hipGetLastError();
}
int sk_coarse_grad_select(const float* grads, const int64_t* sort_inds, int64_t N, int T, int D, int c_pnum, float* out, hipStream_t st) {
    if (N == 0 || c_pnum == 0 || D == 0) return 0;
    hipLaunchKernelGGL(coarse_grad_select_kernel, dim3(blocks_for(N, WAVES_PER_BLOCK)), dim3(256), 0, st, grads, sort_inds, N, T, D, c_pnum, out);
    return (int)hipGetLastError();
}
int sk_weighted_dot_loss(const float* w, const float* a, const float* b, int64_t M, int mode, float scale, float* out, float* workspace, hipStream_t st) {
    int64_t nb = (M + 255) / 256;
    const int blocks = (int)(nb > LOSS_BLOCKS ? LOSS_BLOCKS : (nb < 1 ? 1 : nb));
    hipLaunchKernelGGL(weighted_dot_loss_kernel, dim3(blocks), dim3(256), 64, st, w, a, b, M, mode, workspace);
    hipLaunchKernelGGL(weighted_dot_loss_final_kernel, dim3(1), dim3(256), 64, st, workspace, blocks, scale, out);
    return (int)hipGetLastError();
}
int sk_weighted_dot_loss_backward(const float* g, const float* w, const float* a, const float* b, int64_t M, int mode, float scale, float* d_w, float* d_a,
                                  float* d_b, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(weighted_dot_loss_backward_kernel, dim3(blocks_for(M, 256)), dim3(256), 0, st, g, w, a, b, M, mode, scale, d_w, d_a, d_b);
    return (int)hipGetLastError();
}
int sk_encode_rows(const float* x, int x_stride, int64_t M, int L, int normalize, int elem_bytes, void* out, hipStream_t st) {
    if (M == 0) return 0;
    const dim3 grid(blocks_for(M, 256)), block(256);
    char* o = (char*)out;
    if (L == 10 && elem_bytes == 2) hipLaunchKernelGGL((encode_rows_kernel<10, 2>), grid, block, 0, st, x, x_stride, M, normalize, o);
    else if (L == 10) hipLaunchKernelGGL((encode_rows_kernel<10, 4>), grid, block, 0, st, x, x_stride, M, normalize, o);
    else if (L == 4 && elem_bytes == 2) hipLaunchKernelGGL((encode_rows_kernel<4, 2>), grid, block, 0, st, x, x_stride, M, normalize, o);
    else if (L == 4) hipLaunchKernelGGL((encode_rows_kernel<4, 4>), grid, block, 0, st, x, x_stride, M, normalize, o);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
int sk_frag_rows_mask_blocks() { return 512; }                   // partial rows = blocks * WAVES_PER_BLOCK
int sk_frag_rows_mask(const void* frag, int elem_bytes, int64_t n_sub, int n_kg, int64_t M, void* act_out, void* delta, float* col_sum, hipStream_t st) {
    const size_t lds = (size_t)WAVES_PER_BLOCK * 32 * ((size_t)n_kg * 16 * elem_bytes + 16);
    const dim3 grid(sk_frag_rows_mask_blocks()), block(256);     // fixed grid: every wavefront writes its partial row (zeros without work)
    if (elem_bytes == 2) {
        hipLaunchKernelGGL(frag_rows_mask_kernel<2>, grid, block, lds, st, (const char*)frag, n_sub, n_kg, M, (char*)act_out, (char*)delta, col_sum);
    } else {
        if (int e = nerf_host::allow_dynamic_lds(reinterpret_cast<const void*>(frag_rows_mask_kernel<4>), 140 * 1024)) return e;
        hipLaunchKernelGGL(frag_rows_mask_kernel<4>, grid, block, lds, st, (const char*)frag, n_sub, n_kg, M, (char*)act_out, (char*)delta, col_sum);
    }
    return (int)hipGetLastError();
}

