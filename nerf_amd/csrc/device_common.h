// Shared device helpers for the gfx950 NeRF kernels.  Compiled with -ffp-contract=off: every FMA in
// this code base is an explicit __builtin_fmaf, so position arithmetic (o + z*d, z_base + u*res)
// rounds exactly like the reference's separate torch mul/add (SURVEY.md section 7 "hard parts").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define DEVINL __device__ __forceinline__

DEVINL int lane_id() { return (int)(threadIdx.x & 63u); }

// ------------------------------------------------------------------------------------------------
// sin / cos with exact-to-~1ulp range reduction for |a| <~ 2^15 (PE arguments are 2^f * x <= ~8e3).
// k = rint(a*2/pi); r = a - k*pi/2 by a 3-term Cody-Waite split evaluated with FMAs: the first FMA
// is exact (a - k*c1 is a multiple of 2^-23 below 1), the second rounds once (<= 2^-25).
// `quad` = 0 -> sin(a), 1 -> cos(a) = sin(a + pi/2): both lane halves run the same instruction stream.
// ------------------------------------------------------------------------------------------------
DEVINL float sin_quadrant(float a, int quad) {
    const float k = __builtin_rintf(a * 0.6366197466850281f);
    float r = __builtin_fmaf(-k, 1.57079637050628662109375f, a);       // fl(pi/2)
    r = __builtin_fmaf(-k, -4.371138828673793e-08f, r);                    // fl(pi/2 - c1)
    r = __builtin_fmaf(-k, -1.7151245100058819e-15f, r);                  // fl(pi/2 - c1 - c2)
    const int n = (int)k + quad;
    const float z = r * r;
    // sin(r) on [-pi/4, pi/4]  (fdlibm k_sinf coefficients)
    float ps = __builtin_fmaf(z, 1.5896910177e-10f, -2.5050759689e-08f);
    ps = __builtin_fmaf(z, ps, 2.7557314297e-06f);
    ps = __builtin_fmaf(z, ps, -1.9841270114e-04f);
    ps = __builtin_fmaf(z, ps, 8.3333337680e-03f);
    ps = __builtin_fmaf(z, ps, -1.6666667163e-01f);
    const float s = __builtin_fmaf(r * z, ps, r);
    // cos(r) on [-pi/4, pi/4]  (fdlibm k_cosf coefficients)
    float pc = __builtin_fmaf(z, -1.1359647598e-11f, 2.0875723372e-09f);
    pc = __builtin_fmaf(z, pc, -2.7557314297e-07f);
    pc = __builtin_fmaf(z, pc, 2.4801587642e-05f);
    pc = __builtin_fmaf(z, pc, -1.3888889225e-03f);
    pc = __builtin_fmaf(z, pc, 4.1666667908e-02f);
    const float c = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    float v = (n & 1) ? c : s;
    return (n & 2) ? -v : v;
}

// sin(a) and cos(a) from ONE range reduction: bit-identical to sin_quadrant(a, 0) / sin_quadrant(a, 1)
DEVINL void sincos_quadrant(float a, float& sn, float& cs) {
    const float k = __builtin_rintf(a * 0.6366197466850281f);
    float r = __builtin_fmaf(-k, 1.57079637050628662109375f, a);
    r = __builtin_fmaf(-k, -4.371138828673793e-08f, r);
    r = __builtin_fmaf(-k, -1.7151245100058819e-15f, r);
    const int n = (int)k;
    const float z = r * r;
    float ps = __builtin_fmaf(z, 1.5896910177e-10f, -2.5050759689e-08f);
    ps = __builtin_fmaf(z, ps, 2.7557314297e-06f);
    ps = __builtin_fmaf(z, ps, -1.9841270114e-04f);
    ps = __builtin_fmaf(z, ps, 8.3333337680e-03f);
    ps = __builtin_fmaf(z, ps, -1.6666667163e-01f);
    const float s = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(z, -1.1359647598e-11f, 2.0875723372e-09f);
    pc = __builtin_fmaf(z, pc, -2.7557314297e-07f);
    pc = __builtin_fmaf(z, pc, 2.4801587642e-05f);
    pc = __builtin_fmaf(z, pc, -1.3888889225e-03f);
    pc = __builtin_fmaf(z, pc, 4.1666667908e-02f);
    const float c = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    const float vs = (n & 1) ? c : s, vc = (n & 1) ? s : c;
    sn = (n & 2) ? -vs : vs;
    cs = ((n + 1) & 2) ? -vc : vc;
}

// ------------------------------------------------------------------------------------------------
// Counter-based uniforms (Philox4x32-10, Salmon et al. 2011 -- the generator behind torch's device RNG, restated from the paper's
// constants).  When a caller passes no uniform tensors, the kernels draw them as a PURE FUNCTION of (seed, ray, sample).  One Philox
// block per (ray, slot j) of stream 'RS' -- block(n, j) = Philox(key = seed, counter = (n_lo, n_hi, j, 'RS')) -- carries
//   word 0      : u_strat(n, s = j)                                    stratified jitter (procedures.py:65)
//   words 1..3  : u_inv(n, k) for k = 192 (j / 64) + (j % 64) + {0, 64, 128}   inverse-CDF draws (utils.py:115)
// so in the resampling kernel lane j's ONE Philox call yields everything the lane needs of a ray at the render shapes (S <= 64,
// K <= 192), the proposal pass regenerates the same depths from word 0 without a tensor in between, a render is replayable from
// its seed, and any sub-batch of rays reproduces the same bits (oracle twin: oracle/nerf_oracle.py philox_uniforms).
// uint32 -> [0, 1): the top 24 bits times 2^-24 (torch's device convention).
// The two 32 x 32 -> 64 products of a round are one v_mad_u64_u32 each (hipcc otherwise emits v_mul_lo_u32 + v_mul_hi_u32: two
// quarter-rate instructions instead of one); the multipliers live in SGPRs.
// ------------------------------------------------------------------------------------------------
struct Philox4 { uint32_t w[4]; };
DEVINL uint64_t mul_wide_u32(uint32_t m, uint32_t x) {
    uint64_t p, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(carry) : "s"(m), "v"(x));
    return p;
}
DEVINL Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = mul_wide_u32(0xD2511F53u, c0), p1 = mul_wide_u32(0xCD9E8D57u, c2);
        c0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0; c1 = (uint32_t)p1;
        c2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}
DEVINL float u01_from_bits(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }
constexpr uint32_t PHILOX_STREAM_RENDER = 0x5253u;
DEVINL Philox4 philox_ray_block(uint64_t seed, int64_t n, int j) {
    return philox4x32_10((uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)j, PHILOX_STREAM_RENDER, (uint32_t)seed, (uint32_t)(seed >> 32));
}
DEVINL float philox_u_strat(uint64_t seed, int64_t n, int s) { return u01_from_bits(philox_ray_block(seed, n, s).w[0]); }
// inverse-CDF uniform k of ray n (generic shapes; the render shapes take the words of philox_ray_block directly)
DEVINL float philox_u_inv(uint64_t seed, int64_t n, int k) {
    const int b = k / 192, r = k - 192 * b, q = r >> 6;
    const Philox4 p = philox_ray_block(seed, n, 64 * b + (r & 63));
    return u01_from_bits(q == 0 ? p.w[1] : (q == 1 ? p.w[2] : p.w[3]));
}

// Normal deviates for Ref-NeRF's bottle-neck perturbation (ref_model.py:84-85: `spa_info_b + torch.normal(0, w, shape)`), drawn where
// they are added instead of written to and read back from HBM (1.6 GB and a 0.5 ms launch per 2^14-ray step).  One Philox block per
// (sample m, block q) of stream 'BN' gives EIGHT deviates: its four words are cut into eight 16-bit uniforms (c + 0.5) / 65536 in (0, 1),
// pairs (lo, hi) of a word feed Box-Muller -- r = sqrt(-2 ln u_lo), z = r (cos, sin)(2 pi u_hi).  16-bit uniforms bound |z| by 4.85 sigma
// and put the values on a 65 536 x 65 536 polar lattice: ample for a regularising perturbation of standard deviation `std`, and half the
// Philox calls of full-width uniforms.  Feature f of the 128 of sample m: q = f >> 3 ... the layout the fused kernel's accumulators
// want: q = 2 (f >> 4) + ((f >> 2) & 1), element (f & 3) + 4 ((f >> 3) & 1)  (oracle twin: oracle.philox_normal).
constexpr uint32_t PHILOX_STREAM_BN = 0x424Eu;
DEVINL void philox_normal8(uint64_t seed, int64_t m, int q, float std, float (&z)[8]) {
    const Philox4 p = philox4x32_10((uint32_t)m, (uint32_t)((uint64_t)m >> 32), (uint32_t)q, PHILOX_STREAM_BN, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float u0 = ((float)(p.w[w] & 0xffffu) + 0.5f) * 1.52587890625e-05f;       // (0, 1)
        const float u1 = ((float)(p.w[w] >> 16) + 0.5f) * 1.52587890625e-05f;
        const float r = std * __builtin_sqrtf(-2.0f * __logf(u0));
        z[2 * w] = r * __builtin_amdgcn_cosf(u1);                                        // v_cos_f32 / v_sin_f32 take REVOLUTIONS: cos(2 pi u1)
        z[2 * w + 1] = r * __builtin_amdgcn_sinf(u1);
    }
}

// ------------------------------------------------------------------------------------------------
// Integrated positional encoding, per-frustum part (mip_methods.py:15-33): Gaussian moments of the conical frustum between depths
// z0 < z1 along a ray of pixel radius r (r2 = fl(r*r) computed in double like Python's `r ** 2`), every operation in the reference's
// order (no contraction: this file is compiled with -ffp-contract=off).
// ------------------------------------------------------------------------------------------------
struct ConeMoments { float mu_t, var_t, var_r; };
DEVINL ConeMoments cone_moments(float z0, float z1, float r2) {
    const float mid = (z1 + z0) / 2.0f;                                   // :16
    const float hw = (z1 - z0) / 2.0f;
    const float hw2 = hw * hw;                                            // :17
    const float mid2 = mid * mid;
    const float t1 = 3.0f * mid2 + hw2;                                   // :18
    ConeMoments c;
    c.mu_t = mid + ((2.0f * mid) * hw2) / t1;                             // :20
    const float hw4 = hw2 * hw2;
    c.var_t = hw2 / 3.0f - (((4.0f * hw4) * (12.0f * mid2 - hw2)) / 15.0f) / (t1 * t1);          // :21
    c.var_r = r2 * ((0.25f * mid2 + 0.4166666567325592f * hw2) - (4.0f * hw4) / (15.0f * t1));   // :22  (fl(5/12))
    return c;
}
// mean and diagonal covariance of the frustum in world space (coneMeanCov, :27-33); inv-free: dd / dir_norm like the reference,
// where dir_norm is the norm of the WHOLE (N,3) direction tensor (the reference's `.norm()` without a dim, :31)
DEVINL void cone_mean_cov(const ConeMoments& c, float o, float d, float dir_norm, float& mu, float& diag) {
    mu = o + c.mu_t * d;
    const float dd = d * d;
    diag = c.var_t * dd + c.var_r * (1.0f - dd / dir_norm);
}

// ------------------------------------------------------------------------------------------------
// 64-lane inclusive scans.  fp64 versions mirror torch's CPU cumsum/cumprod, which accumulate float inputs in double and round
// each prefix to float.
// Data movement is DPP (v_mov_b32 with a lane-select modifier, register-file latency), not ds_bpermute (LDS latency): four
// Hillis-Steele steps inside each row of 16 lanes (row_shr), then row 0 -> 1 and 2 -> 3 (row_bcast:15), then rows 0-1 -> 2-3
// (row_bcast:31).  Lanes without a source receive the operation's identity.  The scans sit on the per-ray critical path of the
// wave-per-ray kernels (three per ray in resample_kernel), which are latency-bound, not bandwidth-bound.
// ------------------------------------------------------------------------------------------------
constexpr int DPP_ROW_SHR = 0x110, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
template <int CTRL, int ROW_MASK>
DEVINL int dpp_move(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK>
DEVINL float dpp_move(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
DEVINL double dpp_move(double identity, double v) {
    const int2 o = __builtin_bit_cast(int2, identity), x = __builtin_bit_cast(int2, v);
    int2 r;
    r.x = __builtin_amdgcn_update_dpp(o.x, x.x, CTRL, ROW_MASK, 0xF, false);
    r.y = __builtin_amdgcn_update_dpp(o.y, x.y, CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, r);
}
template <class T, class OP>
DEVINL T wave_incl_scan(T v, T identity, OP&& op) {
    v = op(v, dpp_move<DPP_ROW_SHR + 1, 0xF>(identity, v));
    v = op(v, dpp_move<DPP_ROW_SHR + 2, 0xF>(identity, v));
    v = op(v, dpp_move<DPP_ROW_SHR + 4, 0xF>(identity, v));
    v = op(v, dpp_move<DPP_ROW_SHR + 8, 0xF>(identity, v));
    v = op(v, dpp_move<DPP_ROW_BCAST15, 0xA>(identity, v));
    v = op(v, dpp_move<DPP_ROW_BCAST31, 0xC>(identity, v));
    return v;
}
DEVINL double wave_incl_scan_mul(double v) { return wave_incl_scan(v, 1.0, [](double a, double b) { return a * b; }); }
// the same without the last step: two independent 32-lane scans (lanes 0..31 | 32..63) -- the full scan's value before row_bcast:31
DEVINL double seg32_incl_scan_mul(double v) {
    v = v * dpp_move<DPP_ROW_SHR + 1, 0xF>(1.0, v);
    v = v * dpp_move<DPP_ROW_SHR + 2, 0xF>(1.0, v);
    v = v * dpp_move<DPP_ROW_SHR + 4, 0xF>(1.0, v);
    v = v * dpp_move<DPP_ROW_SHR + 8, 0xF>(1.0, v);
    v = v * dpp_move<DPP_ROW_BCAST15, 0xA>(1.0, v);
    return v;
}
DEVINL double wave_incl_scan_add(double v) { return wave_incl_scan(v, 0.0, [](double a, double b) { return a + b; }); }
DEVINL int wave_incl_scan_add(int v) { return wave_incl_scan(v, 0, [](int a, int b) { return a + b; }); }
DEVINL int wave_max_i(int v) {                  // maximum over the wave, as a wave-uniform scalar
    v = wave_incl_scan(v, (int)0x80000000, [](int a, int b) { return a > b ? a : b; });
    return __builtin_amdgcn_readlane(v, 63);
}
// value of the lane below (lane 0 receives `first`): the exclusive form of a scan
DEVINL double wave_shift_up1(double v, double first) { return dpp_move<DPP_WAVE_SHR1, 0xF>(first, v); }
DEVINL double wave_last(double v) {             // lane 63's value, as a wave-uniform scalar
    const int2 x = __builtin_bit_cast(int2, v);
    int2 r;
    r.x = __builtin_amdgcn_readlane(x.x, 63);
    r.y = __builtin_amdgcn_readlane(x.y, 63);
    return __builtin_bit_cast(double, r);
}
// sum over the wave, returned to every lane: the DPP scan's last lane (register-file lane moves instead of six ds_bpermute round trips)
DEVINL float wave_sum(float v) {
    const float t = wave_incl_scan(v, 0.0f, [](float a, float b) { return a + b; });
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
}
DEVINL double wave_sum_d(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// RefNeRF(use_srgb=True) output transform (ref_model.py:100-102, nerf_helper.py:50-56): linear -> sRGB, and its slope for the backward.
// Arithmetic as torch evaluates it: fl32(323/25) * x;  (211 * max(eps, x)^fl32(5/12) - 11) / 200;  eps = 2^-23.
constexpr float SRGB_KNEE = 0.0031308f, SRGB_EPS = 1.1920928955078125e-07f, SRGB_LOG3 = 1.0986122886681098f;
DEVINL float srgb_from_linear(float x) {
    const float s0 = 12.92f * x;
    const float s1 = (211.0f * powf(fmaxf(SRGB_EPS, x), 0.4166666666666667f) - 11.0f) / 200.0f;
    return x <= SRGB_KNEE ? s0 : s1;
}
DEVINL float srgb_slope(float x) {
    return x <= SRGB_KNEE ? 12.92f : (211.0f / 200.0f) * 0.4166666666666667f * powf(fmaxf(SRGB_EPS, x), -0.5833333333333333f);
}
DEVINL float softplus_f(float x) {            // torch softplus: beta=1, threshold=20
    return x > 20.0f ? x : log1pf(expf(x));
}
DEVINL float density_act(float x, int act) {
    if (act == 0) return fmaxf(x, 0.0f);
    if (act == 2) return softplus_f(x);
    return x;
}
DEVINL float norm3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

// ------------------------------------------------------------------------------------------------
// sigma -> weights for one ray handled by one wavefront (rows 5 / 10).
//   sig(s), zn(s): loaders for sample s (zn already scaled by |d| when required)
//   emit(s, w, zn_s): called by the lane that owns sample s
// ------------------------------------------------------------------------------------------------
template <class SigF, class ZF, class EmitF>
DEVINL void wave_sigma_to_weights(int S, int act, SigF&& sig, ZF&& zn, EmitF&& emit) {
    const int lane = lane_id();
    double carry = 1.0;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool ok = s < S;
        float w = 0.0f, z0 = 0.0f;
        double p = 1.0;
        if (ok) {
            z0 = zn(s);
            const float delta = (s + 1 < S) ? (zn(s + 1) - z0) : 1e10f;
            const float m = expf(-density_act(sig(s), act) * delta);
            w = 1.0f - m;                                    // alpha
            p = (double)(m + 1e-10f);
        }
        const double incl = wave_incl_scan_mul(p);
        const double excl = wave_shift_up1(incl, 1.0);
        const float T = (float)(carry * excl);
        carry *= wave_last(incl);
        if (ok) emit(s, w * T, z0);
    }
}

