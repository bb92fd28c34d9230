// extern "C" boundary of libnerf_amd.so (declared in include/nerf_amd.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nerf_amd.h"
#include "mlp_layout.h"

// launchers defined in the kernel translation units
int mlp_launch_proposal(const void*, int, const nerf_amd_samples&, float*, hipStream_t);
int mlp_launch_proposal128(const void*, int, const nerf_amd_samples&, float*, hipStream_t);
int mlp_launch_mip(const void*, int, const nerf_amd_samples&, float*, hipStream_t);
int mlp_launch_mip128(const void*, int, const nerf_amd_samples&, float*, hipStream_t);
int mlp_launch_mip_composite(const void*, int, const nerf_amd_samples&, float*, float*, float*, int, float, float, hipStream_t);
int mlp_launch_ref(const void*, int, const nerf_amd_samples&, float*, float*, const float*, int, hipStream_t);
size_t mlp_train_layer_stride(int, int64_t);
size_t mlp_train_mask_stride(int, int64_t);
int mlp_launch_proposal_train(const void*, int, const nerf_amd_samples&, float*, void*, hipStream_t);
int mlp_launch_mip_train(const void*, int, const nerf_amd_samples&, float*, void*, hipStream_t);
int sk_frag_to_rows(const void*, int, int64_t, int, int64_t, void*, hipStream_t);
int sk_relu_mask(void*, const void*, int, int64_t, hipStream_t);
int sk_merge_sorted(const float*, const float*, int64_t, int, int, float*, hipStream_t);
int sk_merge_sorted_order(const float*, const float*, const int64_t*, int64_t, int, int, float*, int64_t*, int64_t*, hipStream_t);
int sk_coarse_grad_select(const float*, const int64_t*, int64_t, int, int, int, float*, hipStream_t);
int pack_mfma_stream(int, int, int, float*, hipStream_t);
int sk_weighted_dot_loss(const float*, const float*, const float*, int64_t, int, float, float*, float*, hipStream_t);
int sk_weighted_dot_loss_backward(const float*, const float*, const float*, const float*, int64_t, int, float, float*, float*, float*, hipStream_t);
int sk_encode_rows(const float*, int, int64_t, int, int, int, void*, hipStream_t);
int sk_frag_rows_mask_blocks();
int sk_frag_rows_mask(const void*, int, int64_t, int, int64_t, void*, void*, float*, hipStream_t);
int sk_relu_mask_bias(void*, const void*, int, int64_t, int, float*, hipStream_t);
int pack_ref(int, const float* const*, const float* const*, void*, hipStream_t);
int pack_proposal_bwd(int, const float* const*, void*, hipStream_t);
int pack_mip_bwd(int, const float* const*, void*, hipStream_t);
int pack_ref_bwd(int, const float* const*, void*, hipStream_t);
int mlp_launch_ref_train(const void*, int, const nerf_amd_samples&, float*, float*, const float*, void*, float*, int, hipStream_t, unsigned long long, const unsigned long long*, float);
size_t bwd_density_grad_workspace_bytes(int, int64_t);
int bwd_density_grad(int, const void*, int, int64_t, const void*, const float*, int, const float*, int, float*, void*, hipStream_t, int);
size_t bwd_ref_workspace_bytes(int, int64_t);
int bwd_ref_backward(const void*, int, int64_t, const void*, const float*, const float*, int, const float*, int, const float*, float* const*,
                     float* const*, void*, int, hipStream_t);
int bwd_launch_prop_chain(const void*, int, const float*, int64_t, const void*, void*, hipStream_t);
int bwd_launch_mip_chain(const void*, int, const float*, const float*, int64_t, const void*, void*, hipStream_t);
size_t bwd_wgrad_workspace_bytes(int, int, int64_t);
int bwd_prop_weight_grads(int, int64_t, const void*, const void*, float* const*, float* const*, void*, hipStream_t);
int bwd_mip_weight_grads(int, int64_t, const void*, const void*, const float* const*, const float* const*, float* const*, float* const*, void*,
                         hipStream_t);
int bwd_launch_adam(float* const*, const float* const*, float* const*, float* const*, const long long*, int, float*, double, const double*, double,
                    double, double, float, hipStream_t);
int pack_proposal(int, const float* const*, const float* const*, void*, hipStream_t);
int pack_proposal128(int, const float* const*, const float* const*, void*, hipStream_t);
int pack_mip128(int, const float* const*, const float* const*, void*, hipStream_t);
int pack_mip(int, const float* const*, const float* const*, void*, hipStream_t);
int sk_positional_encoding(const float*, int64_t, int, float*, hipStream_t);
int sk_ipe_feature(const float*, const float*, int64_t, int, int, float, const float*, float*, float*, float*, int, hipStream_t);
int sk_dirs_norm(const float*, int64_t, float*, hipStream_t);
int sk_dirs_norm_scratch(const float*, int64_t, float*, void*, hipStream_t);
int sk_train_sampler(const float*, const int64_t*, int64_t, const float*, const float*, float, float, float, float, int64_t, int, uint64_t, const uint64_t*,
                     float*, float*, float*, float*, hipStream_t);
int sk_philox_uniforms(float*, int64_t, int, uint64_t, const uint64_t*, int64_t, int, hipStream_t);
int sk_philox_normal(float*, int64_t, uint64_t, const uint64_t*, float, int64_t, hipStream_t);
size_t gk_gemm_workspace_bytes(int64_t, int64_t, int64_t);
int gk_gemm(int, int64_t, int64_t, int64_t, const float*, int64_t, int64_t, const float*, int64_t, int64_t, float*, int64_t, const float*, int, const float*, int64_t,
            void*, hipStream_t);
int gk_sigmoid_backward(const float*, int64_t, const float*, int64_t, int64_t, int, float*, int64_t, hipStream_t);
int gr_dir_inputs(const float*, int64_t, const float*, int64_t, int64_t, int, const float*, float*, int64_t, float*, hipStream_t);
int gr_dir_inputs_backward(const float*, int64_t, const float*, int64_t, int64_t, int, const float*, const float*, int64_t, const float*, int64_t, float*, int64_t,
                           hipStream_t);
int gr_combine(const float*, int64_t, const float*, int64_t, int64_t, int, float*, hipStream_t);
int gr_combine_backward(const float*, int64_t, const float*, int64_t, const float*, int64_t, int64_t, int, float*, int64_t, float*, int64_t, hipStream_t);
int gr_pe_backward(const float*, int64_t, const float*, int64_t, int64_t, int, int, float*, hipStream_t);
int gr_add_rows(float*, int64_t, const float*, int64_t, int64_t, int, hipStream_t);
int gr_contract(const float*, int64_t, int64_t, const float*, int64_t, float*, hipStream_t);
int rg_rows_gemm(int64_t, int64_t, int64_t, const void*, int64_t, const void*, int64_t, int64_t, const float*, int, void*, int64_t, int, hipStream_t);
int rg_rows_to_bf16(const float*, int64_t, int64_t, int64_t, int, int, void*, int64_t, hipStream_t);
int sk_advance_seed(uint64_t*, hipStream_t);
int sk_cone_parameters(const float*, int64_t, int, float, float*, float*, float*, hipStream_t);
int sk_generate_rays(const float*, int, int, float, float, int64_t, int64_t, float*, hipStream_t);
int sk_length2pts(const float*, const float*, int64_t, int, float*, hipStream_t);
int sk_sigma_to_weights(const float*, const float*, const float*, int64_t, int, int, float*, hipStream_t);
int sk_max_blur(const float*, int64_t, int, float, float*, hipStream_t);
int sk_inverse_sample(const float*, const float*, const float*, int64_t, int, int, int, int, float*, int64_t*, int64_t*, hipStream_t);
int sk_pixel_rays(const float*, float, float, const int64_t*, int64_t, float*, hipStream_t);
int sk_stratified_points(const float*, const float*, const float*, float, int64_t, int, float*, float*, hipStream_t);
int sk_resample(const float*, const float*, const float*, const float*, float, const float*, int, const float*, int64_t, int,
                int, int, float, uint64_t, int64_t, float*, int64_t*, float*, float*, hipStream_t);
int sk_composite(const float*, const float*, int, const float*, int, int64_t, int, int, int, float, float, float, const float*,
                 const float*, float*, float*, float*, float*, hipStream_t);
int sk_get_bounds(const float*, const int64_t*, int64_t, int, int, float*, hipStream_t);
int sk_weights_backward(const float*, int, int, const float*, int, const float*, int, int64_t, int, int, int, float, const float*, const float*,
                        const float*, const float*, int, float, float, float*, int, int, float*, hipStream_t);
int sk_max_blur_backward(const float*, const float*, int64_t, int, float*, hipStream_t);
int sk_get_bounds_backward(const int64_t*, const float*, int64_t, int, int, float*, hipStream_t);

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a = "") {
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}
int hip_status(int e, const char* where) {
    if (e == 0) return NERF_AMD_OK;
    snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", where, e, hipGetErrorString((hipError_t)e));
    return NERF_AMD_EHIP;
}
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

int check_samples(const nerf_amd_samples* s, bool need_dir) {
    if (!s) return fail(NERF_AMD_EINVAL, "samples descriptor is NULL");
    if (s->M < 0) return fail(NERF_AMD_EINVAL, "negative sample count");
    if (s->mode == 0) {
        if (!s->pts && s->M) return fail(NERF_AMD_EINVAL, "mode 0 needs pts");
        if (s->pts_stride < (need_dir ? 6 : 3)) return fail(NERF_AMD_EINVAL, "pts_stride too small for this network");
    } else if (s->mode == 1 || s->mode == 2) {
        if (s->S <= 0) return fail(NERF_AMD_EINVAL, "S must be positive");
        if (s->mode == 1 && !s->rays && s->M) return fail(NERF_AMD_EINVAL, "mode 1 needs rays");
        if (s->mode == 2 && (s->H <= 0 || s->W <= 0)) return fail(NERF_AMD_EINVAL, "mode 2 needs H, W");
        if (!s->z && !s->z_base && s->M) return fail(NERF_AMD_EINVAL, "need z, or z_base (with u, or without: in-kernel uniforms from rng_seed)");
    } else {
        return fail(NERF_AMD_EINVAL, "unknown sample mode");
    }
    if (s->ipe) {
        if (!need_dir) return fail(NERF_AMD_EUNSUPPORTED, "integrated PE is wired for the MipNeRF kernel only");
        if (s->mode != 1 || !s->z || s->z_stride < s->S + 1) return fail(NERF_AMD_EINVAL, "integrated PE needs mode 1 with S+1 depths per ray (z, z_stride > S)");
        if (!s->ipe_dir_norm || !(s->ipe_radius > 0.0f)) return fail(NERF_AMD_EINVAL, "integrated PE needs ipe_dir_norm (device) and a positive ipe_radius");
    }
    return NERF_AMD_OK;
}
bool bad_prec(int p) { return p != NERF_AMD_F32 && p != NERF_AMD_BF16; }
}  // namespace

extern "C" {

const char* nerf_amd_last_error(void) { return g_err; }
int nerf_amd_version(void) { return 125; }

int nerf_amd_device_info(int* n_cu, int* arch_is_gfx950) {
    int dev = 0;
    hipDeviceProp_t p;
    int e = (int)hipGetDevice(&dev);
    if (!e) e = (int)hipGetDeviceProperties(&p, dev);
    if (e) return hip_status(e, "nerf_amd_device_info");
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (arch_is_gfx950) *arch_is_gfx950 = strncmp(p.gcnArchName, "gfx950", 6) == 0;
    return NERF_AMD_OK;
}

size_t nerf_amd_packed_bytes(int net, int precision) {
    if (bad_prec(precision)) return 0;
    if (net == NERF_AMD_NET_PROPOSAL) return PropLayout::packed_bytes(precision);
    if (net == NERF_AMD_NET_MIP) return MipLayout::packed_bytes(precision);
    if (net == NERF_AMD_NET_REF) return RefLayout::packed_bytes(precision);
    if (net == NERF_AMD_NET_PROPOSAL_128) return PropLayout128::packed_bytes(precision);
    if (net == NERF_AMD_NET_MIP_128) return MipLayout128::packed_bytes(precision);
    return 0;
}
static int launch_mip_any(int flags, const void* packed, int precision, const nerf_amd_samples& s, float* rgbo, hipStream_t st) {
    return (flags & NERF_AMD_FINE_W128) ? mlp_launch_mip128(packed, precision, s, rgbo, st) : mlp_launch_mip(packed, precision, s, rgbo, st);
}
// the proposal pass of an entry point: `flags` = the layout bits of its precision argument
static int launch_proposal_any(int flags, const void* packed, int precision, const nerf_amd_samples& s, float* density, hipStream_t st) {
    return (flags & NERF_AMD_PROP_W128) ? mlp_launch_proposal128(packed, precision, s, density, st) : mlp_launch_proposal(packed, precision, s, density, st);
}

int nerf_amd_pack_weights(int net, int precision, const float* const* weights, const float* const* biases, int n_tensors,
                          void* packed, void* stream) {
    if (bad_prec(precision)) return fail(NERF_AMD_EINVAL, "unknown precision");
    if (!weights || !biases || !packed) return fail(NERF_AMD_EINVAL, "NULL argument");
    const int want = (net == NERF_AMD_NET_PROPOSAL || net == NERF_AMD_NET_PROPOSAL_128) ? 5
                     : ((net == NERF_AMD_NET_MIP || net == NERF_AMD_NET_MIP_128) ? 11 : (net == NERF_AMD_NET_REF ? 20 : -1));
    if (want < 0) return fail(NERF_AMD_EINVAL, "unknown network");
    if (n_tensors != want) return fail(NERF_AMD_EINVAL, "wrong number of weight tensors for this network");
    for (int i = 0; i < want; ++i)
        if (!weights[i] || !biases[i]) return fail(NERF_AMD_EINVAL, "NULL weight or bias tensor");
    const int e = net == NERF_AMD_NET_PROPOSAL ? pack_proposal(precision, weights, biases, packed, S(stream))
                  : net == NERF_AMD_NET_PROPOSAL_128 ? pack_proposal128(precision, weights, biases, packed, S(stream))
                  : net == NERF_AMD_NET_MIP_128 ? pack_mip128(precision, weights, biases, packed, S(stream))
                  : (net == NERF_AMD_NET_MIP ? pack_mip(precision, weights, biases, packed, S(stream))
                                             : pack_ref(precision, weights, biases, packed, S(stream)));
    return hip_status(e, "nerf_amd_pack_weights");
}

int nerf_amd_proposal_forward(const void* packed, int precision, const nerf_amd_samples* src, float* density, void* stream) {
    const int lflags = precision & ~0xff;
    precision &= 0xff;
    if (bad_prec(precision) || (lflags & ~NERF_AMD_PROP_W128)) return fail(NERF_AMD_EINVAL, "unknown precision");
    if (int c = check_samples(src, false)) return c;
    if (src->M == 0) return NERF_AMD_OK;
    if (!packed || !density) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(launch_proposal_any(lflags, packed, precision, *src, density, S(stream)), "nerf_amd_proposal_forward");
}

int nerf_amd_mip_forward(const void* packed, int precision, const nerf_amd_samples* src, float* rgbo, void* stream) {
    const int lflags = precision & ~0xff;
    precision &= 0xff;
    if (bad_prec(precision) || (lflags & ~NERF_AMD_FINE_W128)) return fail(NERF_AMD_EINVAL, "unknown precision");
    if (int c = check_samples(src, true)) return c;
    if (src->M == 0) return NERF_AMD_OK;
    if (!packed || !rgbo) return fail(NERF_AMD_EINVAL, "NULL argument");
    if ((lflags & NERF_AMD_FINE_W128) && src->ipe) return fail(NERF_AMD_EINVAL, "the 128-wide fine layout has no integrated-PE kernel: pack the network 256-wide");
    return hip_status(launch_mip_any(lflags, packed, precision, *src, rgbo, S(stream)), "nerf_amd_mip_forward");
}

int nerf_amd_mip_forward_composite(const void* packed, int precision, const nerf_amd_samples* src, int white_bkg, float near,
                                   float far, float* rgb, float* depth, float* weights, void* stream) {
    const int lflags = precision & ~0xff;                   // layout flags ride in `precision` (nerf_amd.h): strip them before bad_prec
    precision &= 0xff;
    if (bad_prec(precision) || (lflags & ~NERF_AMD_FINE_W128)) return fail(NERF_AMD_EINVAL, "unknown precision");
    if (lflags & NERF_AMD_FINE_W128)                        // (a NET_MIP_128 blob walked by the 256-wide kernel = wrong image + reads past the blob)
        return fail(NERF_AMD_EUNSUPPORTED, "the 128-wide fine layout has no fused-compositing kernel: use nerf_amd_mip_forward + nerf_amd_composite");
    if (int c = check_samples(src, true)) return c;
    if (src->mode != 1 || !src->z) return fail(NERF_AMD_EUNSUPPORTED, "fused compositing needs mode 1 (rays + z)");
    if (src->ipe) return fail(NERF_AMD_EUNSUPPORTED, "integrated PE: use nerf_amd_mip_forward + nerf_amd_composite");
    if (src->S != 32 && src->S != 64 && src->S != 128) return fail(NERF_AMD_EUNSUPPORTED, "fused compositing needs S in {32, 64, 128}");
    if (src->M == 0) return NERF_AMD_OK;
    if (!packed || !rgb) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(mlp_launch_mip_composite(packed, precision, *src, rgb, depth, weights, white_bkg, near, far, S(stream)),
                      "nerf_amd_mip_forward_composite");
}

static bool bad_ref_flags(int f) { return (f & ~NERF_AMD_REF_SRGB) != 0; }
int nerf_amd_ref_forward(const void* packed, int precision, const nerf_amd_samples* src, int ref_flags, float* rgbo, float* normal, void* stream) {
    if (bad_prec(precision) || bad_ref_flags(ref_flags)) return fail(NERF_AMD_EINVAL, "unknown precision or ref_flags");
    if (int c = check_samples(src, true)) return c;
    if (src->M == 0) return NERF_AMD_OK;
    if (!packed || !rgbo) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(mlp_launch_ref(packed, precision, *src, rgbo, normal, nullptr, ref_flags, S(stream)), "nerf_amd_ref_forward");
}
int nerf_amd_ref_forward_train(const void* packed, int precision, const nerf_amd_samples* src, int ref_flags, const float* bn_noise, float* rgbo,
                               float* normal, void* stream) {
    if (bad_prec(precision) || bad_ref_flags(ref_flags)) return fail(NERF_AMD_EINVAL, "unknown precision or ref_flags");
    if (int c = check_samples(src, true)) return c;
    if (src->M == 0) return NERF_AMD_OK;
    if (!packed || !rgbo || !bn_noise) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(mlp_launch_ref(packed, precision, *src, rgbo, normal, bn_noise, ref_flags, S(stream)), "nerf_amd_ref_forward_train");
}

int nerf_amd_positional_encoding(const float* x, int64_t M, int L, float* out, void* stream) {
    if (M < 0 || L < 1 || L > 24) return fail(NERF_AMD_EINVAL, "bad M or L");
    if (M && (!x || !out)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_positional_encoding(x, M, L, out, S(stream)), "nerf_amd_positional_encoding");
}

int nerf_amd_ipe_feature(const float* z, const float* rays, int64_t N, int Sn, int L, float r, const float* dir_norm, float* feat, float* mu,
                         float* mu_t, void* stream) {
    if (N < 0 || Sn < 0 || L < 1 || L > 15) return fail(NERF_AMD_EINVAL, "bad size or L (1..15)");
    if (N * Sn && (!z || !rays || !dir_norm || !feat)) return fail(NERF_AMD_EINVAL, "NULL argument");
    const float r2 = (float)((double)r * (double)r);        // Python's `r ** 2` is a double, rounded when it meets the fp32 tensor
    return hip_status(sk_ipe_feature(z, rays, N, Sn, L, r2, dir_norm, feat, mu, mu_t, 0, S(stream)), "nerf_amd_ipe_feature");
}
int nerf_amd_ipe_feature_contracted(const float* z, const float* rays, int64_t N, int Sn, int L, float r, const float* dir_norm, float* feat, float* mu,
                                    float* mu_t, void* stream) {
    if (N < 0 || Sn < 0 || L < 1 || L > 15) return fail(NERF_AMD_EINVAL, "bad size or L (1..15)");
    if (N * Sn && (!z || !rays || !dir_norm || !feat)) return fail(NERF_AMD_EINVAL, "NULL argument");
    const float r2 = (float)((double)r * (double)r);
    return hip_status(sk_ipe_feature(z, rays, N, Sn, L, r2, dir_norm, feat, mu, mu_t, 1, S(stream)), "nerf_amd_ipe_feature_contracted");
}
int nerf_amd_cone_parameters(const float* z, int64_t N, int Sn, float r, float* mu_t, float* var_t, float* var_r, void* stream) {
    if (N < 0 || Sn < 0) return fail(NERF_AMD_EINVAL, "negative size");
    if (N * Sn && (!z || !mu_t || !var_t || !var_r)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_cone_parameters(z, N, Sn, (float)((double)r * (double)r), mu_t, var_t, var_r, S(stream)), "nerf_amd_cone_parameters");
}
int nerf_amd_dirs_norm(const float* rays, int64_t N, float* out, void* stream) {
    if (N < 0 || !out || (N && !rays)) return fail(NERF_AMD_EINVAL, "bad argument");
    return hip_status(sk_dirs_norm(rays, N, out, S(stream)), "nerf_amd_dirs_norm");
}

int nerf_amd_generate_rays(const float* pose_host, int H, int W, float fx, float fy, int64_t ray_offset, int64_t N, float* rays,
                           void* stream) {
    if (!pose_host || !rays || H <= 0 || W <= 0) return fail(NERF_AMD_EINVAL, "bad argument");
    if (ray_offset < 0 || N < 0 || ray_offset + N > (int64_t)H * W) return fail(NERF_AMD_EINVAL, "ray range outside the image");
    return hip_status(sk_generate_rays(pose_host, H, W, fx, fy, ray_offset, N, rays, S(stream)), "nerf_amd_generate_rays");
}

int nerf_amd_length2pts(const float* rays, const float* z, int64_t N, int Sn, float* out, void* stream) {
    if (N < 0 || Sn < 0) return fail(NERF_AMD_EINVAL, "negative size");
    if (N * Sn && (!rays || !z || !out)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_length2pts(rays, z, N, Sn, out, S(stream)), "nerf_amd_length2pts");
}

int nerf_amd_sigma_to_weights(const float* sigma, const float* z, const float* dirs, int64_t N, int Sn, int act, float* w,
                              void* stream) {
    if (N < 0 || Sn < 0 || act < 0 || act > 2) return fail(NERF_AMD_EINVAL, "bad size or activation");
    if (N * Sn && (!sigma || !z || !w)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_sigma_to_weights(sigma, z, dirs, N, Sn, act, w, S(stream)), "nerf_amd_sigma_to_weights");
}

int nerf_amd_max_blur(const float* w, int64_t N, int Sn, float alpha, float* out, void* stream) {
    if (N < 0 || Sn < 0) return fail(NERF_AMD_EINVAL, "negative size");
    if (N * Sn && (!w || !out)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_max_blur(w, N, Sn, alpha, out, S(stream)), "nerf_amd_max_blur");
}

int nerf_amd_inverse_sample(const float* w, const float* z, const float* u, int64_t N, int C, int K, int sort, float* z_out,
                            int64_t* below, void* stream) {
    if (N < 0 || C < 3 || C > 256 || K < 1 || K > 1024) return fail(NERF_AMD_EINVAL, "need 3 <= C <= 256 and 1 <= K <= 1024");
    if (N && (!w || !z || !u || !z_out)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_inverse_sample(w, z, u, N, C, K, sort, 0, z_out, below, nullptr, S(stream)), "nerf_amd_inverse_sample");
}

int nerf_amd_sample_pdf(const float* bins, const float* weights, const float* u, int64_t N, int B, int K, float* samples,
                        int64_t* below, int64_t* above, void* stream) {
    if (N < 0 || B < 2 || B > 256 || K < 1 || K > 1024) return fail(NERF_AMD_EINVAL, "need 2 <= B <= 256 and 1 <= K <= 1024");
    if (N && (!bins || !weights || !u || !samples)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_inverse_sample(weights, bins, u, N, B, K, 0, 1, samples, below, above, S(stream)), "nerf_amd_sample_pdf");
}

int nerf_amd_pixel_rays(const float* pose_host, float fx, float fy, const int64_t* coords, int64_t N, float* rays, void* stream) {
    if (!pose_host || N < 0 || (N && (!coords || !rays))) return fail(NERF_AMD_EINVAL, "bad argument");
    return hip_status(sk_pixel_rays(pose_host, fx, fy, coords, N, rays, S(stream)), "nerf_amd_pixel_rays");
}

int nerf_amd_sample_training_rays(const float* rgbs, const int64_t* coords, int64_t n_pixels, const float* pose_host, float fx, float fy,
                                  float near, float far, int64_t N, int C, uint64_t rng_seed, float* pts, float* lengths, float* rgb, float* rays,
                                  void* stream) {
    if (N < 0 || C < 0 || n_pixels < 1) return fail(NERF_AMD_EINVAL, "bad size");
    if (!pose_host || (N && (!rgbs || !coords || !rgb || !rays))) return fail(NERF_AMD_EINVAL, "NULL argument");
    if ((pts == nullptr) != (lengths == nullptr) || (pts && C < 1)) return fail(NERF_AMD_EINVAL, "pts and lengths go together (C >= 1)");
    return hip_status(sk_train_sampler(rgbs, coords, n_pixels, pose_host, nullptr, fx, fy, near, far, N, C, rng_seed, nullptr, pts, lengths, rgb, rays, S(stream)),
                      "nerf_amd_sample_training_rays");
}
int nerf_amd_sample_training_rays_dev(const float* rgbs, const int64_t* coords, int64_t n_pixels, const float* pose_dev, float fx, float fy,
                                      float near, float far, int64_t N, int C, const uint64_t* seed_dev, float* pts, float* lengths, float* rgb,
                                      float* rays, void* stream) {
    if (N < 0 || C < 0 || n_pixels < 1) return fail(NERF_AMD_EINVAL, "bad size");
    if (!pose_dev || !seed_dev || (N && (!rgbs || !coords || !rgb || !rays))) return fail(NERF_AMD_EINVAL, "NULL argument");
    if ((pts == nullptr) != (lengths == nullptr) || (pts && C < 1)) return fail(NERF_AMD_EINVAL, "pts and lengths go together (C >= 1)");
    return hip_status(sk_train_sampler(rgbs, coords, n_pixels, nullptr, pose_dev, fx, fy, near, far, N, C, 0, seed_dev, pts, lengths, rgb, rays, S(stream)),
                      "nerf_amd_sample_training_rays_dev");
}
int nerf_amd_philox_uniforms(float* out, int64_t N, int K, uint64_t rng_seed, const uint64_t* seed_dev, void* stream) {
    if (N < 0 || K < 0) return fail(NERF_AMD_EINVAL, "negative size");
    if (N * K && !out) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_philox_uniforms(out, N, K, rng_seed, seed_dev, 0, 0, S(stream)), "nerf_amd_philox_uniforms");
}
int nerf_amd_philox_stream(float* out, int64_t N, int K, uint64_t rng_seed, const uint64_t* seed_dev, int64_t ray_offset, int stream_id, void* stream) {
    if (N < 0 || K < 0 || ray_offset < 0) return fail(NERF_AMD_EINVAL, "negative size");
    if (stream_id != NERF_AMD_PHILOX_INV && stream_id != NERF_AMD_PHILOX_STRAT) return fail(NERF_AMD_EINVAL, "unknown stream_id");
    if (stream_id == NERF_AMD_PHILOX_STRAT && K > 64) return fail(NERF_AMD_EINVAL, "the stratified stream has 64 slots per ray");
    if (N * K && !out) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_philox_uniforms(out, N, K, rng_seed, seed_dev, ray_offset, stream_id == NERF_AMD_PHILOX_STRAT, S(stream)), "nerf_amd_philox_stream");
}
int nerf_amd_advance_seed(uint64_t* seed_dev, void* stream) {
    if (!seed_dev) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_advance_seed(seed_dev, S(stream)), "nerf_amd_advance_seed");
}

int nerf_amd_stratified_points(const float* rays, const float* z_base, const float* u, float z_jitter, int64_t N, int Sn,
                               float* z_out, float* pts, void* stream) {
    if (N < 0 || Sn < 0) return fail(NERF_AMD_EINVAL, "negative size");
    if (N * Sn && (!z_base || !u || !z_out || (pts && !rays))) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_stratified_points(rays, z_base, u, z_jitter, N, Sn, z_out, pts, S(stream)), "nerf_amd_stratified_points");
}

int nerf_amd_resample(const float* density, const float* z, const float* z_base, const float* u_strat, float z_jitter,
                      const float* dirs, int dirs_stride, const float* u_inv, int64_t N, int C, int K, int softplus_density,
                      float blur_alpha, uint64_t rng_seed, int64_t rng_ray_offset, float* z_fine, int64_t* below, float* w_prop,
                      float* z_coarse, void* stream) {
    if (N < 0 || C < 3 || C > 256 || K < 1 || K > 1024) return fail(NERF_AMD_EINVAL, "need 3 <= C <= 256 and 1 <= K <= 1024");
    if (N && (!density || !dirs || !z_fine)) return fail(NERF_AMD_EINVAL, "NULL argument");
    if (N && !z && !z_base) return fail(NERF_AMD_EINVAL, "need z, or z_base (with u_strat, or without: in-kernel uniforms)");
    return hip_status(sk_resample(density, z, z_base, u_strat, z_jitter, dirs, dirs_stride, u_inv, N, C, K, softplus_density,
                                  blur_alpha, rng_seed, rng_ray_offset, z_fine, below, w_prop, z_coarse, S(stream)), "nerf_amd_resample");
}

int nerf_amd_composite(const float* rgbo, const float* z, int z_stride, const float* dirs, int dirs_stride, int64_t N, int Sn,
                       int flags, int act, float sigma_shift, float near, float far, const float* normal, const float* cam_dir, float* rgb,
                       float* weights, float* depth, float* normal_img, void* stream) {
    if (N < 0 || Sn < 1 || act < 0 || act > 2 || z_stride < Sn) return fail(NERF_AMD_EINVAL, "bad size, stride or activation");
    if (N && (!rgbo || !z || !dirs || !rgb)) return fail(NERF_AMD_EINVAL, "NULL argument");
    if (normal_img && !(normal && cam_dir)) return fail(NERF_AMD_EINVAL, "normal_img needs normal and cam_dir");
    return hip_status(sk_composite(rgbo, z, z_stride, dirs, dirs_stride, N, Sn, flags, act, sigma_shift, near, far, normal, cam_dir,
                                   rgb, weights, depth, normal_img, S(stream)), "nerf_amd_composite");
}

int nerf_amd_get_bounds(const float* w_prop, const int64_t* below, int64_t N, int C, int K, float* bounds, void* stream) {
    if (N < 0 || C < 1 || C > 4096 || K < 2) return fail(NERF_AMD_EINVAL, "bad size");
    if (N && (!w_prop || !below || !bounds)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_get_bounds(w_prop, below, N, C, K, bounds, S(stream)), "nerf_amd_get_bounds");
}

int nerf_amd_merge_depths(const float* z_fine, const float* z_coarse, int64_t N, int K, int C, float* z_out, void* stream) {
    if (N < 0 || K < 1 || C < 1 || K + C > 2048) return fail(NERF_AMD_EINVAL, "bad size (K + C <= 2048: four rays of depths and their sort scratch per workgroup live in 64 KiB of LDS)");
    if (N && (!z_fine || !z_coarse || !z_out)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_merge_sorted(z_fine, z_coarse, N, K, C, z_out, S(stream)), "nerf_amd_merge_depths");
}

int nerf_amd_merge_depths_order(const float* z_fine, const float* z_coarse, const int64_t* f_inds, int64_t N, int K, int C, float* z_out, int64_t* order,
                                int64_t* all_inds, void* stream) {
    if (N < 0 || K < 1 || C < 1 || K + C > 1024) return fail(NERF_AMD_EINVAL, "bad size (K + C <= 1024: four rays of depths, indices and their sort scratch per workgroup live in 64 KiB of LDS)");
    if (N && (!z_fine || !z_coarse || !z_out || !order)) return fail(NERF_AMD_EINVAL, "NULL argument");
    if (N && all_inds && !f_inds) return fail(NERF_AMD_EINVAL, "all_inds needs f_inds");
    return hip_status(sk_merge_sorted_order(z_fine, z_coarse, f_inds, N, K, C, z_out, order, all_inds, S(stream)), "nerf_amd_merge_depths_order");
}

int nerf_amd_coarse_grad_select(const float* grads, const int64_t* sort_inds, int64_t N, int T, int D, int c_pnum, float* out, void* stream) {
    if (N < 0 || T < 1 || D < 1 || c_pnum < 0 || c_pnum > T) return fail(NERF_AMD_EINVAL, "bad size");
    if (N && c_pnum && (!grads || !sort_inds || !out)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_coarse_grad_select(grads, sort_inds, N, T, D, c_pnum, out, S(stream)), "nerf_amd_coarse_grad_select");
}

int nerf_amd_weighted_dot_loss(const float* w, const float* a, const float* b, int64_t M, int mode, float scale, float* out, float* workspace, void* stream) {
    if (M < 0 || (mode != 0 && mode != 1)) return fail(NERF_AMD_EINVAL, "bad size / mode");
    if (!out || !workspace || (M && (!w || !a || !b))) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_weighted_dot_loss(w, a, b, M, mode, scale, out, workspace, S(stream)), "nerf_amd_weighted_dot_loss");
}

int nerf_amd_weighted_dot_loss_backward(const float* g, const float* w, const float* a, const float* b, int64_t M, int mode, float scale, float* d_w,
                                        float* d_a, float* d_b, void* stream) {
    if (M < 0 || (mode != 0 && mode != 1)) return fail(NERF_AMD_EINVAL, "bad size / mode");
    if (M && (!g || !w || !a || !b)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_weighted_dot_loss_backward(g, w, a, b, M, mode, scale, d_w, d_a, d_b, S(stream)), "nerf_amd_weighted_dot_loss_backward");
}

int nerf_amd_mfma_stream(int iters, int workgroups, int mode, float* sink, void* stream) {
    if (iters < 0 || workgroups < 1 || mode < 0 || mode > 3 || !sink) return fail(NERF_AMD_EINVAL, "bad argument");
    return hip_status(pack_mfma_stream(iters, workgroups, mode, sink, S(stream)), "nerf_amd_mfma_stream");
}

// ---- training forward: the MLP kernels also dump their hidden activations ----
static int train_layers(int net) {
    return net == NERF_AMD_NET_PROPOSAL ? PROP_DUMP_SLOTS : (net == NERF_AMD_NET_MIP ? MIP_DUMP_SLOTS : (net == NERF_AMD_NET_REF ? REF_DUMP_SLOTS : 0));
}
static bool bad_train_prec(int p, int net = NERF_AMD_NET_MIP) {           // the fp8-dump mode exists for the proposal and MipNeRF networks
    return p != NERF_AMD_F32 && p != NERF_AMD_BF16 && !(p == NERF_AMD_BF16_F8 && net != NERF_AMD_NET_REF);
}
size_t nerf_amd_train_dump_bytes(int net, int precision, int64_t M) {
    if (M < 0 || !train_layers(net) || bad_train_prec(precision, net)) return 0;
    // activation slots + one ReLU bit per activation: 1 KiB per slot and 32-sample subtile (Ref-NeRF too since round 4)
    const size_t bits = (size_t)train_layers(net) * mlp_train_mask_stride(precision, M);
    return (size_t)train_layers(net) * mlp_train_layer_stride(precision, M) + bits;
}
int nerf_amd_proposal_forward_train(const void* packed, int precision, const nerf_amd_samples* src, float* density, void* dump, void* stream) {
    if (!packed || !src || !dump) return fail(NERF_AMD_EINVAL, "NULL argument");
    if (bad_train_prec(precision)) return fail(NERF_AMD_EINVAL, "bad precision");
    if (src->M && !density) return fail(NERF_AMD_EINVAL, "NULL output");
    if (int c = check_samples(src, false)) return c;
    return hip_status(mlp_launch_proposal_train(packed, precision, *src, density, dump, S(stream)), "nerf_amd_proposal_forward_train");
}
int nerf_amd_mip_forward_train(const void* packed, int precision, const nerf_amd_samples* src, float* rgbo, void* dump, void* stream) {
    if (!packed || !src || !dump) return fail(NERF_AMD_EINVAL, "NULL argument");
    if (bad_train_prec(precision)) return fail(NERF_AMD_EINVAL, "bad precision");
    if (src->M && !rgbo) return fail(NERF_AMD_EINVAL, "NULL output");
    if (int c = check_samples(src, true)) return c;         // (every sample mode, integrated PE and scene contraction included)
    return hip_status(mlp_launch_mip_train(packed, precision, *src, rgbo, dump, S(stream)), "nerf_amd_mip_forward_train");
}
int nerf_amd_train_dump_to_rows(const void* dump, int net, int precision, int64_t M, int layer, int n_features, void* out, void* stream) {
    if (M < 0 || layer < 0 || layer >= train_layers(net) || n_features < 16 || n_features > 256 || (n_features & 15))
        return fail(NERF_AMD_EINVAL, "bad layer or feature count");
    if (M && (!dump || !out)) return fail(NERF_AMD_EINVAL, "NULL argument");
    const size_t stride = mlp_train_layer_stride(precision, M);
    const int elem = precision == NERF_AMD_BF16 ? 2 : 4;
    const int64_t n_sub = (int64_t)(stride / (16 * (size_t)512 * elem));
    return hip_status(sk_frag_to_rows(reinterpret_cast<const char*>(dump) + (size_t)layer * stride, elem, n_sub, n_features / 16, M, out, S(stream)),
                      "nerf_amd_train_dump_to_rows");
}

int64_t nerf_amd_train_dump_rows_mask_partials(void) { return (int64_t)sk_frag_rows_mask_blocks() * 4; }
int nerf_amd_train_dump_rows_mask(const void* dump, int net, int precision, int64_t M, int layer, int n_features, void* act_out, void* delta,
                                  float* col_sum, void* stream) {
    if (M < 0 || layer < 0 || layer >= train_layers(net) || (n_features != 128 && n_features != 256))
        return fail(NERF_AMD_EINVAL, "bad layer or feature count (128 or 256)");
    if (precision != NERF_AMD_F32 && precision != NERF_AMD_BF16) return fail(NERF_AMD_EINVAL, "bad precision");
    if (!col_sum || (M && (!dump || !act_out || !delta))) return fail(NERF_AMD_EINVAL, "NULL argument");
    const size_t stride = mlp_train_layer_stride(precision, M);
    const int elem = precision == NERF_AMD_BF16 ? 2 : 4;
    const int64_t n_sub = (int64_t)(stride / (16 * (size_t)512 * elem));
    return hip_status(sk_frag_rows_mask(reinterpret_cast<const char*>(dump) + (size_t)layer * stride, elem, n_sub, n_features / 16, M, act_out, delta, col_sum,
                                        S(stream)), "nerf_amd_train_dump_rows_mask");
}

int nerf_amd_relu_mask(void* delta, const void* act, int precision, int64_t n, void* stream) {
    if (n < 0 || (precision == NERF_AMD_BF16 && (n & 1))) return fail(NERF_AMD_EINVAL, "bad element count (bf16: even)");
    if (n && (!delta || !act)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_relu_mask(delta, act, precision == NERF_AMD_BF16 ? 2 : 4, n, S(stream)), "nerf_amd_relu_mask");
}

int64_t nerf_amd_relu_mask_bias_partials(int precision, int64_t rows, int cols) {
    const int W = cols * (precision == NERF_AMD_BF16 ? 2 : 4) / 4;
    if (rows <= 0 || W < 1 || W > 256 || (256 % W) != 0) return 0;
    const int rpb = 256 / W;
    int64_t blocks = (rows + rpb - 1) / rpb;
    if (blocks > 1024) blocks = 1024;
    return blocks * rpb;
}
int nerf_amd_relu_mask_bias(void* delta, const void* act, int precision, int64_t rows, int cols, float* col_sum, void* stream) {
    if (rows < 0 || cols < 2 || (precision != NERF_AMD_F32 && precision != NERF_AMD_BF16)) return fail(NERF_AMD_EINVAL, "bad size or precision");
    if (rows && (!delta || !act || !col_sum)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_relu_mask_bias(delta, act, precision == NERF_AMD_BF16 ? 2 : 4, rows, cols, col_sum, S(stream)), "nerf_amd_relu_mask_bias");
}

int nerf_amd_encode_rows(const float* x, int x_stride, int64_t M, int L, int normalize, int precision, void* out, void* stream) {
    if (M < 0 || x_stride < 3 || (L != 4 && L != 10) || (precision != NERF_AMD_F32 && precision != NERF_AMD_BF16)) return fail(NERF_AMD_EINVAL, "bad size, L (4 or 10) or precision");
    if (M && (!x || !out)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_encode_rows(x, x_stride, M, L, normalize, precision == NERF_AMD_BF16 ? 2 : 4, out, S(stream)), "nerf_amd_encode_rows");
}

// ---- backward of the MLPs: dgrad chain on transposed packed weights, MFMA weight gradients, Adam (bwd_kernels.hip) ----
size_t nerf_amd_packed_backward_bytes(int net, int precision) {
    if (bad_prec(precision)) return 0;
    if (net == NERF_AMD_NET_PROPOSAL) return PropBwdLayout::packed_bytes(precision);
    if (net == NERF_AMD_NET_MIP) return MipBwdLayout::packed_bytes(precision);
    if (net == NERF_AMD_NET_REF) return RefBwdLayout::packed_bytes(precision);
    return 0;
}
int nerf_amd_pack_weights_backward(int net, int precision, const float* const* weights, int n_tensors, void* packed_bwd, void* stream) {
    if (bad_prec(precision)) return fail(NERF_AMD_EINVAL, "unknown precision");
    if (!weights || !packed_bwd) return fail(NERF_AMD_EINVAL, "NULL argument");
    const int want = (net == NERF_AMD_NET_PROPOSAL || net == NERF_AMD_NET_PROPOSAL_128) ? 5 : (net == NERF_AMD_NET_MIP ? 11 : (net == NERF_AMD_NET_REF ? 20 : -1));
    if (want < 0) return fail(NERF_AMD_EINVAL, "unknown network");
    if (n_tensors != want) return fail(NERF_AMD_EINVAL, "wrong number of weight tensors for this network");
    for (int i = 0; i < want; ++i)
        if (!weights[i]) return fail(NERF_AMD_EINVAL, "NULL weight tensor");
    const int e = net == NERF_AMD_NET_PROPOSAL ? pack_proposal_bwd(precision, weights, packed_bwd, S(stream))
                  : (net == NERF_AMD_NET_MIP ? pack_mip_bwd(precision, weights, packed_bwd, S(stream)) : pack_ref_bwd(precision, weights, packed_bwd, S(stream)));
    return hip_status(e, "nerf_amd_pack_weights_backward");
}
int nerf_amd_proposal_backward_chain(const void* packed_bwd, int precision, const float* g_density, int64_t M, const void* act_dump,
                                     void* delta_dump, void* stream) {
    if (bad_train_prec(precision) || M < 0) return fail(NERF_AMD_EINVAL, "bad precision or size");
    if (M == 0) return NERF_AMD_OK;
    if (!packed_bwd || !g_density || !act_dump || !delta_dump) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(bwd_launch_prop_chain(packed_bwd, precision, g_density, M, act_dump, delta_dump, S(stream)), "nerf_amd_proposal_backward_chain");
}
int nerf_amd_mip_backward_chain(const void* packed_bwd, int precision, const float* g_rgbo, const float* rgbo, int64_t M, const void* act_dump,
                                void* delta_dump, void* stream) {
    if (bad_train_prec(precision) || M < 0) return fail(NERF_AMD_EINVAL, "bad precision or size");
    if (M == 0) return NERF_AMD_OK;
    if (!packed_bwd || !g_rgbo || !rgbo || !act_dump || !delta_dump) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(bwd_launch_mip_chain(packed_bwd, precision, g_rgbo, rgbo, M, act_dump, delta_dump, S(stream)), "nerf_amd_mip_backward_chain");
}
size_t nerf_amd_weight_grads_workspace_bytes(int net, int precision, int64_t M) {
    if (bad_train_prec(precision, net) || M < 0) return 0;
    return bwd_wgrad_workspace_bytes(net, precision == NERF_AMD_BF16_F8 ? NERF_AMD_BF16 : precision, M);
}
int nerf_amd_proposal_weight_grads(int precision, int64_t M, const void* act_dump, const void* delta_dump, float* const* d_weights,
                                   float* const* d_biases, void* workspace, void* stream) {
    if (bad_train_prec(precision) || M < 0) return fail(NERF_AMD_EINVAL, "bad precision or size");
    if (!d_weights || !d_biases) return fail(NERF_AMD_EINVAL, "NULL argument");
    for (int i = 0; i < 5; ++i)
        if (!d_weights[i] || !d_biases[i]) return fail(NERF_AMD_EINVAL, "NULL gradient tensor");
    if (M == 0) return fail(NERF_AMD_EINVAL, "no samples (the caller zero-fills the gradients of an empty batch)");
    if (!act_dump || !delta_dump || !workspace) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(bwd_prop_weight_grads(precision, M, act_dump, delta_dump, d_weights, d_biases, workspace, S(stream)), "nerf_amd_proposal_weight_grads");
}
int nerf_amd_mip_weight_grads(int precision, int64_t M, const void* act_dump, const void* delta_dump, const float* const* weights,
                              const float* const* biases, float* const* d_weights, float* const* d_biases, void* workspace, void* stream) {
    if (bad_train_prec(precision) || M < 0) return fail(NERF_AMD_EINVAL, "bad precision or size");
    if (!weights || !biases || !d_weights || !d_biases) return fail(NERF_AMD_EINVAL, "NULL argument");
    for (int i = 0; i < 11; ++i)
        if (!weights[i] || !biases[i] || !d_weights[i] || !d_biases[i]) return fail(NERF_AMD_EINVAL, "NULL tensor");
    if (M == 0) return fail(NERF_AMD_EINVAL, "no samples (the caller zero-fills the gradients of an empty batch)");
    if (!act_dump || !delta_dump || !workspace) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(bwd_mip_weight_grads(precision, M, act_dump, delta_dump, weights, biases, d_weights, d_biases, workspace, S(stream)),
                      "nerf_amd_mip_weight_grads");
}
int nerf_amd_ref_forward_train_dump(const void* packed, int precision, const nerf_amd_samples* src, int ref_flags, const float* bn_noise, float* rgbo,
                                    float* normal, void* dump, float* aux, void* stream) {
    if (bad_prec(precision) || bad_ref_flags(ref_flags)) return fail(NERF_AMD_EINVAL, "unknown precision or ref_flags");
    if (int c = check_samples(src, true)) return c;
    if (src->M == 0) return NERF_AMD_OK;
    if (!packed || !rgbo || !dump || !aux) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(mlp_launch_ref_train(packed, precision, *src, rgbo, normal, bn_noise, dump, aux, ref_flags, S(stream), 0, nullptr, 0.0f), "nerf_amd_ref_forward_train_dump");
}
int nerf_amd_ref_forward_train_dump_rng(const void* packed, int precision, const nerf_amd_samples* src, int ref_flags, uint64_t noise_seed,
                                        const uint64_t* noise_seed_dev, float noise_std, float* rgbo, float* normal, void* dump, float* aux, void* stream) {
    if (bad_prec(precision) || bad_ref_flags(ref_flags)) return fail(NERF_AMD_EINVAL, "unknown precision or ref_flags");
    if (!(noise_std >= 0.0f)) return fail(NERF_AMD_EINVAL, "noise_std must be >= 0");
    if (int c = check_samples(src, true)) return c;
    if (src->M == 0) return NERF_AMD_OK;
    if (!packed || !rgbo || !dump || !aux) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(mlp_launch_ref_train(packed, precision, *src, rgbo, normal, nullptr, dump, aux, ref_flags, S(stream), noise_seed,
                                           reinterpret_cast<const unsigned long long*>(noise_seed_dev), noise_std), "nerf_amd_ref_forward_train_dump_rng");
}
int nerf_amd_philox_normal(float* out, int64_t M, uint64_t rng_seed, const uint64_t* seed_dev, float std, int64_t sample_offset, void* stream) {
    if (M < 0 || sample_offset < 0 || !(std >= 0.0f)) return fail(NERF_AMD_EINVAL, "negative size or std");
    if (M && !out) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_philox_normal(out, M, rng_seed, seed_dev, std, sample_offset, S(stream)), "nerf_amd_philox_normal");
}
size_t nerf_amd_density_grad_workspace_bytes(int net, int precision, int64_t M) {
    if (bad_prec(precision) || M < 0 || (net != NERF_AMD_NET_PROPOSAL && net != NERF_AMD_NET_REF)) return 0;
    return bwd_density_grad_workspace_bytes(precision, M);
}
int nerf_amd_density_grad(int net, const void* packed_bwd, int precision, int64_t M, const void* act_dump, const float* x, int x_stride,
                          const float* scale, int scale_stride, float* grad, void* workspace, void* stream) {
    const int contract = (net & NERF_AMD_CONTRACTED) ? 1 : 0;   // flag in `net`: the forward saw contracted positions (nerf_amd_samples.contract)
    net &= ~NERF_AMD_CONTRACTED;
    if (bad_prec(precision) || M < 0 || x_stride < 3) return fail(NERF_AMD_EINVAL, "bad precision, size or stride");
    if (net != NERF_AMD_NET_PROPOSAL && net != NERF_AMD_NET_REF) return fail(NERF_AMD_EUNSUPPORTED, "density gradients exist for the proposal and Ref-NeRF networks");
    if (M == 0) return NERF_AMD_OK;
    if (!packed_bwd || !act_dump || !x || !grad || !workspace) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(bwd_density_grad(net, packed_bwd, precision, M, act_dump, x, x_stride, scale, scale_stride, grad, workspace, S(stream), contract), "nerf_amd_density_grad");
}
size_t nerf_amd_ref_backward_workspace_bytes(int precision, int64_t M) {
    if (bad_prec(precision) || M < 0) return 0;
    return bwd_ref_workspace_bytes(precision, M);
}
int nerf_amd_ref_backward(const void* packed_bwd, int precision, int ref_flags, int64_t M, const void* act_dump, const float* aux, const float* dirs,
                          int dir_stride, const float* g_out, int g_stride, const float* ide_table, float* const* d_weights, float* const* d_biases,
                          void* workspace, void* stream) {
    if (bad_prec(precision) || bad_ref_flags(ref_flags) || M < 0 || dir_stride < 3 || g_stride < 7) return fail(NERF_AMD_EINVAL, "bad precision, flags, size or stride");
    if (!d_weights || !d_biases) return fail(NERF_AMD_EINVAL, "NULL argument");
    for (int i = 0; i < 20; ++i)
        if (!d_weights[i] || !d_biases[i]) return fail(NERF_AMD_EINVAL, "NULL gradient tensor");
    if (M == 0) return fail(NERF_AMD_EINVAL, "no samples (the caller zero-fills the gradients of an empty batch)");
    if (!packed_bwd || !act_dump || !aux || !dirs || !g_out || !ide_table || !workspace) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(bwd_ref_backward(packed_bwd, precision, M, act_dump, aux, dirs, dir_stride, g_out, g_stride, ide_table, d_weights, d_biases, workspace,
                                       ref_flags, S(stream)), "nerf_amd_ref_backward");
}

int nerf_amd_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                       int n_tensors, float* step, double lr, const double* lr_dev, double beta1, double beta2, double eps, float grad_scale,
                       void* stream) {
    if (n_tensors < 0 || (n_tensors && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)) || !step) return fail(NERF_AMD_EINVAL, "NULL argument");
    for (int i = 0; i < n_tensors; ++i)
        if (numel[i] < 0 || (numel[i] && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]))) return fail(NERF_AMD_EINVAL, "NULL tensor");
    static_assert(sizeof(long long) == sizeof(int64_t), "int64");
    return hip_status(bwd_launch_adam(params, grads, exp_avg, exp_avg_sq, reinterpret_cast<const long long*>(numel), n_tensors, step, lr, lr_dev,
                                      beta1, beta2, eps, grad_scale, S(stream)), "nerf_amd_adam_step");
}

// ---- backward of the sampling / compositing rows ----
int nerf_amd_sigma_to_weights_backward(const float* sigma, const float* z, const float* dirs, int64_t N, int Sn, int act,
                                       const float* d_weights, float* d_sigma, void* stream) {
    if (N < 0 || Sn < 1 || Sn > 1024) return fail(NERF_AMD_EINVAL, "bad size (S must be 1..1024)");
    if (N && (!sigma || !z || !d_weights || !d_sigma)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_weights_backward(sigma, 1, 0, z, Sn, dirs, 3, N, Sn, dirs ? 1 : 0, act, 0.0f, nullptr, nullptr, d_weights, nullptr, 0, 0.0f,
                                          1.0f, d_sigma, 1, 0, nullptr, S(stream)), "nerf_amd_sigma_to_weights_backward");
}
int nerf_amd_composite_backward(const float* rgbo, const float* z, int z_stride, const float* dirs, int dirs_stride, int64_t N, int Sn,
                                int flags, int act, float sigma_shift, float near, float far, const float* d_rgb,
                                const float* d_weights, const float* d_depth, float* d_rgbo, void* stream) {
    if (N < 0 || Sn < 1 || Sn > 1024 || z_stride < Sn) return fail(NERF_AMD_EINVAL, "bad size (S must be 1..1024)");
    if (N && (!rgbo || !z || !dirs || !d_rgbo)) return fail(NERF_AMD_EINVAL, "NULL argument");
    if (N && !d_rgb) return fail(NERF_AMD_EINVAL, "d_rgb is required (pass zeros when only the weights carry a gradient)");
    return hip_status(sk_weights_backward(rgbo, 4, 3, z, z_stride, dirs, dirs_stride, N, Sn, flags & 1, act, sigma_shift, rgbo, d_rgb, d_weights,
                                          d_depth, (flags >> 1) & 1, near, far, d_rgbo, 4, 3, d_rgbo, S(stream)), "nerf_amd_composite_backward");
}
int nerf_amd_max_blur_backward(const float* weights, const float* d_out, int64_t N, int Sn, float* d_weights, void* stream) {
    if (N < 0 || Sn < 1) return fail(NERF_AMD_EINVAL, "bad size");
    if (N && (!weights || !d_out || !d_weights)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_max_blur_backward(weights, d_out, N, Sn, d_weights, S(stream)), "nerf_amd_max_blur_backward");
}
int nerf_amd_get_bounds_backward(const int64_t* below, const float* d_bounds, int64_t N, int C, int K, float* d_w_prop, void* stream) {
    if (N < 0 || C < 1 || C > 4096 || K < 2 || K > 2048) return fail(NERF_AMD_EINVAL, "bad size (C <= 4096, 2 <= K <= 2048)");
    if (N && (!below || !d_bounds || !d_w_prop)) return fail(NERF_AMD_EINVAL, "NULL argument");
    return hip_status(sk_get_bounds_backward(below, d_bounds, N, C, K, d_w_prop, S(stream)), "nerf_amd_get_bounds_backward");
}

// workspace: density (N,64) | z_fine (N, n_fine+1) | rgbo (N, n_fine, 4) | rays (N, 6)
size_t nerf_amd_render_workspace_bytes(int64_t N, int n_fine) {
    if (N < 0 || n_fine < 1) return 0;
    const size_t a = ((size_t)N * 64 * 4 + 255) & ~(size_t)255;
    const size_t b = ((size_t)N * (n_fine + 1) * 4 + 255) & ~(size_t)255;
    const size_t c = ((size_t)N * n_fine * 16 + 255) & ~(size_t)255;
    const size_t d = ((size_t)N * 24 + 255) & ~(size_t)255;
    return a + b + c + d + 512;                             // + alignment slack + one scalar slot (direction norm of the IPE mode)
}

int nerf_amd_render_rays(const void* packed_prop, const void* packed_mip, int precision, const float* rays,
                         const nerf_amd_samples* camera, int64_t ray_offset, const float* z_base, const float* u_strat,
                         const float* u_inv, int64_t N, int n_fine, float near, float far, int white_bkg, float* rgb,
                         float* depth, float* weights, void* workspace, void* stream) {
    const int lflags = precision & ~0xff;
    precision &= 0xff;
    if (bad_prec(precision) || (lflags & ~(NERF_AMD_PROP_W128 | NERF_AMD_FINE_W128))) return fail(NERF_AMD_EINVAL, "unknown precision");
    if ((lflags & NERF_AMD_FINE_W128) && camera && camera->ipe) return fail(NERF_AMD_EINVAL, "the 128-wide fine layout has no integrated-PE kernel: pack the network 256-wide");
    if (N < 0 || n_fine < 1 || n_fine > 1023) return fail(NERF_AMD_EINVAL, "bad N or n_fine");
    if (N == 0) return NERF_AMD_OK;
    if (!packed_prop || !packed_mip || !z_base || !rgb || !workspace) return fail(NERF_AMD_EINVAL, "NULL argument");
    if ((u_strat == nullptr) != (u_inv == nullptr)) return fail(NERF_AMD_EINVAL, "u_strat and u_inv are both given or both NULL (in-kernel uniforms)");
    if (!u_strat && !camera) return fail(NERF_AMD_EINVAL, "in-kernel uniforms need the descriptor (rng_seed, rng_ray_offset)");
    if (!rays && !camera) return fail(NERF_AMD_EINVAL, "need rays or camera");
    const uint64_t seed = camera ? camera->rng_seed : 0;
    const int64_t ray0 = camera ? camera->rng_ray_offset : 0;
    constexpr int C = 64;                                   // procedures.py:22 RENDER_COARSE_PNUM
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float* density = reinterpret_cast<float*>(ws);
    ws += ((size_t)N * C * 4 + 255) & ~(size_t)255;
    float* z_fine = reinterpret_cast<float*>(ws);
    ws += ((size_t)N * (n_fine + 1) * 4 + 255) & ~(size_t)255;
    float* rgbo = reinterpret_cast<float*>(ws);
    ws += ((size_t)N * n_fine * 16 + 255) & ~(size_t)255;
    hipStream_t st = S(stream);
    if (!rays) {                                            // row 1: procedures.py:43-51,64
        if (camera->H <= 0 || camera->W <= 0 || ray_offset < 0 || ray_offset + N > (int64_t)camera->H * camera->W)
            return fail(NERF_AMD_EINVAL, "ray range outside the camera image");
        float* gen = reinterpret_cast<float*>(ws);
        if (int e = sk_generate_rays(camera->pose, camera->H, camera->W, camera->fx, camera->fy, ray_offset, N, gen, st))
            return hip_status(e, "ray generation");
        rays = gen;
    }
    ws += ((size_t)N * 24 + 255) & ~(size_t)255;
    float* dir_norm = reinterpret_cast<float*>(ws);
    const bool ipe = camera && camera->ipe;
    if (ipe) {                                              // row 12 inside the fine pass: the direction norm of this ray batch
        if (!(camera->ipe_radius > 0.0f)) return fail(NERF_AMD_EINVAL, "integrated PE needs a positive ipe_radius");
        // (the density buffer is scratch until the proposal pass below writes it: room for the 256 fp64 workgroup partials when N >= 8)
        // a caller that renders a SHARD of a ray list hands over the norm of the whole list (camera->ipe_dir_norm): the norm of
        // mip_methods.py:31 is over all rays of the reference's call, not over the rays this launch happens to hold
        if (camera->ipe_dir_norm) dir_norm = const_cast<float*>(camera->ipe_dir_norm);
        else if (int e = (N >= 8) ? sk_dirs_norm_scratch(rays, N, dir_norm, density, st) : sk_dirs_norm(rays, N, dir_norm, st))
            return hip_status(e, "direction norm");
    }
    const float jitter = (far - near) / (float)n_fine;      // procedures.py:59

    nerf_amd_samples sc{};                                  // rows 2-4: stratified z fused into the proposal MLP
    sc.mode = 1; sc.rays = rays; sc.S = C; sc.M = N * C; sc.z = nullptr; sc.z_base = z_base; sc.u = u_strat;
    sc.z_jitter = jitter; sc.z_stride = C;
    sc.contract = camera ? camera->contract : 0;           // (the descriptor may accompany explicit rays just to carry this flag)
    sc.rng_seed = seed; sc.rng_ray_offset = ray0;          // (read only when u_strat == NULL)
    if (int e = launch_proposal_any(lflags, packed_prop, precision, sc, density, st)) return hip_status(e, "proposal MLP");
    // rows 5-7: weights -> max-blur(0.01) -> inverse sampling of n_fine+1 sorted depths (procedures.py:68-70)
    if (int e = sk_resample(density, nullptr, z_base, u_strat, jitter, rays + 3, 6, u_inv, N, C, n_fine + 1, 0, 0.01f, seed, ray0, z_fine,
                            nullptr, nullptr, nullptr, st)) return hip_status(e, "resample");
    nerf_amd_samples sf{};                                  // rows 8-9: drop the last depth, length2pts fused into the MLP
    sf.mode = 1; sf.rays = rays; sf.S = n_fine; sf.M = N * n_fine; sf.z = z_fine; sf.z_stride = n_fine + 1;
    sf.contract = sc.contract;
    if (ipe) { sf.ipe = 1; sf.ipe_radius = camera->ipe_radius; sf.ipe_dir_norm = dir_norm; }   // frustum s = [z_fine[s], z_fine[s+1]]
    // rows 9 and 10 as two launches: measured 2-3 % faster than the fused epilogue of nerf_amd_mip_forward_composite on
    // MI355X, and the composite kernel's HBM rate stays individually measurable
    if (int e = launch_mip_any(lflags, packed_mip, precision, sf, rgbo, st)) return hip_status(e, "fine MLP");
    const int flags = 1 | (white_bkg ? 2 : 0);              // row 10
    if (int e = sk_composite(rgbo, z_fine, n_fine + 1, rays + 3, 6, N, n_fine, flags, NERF_AMD_ACT_RELU, 0.0f, near, far, nullptr,
                             nullptr, rgb, weights, depth, nullptr, st)) return hip_status(e, "composite");
    return NERF_AMD_OK;
}

// workspace of nerf_amd_render_rays_ref: density (N,64) | z_fine (N, n_fine+1) | z_coarse (N,64) | z_all (N, n_fine+64) | rgbo (N, n_fine+64, 4)
//                                        | normals (N, n_fine+64, 3) | rays (N, 6)
static size_t ref_ws_part(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
size_t nerf_amd_render_ref_workspace_bytes(int64_t N, int n_fine) {
    if (N < 0 || n_fine < 1) return 0;
    const size_t n = (size_t)N, S = (size_t)n_fine + 64;
    return ref_ws_part(n * 64 * 4) * 2 + ref_ws_part(n * (n_fine + 1) * 4) + ref_ws_part(n * S * 4) + ref_ws_part(n * S * 16) + ref_ws_part(n * S * 12) +
           n * 24 + 256;
}

int nerf_amd_render_rays_ref(const void* packed_prop, const void* packed_ref, int precision, int ref_flags, const float* rays,
                             const nerf_amd_samples* camera, int64_t ray_offset, const float* z_base, const float* u_strat,
                             const float* u_inv, int64_t N, int n_fine, float near, float far, int white_bkg, const float* cam_dir,
                             float* rgb, float* depth, float* normal_img, void* workspace, void* stream) {
    const int lflags = precision & ~0xff;
    precision &= 0xff;
    if (bad_prec(precision) || (lflags & ~NERF_AMD_PROP_W128) || bad_ref_flags(ref_flags)) return fail(NERF_AMD_EINVAL, "unknown precision or ref_flags");
    if (N < 0 || n_fine < 1 || n_fine > 1023) return fail(NERF_AMD_EINVAL, "bad N or n_fine");
    if (N == 0) return NERF_AMD_OK;
    if (!packed_prop || !packed_ref || !z_base || !rgb || !workspace) return fail(NERF_AMD_EINVAL, "NULL argument");
    if ((u_strat == nullptr) != (u_inv == nullptr)) return fail(NERF_AMD_EINVAL, "u_strat and u_inv are both given or both NULL (in-kernel uniforms)");
    if (!u_strat && !camera) return fail(NERF_AMD_EINVAL, "in-kernel uniforms need the descriptor (rng_seed, rng_ray_offset)");
    if (!rays && !camera) return fail(NERF_AMD_EINVAL, "need rays or camera");
    const uint64_t seed = camera ? camera->rng_seed : 0;
    const int64_t ray0 = camera ? camera->rng_ray_offset : 0;
    if ((normal_img != nullptr) != (cam_dir != nullptr)) return fail(NERF_AMD_EINVAL, "normal_img and cam_dir go together");
    const int contract = camera ? camera->contract : 0;    // (round 4: a flag of the sample fetch for this path too; the build's own definition)
    constexpr int C = 64;                                   // procedures.py:22 RENDER_COARSE_PNUM
    const int S_all = n_fine + C;                           // (n_fine + 1) fine + 64 coarse depths, the last one dropped
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    auto take = [&](size_t bytes) { float* p = reinterpret_cast<float*>(ws); ws += ref_ws_part(bytes); return p; };
    float* density = take((size_t)N * C * 4);
    float* z_fine = take((size_t)N * (n_fine + 1) * 4);
    float* z_coarse = take((size_t)N * C * 4);
    float* z_all = take((size_t)N * S_all * 4);
    float* rgbo = take((size_t)N * S_all * 16);
    float* normals = take((size_t)N * S_all * 12);
    hipStream_t st = S(stream);
    if (!rays) {                                            // row 1: procedures.py:43-51,64
        if (camera->H <= 0 || camera->W <= 0 || ray_offset < 0 || ray_offset + N > (int64_t)camera->H * camera->W)
            return fail(NERF_AMD_EINVAL, "ray range outside the camera image");
        float* gen = reinterpret_cast<float*>(ws);
        if (int e = sk_generate_rays(camera->pose, camera->H, camera->W, camera->fx, camera->fy, ray_offset, N, gen, st))
            return hip_status(e, "ray generation");
        rays = gen;
    }
    const float jitter = (far - near) / (float)n_fine;      // procedures.py:59
    nerf_amd_samples sc{};                                  // rows 2-4
    sc.mode = 1; sc.rays = rays; sc.S = C; sc.M = N * C; sc.z = nullptr; sc.z_base = z_base; sc.u = u_strat;
    sc.z_jitter = jitter; sc.z_stride = C;
    sc.contract = contract;
    sc.rng_seed = seed; sc.rng_ray_offset = ray0;          // (read only when u_strat == NULL)
    if (int e = launch_proposal_any(lflags, packed_prop, precision, sc, density, st)) return hip_status(e, "proposal MLP");
    // rows 5-7 (procedures.py:68-70), also returning the stratified depths the proposal pass used
    if (int e = sk_resample(density, nullptr, z_base, u_strat, jitter, rays + 3, 6, u_inv, N, C, n_fine + 1, 0, 0.01f, seed, ray0, z_fine,
                            nullptr, nullptr, z_coarse, st)) return hip_status(e, "resample");
    // row 8, Ref-NeRF branch (procedures.py:71-74): fine and coarse depths merged, the last one dropped
    if (int e = sk_merge_sorted(z_fine, z_coarse, N, n_fine + 1, C, z_all, st)) return hip_status(e, "depth merge");
    nerf_amd_samples sf{};                                  // row 13
    sf.mode = 1; sf.rays = rays; sf.S = S_all; sf.M = N * S_all; sf.z = z_all; sf.z_stride = S_all;
    sf.contract = contract;
    if (int e = mlp_launch_ref(packed_ref, precision, sf, rgbo, normal_img ? normals : nullptr, nullptr, ref_flags, st)) return hip_status(e, "Ref-NeRF MLP");
    const int flags = 1 | (white_bkg ? 2 : 0);              // row 10 with sigma -> softplus(sigma + 0.5) (procedures.py:73)
    if (int e = sk_composite(rgbo, z_all, S_all, rays + 3, 6, N, S_all, flags, NERF_AMD_ACT_SOFTPLUS, 0.5f, near, far,
                             normal_img ? normals : nullptr, cam_dir, rgb, nullptr, depth, normal_img, st)) return hip_status(e, "composite");
    return NERF_AMD_OK;
}

size_t nerf_amd_gemm_workspace_bytes(int64_t M, int64_t N, int64_t P) {
    if (M <= 0 || N <= 0 || P <= 0) return 0;
    return gk_gemm_workspace_bytes(M, N, P);
}

int nerf_amd_gemm(int precision, int64_t M, int64_t N, int64_t P, const float* A, int64_t a_si, int64_t a_sp, const float* B, int64_t b_sp, int64_t b_sj,
                  float* C, int64_t ldc, const float* bias, int act, const float* mask, int64_t ldm, void* workspace, void* stream) {
    if (precision != NERF_AMD_F32 && precision != NERF_AMD_BF16) return fail(NERF_AMD_EINVAL, "nerf_amd_gemm: precision must be NERF_AMD_F32 or NERF_AMD_BF16");
    if (M < 0 || N < 0 || P < 0 || N > 65535LL * 128) return fail(NERF_AMD_EINVAL, "nerf_amd_gemm: bad size");
    if (M == 0 || N == 0) return 0;
    if (!C || (P && (!A || !B))) return fail(NERF_AMD_EINVAL, "nerf_amd_gemm: NULL argument");
    if ((a_si != 1 && a_sp != 1) || (b_sp != 1 && b_sj != 1)) return fail(NERF_AMD_EINVAL, "nerf_amd_gemm: one stride of each operand must be 1");
    if (ldc < N || (mask && ldm < N)) return fail(NERF_AMD_EINVAL, "nerf_amd_gemm: row stride smaller than N");
    if (act < 0 || act > 2) return fail(NERF_AMD_EINVAL, "nerf_amd_gemm: act must be 0 (none), 1 (ReLU) or 2 (sigmoid)");
    if (gk_gemm_workspace_bytes(M, N, P) && !workspace) return fail(NERF_AMD_EINVAL, "nerf_amd_gemm: this shape needs nerf_amd_gemm_workspace_bytes of workspace");
    return hip_status(gk_gemm(precision == NERF_AMD_BF16, M, N, P, A, a_si, a_sp, B, b_sp, b_sj, C, ldc, bias, act, mask, ldm, workspace, S(stream)), "nerf_amd_gemm");
}

int nerf_amd_sigmoid_backward(const float* g, int64_t g_stride, const float* y, int64_t y_stride, int64_t M, int cols, float* out, int64_t out_stride,
                              void* stream) {
    if (M < 0 || cols < 0) return fail(NERF_AMD_EINVAL, "nerf_amd_sigmoid_backward: bad size");
    if (M * cols && (!g || !y || !out)) return fail(NERF_AMD_EINVAL, "nerf_amd_sigmoid_backward: NULL argument");
    if (g_stride < cols || y_stride < cols || out_stride < cols) return fail(NERF_AMD_EINVAL, "nerf_amd_sigmoid_backward: row stride smaller than cols");
    return hip_status(gk_sigmoid_backward(g, g_stride, y, y_stride, M, cols, out, out_stride, S(stream)), "nerf_amd_sigmoid_backward");
}

// ---- generic-shape Ref-NeRF: the element-wise stages between its layer products (generic_ref_kernels.hip) ----
int nerf_amd_ref_dir_inputs(const float* heads, int64_t heads_stride, const float* dirs, int64_t dirs_stride, int64_t M, int ide_level, const float* ide_table,
                            float* out, int64_t out_stride, float* normal, void* stream) {
    if (M < 0 || ide_level < 1 || ide_level > 5) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_dir_inputs: bad size / ide_level (1..5)");
    if (M && (!heads || !dirs || !ide_table || !out || !normal)) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_dir_inputs: NULL argument");
    const int T = (1 << ide_level) - 1 + ide_level;
    if (heads_stride < 11 || dirs_stride < 3 || out_stride < 2 * T + 1) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_dir_inputs: row stride too small");
    return hip_status(gr_dir_inputs(heads, heads_stride, dirs, dirs_stride, M, ide_level, ide_table, out, out_stride, normal, S(stream)), "nerf_amd_ref_dir_inputs");
}

int nerf_amd_ref_dir_inputs_backward(const float* heads, int64_t heads_stride, const float* dirs, int64_t dirs_stride, int64_t M, int ide_level,
                                     const float* ide_table, const float* d_out, int64_t d_out_stride, const float* g_normal, int64_t g_normal_stride,
                                     float* d_heads, int64_t d_heads_stride, void* stream) {
    if (M < 0 || ide_level < 1 || ide_level > 5) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_dir_inputs_backward: bad size / ide_level (1..5)");
    if (M && (!heads || !dirs || !ide_table || !d_out || !g_normal || !d_heads)) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_dir_inputs_backward: NULL argument");
    const int T = (1 << ide_level) - 1 + ide_level;
    if (heads_stride < 11 || dirs_stride < 3 || d_out_stride < 2 * T + 1 || g_normal_stride < 3 || d_heads_stride < 11)
        return fail(NERF_AMD_EINVAL, "nerf_amd_ref_dir_inputs_backward: row stride too small");
    return hip_status(gr_dir_inputs_backward(heads, heads_stride, dirs, dirs_stride, M, ide_level, ide_table, d_out, d_out_stride, g_normal, g_normal_stride, d_heads,
                                             d_heads_stride, S(stream)), "nerf_amd_ref_dir_inputs_backward");
}

int nerf_amd_ref_combine(const float* heads, int64_t heads_stride, const float* spec, int64_t spec_stride, int64_t M, int ref_flags, float* rgbo, void* stream) {
    if (M < 0 || bad_ref_flags(ref_flags)) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_combine: bad size / ref_flags");
    if (M && (!heads || !spec || !rgbo)) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_combine: NULL argument");
    if (heads_stride < 11 || spec_stride < 3) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_combine: row stride too small");
    return hip_status(gr_combine(heads, heads_stride, spec, spec_stride, M, (ref_flags & NERF_AMD_REF_SRGB) ? 1 : 0, rgbo, S(stream)), "nerf_amd_ref_combine");
}

int nerf_amd_ref_combine_backward(const float* g_rgbo, int64_t g_stride, const float* heads, int64_t heads_stride, const float* spec, int64_t spec_stride, int64_t M,
                                  int ref_flags, float* d_spec, int64_t d_spec_stride, float* d_heads, int64_t d_heads_stride, void* stream) {
    if (M < 0 || bad_ref_flags(ref_flags)) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_combine_backward: bad size / ref_flags");
    if (M && (!g_rgbo || !heads || !spec || !d_spec || !d_heads)) return fail(NERF_AMD_EINVAL, "nerf_amd_ref_combine_backward: NULL argument");
    if (g_stride < 4 || heads_stride < 11 || spec_stride < 3 || d_spec_stride < 3 || d_heads_stride < 11)
        return fail(NERF_AMD_EINVAL, "nerf_amd_ref_combine_backward: row stride too small");
    return hip_status(gr_combine_backward(g_rgbo, g_stride, heads, heads_stride, spec, spec_stride, M, (ref_flags & NERF_AMD_REF_SRGB) ? 1 : 0, d_spec, d_spec_stride,
                                          d_heads, d_heads_stride, S(stream)), "nerf_amd_ref_combine_backward");
}

int nerf_amd_positional_encoding_backward(const float* d_enc, int64_t d_enc_stride, const float* x, int64_t x_stride, int64_t M, int L, int cat_origin, float* d_x,
                                          void* stream) {
    if (M < 0 || L < 0) return fail(NERF_AMD_EINVAL, "nerf_amd_positional_encoding_backward: bad size");
    if (M && (!d_enc || !x || !d_x)) return fail(NERF_AMD_EINVAL, "nerf_amd_positional_encoding_backward: NULL argument");
    if (d_enc_stride < 6 * L + (cat_origin ? 3 : 0) || x_stride < 3) return fail(NERF_AMD_EINVAL, "nerf_amd_positional_encoding_backward: row stride too small");
    return hip_status(gr_pe_backward(d_enc, d_enc_stride, x, x_stride, M, L, cat_origin ? 1 : 0, d_x, S(stream)), "nerf_amd_positional_encoding_backward");
}

int nerf_amd_contract_positions(const float* x, int64_t x_stride, int64_t M, const float* g, int64_t g_stride, float* out, void* stream) {
    if (M < 0 || x_stride < 3 || (g && g_stride < 3)) return fail(NERF_AMD_EINVAL, "nerf_amd_contract_positions: bad size or stride");
    if (M && (!x || !out)) return fail(NERF_AMD_EINVAL, "nerf_amd_contract_positions: NULL argument");
    return hip_status(gr_contract(x, x_stride, M, g, g_stride, out, S(stream)), "nerf_amd_contract_positions");
}

int nerf_amd_add_rows(float* dst, int64_t dst_stride, const float* src, int64_t src_stride, int64_t M, int cols, void* stream) {
    if (M < 0 || cols < 0) return fail(NERF_AMD_EINVAL, "nerf_amd_add_rows: bad size");
    if (M * cols && (!dst || !src)) return fail(NERF_AMD_EINVAL, "nerf_amd_add_rows: NULL argument");
    if (dst_stride < cols || src_stride < cols) return fail(NERF_AMD_EINVAL, "nerf_amd_add_rows: row stride smaller than cols");
    return hip_status(gr_add_rows(dst, dst_stride, src, src_stride, M, cols, S(stream)), "nerf_amd_add_rows");
}

int nerf_amd_rows_gemm(int64_t M, int64_t N, int64_t K, const void* X, int64_t ldx, const void* W, int64_t ldw, int64_t n_pad, const float* bias, int act, void* C,
                       int64_t ldc, int out_bf16, void* stream) {
    if (M < 0 || N < 0 || K < 1 || N > 65535LL * 256) return fail(NERF_AMD_EINVAL, "nerf_amd_rows_gemm: bad size");
    if (M == 0 || N == 0) return 0;
    if (!X || !W || !bias || !C) return fail(NERF_AMD_EINVAL, "nerf_amd_rows_gemm: NULL argument");
    if (act < 0 || act > 2) return fail(NERF_AMD_EINVAL, "nerf_amd_rows_gemm: act must be 0 (none), 1 (ReLU) or 2 (sigmoid)");
    if ((ldx & 7) || ldx < (K + 7) / 8 * 8 || (reinterpret_cast<uintptr_t>(X) & 15u))
        return fail(NERF_AMD_EINVAL, "nerf_amd_rows_gemm: X must be 16-byte aligned bf16 rows with a stride that is a multiple of 8 elements >= roundup(K, 8)");
    if ((ldw & 63) || ldw < K || (n_pad & 255) || n_pad < N || (reinterpret_cast<uintptr_t>(W) & 15u) || (reinterpret_cast<uintptr_t>(bias) & 15u))
        return fail(NERF_AMD_EINVAL, "nerf_amd_rows_gemm: W must be packed (n_pad % 256 == 0 rows >= N, ldw % 64 == 0 >= K, 16-byte aligned), bias 16-byte aligned");
    if (ldc < N) return fail(NERF_AMD_EINVAL, "nerf_amd_rows_gemm: row stride of C smaller than N");
    if (out_bf16 && ((N & 3) || (ldc & 3) || (reinterpret_cast<uintptr_t>(C) & 7u)))
        return fail(NERF_AMD_EINVAL, "nerf_amd_rows_gemm: bf16 output needs N % 4 == 0, ldc % 4 == 0 and an 8-byte aligned C");
    return hip_status(rg_rows_gemm(M, N, K, X, ldx, W, ldw, n_pad, bias, act, C, ldc, out_bf16, S(stream)), "nerf_amd_rows_gemm");
}

int nerf_amd_rows_to_bf16(const float* src, int64_t rows_src, int64_t src_stride, int64_t rows, int cols, int fill, void* dst, int64_t dst_stride, void* stream) {
    if (rows < 0 || rows_src < 0 || rows_src > rows || cols < 0 || fill < cols) return fail(NERF_AMD_EINVAL, "nerf_amd_rows_to_bf16: bad size");
    if (rows * fill == 0) return 0;
    if (!dst || (rows_src * cols && !src)) return fail(NERF_AMD_EINVAL, "nerf_amd_rows_to_bf16: NULL argument");
    if (src_stride < cols || dst_stride < fill) return fail(NERF_AMD_EINVAL, "nerf_amd_rows_to_bf16: row stride smaller than the columns written");
    return hip_status(rg_rows_to_bf16(src, rows_src, src_stride, rows, cols, fill, dst, dst_stride, S(stream)), "nerf_amd_rows_to_bf16");
}

}  // extern "C"
