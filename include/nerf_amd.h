/*
 * nerf_amd.h -- C-ABI of libnerf_amd.so: the MI355X (gfx950) native NeRF ray-march hot path.
 *
 * The reference (Enigmatisms/NeRF) has no FFI layer: its hot path is plain Python/torch functions
 * (SURVEY.md section 8b).  This header declares, one entry point per reference callee, what a
 * maintainer of the reference would bind from Python (ctypes stub: INTEGRATION.md) to replace the
 * torch expressions with the hand-written HIP kernels.  Citations are file:line under the
 * reference checkout.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 (or int64 where stated) unless
 *     marked "host"; no torch types cross this boundary;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     asynchronous on that stream, nothing synchronises the device;
 *   - every function returns 0 on success, a negative NERF_AMD_E* code otherwise;
 *     nerf_amd_last_error() returns a host string describing the last failure of the calling thread;
 *   - inputs are never written; outputs never alias inputs.
 */
#ifndef NERF_AMD_H
#define NERF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_AMD_OK            0
#define NERF_AMD_EINVAL       -1   /* bad argument (NULL pointer, unsupported size)            */
#define NERF_AMD_EUNSUPPORTED -2   /* a configuration the HIP kernels are not instantiated for */
#define NERF_AMD_EHIP         -3   /* a HIP runtime call failed (see nerf_amd_last_error)      */

/* arithmetic of the MLP matrix products */
#define NERF_AMD_F32   0   /* v_mfma_f32_32x32x2_f32: exact fp32 products + fp32 accumulate (parity mode, <=1e-4) */
#define NERF_AMD_BF16  1   /* v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate (throughput mode)          */
/* NERF_AMD_BF16 arithmetic with the TRAINING DUMPS of the 256-wide hidden layers (activations written by nerf_amd_*_forward_train, deltas
 * written by nerf_amd_*_backward_chain, both read by nerf_amd_*_weight_grads) stored as OCP e4m3 with one power-of-two scale per sample and
 * 16-feature K group: 288 B instead of 512 B per sample and layer in each of the four passes over a dump.  Accepted by exactly those entry
 * points and nerf_amd_train_dump_bytes / nerf_amd_weight_grads_workspace_bytes (pack with NERF_AMD_BF16); encodings and head deltas stay bf16. */
#define NERF_AMD_BF16_F8 2

/* which network a packed weight blob belongs to */
#define NERF_AMD_NET_PROPOSAL 0    /* ProposalNetwork(10, 256)        addtional.py:53-96  */
#define NERF_AMD_NET_MIP      1    /* MipNeRF(10, 4, 256)             mip_model.py:14-60  */
#define NERF_AMD_NET_REF      2    /* RefNeRF(10, 4, 128, 256, 256)   ref_model.py:16-66  */
/* ProposalNetwork(10, 128) -- the reference's class default (addtional.py:61) and `--prop_net_width 128` (procedures.py:176): its own packed
 * layout (five tensors in their 128-wide shapes) and a narrow-tile kernel at a quarter of the 256-wide MACs.  Forward / render only: a blob
 * packed for this network is passed as `packed_prop` together with NERF_AMD_PROP_W128 OR-ed into the call's `precision` argument
 * (nerf_amd_proposal_forward, nerf_amd_render_rays, nerf_amd_render_rays_ref).  Training a narrow network runs the 256-wide kernels on
 * zero-padded tensors (exactly the same function). */
#define NERF_AMD_NET_PROPOSAL_128 3
#define NERF_AMD_PROP_W128 0x100   /* layout flag in `precision`: packed_prop is a NERF_AMD_NET_PROPOSAL_128 blob */
/* MipNeRF(10, 4, 128) -- `--nerf_net_width 128` (procedures.py:177): the eleven tensors in their own shapes (lin_block1 128 wide, lin_block2.0
 * (128, 191), lin_block2.4 (256, 128), the heads as at width 256), a narrow-tile kernel at a third of the 256-wide MACs.  Forward / render with
 * the point PE only: NERF_AMD_FINE_W128 in `precision` of nerf_amd_mip_forward / nerf_amd_render_rays (not with the integrated PE). */
#define NERF_AMD_NET_MIP_128 4
#define NERF_AMD_FINE_W128 0x200   /* layout flag in `precision`: the fine-network blob is a NERF_AMD_NET_MIP_128 blob */

/* density activation applied inside sigma->alpha (nerf_base.py:82 `density_act`) */
#define NERF_AMD_ACT_RELU     0
#define NERF_AMD_ACT_IDENTITY 1    /* RefNeRF passes `lambda x: x` (ref_model.py:26)          */
#define NERF_AMD_ACT_SOFTPLUS 2

const char* nerf_amd_last_error(void);
int         nerf_amd_version(void);                       /* 100*major + minor */
int         nerf_amd_device_info(int* n_cu, int* arch_is_gfx950);

/* ------------------------------------------------------------------------------------------------
 * Where the samples of an MLP launch come from.  Three sources, so that positions never have to be
 * materialised in HBM on the render path:
 *   mode 0  points in memory        pts (M, pts_stride): xyz at [0:3], raw direction at [3:6]
 *                                   (MipNeRF.forward input, mip_model.py:41; ProposalNetwork.forward
 *                                   input, addtional.py:88)
 *   mode 1  rays + depths           x = o + z*d  (NeRF.length2pts nerf_base.py:53-56;
 *                                   procedures.py:66).  rays (N,6) = [o|d]; depth of sample s on ray n:
 *                                   z[n*z_stride + s]                         if z  != NULL
 *                                   z_base[s] + u[n*S + s] * z_jitter          otherwise (stratified
 *                                   draw of procedures.py:65 / utils.py:89 fused in); with u == NULL too
 *                                   the uniform is drawn in the kernel: Philox4x32-10 keyed by rng_seed,
 *                                   a pure function of (ray n + rng_ray_offset, s) -- see below
 *   mode 2  camera + depths         like mode 1 with o = pose[:,3] and d = R.((col-W/2+.5)/fx,
 *                                   (H/2-row+.5)/fy, -1) generated in-kernel for ray n = row*W + col
 *                                   (procedures.py:43-51)
 * M = total samples = N*S in modes 1/2.
 * ------------------------------------------------------------------------------------------------ */
typedef struct nerf_amd_samples {
    int32_t      mode;
    int32_t      S;            /* samples per ray (modes 1, 2)                 */
    int64_t      M;            /* total number of samples                      */
    const float* pts;          /* mode 0                                       */
    int32_t      pts_stride;   /* mode 0: floats per sample (3 or 6)           */
    int32_t      z_stride;     /* modes 1/2: floats between rays in z          */
    const float* rays;         /* mode 1: (N, 6)                               */
    const float* z;            /* modes 1/2 or NULL                            */
    const float* z_base;       /* (S) when z == NULL                           */
    const float* u;            /* (N, S) uniforms when z == NULL               */
    float        z_jitter;     /* scale of u when z == NULL                    */
    int32_t      H, W;         /* mode 2                                       */
    float        fx, fy;       /* mode 2: x is divided by fx, y by fy          */
    float        pose[12];     /* mode 2: row-major 3x4 camera-to-world (host) */
    int32_t      contract;     /* != 0: Mip-NeRF 360 scene contraction of the sample POSITION before it is encoded
                                  (Barron et al. 2022, eq. 10): x -> x if |x| <= 1 else (2 - 1/|x|) x/|x|.  Not in the
                                  reference (BASELINE config 5 only); occupies former tail padding, 0 = off.        */
    int32_t      ipe;          /* != 0 (modes 1 with z only, MipNeRF kernels): integrated positional encoding.  Sample s of ray n is the
                                  conical frustum between z[n*z_stride+s] and z[n*z_stride+s+1] (so z rows hold S+1 depths); the network
                                  input is [mu | ipe_feature(mu, diag Sigma)] (mip_methods.py:15-58) instead of [x | PE(x)].        */
    float        ipe_radius;   /* pixel radius r of coneParameters (mip_methods.py:15)                                              */
    const float* ipe_dir_norm; /* DEVICE pointer to one float: norm of the whole (N,3) direction tensor (mip_methods.py:31;
                                  nerf_amd_dirs_norm)                                                                              */
    uint64_t     rng_seed;     /* in-kernel uniforms (z == NULL and u == NULL; modes 1, 2): the stratified draw of sample s of ray n is    */
    int64_t      rng_ray_offset; /* word 0 of Philox4x32-10(key = rng_seed, counter = (n + rng_ray_offset [64 bit], s, 'RS' = 0x5253)),
                                  top 24 bits * 2^-24.  The reference draws these on the CPU generator (procedures.py:65, utils.py:89);
                                  the counter form needs no tensor, replays from the seed, and is independent of how rays are batched.
                                  nerf_amd_resample / nerf_amd_render_rays regenerate the same values from the same (seed, offset). */
} nerf_amd_samples;

/* ------------------------------------------------------------------------------------------------
 * Weights.  The kernels stream MFMA-fragment-ordered weights through LDS; packing turns the
 * reference's nn.Linear tensors ((out,in) row-major fp32, state_dict order below) into that stream.
 * Re-pack after every optimiser step (a few microseconds).
 *   proposal: layers.{0,2,4,6,8}                                   (addtional.py:67-71)
 *   mip     : lin_block1.{0,2,4,6}, lin_block2.{0,2,4}, bottle_neck.0, opacity_head.0,
 *             rgb_layer.{0,2}                                        (mip_model.py:19-37)
 *   ref     : spa_block1.{0,2,4,6}, spa_block2.{0,2,4,6}, bottle_neck, heads, dir_block1.{0,2,4,6},
 *             dir_block2.{0,2,4,6}, spec_rgb_head.0, ide_table                       (ref_model.py:31-62)
 *             where `heads` is the (11,256) row-concatenation [norm_col_tint_head[0:3], rho_tau_head[0:1],
 *             norm_col_tint_head[3:6], rho_tau_head[1:2], norm_col_tint_head[6:9]] (+ its (11) bias), and
 *             `ide_table` is the (9,19) fp32 coefficient matrix of ref_func.py:60-74 passed in the `weights`
 *             slot (its `biases` slot is ignored and may repeat any valid pointer).
 * `weights` / `biases` are HOST arrays of DEVICE pointers in that order.
 * ------------------------------------------------------------------------------------------------ */
size_t nerf_amd_packed_bytes(int net, int precision);
int    nerf_amd_pack_weights(int net, int precision, const float* const* weights, const float* const* biases,
                             int n_tensors, void* packed, void* stream);

/* ProposalNetwork.forward (addtional.py:88-96): density (M) fp32, no activation. */
int nerf_amd_proposal_forward(const void* packed, int precision, const nerf_amd_samples* src,
                              float* density, void* stream);
/* MipNeRF.forward (mip_model.py:41-60): rgbo (M, 4) = [sigmoid rgb | raw sigma]. */
int nerf_amd_mip_forward(const void* packed, int precision, const nerf_amd_samples* src,
                         float* rgbo, void* stream);

/* MipNeRF.forward + NeRF.render fused (mip_model.py:41-60 + nerf_base.py:91-113, mul_norm = True, relu density): the
 * (N,S,4) network output stays on chip; the last wavefront of each ray composites it.  `src` must be mode 1 (rays + z) with
 * S in {32, 64, 128}.  Outputs rgb (N,3), depth (N) or NULL, weights (N,S) or NULL.  256-wide blobs only: NERF_AMD_FINE_W128 in
 * `precision` is refused (EUNSUPPORTED) -- a NERF_AMD_NET_MIP_128 blob takes nerf_amd_mip_forward + nerf_amd_composite. */
int nerf_amd_mip_forward_composite(const void* packed, int precision, const nerf_amd_samples* src, int white_bkg,
                                   float near, float far, float* rgb, float* depth, float* weights, void* stream);

/* RefNeRF.forward in eval mode (ref_model.py:68-106): rgbo (M,4) = [rgb | raw density], normal (M,3) (NULL to skip).  Samples need a
 * direction (pts_stride >= 6 in mode 0).  ref_flags: NERF_AMD_REF_SRGB = the module's use_srgb (ref_model.py:100-102:
 * rgb = linear_to_srgb(specular + sigmoid(diffuse - log 3))), 0 = ref_model.py:104-105.  The same flag goes to every Ref-NeRF entry point. */
#define NERF_AMD_REF_SRGB 1
int nerf_amd_ref_forward(const void* packed, int precision, const nerf_amd_samples* src, int ref_flags,
                         float* rgbo, float* normal, void* stream);
/* Training-mode forward (ref_model.py:84-85): the same kernel with the bottle-neck perturbation `bn_noise` (M,128), which the
 * caller draws (torch.normal(0, perturb_bottle_neck_w)) so that the backward can re-evaluate with the same noise. */
int nerf_amd_ref_forward_train(const void* packed, int precision, const nerf_amd_samples* src, int ref_flags, const float* bn_noise,
                               float* rgbo, float* normal, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sampling / compositing kernels (one 64-lane wavefront per ray; HBM-bound).
 * ------------------------------------------------------------------------------------------------ */

/* positional_encoding (nerf_helper.py:38-48): x (M,3) -> out (M, 6L) = [sin 2^0 x, cos 2^0 x, sin 2^1 x, ...] */
int nerf_amd_positional_encoding(const float* x, int64_t M, int L, float* out, void* stream);

/* ipe_feature with its helpers coneParameters / coneMeanCov / multFreq (mip_methods.py:15-58): z (N, S+1) depths, rays (N,6),
 * L frequency levels, pixel radius r -> feat (N, S, 6L) = per level [sin(2^l mu) e^(-4^l diag/2) xyz | cos(...) xyz], mu (N,S,3) or NULL,
 * mu_t (N,S) or NULL.  `dir_norm` is a DEVICE pointer to the norm of the whole (N,3) direction tensor (the reference's `.norm()`
 * without a dim, :31) -- nerf_amd_dirs_norm computes it.  The reference never calls ipe_feature from its entry scripts
 * (SURVEY.md 8a row 12); golden G12 pins the function. */
int nerf_amd_ipe_feature(const float* z, const float* rays, int64_t N, int S, int L, float r, const float* dir_norm,
                         float* feat, float* mu, float* mu_t, void* stream);
/* (ABI 125) the same with the Mip-NeRF 360 scene contraction of the frustum MEAN before the lift (the diagonal covariance stays metric):
 * what the fused kernels' sample fetch computes for `ipe` + `contract` together (BASELINE configs[2] + configs[4]); `mu` receives the
 * CONTRACTED mean.  Not in the reference (mip_methods.py has no contraction): the build's own definition, oracle.ipe_feature(contracted=True).
 * The layer-by-layer route of networks larger than the compiled shapes encodes with it. */
int nerf_amd_ipe_feature_contracted(const float* z, const float* rays, int64_t N, int S, int L, float r, const float* dir_norm,
                                    float* feat, float* mu, float* mu_t, void* stream);
/* coneParameters (mip_methods.py:15-23): z (N, S+1) -> mu_t, sigma_t^2, sigma_r^2, each (N, S). */
int nerf_amd_cone_parameters(const float* z, int64_t N, int S, float r, float* mu_t, float* var_t, float* var_r, void* stream);
/* sqrt(sum of squares) of the direction halves of rays (N,6) -> out (1 float on the device); fp64 accumulation, fixed order. */
int nerf_amd_dirs_norm(const float* rays, int64_t N, float* out, void* stream);

/* Ray table of render_image (procedures.py:43-51,64): rays (N, 6) = [pose[:,3] | R.c] for pixels
 * n = row*W + col in [ray_offset, ray_offset+N); `pose_host` is a HOST row-major 3x4. */
int nerf_amd_generate_rays(const float* pose_host, int H, int W, float fx, float fy, int64_t ray_offset, int64_t N,
                           float* rays, void* stream);

/* NeRF.length2pts (nerf_base.py:53-56): out (N, S, 6) = [o + z d | d]. */
int nerf_amd_length2pts(const float* rays, const float* z, int64_t N, int S, float* out, void* stream);

/* ProposalNetwork.get_weights (addtional.py:100-107) / NeRF.getNormedWeight (nerf_base.py:80-86):
 * sigma (N,S), z (N,S), dirs (N,3) or NULL (NULL = z already scaled) -> w (N,S). */
int nerf_amd_sigma_to_weights(const float* sigma, const float* z, const float* dirs, int64_t N, int S, int act,
                              float* w, void* stream);

/* maxBlurFilter (mip_methods.py:61-66): w (N,S) -> out (N,S). */
int nerf_amd_max_blur(const float* w, int64_t N, int S, float alpha, float* out, void* stream);

/* inverseSample (utils.py:34-44) + sample_pdf (utils.py:108-133) with the uniforms explicit:
 * w (N,C), z (N,C), u (N,K) -> z_out (N,K) [sorted when sort!=0], below (N,K) int64 (NULL to skip). C <= 256. */
int nerf_amd_inverse_sample(const float* w, const float* z, const float* u, int64_t N, int C, int K, int sort,
                            float* z_out, int64_t* below, void* stream);

/* sample_pdf (utils.py:108-133): bins (N,B), weights (N,B-1), u (N,K) -> samples (N,K), below/above (N,K) int64 or NULL. */
int nerf_amd_sample_pdf(const float* bins, const float* weights, const float* u, int64_t N, int B, int K, float* samples,
                        int64_t* below, int64_t* above, void* stream);

/* Training twin of the ray table (validSampler, utils.py:78-85): integer pixel coords (N,2) int64 =
 * (col - W//2, H//2 - row) from randomFromOneImage -> rays (N,6). */
int nerf_amd_pixel_rays(const float* pose_host, float fx, float fy, const int64_t* coords, int64_t N, float* rays, void* stream);

/* validSampler (utils.py:72-94) with every random number drawn in the kernel: N rays through uniformly drawn pixels of
 * the flattened pixel table rgbs (n_pixels,3) / coords (n_pixels,2) int64 of randomFromOneImage, their ground-truth colours rgb (N,3),
 * rays (N,6) = [pose[:,3] | R.c], and -- when pts / lengths are given -- C stratified depths lengths (N,C) = linspace(near, far - res, C)
 * + u res, res = (far-near)/C, with the positions pts (N,C,3).  Uniforms: Philox4x32-10 keyed by rng_seed; pixel index of ray n =
 * floor(n_pixels * u64 / 2^64) of counter (n, 0, 'IX'), depth uniform = word s&3 of counter (n, s>>2, 'TS'). */
int nerf_amd_sample_training_rays(const float* rgbs, const int64_t* coords, int64_t n_pixels, const float* pose_host, float fx, float fy,
                                  float near, float far, int64_t N, int C, uint64_t rng_seed, float* pts, float* lengths, float* rgb,
                                  float* rays, void* stream);
/* The same with the camera pose (3,4) and the seed read from DEVICE memory at run time: nothing about the step is baked into the launch
 * arguments, so a training step captured once in a hipGraph sees a new image pose and fresh random numbers on every replay
 * (nerf_amd/training.py).  nerf_amd_advance_seed replaces *seed_dev by an unrelated key (one thread; put it at the end of the step);
 * nerf_amd_philox_uniforms fills u (N,K) with the inverse-CDF stream of nerf_amd_resample (key = rng_seed, or *seed_dev when given) for
 * callers that keep inverseSample(weights, depths, u) as a separate op (utils.py:34-44, 115). */
int nerf_amd_sample_training_rays_dev(const float* rgbs, const int64_t* coords, int64_t n_pixels, const float* pose_dev, float fx, float fy,
                                      float near, float far, int64_t N, int C, const uint64_t* seed_dev, float* pts, float* lengths,
                                      float* rgb, float* rays, void* stream);
int nerf_amd_philox_uniforms(float* out, int64_t N, int K, uint64_t rng_seed, const uint64_t* seed_dev, void* stream);
/* (ABI 121) Either Philox stream of the render kernels as a tensor, for rows that are GLOBAL rays ray_offset .. ray_offset + N - 1: the
 * uniforms the reference draws per tile with torch.rand (procedures.py:65 stratified jitter -> NERF_AMD_PHILOX_STRAT, K <= 64;
 * utils.py:115 inverse-CDF -> NERF_AMD_PHILOX_INV), bit-identical to what nerf_amd_render_rays draws in place for the same seed and ray
 * index -- so a network outside the fused kernels' shapes (the layer-by-layer route) renders with the same random numbers. */
#define NERF_AMD_PHILOX_INV 0
#define NERF_AMD_PHILOX_STRAT 1
int nerf_amd_philox_stream(float* out, int64_t N, int K, uint64_t rng_seed, const uint64_t* seed_dev, int64_t ray_offset, int stream_id,
                           void* stream);
int nerf_amd_advance_seed(uint64_t* seed_dev, void* stream);

/* Stratified depths and points (utils.py:87-90, procedures.py:65-66): z = z_base[s] + u*z_jitter (N,S);
 * pts (N,S,3) = o + d*z, or NULL to skip. */
int nerf_amd_stratified_points(const float* rays, const float* z_base, const float* u, float z_jitter, int64_t N, int S,
                               float* z_out, float* pts, void* stream);

/* Fused proposal resampling of the render/train loop (procedures.py:68-70, train.py:169-172):
 * density -> [softplus] -> get_weights(|d| scaling, relu) -> maxBlur(alpha) -> inverse sample (sorted).
 * Depths as in nerf_amd_samples (z, or z_base + u_strat*z_jitter).  dirs row n at dirs[n*dirs_stride .. +3]
 * (pass rays+3 with stride 6).  Optional outputs (NULL to skip): w_prop (N,C), below (N,K) int64, z_coarse (N,C).
 * A NULL u_strat (with z == NULL) / a NULL u_inv is drawn in the kernel from (rng_seed, ray + rng_ray_offset) exactly like
 * nerf_amd_samples describes; words 1..3 of the same blocks are the inverse-CDF draws: u_inv(n, k), k = 192 b + r (r < 192), is word
 * 1 + r/64 of Philox4x32-10(key = rng_seed, counter = (n + rng_ray_offset, 64 b + r%64, 'RS')) -- at the render shapes one block per
 * (ray, lane) carries everything that lane needs. */
int nerf_amd_resample(const float* density, const float* z, const float* z_base, const float* u_strat, float z_jitter,
                      const float* dirs, int dirs_stride, const float* u_inv, int64_t N, int C, int K,
                      int softplus_density, float blur_alpha, uint64_t rng_seed, int64_t rng_ray_offset,
                      float* z_fine, int64_t* below, float* w_prop, float* z_coarse, void* stream);

/* NeRF.render (nerf_base.py:91-113).  rgbo (N,S,4), z (N,z_stride) [first S used], dirs as above.
 * flags: bit0 mul_norm, bit1 white_bkg.  density = act(sigma + sigma_shift) (sigma_shift 0.5 + softplus is
 * the Ref-NeRF render path, procedures.py:74).  Outputs: rgb (N,3), weights (N,S) or NULL, depth (N) or NULL
 * ((sum w z - near)/(far-near)), normal_img (N) or NULL when normal (N,S,3) and cam_dir (3, device) given. */
int nerf_amd_composite(const float* rgbo, const float* z, int z_stride, const float* dirs, int dirs_stride,
                       int64_t N, int S, int flags, int act, float sigma_shift, float near, float far,
                       const float* normal, const float* cam_dir,
                       float* rgb, float* weights, float* depth, float* normal_img, void* stream);

/* The depth sort of NeRF.coarseFineMerge (nerf_base.py:59-73) for the render path (procedures.py:72): z_fine (N,K) and z_coarse
 * (N,C) -> z_out (N, K+C-1) = sort(cat(z_fine, z_coarse))[..., :-1].  Both inputs are normally ascending along the last dimension
 * (then this is a merge); a ray whose input is out of order -- stratified depths with a jitter above the bin spacing, n_fine < 63 --
 * is sorted first, so the result always equals the sort.  K + C <= 2048. */
int nerf_amd_merge_depths(const float* z_fine, const float* z_coarse, int64_t N, int K, int C, float* z_out, void* stream);

/* NeRF.coarseFineMerge as the TRAINING loop calls it (nerf_base.py:59-73 with f_inds; train.py:176): z_out as above, plus
 * order (N, K+C) int64 = the stable argsort of cat(z_fine, z_coarse) (fine index i before coarse index K + j among equal depths --
 * what torch.sort's device radix sort returns; the reference hands order[..., :-1] to RefNeRF.coarse_grad_select), and, when
 * f_inds (N,K) int64 is given, all_inds (N, K+C) int64 = gather(cat(f_inds, arange(C)), order) (the bin indices getBounds reads).
 * f_inds / all_inds may be NULL.  K + C <= 1024. */
int nerf_amd_merge_depths_order(const float* z_fine, const float* z_coarse, const int64_t* f_inds, int64_t N, int K, int C, float* z_out,
                                int64_t* order, int64_t* all_inds, void* stream);

/* RefNeRF.coarse_grad_select (ref_model.py:108-117): grads (N,T,D) fp32, sort_inds (N,T) int64 -> out (N,c_pnum,D): the rows of the
 * first c_pnum positions p (ascending) with sort_inds[n,p] >= T - c_pnum -- the reference's boolean mask
 * gather(cat(zeros(T-c), ones(c)), sort_inds) without its data-dependent output size; rows with fewer flagged positions continue with
 * the unflagged ones in order. */
int nerf_amd_coarse_grad_select(const float* grads, const int64_t* sort_inds, int64_t N, int T, int D, int c_pnum, float* out, void* stream);

/* Ref-NeRF's normal losses (ref_model.py:127-143) as one streaming pass + a fixed-order two-stage sum (deterministic):
 *   mode 0  WeightedNormalLoss(size_average=False): out[0] = scale * sum_i w_i (1 - <a_i, b_i>)      (a = d_norm, b = p_norm; scale 1, or 1/M for size_average)
 *   mode 1  BackFaceLoss:                            out[0] = scale * sum_i w_i relu(<a_i, b_i>)      (a = normal, b = ray_d; scale = 1 / M: torch.mean)
 * w (M), a, b (M,3) fp32 contiguous; workspace: NERF_AMD_DOT_LOSS_WORKSPACE_FLOATS floats.
 * Backward: g = d loss / d out (ONE float in device memory, so the call needs no synchronisation) -> d_w (M), d_a, d_b (M,3); any of the
 * three may be NULL. */
#define NERF_AMD_DOT_LOSS_WORKSPACE_FLOATS 512
int nerf_amd_weighted_dot_loss(const float* w, const float* a, const float* b, int64_t M, int mode, float scale, float* out, float* workspace, void* stream);
int nerf_amd_weighted_dot_loss_backward(const float* g, const float* w, const float* a, const float* b, int64_t M, int mode, float scale, float* d_w,
                                        float* d_a, float* d_b, void* stream);

/* Measurement aid, no reference counterpart (the roofline's denominator checked on the box): `workgroups` workgroups of four
 * waves (one per SIMD; 160 KiB of LDS each, so one workgroup per CU) issue iters * 64 v_mfma_f32_32x32x16_bf16 per wave and nothing else;
 * timed by the caller, 32 768 flop per MFMA and wave.  mode 0: constant operands (optimistic: the datapath does not toggle); 1: a rotating
 * pool of pseudo-random bf16 A / B register groups (pessimistic); 2: random weights-like A, post-ReLU-like B (half zeros); 3: mode 2 with
 * every A operand read from LDS four fragments ahead and feeding two MFMAs (the weight ring's cadence).  sink: 256 floats (never written
 * in practice). */
int nerf_amd_mfma_stream(int iters, int workgroups, int mode, float* sink, void* stream);

/* getBounds (addtional.py:14-18): w_prop (N,C), below (N,K) int64 -> bounds (N,K-1). */
int nerf_amd_get_bounds(const float* w_prop, const int64_t* below, int64_t N, int C, int K, float* bounds, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training forward of the MLPs: same kernels and results as nerf_amd_{proposal,mip}_forward, and in
 * addition every hidden layer's post-ReLU activations are written to `dump` (nerf_amd_train_dump_bytes bytes, device) in the
 * kernels' fragment order.  nerf_amd_train_dump_to_rows turns one layer of a dump into a row-major (M, n_features) matrix
 * (bf16 for NERF_AMD_BF16, fp32 for NERF_AMD_F32) for the dgrad / wgrad GEMMs.
 * Layers: proposal 0..3 = layers.{0,2,4,6} outputs (256 wide), 4 = the input encoding [x | PE10(x)] in the kernels' slot order;
 * MipNeRF 0..3 = lin_block1.{0,2,4,6}, 4..6 = lin_block2.{0,2,4} (256 wide), 7 = rgb_layer.0 output (128 wide), 8 = the position and
 * direction encodings in slot order (operands of the first-layer / skip-layer / colour-head weight gradients).
 * ------------------------------------------------------------------------------------------------ */
size_t nerf_amd_train_dump_bytes(int net, int precision, int64_t M);
int nerf_amd_proposal_forward_train(const void* packed, int precision, const nerf_amd_samples* src, float* density, void* dump,
                                    void* stream);
int nerf_amd_mip_forward_train(const void* packed, int precision, const nerf_amd_samples* src, float* rgbo, void* dump, void* stream);
int nerf_amd_train_dump_to_rows(const void* dump, int net, int precision, int64_t M, int layer, int n_features, void* out,
                                void* stream);
/* ReLU adjoint of the dgrad chain, in place: delta[i] = act[i] > 0 ? delta[i] : 0 over n elements (bf16 / fp32 as above). */
int nerf_amd_relu_mask(void* delta, const void* act, int precision, int64_t n, void* stream);
/* The same over a (rows, cols) matrix with the bias gradient fused in: `col_sum` receives nerf_amd_relu_mask_bias_partials(...) rows of
 * `cols` fp32 partial column sums of the masked delta (written, not accumulated: no atomics, reproducible); their sum over the rows is
 * the bias gradient.  cols * element size / 4 must divide 256. */
int64_t nerf_amd_relu_mask_bias_partials(int precision, int64_t rows, int cols);
int nerf_amd_relu_mask_bias(void* delta, const void* act, int precision, int64_t rows, int cols, float* col_sum, void* stream);

/* nerf_amd_train_dump_to_rows and nerf_amd_relu_mask_bias of ONE layer in one pass: act_out (M, n_features) = the layer's activations
 * row-major, delta (M, n_features, same element type) masked in place by [act > 0], and col_sum = nerf_amd_train_dump_rows_mask_partials()
 * rows of n_features fp32 partial column sums of the masked delta (all rows written; their sum is the bias gradient).
 * n_features = 128 or 256. */
int64_t nerf_amd_train_dump_rows_mask_partials(void);
int nerf_amd_train_dump_rows_mask(const void* dump, int net, int precision, int64_t M, int layer, int n_features, void* act_out, void* delta,
                                  float* col_sum, void* stream);

/* Input operand of the first-layer / skip-layer weight gradients: row m = [x | positional_encoding_L(x) (nerf_helper.py:38-48) | 0 ...]
 * with 3 + 6L columns rounded up to a multiple of 8, bf16 (NERF_AMD_BF16) or fp32 rows.  x (M, >=3) with row stride x_stride floats;
 * normalize != 0 divides x by its norm first (the view direction, mip_model.py:52).  L = 4 or 10. */
int nerf_amd_encode_rows(const float* x, int x_stride, int64_t M, int L, int normalize, int precision, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the two MLPs (what torch.autograd computes for addtional.py:88-96 / mip_model.py:41-60 inside train.py:164-199, and the
 * optimizer step of train.py:117-118,200-218) as hand-written kernels -- no library GEMM:
 *   1. nerf_amd_pack_weights_backward: the TRANSPOSED weights in the kernels' fragment order (re-pack after every optimizer step,
 *      like the forward blob; `weights` in the same order as nerf_amd_pack_weights);
 *   2. nerf_amd_{proposal,mip}_backward_chain: the dgrad chain.  Inputs: the gradient w.r.t. the network output (g_density (M) /
 *      g_rgbo (M,4) together with the forward output rgbo (M,4) for the sigmoid adjoint) and the activation dump of the training
 *      forward; output: the "delta dump" (same size and fragment order as the activation dump: nerf_amd_train_dump_bytes);
 *   3. nerf_amd_{proposal,mip}_weight_grads: delta^T . activations on the matrix cores, written as (out, in) row-major fp32 tensors
 *      in the networks' tensor order (d_weights / d_biases: HOST arrays of DEVICE pointers, every tensor fully overwritten);
 *      workspace: nerf_amd_weight_grads_workspace_bytes bytes of device scratch.  Deterministic (no atomics).
 *   4. nerf_amd_adam_step: torch.optim.Adam (no weight decay, no amsgrad) over a table of tensors; `step` is a DEVICE float holding
 *      the number of steps taken so far (incremented by the call, so that a captured graph replays correctly); grads are multiplied
 *      by grad_scale first (1 = plain; 1/world_size after a summing all-reduce).  lr / betas / eps are doubles like torch's Python
 *      scalars (the bias corrections 1 - beta^step are formed in double, as torch does); a non-NULL lr_dev (one double on the device)
 *      overrides lr when the kernel runs, so that a captured graph follows a learning-rate schedule.
 * Gradients w.r.t. the sample positions are not produced (the reference detaches them for these two networks).
 * ------------------------------------------------------------------------------------------------ */
size_t nerf_amd_packed_backward_bytes(int net, int precision);
int    nerf_amd_pack_weights_backward(int net, int precision, const float* const* weights, int n_tensors, void* packed_bwd, void* stream);
int    nerf_amd_proposal_backward_chain(const void* packed_bwd, int precision, const float* g_density, int64_t M, const void* act_dump,
                                        void* delta_dump, void* stream);
int    nerf_amd_mip_backward_chain(const void* packed_bwd, int precision, const float* g_rgbo, const float* rgbo, int64_t M,
                                   const void* act_dump, void* delta_dump, void* stream);
size_t nerf_amd_weight_grads_workspace_bytes(int net, int precision, int64_t M);
int    nerf_amd_proposal_weight_grads(int precision, int64_t M, const void* act_dump, const void* delta_dump, float* const* d_weights,
                                      float* const* d_biases, void* workspace, void* stream);
int    nerf_amd_mip_weight_grads(int precision, int64_t M, const void* act_dump, const void* delta_dump, const float* const* weights,
                                 const float* const* biases, float* const* d_weights, float* const* d_biases, void* workspace, void* stream);
/* Ref-NeRF training (ref_model.py:68-143, train.py:176-199).
 *   nerf_amd_ref_forward_train_dump: nerf_amd_ref_forward_train + the activation dump (nerf_amd_train_dump_bytes(NERF_AMD_NET_REF, ..))
 *     and aux (M,16) fp32 = the pre-activation head values the backward's element-wise stage needs.
 *   nerf_amd_density_grad: RefNeRF.get_grad's gradient (ref_model.py:119-125 before the normalisation) -- d density / d position of
 *     every sample, times scale[m * scale_stride] when `scale` is given -- for the proposal network (train.py:165-168) or Ref-NeRF's
 *     spatial network (train.py:178): a dgrad-only chain from the density row through the hidden layers (ReLU masks from the dump)
 *     to the encoded position, then the positional encoding's derivative.  x (M, x_stride >= 3) = the sample positions.
 *   nerf_amd_ref_backward: every parameter gradient.  g_out (M, g_stride >= 7) = the gradient w.r.t. [rgb 3 | raw density 1 |
 *     predicted normal 3]; dirs = the view directions the forward saw; ide_table = the (9,19) table given to nerf_amd_pack_weights.
 *     d_weights / d_biases: 20 tensors each -- 0..7 spa_block1.{0,2,4,6}, spa_block2.{0,2,4,6}; 8 bottle_neck; 9 norm_col_tint_head;
 *     10 rho_tau_head; 11..18 dir_block1.{0,2,4,6}, dir_block2.{0,2,4,6}; 19 spec_rgb_head.0 -- (out, in) row-major, fully overwritten.
 *   packed_bwd: nerf_amd_pack_weights_backward(NERF_AMD_NET_REF, ...) with the 20 tensors of nerf_amd_pack_weights. */
int    nerf_amd_ref_forward_train_dump(const void* packed, int precision, const nerf_amd_samples* src, int ref_flags, const float* bn_noise,
                                       float* rgbo, float* normal, void* dump, float* aux, void* stream);
/* (ABI 122) The same with the bottle-neck perturbation (ref_model.py:84-85, `spa_info_b + torch.normal(0, w, shape)`) drawn INSIDE the
 * kernel: N(0, noise_std) deviates from Philox4x32-10 + Box-Muller keyed by (noise_seed -- or *noise_seed_dev, read at run time so that a
 * captured hipGraph replays fresh noise --, sample index): no (M, 128) noise tensor is written and read back (1.6 GB and one launch per
 * 2^14-ray step).  nerf_amd_philox_normal materialises exactly those deviates as out (M, 128) for samples sample_offset .. + M - 1
 * (nerf_amd_ref_forward_train_dump with that tensor gives bit-identical outputs). */
int    nerf_amd_ref_forward_train_dump_rng(const void* packed, int precision, const nerf_amd_samples* src, int ref_flags, uint64_t noise_seed,
                                           const uint64_t* noise_seed_dev, float noise_std, float* rgbo, float* normal, void* dump, float* aux,
                                           void* stream);
int    nerf_amd_philox_normal(float* out, int64_t M, uint64_t rng_seed, const uint64_t* seed_dev, float std, int64_t sample_offset, void* stream);
/* NERF_AMD_CONTRACTED OR-ed into `net` of nerf_amd_density_grad (round 5): the training forward fetched its samples with
 * nerf_amd_samples.contract = 1 -- x are the UNcontracted positions; the encoding's derivative is taken at contract(x) and pulled back
 * through the contraction's Jacobian, so the result is still d density / d x. */
#define NERF_AMD_CONTRACTED 0x100
size_t nerf_amd_density_grad_workspace_bytes(int net, int precision, int64_t M);
int    nerf_amd_density_grad(int net, const void* packed_bwd, int precision, int64_t M, const void* act_dump, const float* x, int x_stride,
                             const float* scale, int scale_stride, float* grad, void* workspace, void* stream);
size_t nerf_amd_ref_backward_workspace_bytes(int precision, int64_t M);
int    nerf_amd_ref_backward(const void* packed_bwd, int precision, int ref_flags, int64_t M, const void* act_dump, const float* aux,
                             const float* dirs, int dir_stride, const float* g_out, int g_stride, const float* ide_table,
                             float* const* d_weights, float* const* d_biases, void* workspace, void* stream);
int    nerf_amd_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                          const int64_t* numel, int n_tensors, float* step, double lr, const double* lr_dev, double beta1, double beta2,
                          double eps, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the sampling / compositing rows (what torch.autograd computes for train.py:169-199).
 * One wavefront per ray; S <= 1024 (4, 8 or 16 register chunks of 64 samples, picked by S).  Depths and directions are not differentiated (the reference detaches them too).
 * ------------------------------------------------------------------------------------------------ */
/* d(get_weights)/d(density) (addtional.py:100-107, nerf_base.py:80-86): d_weights (N,S) -> d_sigma (N,S). */
int nerf_amd_sigma_to_weights_backward(const float* sigma, const float* z, const float* dirs, int64_t N, int S, int act,
                                       const float* d_weights, float* d_sigma, void* stream);
/* d(NeRF.render)/d(rgbo) (nerf_base.py:91-113): d_rgb (N,3), optional d_weights (N,S) and d_depth (N) -> d_rgbo (N,S,4);
 * same flags / act / sigma_shift / near / far as nerf_amd_composite. */
int nerf_amd_composite_backward(const float* rgbo, const float* z, int z_stride, const float* dirs, int dirs_stride, int64_t N, int S,
                                int flags, int act, float sigma_shift, float near, float far, const float* d_rgb,
                                const float* d_weights, const float* d_depth, float* d_rgbo, void* stream);
/* d(maxBlurFilter)/d(weights) (mip_methods.py:61-66); ties of torch.maximum split the gradient in halves. */
int nerf_amd_max_blur_backward(const float* weights, const float* d_out, int64_t N, int S, float* d_weights, void* stream);
/* d(getBounds)/d(w_prop) (addtional.py:14-18): d_bounds (N,K-1) -> d_w_prop (N,C). */
int nerf_amd_get_bounds_backward(const int64_t* below, const float* d_bounds, int64_t N, int C, int K, float* d_w_prop, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The whole tile body of render_image (procedures.py:64-85, non-Ref) for N rays in four launches:
 *   proposal MLP (stratified z fused) -> resample -> fine MLP (length2pts fused) -> composite.
 * rays: (N,6) device, or NULL to generate them in-kernel from `camera` (mode 2 of nerf_amd_samples;
 * only H, W, fx, fy, pose are read; ray n = row*W + col, n in [ray_offset, ray_offset+N)).
 * z_base (64) = linspace(near, far, 64) (procedures.py:52; an input so that it is bit-identical to the
 * caller's torch.linspace), u_strat (N,64), u_inv (N, n_fine+1) -- or both NULL: every uniform is then drawn inside the kernels
 * (Philox4x32-10, camera->rng_seed / camera->rng_ray_offset; `camera` may accompany explicit rays just to carry them; n_fine <= 255).
 * Outputs rgb (N,3), depth (N) or NULL, weights (N,n_fine) or NULL.
 * workspace: nerf_amd_render_workspace_bytes(N, n_fine) bytes of device scratch.
 * ------------------------------------------------------------------------------------------------ */
/* (With camera != NULL and camera->ipe != 0 -- the descriptor may accompany explicit rays just to carry flags -- the FINE pass uses
 * the integrated positional encoding of the frusta between consecutive fine depths, radius camera->ipe_radius; `rays` must be given.
 * The direction norm of mip_methods.py:31 is taken over the N rays of this call, or -- camera->ipe_dir_norm != NULL -- read from that
 * device scalar: a caller rendering a SHARD of a ray list passes the norm of the whole list (nerf_amd_dirs_norm), so that the shards of an
 * image encode exactly like the single call over all of its rays.) */
size_t nerf_amd_render_workspace_bytes(int64_t N, int n_fine);
int    nerf_amd_render_rays(const void* packed_prop, const void* packed_mip, int precision,
                            const float* rays, const nerf_amd_samples* camera, int64_t ray_offset,
                            const float* z_base, const float* u_strat, const float* u_inv, int64_t N, int n_fine,
                            float near, float far, int white_bkg,
                            float* rgb, float* depth, float* weights, void* workspace, void* stream);

/* The same tile body for a Ref-NeRF fine network (procedures.py:64-85 with the is_ref_model branch, lines 71-74): proposal pass,
 * resampling, the n_fine+1 fine and 64 coarse depths merged (the last one dropped), Ref-NeRF MLP on the n_fine+64 samples,
 * compositing with sigma -> softplus(sigma + 0.5).  normal_img (N) and cam_dir (3 floats on the device: render_pose[:, -2]) are
 * both given or both NULL (procedures.py:79-81).  Six launches.  Workspace: nerf_amd_render_ref_workspace_bytes(N, n_fine). */
size_t nerf_amd_render_ref_workspace_bytes(int64_t N, int n_fine);
int    nerf_amd_render_rays_ref(const void* packed_prop, const void* packed_ref, int precision, int ref_flags, const float* rays,
                                const nerf_amd_samples* camera, int64_t ray_offset, const float* z_base, const float* u_strat,
                                const float* u_inv, int64_t N, int n_fine, float near, float far, int white_bkg, const float* cam_dir,
                                float* rgb, float* depth, float* normal_img, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Generic-shape layer product (ABI 119): the path of networks LARGER than the shapes the fused MLP kernels are compiled for -- hidden
 * widths above 256 (`--nerf_net_width` / `--prop_net_width`, procedures.py:176-177), more than 10 encoding octaves (mip_model.py:15-18,
 * addtional.py:61).  Such a network is evaluated layer by layer (nerf_helper.py:28-36 makeMLP = nn.Linear + activation) with fp32
 * row-major activations in HBM, forward and backward, by ONE hand-written MFMA GEMM with explicit element strides:
 *
 *     C[i, j] = act( sum_{p < P} A[i*a_si + p*a_sp] * B[p*b_sp + j*b_sj] + bias[j] ) * [mask[i*ldm + j] > 0]        i < M, j < N
 *
 *   forward  y = act(x W^T + b): A = x, B(p, j) = W[j, p];   input gradient  dx = (dy W) . [x > 0]: A = dy, B = W, mask = x;
 *   weight gradient  dW = dy^T x: A(i, p) = dy[p, i], B = x (db: B = a column of ones) -- a long contraction over the samples is split
 *   over workgroups, partial sums in `workspace` (nerf_amd_gemm_workspace_bytes bytes; may be NULL when that is 0), added in a fixed
 *   order: deterministic, no atomics.
 * Of each operand's two strides one must be 1.  C row-major, row stride ldc.  bias (N) or NULL; act 0 none / 1 ReLU / 2 sigmoid; mask or NULL.
 * precision: NERF_AMD_F32 = v_mfma_f32_32x32x2_f32 on the fp32 operands; NERF_AMD_BF16 = operands rounded to bf16 (RNE), fp32 accumulation.
 * All matrices fp32 on the device.  Every compiled shape keeps its fused kernel; this is the compatibility path of the shape arguments. */
size_t nerf_amd_gemm_workspace_bytes(int64_t M, int64_t N, int64_t P);
int    nerf_amd_gemm(int precision, int64_t M, int64_t N, int64_t P, const float* A, int64_t a_si, int64_t a_sp,
                     const float* B, int64_t b_sp, int64_t b_sj, float* C, int64_t ldc, const float* bias, int act,
                     const float* mask, int64_t ldm, void* workspace, void* stream);
/* out[m, c] = g[m, c] * y[m, c] * (1 - y[m, c]), c < cols: the adjoint of y = sigmoid(.) (rgb_layer.2, mip_model.py:35); row strides in floats */
int    nerf_amd_sigmoid_backward(const float* g, int64_t g_stride, const float* y, int64_t y_stride, int64_t M, int cols,
                                 float* out, int64_t out_stride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Generic-shape Ref-NeRF (ABI 120): a RefNeRF the fused kernel is not compiled for -- hidden width above 256, more than 10 position
 * octaves, `--ide_level 5` (procedures.py:211; 36 spherical-harmonic terms) -- runs layer by layer on nerf_amd_gemm; these are the
 * element-wise stages between its layer products (ref_model.py:78-105), forward and adjoint.  One row per sample, fp32, explicit row
 * strides in floats (column ranges of a concatenated buffer are views).  `heads` (M, >= 11) = the raw outputs of
 * cat(norm_col_tint_head, rho_tau_head):  [normal 0-2 | diffuse 3-5 | tint 6-8 | roughness 9 | density 10].
 *
 *   nerf_amd_ref_dir_inputs            ref_model.py:80-92: roughness = softplus(rho - 1), n = -n / (|n| + 1e-7), w_r = d - 2 (d.n) n, the
 *                                      integrated directional encoding of ref_func.py:76-108 at `ide_level` 1..5 (T = 2^level - 1 + level
 *                                      terms; `ide_table` = its (2^(level-1) + 1, T) coefficient matrix, ref_func.py:60-74) and n.d
 *                                      -> out (M, >= 2T+1) = [IDE real T | IDE imag T | n.d], normal (M,3) contiguous
 *   nerf_amd_ref_dir_inputs_backward   d_out (M, >= 2T+1), g_normal (M, >= 3) -> d_heads columns 0-2 (normal) and 9 (roughness)
 *   nerf_amd_ref_combine               ref_model.py:98-105: rgb = spec * sigmoid(tint) + sigmoid(diffuse) (NERF_AMD_REF_SRGB: linear_to_srgb(
 *                                      spec * sigmoid(tint) + sigmoid(diffuse - log 3))), `spec` (M, >= 3) = sigmoid(spec_rgb_head);
 *                                      rgbo (M,4) contiguous = [rgb | raw density]
 *   nerf_amd_ref_combine_backward      g_rgbo (M, >= 4) -> d_spec (M, >= 3) w.r.t. spec_rgb_head's PRE-activation, d_heads columns 3-8, 10
 *   nerf_amd_positional_encoding_backward   d_x (M,3) = (d [x | sin 2^f x | cos 2^f x] / d x)^T d_enc, f < L (nerf_helper.py:38-48 layout behind
 *                                      the raw position when cat_origin): RefNeRF.get_grad's last step (ref_model.py:119-125)
 *   nerf_amd_add_rows                  dst[m, c] += src[m, c], c < cols
 */
int nerf_amd_ref_dir_inputs(const float* heads, int64_t heads_stride, const float* dirs, int64_t dirs_stride, int64_t M, int ide_level,
                            const float* ide_table, float* out, int64_t out_stride, float* normal, void* stream);
int nerf_amd_ref_dir_inputs_backward(const float* heads, int64_t heads_stride, const float* dirs, int64_t dirs_stride, int64_t M, int ide_level,
                                     const float* ide_table, const float* d_out, int64_t d_out_stride, const float* g_normal,
                                     int64_t g_normal_stride, float* d_heads, int64_t d_heads_stride, void* stream);
int nerf_amd_ref_combine(const float* heads, int64_t heads_stride, const float* spec, int64_t spec_stride, int64_t M, int ref_flags,
                         float* rgbo, void* stream);
int nerf_amd_ref_combine_backward(const float* g_rgbo, int64_t g_stride, const float* heads, int64_t heads_stride, const float* spec,
                                  int64_t spec_stride, int64_t M, int ref_flags, float* d_spec, int64_t d_spec_stride, float* d_heads,
                                  int64_t d_heads_stride, void* stream);
int nerf_amd_positional_encoding_backward(const float* d_enc, int64_t d_enc_stride, const float* x, int64_t x_stride, int64_t M, int L,
                                          int cat_origin, float* d_x, void* stream);
int nerf_amd_add_rows(float* dst, int64_t dst_stride, const float* src, int64_t src_stride, int64_t M, int cols, void* stream);
/* (ABI 123) Mip-NeRF 360 scene contraction as a stage of the layer-by-layer route (the fused kernels apply it in their sample fetch:
 * nerf_amd_samples.contract): g == NULL -> out (M,3) = contract(x); g (M, g_stride >= 3) = a gradient w.r.t. contract(x) -> out (M,3) = its
 * pull-back through the contraction's Jacobian (what RefNeRF.get_grad needs, ref_model.py:119-125).  Not in the reference (BASELINE
 * configs[4]); the definition is oracle.contract. */
int nerf_amd_contract_positions(const float* x, int64_t x_stride, int64_t M, const float* g, int64_t g_stride, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layer products on bf16 rows (ABI 124): the INFERENCE route of the networks above under bf16 precision (forward only; nerf_amd_gemm
 * stays the fp32 parity mode and the backward).  One nn.Linear (+ activation) of mip_model.py:41-60 / addtional.py:88-96 /
 * ref_model.py:68-106:
 *
 *     C[m, n] = act( sum_k X[m, k] * W[n, k] + bias[n] ),   m < M, n < N, k < K
 *
 *   X     bf16 rows, row stride ldx ELEMENTS (a multiple of 8), 16-byte aligned; elements K .. roundup(K, 8) - 1 of a row must be finite
 *         (they meet packed zero weights): the padding nerf_amd_rows_to_bf16 writes, or the neighbouring columns of a wider buffer
 *   W     the layer's weight (N, K) as bf16 rows, zero-padded to (n_pad, ldw): n_pad a multiple of 256 >= N, ldw a multiple of 64 >= K
 *         (nerf_amd_rows_to_bf16 with rows = n_pad, fill = ldw packs an fp32 nn.Linear.weight)
 *   bias  n_pad floats (zeros beyond N), 16-byte aligned
 *   C     out_bf16 != 0: bf16 rows for the next layer (N % 4 == 0, ldc % 4 == 0, 8-byte aligned); else fp32 rows (any N / ldc): the
 *         heads, and the inputs of the element-wise stages
 *   act   0 none / 1 ReLU / 2 sigmoid
 * 256 x 256 outputs per workgroup, both operands global -> LDS by DMA through a 4-slot ring, v_mfma_f32_32x32x16_bf16, fp32 accumulation:
 * the values nerf_amd_gemm(NERF_AMD_BF16) computes (it rounds the same operands on their way into LDS), at 3x its rate. */
int nerf_amd_rows_gemm(int64_t M, int64_t N, int64_t K, const void* X, int64_t ldx, const void* W, int64_t ldw, int64_t n_pad,
                       const float* bias, int act, void* C, int64_t ldc, int out_bf16, void* stream);
/* dst[m, j] = bf16(src[m, j]) (RNE) for m < rows_src, j < cols; 0 for the rest of rows x fill: fp32 rows (row stride src_stride floats)
 * into a column range of bf16 rows (dst = first element of the range, row stride dst_stride elements) with the zero padding above */
int nerf_amd_rows_to_bf16(const float* src, int64_t rows_src, int64_t src_stride, int64_t rows, int cols, int fill, void* dst,
                          int64_t dst_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_AMD_H */
