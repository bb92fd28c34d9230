// Weight packing: nn.Linear tensors ((out,in) row-major fp32, reference state_dict order) -> the
// MFMA-fragment-ordered stream + bias table consumed by mlp_kernels.hip (layout: mlp_layout.h).
// Runs on the GPU (a few microseconds) so it can be repeated after every optimiser step.
#include "device_common.h"
#include "mlp_layout.h"

namespace {

enum { SEG_DMAP = 0, SEG_PE = 1, SEG_IDE = 2 };

struct PackLayer {
    const float* b;         // bias of rows [0, rows)
    int rows;               // real output rows (the rest of the 32*nfb rows are zero)
    // optional second row segment [rows, rows + rows2) taken from its own matrix/bias with the same K map
    const float* w2; const float* b2; int rows2; int stride2;
    int nkg, nfb;
    int frag_start, bias_off;
    // up to three K segments, each reading its own matrix: kind, #K groups, first column, PE levels, valid width
    const float* seg_w[3];
    int seg_stride[3];
    int seg_kind[3], seg_nkg[3], seg_col[3], seg_L[3], seg_width[3];
    // backward chains (TRANSPOSED weights): element (row i, slot feature f) = seg_w[(f - seg_first) * stride + seg_col + i] for
    // seg_first <= f < seg_first + seg_width -- row i of the fragment is an INPUT column of the reference matrix, the K slot an output row
    int trans;
    int seg_first[3];
};

// all layers of a network in ONE launch (blockIdx.y = layer): re-packing after every optimiser step is part of the training step,
// where the 5 + 10 separate launches were ~3 % of a 512-ray step
constexpr int PACK_MAX_LAYERS = 24;
struct PackBatch { PackLayer L[PACK_MAX_LAYERS]; };

template <bool BF16>
__global__ void pack_layer_kernel(PackBatch B, char* __restrict__ stream, float* __restrict__ bias) {
    const PackLayer& L = B.L[blockIdx.y];
    const int n_elem = L.nfb * L.nkg * 512;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_elem; i += gridDim.x * blockDim.x) {
        const int e = i & 7, lane = (i >> 3) & 63, frag = i >> 9;
        // consumption order of mlp_kernels.hip: feature blocks in pairs, a pair's fragments interleaved kg-major;
        // a trailing single block (odd nfb) in plain kg order
        const int grp = frag / (2 * L.nkg);
        int fb, kg;
        if (2 * grp + 1 < L.nfb) { const int r = frag - grp * 2 * L.nkg; kg = r >> 1; fb = 2 * grp + (r & 1); }
        else { fb = 2 * grp; kg = frag - 2 * grp * L.nkg; }
        const int row = 32 * fb + (lane & 31), h = lane >> 5;
        int seg = 0, lkg = kg;
        if (lkg >= L.seg_nkg[0]) { lkg -= L.seg_nkg[0]; seg = 1; if (lkg >= L.seg_nkg[1]) { lkg -= L.seg_nkg[1]; seg = 2; } }
        int col;
        if (L.seg_kind[seg] == SEG_PE) col = pe_slot_column(8 * lkg + e, h, L.seg_L[seg]);
        else if (L.seg_kind[seg] == SEG_IDE) col = ide_slot_column(8 * lkg + e, h);
        else col = dmap_feature(lkg, h, e);
        if (col >= L.seg_width[seg]) col = -1;
        float v = 0.0f;
        if (L.trans) {
            const int f = dmap_feature(lkg, h, e) - L.seg_first[seg];
            if (f >= 0 && f < L.seg_width[seg] && row < L.rows) v = L.seg_w[seg][(size_t)f * L.seg_stride[seg] + L.seg_col[seg] + row];
        } else if (col >= 0) {
            if (row < L.rows) v = L.seg_w[seg][(size_t)row * L.seg_stride[seg] + L.seg_col[seg] + col];
            else if (row < L.rows + L.rows2) v = L.w2[(size_t)(row - L.rows) * L.stride2 + L.seg_col[seg] + col];
        }
        const size_t f = (size_t)(L.frag_start + frag);
        if (BF16) reinterpret_cast<__bf16*>(stream + f * 1024)[lane * 8 + e] = (__bf16)v;
        else reinterpret_cast<float*>(stream + f * 2048)[(e >> 2) * 256 + lane * 4 + (e & 3)] = v;
    }
    const int n_b = (bias != nullptr) ? 32 * L.nfb : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_b; i += gridDim.x * blockDim.x) {
        bias[L.bias_off + i] = (i < L.rows) ? L.b[i] : ((i < L.rows + L.rows2) ? L.b2[i - L.rows] : 0.0f);
    }
}

PackLayer make_layer(const float* w, const float* b, int rows, int in_f, int nkg, int nfb, int start, int bias_off) {
    PackLayer L{};
    L.b = b; L.rows = rows; L.nkg = nkg; L.nfb = nfb; L.frag_start = start; L.bias_off = bias_off;
    L.w2 = nullptr; L.b2 = nullptr; L.rows2 = 0; L.stride2 = 0;
    L.seg_w[2] = w; L.seg_stride[2] = in_f; L.seg_kind[2] = SEG_DMAP; L.seg_nkg[2] = 0; L.seg_col[2] = 0; L.seg_L[2] = 0; L.seg_width[2] = 0;
    L.seg_w[0] = w; L.seg_stride[0] = in_f; L.seg_kind[0] = SEG_DMAP; L.seg_nkg[0] = nkg; L.seg_col[0] = 0; L.seg_L[0] = 0;
    L.seg_width[0] = 16 * nkg;
    L.seg_w[1] = w; L.seg_stride[1] = in_f; L.seg_kind[1] = SEG_DMAP; L.seg_nkg[1] = 0; L.seg_col[1] = 0; L.seg_L[1] = 0;
    L.seg_width[1] = 0;
    return L;
}
void set_seg(PackLayer& L, int s, int kind, int nkg, int col, int pe_L, int width) {
    L.seg_kind[s] = kind; L.seg_nkg[s] = nkg; L.seg_col[s] = col; L.seg_L[s] = pe_L; L.seg_width[s] = width;
}

// fold bottle_neck into rgb_layer.0:  Wf (128,256) = W8[:, :256] @ Wb,  bf (128) = W8[:, :256] @ bb + b8   (fp32)
__global__ void fold_bottleneck_kernel(const float* __restrict__ w8, const float* __restrict__ b8, const float* __restrict__ wb,
                                       const float* __restrict__ bb, float* __restrict__ wf, float* __restrict__ bf) {
    const int i = blockIdx.x;                        // output row 0..127
    const int j = threadIdx.x;                       // output column 0..255
    float acc = 0.0f;
    for (int k = 0; k < 256; ++k) acc = __builtin_fmaf(w8[i * 283 + k], wb[k * 256 + j], acc);
    wf[i * 256 + j] = acc;
    if (j == 0) {
        float s = 0.0f;
        for (int k = 0; k < 256; ++k) s = __builtin_fmaf(w8[i * 283 + k], bb[k], s);
        bf[i] = s + b8[i];
    }
}

int launch_pack(const PackBatch& B, int n_layers, int precision, char* stream, float* bias, hipStream_t st) {
    int blocks = 1;
    for (int l = 0; l < n_layers; ++l) {
        const int n = B.L[l].nfb * B.L[l].nkg * 512;
        blocks = ((n + 255) / 256 > blocks) ? (n + 255) / 256 : blocks;
    }
    if (precision == NERF_AMD_BF16) hipLaunchKernelGGL(pack_layer_kernel<true>, dim3(blocks, n_layers), dim3(256), 0, st, B, stream, bias);
    else hipLaunchKernelGGL(pack_layer_kernel<false>, dim3(blocks, n_layers), dim3(256), 0, st, B, stream, bias);
    return (int)hipGetLastError();
}

}  // namespace

int pack_proposal(int precision, const float* const* w, const float* const* b, void* packed, hipStream_t st) {
    using Lay = PropLayout;
    char* stream = reinterpret_cast<char*>(packed);
    float* bias = reinterpret_cast<float*>(stream + Lay::stream_bytes(precision));
    const int rows[5] = {256, 256, 256, 256, 1};
    const int inf[5] = {63, 256, 256, 256, 256};
    PackBatch B = {};
    for (int l = 0; l < 5; ++l) {
        PackLayer& L = B.L[l] = make_layer(w[l], b[l], rows[l], inf[l], Lay::NKG[l], Lay::NFB[l], Lay::START[l], Lay::BIAS_OFF[l]);
        if (l == 0) set_seg(L, 0, SEG_PE, 4, 0, 10, 63);
    }
    return launch_pack(B, 5, precision, stream, bias, st);
}

// ProposalNetwork(10, 128): w = layers.{0,2,4,6,8}.weight in their own (128-wide) shapes; the stream's last 8 fragments are padding
// (never multiplied: the kernel fetches and drops them), zeroed once here so that the blob is deterministic
int pack_proposal128(int precision, const float* const* w, const float* const* b, void* packed, hipStream_t st) {
    using Lay = PropLayout128;
    char* stream = reinterpret_cast<char*>(packed);
    float* bias = reinterpret_cast<float*>(stream + Lay::stream_bytes(precision));
    const size_t frag = (precision == NERF_AMD_BF16) ? 1024 : 2048;
    if (int e = (int)hipMemsetAsync(stream + Lay::USED_FRAGS * frag, 0, (Lay::N_FRAGS - Lay::USED_FRAGS) * frag, st)) return e;
    const int rows[5] = {128, 128, 128, 128, 1};
    const int inf[5] = {63, 128, 128, 128, 128};
    PackBatch B = {};
    for (int l = 0; l < 5; ++l) {
        PackLayer& L = B.L[l] = make_layer(w[l], b[l], rows[l], inf[l], Lay::NKG[l], Lay::NFB[l], Lay::START[l], Lay::BIAS_OFF[l]);
        if (l == 0) set_seg(L, 0, SEG_PE, 4, 0, 10, 63);
    }
    return launch_pack(B, 5, precision, stream, bias, st);
}

int pack_mip(int precision, const float* const* w, const float* const* b, void* packed, hipStream_t st) {
    using Lay = MipLayout;
    char* stream = reinterpret_cast<char*>(packed);
    float* bias = reinterpret_cast<float*>(stream + Lay::stream_bytes(precision));
    float* wf = bias + Lay::N_BIAS;                  // (128,256) folded weight, then (128) folded bias
    float* bf = wf + 128 * 256;
    // tensors: 0..3 lin_block1.{0,2,4,6}; 4..6 lin_block2.{0,2,4}; 7 bottle_neck.0; 8 opacity_head.0; 9,10 rgb_layer.{0,2}
    hipLaunchKernelGGL(fold_bottleneck_kernel, dim3(128), dim3(256), 0, st, w[9], b[9], w[7], b[7], wf, bf);
    if (int e = (int)hipGetLastError()) return e;
    const float* lw[10] = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[8], wf, w[10]};
    const float* lb[10] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[8], bf, b[10]};
    const int rows[10] = {256, 256, 256, 256, 256, 256, 256, 1, 128, 3};
    const int inf[10] = {63, 256, 256, 256, 319, 256, 256, 256, 256, 128};
    PackBatch B = {};
    for (int l = 0; l < 10; ++l) {
        PackLayer& L = B.L[l] = make_layer(lw[l], lb[l], rows[l], inf[l], Lay::NKG[l], Lay::NFB[l], Lay::START[l], Lay::BIAS_OFF[l]);
        if (l == 0) set_seg(L, 0, SEG_PE, 4, 0, 10, 63);
        if (l == 4) { set_seg(L, 0, SEG_PE, 4, 0, 10, 63); set_seg(L, 1, SEG_DMAP, 16, 63, 0, 256); }
        if (l == 8) {                                 // K = [folded 256 | direction encoding 27 from rgb_layer.0[:, 256:]]
            set_seg(L, 0, SEG_DMAP, 16, 0, 0, 256);
            set_seg(L, 1, SEG_PE, 2, 256, 4, 27);
            L.seg_w[1] = w[9]; L.seg_stride[1] = 283;
        }
    }
    return launch_pack(B, 10, precision, stream, bias, st);
}

// MipNeRF(10, 4, 128): the 11 tensors in their own shapes (lin_block1 128 wide, lin_block2.0 (128, 191), lin_block2.4 (256, 128), heads as at 256)
int pack_mip128(int precision, const float* const* w, const float* const* b, void* packed, hipStream_t st) {
    using Lay = MipLayout128;
    char* stream = reinterpret_cast<char*>(packed);
    float* bias = reinterpret_cast<float*>(stream + Lay::stream_bytes(precision));
    float* wf = bias + Lay::N_BIAS;
    float* bf = wf + 128 * 256;
    hipLaunchKernelGGL(fold_bottleneck_kernel, dim3(128), dim3(256), 0, st, w[9], b[9], w[7], b[7], wf, bf);
    if (int e = (int)hipGetLastError()) return e;
    const float* lw[10] = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[8], wf, w[10]};
    const float* lb[10] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[8], bf, b[10]};
    const int rows[10] = {128, 128, 128, 128, 128, 128, 256, 1, 128, 3};
    const int inf[10] = {63, 128, 128, 128, 191, 128, 128, 256, 256, 128};
    PackBatch B = {};
    for (int l = 0; l < 10; ++l) {
        PackLayer& L = B.L[l] = make_layer(lw[l], lb[l], rows[l], inf[l], Lay::NKG[l], Lay::NFB[l], Lay::START[l], Lay::BIAS_OFF[l]);
        if (l == 0) set_seg(L, 0, SEG_PE, 4, 0, 10, 63);
        if (l == 4) { set_seg(L, 0, SEG_PE, 4, 0, 10, 63); set_seg(L, 1, SEG_DMAP, 8, 63, 0, 128); }
        if (l == 8) {
            set_seg(L, 0, SEG_DMAP, 16, 0, 0, 256);
            set_seg(L, 1, SEG_PE, 2, 256, 4, 27);
            L.seg_w[1] = w[9]; L.seg_stride[1] = 283;
        }
    }
    return launch_pack(B, 10, precision, stream, bias, st);
}

// tensors: 0-3 spa_block1.{0,2,4,6}; 4-7 spa_block2.{0,2,4,6}; 8 bottle_neck; 9 heads (11,256); 10-13 dir_block1.{0,2,4,6};
//          14-17 dir_block2.{0,2,4,6}; 18 spec_rgb_head.0; 19 ide_table (9,19) in the weights slot
int pack_ref(int precision, const float* const* w, const float* const* b, void* packed, hipStream_t st) {
    using Lay = RefLayout;
    char* stream = reinterpret_cast<char*>(packed);
    float* bias = reinterpret_cast<float*>(stream + Lay::stream_bytes(precision));
    const int tensor_of[18] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16, 17, 18};
    const int rows[18] = {256, 256, 256, 256, 256, 256, 256, 256, 128, 256, 256, 256, 256, 256, 256, 256, 256, 3};
    const int inf[18] = {63, 256, 256, 256, 319, 256, 256, 256, 256, 167, 256, 256, 256, 423, 256, 256, 256, 256};
    PackBatch B = {};
    for (int l = 0; l < 18; ++l) {
        const int t = tensor_of[l];
        PackLayer& L = B.L[l] = make_layer(w[t], b[t], rows[l], inf[l], Lay::NKG[l], Lay::NFB[l], Lay::START[l], Lay::BIAS_OFF[l]);
        if (l == 0) set_seg(L, 0, SEG_PE, 4, 0, 10, 63);
        if (l == 4) { set_seg(L, 0, SEG_PE, 4, 0, 10, 63); set_seg(L, 1, SEG_DMAP, 16, 63, 0, 256); }
        if (l == 8) { L.w2 = w[9]; L.b2 = b[9]; L.rows2 = 11; L.stride2 = 256; }              // bottle_neck rows + the 11 head rows
        if (l == 9) { set_seg(L, 0, SEG_DMAP, 8, 0, 0, 128); set_seg(L, 1, SEG_IDE, 3, 128, 0, 39); }
        if (l == 13) { set_seg(L, 0, SEG_DMAP, 8, 0, 0, 128); set_seg(L, 1, SEG_IDE, 3, 128, 0, 39); set_seg(L, 2, SEG_DMAP, 16, 167, 0, 256); }
    }
    if (int e = launch_pack(B, 18, precision, stream, bias, st)) return e;
    if (int e = (int)hipMemcpyAsync(bias + Lay::N_BIAS, w[19], 9 * 19 * sizeof(float), hipMemcpyDeviceToDevice, st)) return e;
    return 0;
}

// ------------------------------------------------------------------------------------------------ backward chains (mlp_layout.h *BwdLayout)
namespace {
// W_fold (128,256) = rgb_layer.0[:, :256] @ bottle_neck.0 (the same product as the forward fold, without the bias)
__global__ void fold_weight_kernel(const float* __restrict__ w9, const float* __restrict__ wb, float* __restrict__ wf) {
    const int i = blockIdx.x, j = threadIdx.x;
    float acc = 0.0f;
    for (int k = 0; k < 256; ++k) acc = __builtin_fmaf(w9[i * 283 + k], wb[k * 256 + j], acc);
    wf[i * 256 + j] = acc;
}
// one transposed layer: K = `nkg` groups of delta features (rows of w), fragment rows = `rows` input columns of w starting at `col0`
PackLayer make_trans_layer(const float* w, int stride, int col0, int rows, int k_width, int nkg, int nfb, int start) {
    PackLayer L{};
    L.b = nullptr; L.rows = rows; L.nkg = nkg; L.nfb = nfb; L.frag_start = start; L.bias_off = 0; L.trans = 1;
    for (int s = 0; s < 3; ++s) { L.seg_w[s] = w; L.seg_stride[s] = stride; L.seg_kind[s] = SEG_DMAP; L.seg_nkg[s] = 0; L.seg_col[s] = col0; L.seg_first[s] = 0; L.seg_width[s] = 0; }
    L.seg_nkg[0] = nkg; L.seg_width[0] = k_width;
    return L;
}
}  // namespace

// proposal: w = layers.{0,2,4,6,8}.weight
int pack_proposal_bwd(int precision, const float* const* w, void* packed, hipStream_t st) {
    using Lay = PropBwdLayout;
    PackBatch B = {};
    B.L[0] = make_trans_layer(w[4], 256, 0, 256, 1, Lay::NKG[0], Lay::NFB[0], Lay::START[0]);       // d3 = layers.8^T g (+ one zero K group)
    for (int l = 1; l < 4; ++l) B.L[l] = make_trans_layer(w[4 - l], 256, 0, 256, 256, Lay::NKG[l], Lay::NFB[l], Lay::START[l]);
    B.L[4] = make_trans_layer(w[0], 63, 0, 63, 256, 16, 2, Lay::ENC_START);                        // d enc = layers.0^T delta_0 (density-gradient chain)
    return launch_pack(B, 5, precision, reinterpret_cast<char*>(packed), nullptr, st);
}

// MipNeRF: w in _linear_layers() order (0..3 lin_block1, 4..6 lin_block2, 7 bottle_neck.0, 8 opacity_head.0, 9, 10 rgb_layer.{0,2})
int pack_mip_bwd(int precision, const float* const* w, void* packed, hipStream_t st) {
    using Lay = MipBwdLayout;
    char* stream = reinterpret_cast<char*>(packed);
    float* wf = reinterpret_cast<float*>(stream + Lay::stream_bytes(precision));
    hipLaunchKernelGGL(fold_weight_kernel, dim3(128), dim3(256), 0, st, w[9], w[7], wf);
    if (int e = (int)hipGetLastError()) return e;
    PackBatch B = {};
    B.L[0] = make_trans_layer(w[10], 128, 0, 128, 3, Lay::NKG[0], Lay::NFB[0], Lay::START[0]);       // dc = rgb_layer.2^T dpre (slot features 0..2)
    PackLayer& L1 = B.L[1] = make_trans_layer(wf, 256, 0, 256, 128, Lay::NKG[1], Lay::NFB[1], Lay::START[1]);   // d6 = W_fold^T dc ...
    L1.seg_nkg[0] = 8;
    L1.seg_w[1] = w[8]; L1.seg_stride[1] = 256; L1.seg_nkg[1] = 1; L1.seg_first[1] = 3; L1.seg_width[1] = 1; // ... + opacity_head^T dsigma (slot feature 3)
    B.L[2] = make_trans_layer(w[6], 256, 0, 256, 256, 16, 8, Lay::START[2]);
    B.L[3] = make_trans_layer(w[5], 256, 0, 256, 256, 16, 8, Lay::START[3]);
    B.L[4] = make_trans_layer(w[4], 319, 63, 256, 256, 16, 8, Lay::START[4]);                        // skip layer: the hidden columns
    B.L[5] = make_trans_layer(w[3], 256, 0, 256, 256, 16, 8, Lay::START[5]);
    B.L[6] = make_trans_layer(w[2], 256, 0, 256, 256, 16, 8, Lay::START[6]);
    B.L[7] = make_trans_layer(w[1], 256, 0, 256, 256, 16, 8, Lay::START[7]);
    return launch_pack(B, 8, precision, stream, nullptr, st);
}

// Ref-NeRF: w = the 20 tensors of pack_ref (0-3 spa_block1, 4-7 spa_block2, 8 bottle_neck, 9 heads (11,256), 10-13 dir_block1,
// 14-17 dir_block2, 18 spec_rgb_head.0, 19 ide_table); layer table: mlp_layout.h RefBwdLayout
int pack_ref_bwd(int precision, const float* const* w, void* packed, hipStream_t st) {
    using Lay = RefBwdLayout;
    PackBatch B = {};
    auto full = [&](int l, const float* m) { B.L[l] = make_trans_layer(m, 256, 0, 256, 256, 16, 8, Lay::START[l]); };
    // DIR chain
    B.L[0] = make_trans_layer(w[18], 256, 0, 256, 3, Lay::NKG[0], 8, Lay::START[0]);                // spec head (slot features 0..2; K group 1: zero)
    full(1, w[17]); full(2, w[16]); full(3, w[15]);
    B.L[4] = make_trans_layer(w[14], 423, 167, 256, 256, 16, 8, Lay::START[4]);                     // dir_block2.0: hidden columns
    B.L[5] = make_trans_layer(w[14], 423, 0, 167, 256, 16, 6, Lay::START[5]);                       //               input-vector columns
    full(6, w[13]); full(7, w[12]); full(8, w[11]);
    B.L[9] = make_trans_layer(w[10], 167, 0, 167, 256, 16, 6, Lay::START[9]);                       // dir_block1.0
    // SPA chain
    PackLayer& H = B.L[10] = make_trans_layer(w[8], 256, 0, 256, 128, Lay::NKG[10], 8, Lay::START[10]);   // [bottle_neck (K 0..127) | heads (K group 8) | zero]
    H.seg_nkg[0] = 8;
    H.seg_w[1] = w[9]; H.seg_stride[1] = 256; H.seg_nkg[1] = 1; H.seg_first[1] = 0; H.seg_width[1] = 11;
    H.seg_w[2] = w[9]; H.seg_stride[2] = 256; H.seg_nkg[2] = 1; H.seg_first[2] = 0; H.seg_width[2] = 0;    // (a K group of width 0: all padding)
    full(11, w[7]); full(12, w[6]); full(13, w[5]);
    B.L[14] = make_trans_layer(w[4], 319, 63, 256, 256, 16, 8, Lay::START[14]);                     // spa_block2.0: hidden columns
    full(15, w[3]); full(16, w[2]); full(17, w[1]);
    if (int e = launch_pack(B, 18, precision, reinterpret_cast<char*>(packed), nullptr, st)) return e;
    // DEN chain (second copies of the spatial layers + the encoding columns)
    PackBatch C = {};
    auto fullc = [&](int l, const float* m) { C.L[l - 18] = make_trans_layer(m, 256, 0, 256, 256, 16, 8, Lay::START[l]); };
    C.L[0] = make_trans_layer(w[9] + 7 * 256, 256, 0, 256, 1, Lay::NKG[18], 8, Lay::START[18]);     // the density row of the heads alone
    fullc(19, w[7]); fullc(20, w[6]); fullc(21, w[5]);
    C.L[4] = make_trans_layer(w[4], 319, 63, 256, 256, 16, 8, Lay::START[22]);
    C.L[5] = make_trans_layer(w[4], 319, 0, 63, 256, 16, 2, Lay::START[23]);                        // spa_block2.0: encoding columns
    fullc(24, w[3]); fullc(25, w[2]); fullc(26, w[1]);
    C.L[9] = make_trans_layer(w[0], 63, 0, 63, 256, 16, 2, Lay::START[27]);                         // spa_block1.0
    return launch_pack(C, 10, precision, reinterpret_cast<char*>(packed), nullptr, st);
}

// ------------------------------------------------------------------------------------------------ measurement aid (bench.py roofline.mfma_stream_ref)
// Streams of nothing but v_mfma_f32_32x32x16_bf16 on four independent accumulators, one wave per SIMD (160 KiB of LDS per workgroup keeps
// everything else off the CU): what the matrix cores of THIS box sustain under its power limit -- the measured ceiling next to the
// datasheet peak (scripts/mfma_probe.hip is the stand-alone, more detailed form).  64 MFMAs per iteration and wave.
//   mode 0  constant operands (round 3's probe): the datapath barely toggles -- an OPTIMISTIC ceiling (a zero-weight run
//           of the fine kernel is 19 % faster than a real one at identical cycles)
//   mode 1  operands that toggle: a rotating pool of 8 A and 8 B register groups of pseudo-random bf16 (random sign, mantissa and a few
//           exponents: weights-like A, signed B) -- every MFMA sees other operands than its predecessor: the PESSIMISTIC ceiling
//   mode 2  data like the networks': A as in mode 1, B post-ReLU-like (half the elements zero, the rest positive)
//   mode 3  mode 2 + the weight ring's LDS cadence: every second MFMA's A operand arrives through a ds_read_b128 issued four fragments
//           ahead from 128 KiB of random fragments in LDS (each A fragment feeds two MFMAs, like the 64-sample tile of the MLP kernels)
namespace {
typedef __attribute__((ext_vector_type(16))) float pk_f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t pk_u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 pk_bf16x8;
__device__ __forceinline__ uint32_t stream_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two bf16 per dword: sign and mantissa random, exponent 2^-3 .. 2^0; relu_like: half the elements +0, the others positive
__device__ __forceinline__ uint32_t stream_bf16_pair(uint32_t key, bool relu_like) {
    const uint32_t r = stream_hash(key);
    uint32_t lo = (r & 0x807fu) | ((124u + ((r >> 8) & 3u)) << 7), hi = ((r >> 16) & 0x807fu) | ((124u + ((r >> 24) & 3u)) << 7);
    if (relu_like) {
        lo = (r & 0x8000u) ? 0u : (lo & 0x7fffu);
        hi = (r & 0x80000000u) ? 0u : (hi & 0x7fffu);
    }
    return lo | (hi << 16);
}
template <int MODE>
__global__ __launch_bounds__(256) void mfma_stream_kernel(int iters, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char stream_lds[];
    const int lane = threadIdx.x & 63;
    pk_f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    if constexpr (MODE == 0) {
        pk_u32x4 a = {0x3c003c00u + (uint32_t)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
            }
        }
    } else {
        pk_u32x4 a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[i][e] = stream_bf16_pair((uint32_t)(threadIdx.x * 64 + i * 4 + e) * 2654435761u + blockIdx.x, false);
                b[i][e] = stream_bf16_pair((uint32_t)(threadIdx.x * 64 + 32 + i * 4 + e) * 2246822519u + blockIdx.x, MODE >= 2);
            }
        if constexpr (MODE == 3) {
            // 128 fragments of 1 KiB (lane-linear 16-byte slots, conflict-free like the weight ring) of weights-like random data
            for (int i = threadIdx.x; i < 128 * 64; i += 256) {
                pk_u32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = stream_bf16_pair((uint32_t)(i * 4 + e) * 40503u + 977u * blockIdx.x, false);
                *reinterpret_cast<pk_u32x4*>(stream_lds + (size_t)i * 16) = v;
            }
            __syncthreads();
            const uint32_t base = lane * 16;
            pk_u32x4 q[4];
            uint32_t f = (threadIdx.x >> 6) * 8;                      // (the four waves read different fragments)
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const pk_u32x4*>(stream_lds + base + ((f + i) & 127u) * 1024);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 32; ++k) {                         // 32 fragments, two MFMAs (two column tiles) each
                    const pk_u32x4 af = q[k & 3];
                    q[k & 3] = *reinterpret_cast<const pk_u32x4*>(stream_lds + base + ((f + k + 4) & 127u) * 1024);
                    if (k & 1) {
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(af), "v"(b[(k >> 1) & 7]));
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc3) : "v"(af), "v"(b[((k >> 1) + 3) & 7]));
                    } else {
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(af), "v"(b[(k >> 1) & 7]));
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(af), "v"(b[((k >> 1) + 3) & 7]));
                    }
                }
                f = (f + 32) & 127u;
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a[k & 7]), "v"(b[(k + 1) & 7]));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a[(k + 2) & 7]), "v"(b[(k + 5) & 7]));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a[(k + 4) & 7]), "v"(b[(k + 3) & 7]));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc3) : "v"(a[(k + 6) & 7]), "v"(b[(k + 7) & 7]));
                }
            }
        }
    }
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += (acc0[i] + acc1[i]) + (acc2[i] + acc3[i]);
    if (r == 123.456f) sink[threadIdx.x] = r;
}
template <int MODE>
int launch_mfma_stream(int iters, int workgroups, float* sink, hipStream_t st) {
    const size_t lds = 160 * 1024;
    if (int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_stream_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) return e;
    hipLaunchKernelGGL(mfma_stream_kernel<MODE>, dim3(workgroups), dim3(256), lds, st, iters, sink);
    return (int)hipGetLastError();
}
}  // namespace
int pack_mfma_stream(int iters, int workgroups, int mode, float* sink, hipStream_t st) {
    switch (mode) {
        case 0: return launch_mfma_stream<0>(iters, workgroups, sink, st);
        case 1: return launch_mfma_stream<1>(iters, workgroups, sink, st);
        case 2: return launch_mfma_stream<2>(iters, workgroups, sink, st);
        default: return launch_mfma_stream<3>(iters, workgroups, sink, st);
    }
}
