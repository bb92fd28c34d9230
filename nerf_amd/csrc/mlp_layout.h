// Packed-weight stream layout shared by the MLP kernels and the pack kernels.
//
// A packed blob = [fragment stream][bias table].
//   fragment (layer, fb, kg) holds A-operand registers of one 32x32 MFMA tile: rows = output features
//   32*fb .. 32*fb+31, K = the 16 input "slots" of K group kg.  Element (lane, e) with i = lane&31,
//   h = lane>>5, e in 0..7 is W[32*fb + i][ column(kg, h, e) ]:
//       bf16: [lane][e]             1 KiB per fragment  (one ds_read_b128 per lane)
//       fp32: [e>>2][lane][e&3]     2 KiB per fragment  (two conflict-free ds_read_b128 per lane)
//   Fragments are stored in exactly the order the kernels consume them: layer by layer; inside a layer the
//   feature blocks go in pairs (2g, 2g+1) with the pair's fragments interleaved kg-major:
//   (2g,kg0) (2g+1,kg0) (2g,kg1) (2g+1,kg1) ...; an odd trailing block follows in plain kg order.
//
// Slot -> feature maps (why the activations never leave registers):
//   D map   a layer's MFMA output registers (lane half h, reg r) hold feature (r&3) + 8*(r>>2) + 4*h of
//           each 32-row block; read back as B operand of the next layer they are K slot
//           (kg, h, e) = feature 16*kg + 8*(e>>2) + 4*h + (e&3).
//   PE map  encoded inputs are generated in-register; slot q = 8*kg + e:
//           q < 3L      -> frequency q/3, component q%3; half 0 holds sin, half 1 holds cos
//           q == 3L     -> half 0: x, half 1: z      q == 3L+1 -> half 0: y, half 1: (zero pad)
//           reference column order (nerf_helper.py:38-48 + cat_origin): [x y z | sin f0 xyz | cos f0 xyz | ...]
#pragma once
#include "../../include/nerf_amd.h"

#define MLP_CHUNK_BYTES 8192
#ifndef MLP_NSLOT
#define MLP_NSLOT 8          /* ring slots of the proposal / MipNeRF kernels */
#endif
#define MLP_NSLOT_REF 6      /* Ref-NeRF needs the LDS for its 11 KiB/wave activation stash */
#define MLP_RING_BYTES (MLP_CHUNK_BYTES * MLP_NSLOT)
#define MLP_NW_BF16 8
#define MLP_NW_F32 4

#if defined(__HIPCC__)
#define LAYOUT_HD __host__ __device__
#else
#define LAYOUT_HD
#endif

// column of the reference weight matrix that PE slot (q, h) multiplies, or -1 for zero padding
LAYOUT_HD inline int pe_slot_column(int q, int h, int L) {
    if (q < 3 * L) return 3 + 6 * (q / 3) + 3 * h + (q % 3);
    if (q == 3 * L) return h ? 2 : 0;
    if (q == 3 * L + 1) return h ? -1 : 1;
    return -1;
}
LAYOUT_HD inline int dmap_feature(int kg, int h, int e) { return 16 * kg + 8 * (e >> 2) + 4 * h + (e & 3); }

struct PropLayout {
    static constexpr int N_LAYERS = 5;
    static constexpr int NKG[5] = {4, 16, 16, 16, 16};
    static constexpr int NFB[5] = {8, 8, 8, 8, 1};
    static constexpr int START[5] = {0, 32, 160, 288, 416};
    static constexpr int BIAS_OFF[5] = {0, 256, 512, 768, 1024};
    static constexpr int N_FRAGS = 432;
    static constexpr int N_BIAS = 1056;
    LAYOUT_HD static constexpr size_t stream_bytes(int prec) { return (size_t)N_FRAGS * (prec == NERF_AMD_BF16 ? 1024 : 2048); }
    LAYOUT_HD static constexpr size_t packed_bytes(int prec) { return stream_bytes(prec) + N_BIAS * 4; }
};

// ProposalNetwork(10, hidden <= 128) -- the reference's CLASS DEFAULT is 128 (addtional.py:61) and `--prop_net_width` selects it
// (procedures.py:176): half the K groups and half the feature blocks per hidden layer (a quarter of the MACs), evaluated by a tile policy
// with FOUR 32-sample column tiles per wave (every A fragment read from LDS feeds four MFMAs).  The stream is padded by one chunk's worth
// of fragments (120 -> 128) that the kernel fetches and drops: every cyclic stream holds an even number of chunks (see PropBwdLayout).
struct PropLayout128 {
    static constexpr int N_LAYERS = 5;
    static constexpr int HK = 8, HFB = 4;                // hidden K groups / feature blocks
    static constexpr int NKG[5] = {4, 8, 8, 8, 8};
    static constexpr int NFB[5] = {4, 4, 4, 4, 1};
    static constexpr int START[5] = {0, 16, 48, 80, 112};
    static constexpr int BIAS_OFF[5] = {0, 128, 256, 384, 512};
    static constexpr int USED_FRAGS = 120, N_FRAGS = 128;
    static constexpr int N_BIAS = 544;
    LAYOUT_HD static constexpr size_t stream_bytes(int prec) { return (size_t)N_FRAGS * (prec == NERF_AMD_BF16 ? 1024 : 2048); }
    LAYOUT_HD static constexpr size_t packed_bytes(int prec) { return stream_bytes(prec) + N_BIAS * 4; }
};

// MipNeRF.  The bottle_neck layer has no activation (mip_model.py:31,58), so it is folded into rgb_layer.0 at pack time:
//   relu(W8a.(Wb g + bb) + W8b.r + b8) = relu((W8a Wb) g + W8b r + (W8a bb + b8))     -- exact algebra, 12 % fewer MFMAs.
struct MipLayout {
    static constexpr int N_LAYERS = 10;
    //                             l1.0 l1.2 l1.4 l1.6 l2.0 l2.2 l2.4 sigma rgb0' rgb2
    static constexpr int NKG[10] = {4, 16, 16, 16, 20, 16, 16, 16, 18, 8};
    static constexpr int NFB[10] = {8, 8, 8, 8, 8, 8, 8, 1, 4, 1};
    static constexpr int START[10] = {0, 32, 160, 288, 416, 576, 704, 832, 848, 920};
    static constexpr int BIAS_OFF[10] = {0, 256, 512, 768, 1024, 1280, 1536, 1792, 1824, 1952};
    static constexpr int N_FRAGS = 928;
    static constexpr int N_BIAS = 1984;
    static constexpr size_t FOLD_SCRATCH = (128 * 256 + 128) * 4;    // folded weight + bias, kept behind the bias table
    LAYOUT_HD static constexpr size_t stream_bytes(int prec) { return (size_t)N_FRAGS * (prec == NERF_AMD_BF16 ? 1024 : 2048); }
    LAYOUT_HD static constexpr size_t packed_bytes(int prec) { return stream_bytes(prec) + N_BIAS * 4 + FOLD_SCRATCH; }
};

// MipNeRF(10, 4, hidden <= 128) (`--nerf_net_width 128`, procedures.py:177): lin_block1 63 -> 128 x 4, lin_block2 (128 + 63) -> 128 -> 128 -> 256
// (its last layer is 256 wide whatever the hidden width, mip_model.py:28), heads as in MipLayout (bottle_neck folded into rgb_layer.0).
// 233 216 MAC per sample (the reference's layers, bottle_neck counted) against 527 872 at width 256; executed: 352 fragments against 928.
struct MipLayout128 {
    static constexpr int N_LAYERS = 10;
    //                             l1.0 l1.2 l1.4 l1.6 l2.0 l2.2 l2.4 sigma rgb0' rgb2
    static constexpr int NKG[10] = {4, 8, 8, 8, 12, 8, 8, 16, 18, 8};
    static constexpr int NFB[10] = {4, 4, 4, 4, 4, 4, 8, 1, 4, 1};
    static constexpr int START[10] = {0, 16, 48, 80, 112, 160, 192, 256, 272, 344};
    static constexpr int BIAS_OFF[10] = {0, 128, 256, 384, 512, 640, 768, 1024, 1056, 1184};
    static constexpr int N_FRAGS = 352;
    static constexpr int N_BIAS = 1216;
    static constexpr size_t FOLD_SCRATCH = (128 * 256 + 128) * 4;
    LAYOUT_HD static constexpr size_t stream_bytes(int prec) { return (size_t)N_FRAGS * (prec == NERF_AMD_BF16 ? 1024 : 2048); }
    LAYOUT_HD static constexpr size_t packed_bytes(int prec) { return stream_bytes(prec) + N_BIAS * 4 + FOLD_SCRATCH; }
};

// ------------------------------------------------------------------------------------------------
// Backward (dgrad) chains: the same stream format, carrying TRANSPOSED weights.  Layer "dL" turns delta_{L+1} (gradient w.r.t. the
// pre-activation of forward layer L+1, K operand) into delta_L = (W_{L+1}^T delta_{L+1}) * [y_L > 0]: rows = the inputs of W_{L+1}.
//   proposal: head (g_density in slot feature 0) -> d3 -> d2 -> d1 -> d0                       (layers.8, .6, .4, .2 transposed)
//   MipNeRF : head K group = [d(rgb pre-sigmoid) 0..2 | d(sigma) 3]
//             dc (128)  = W_rgb2^T head[0..2]                       (NKG 2: the second K group is zero padding to keep whole chunks)
//             d6 (256)  = [W_fold^T | W_sigma^T] [dc | head]         (W_fold = rgb_layer.0[:, :256] . bottle_neck.0, as in the forward fold)
//             d5, d4    = lin_block2.4^T, lin_block2.2^T;  d3 = lin_block2.0[:, 63:]^T (skip layer: the hidden part);  d2, d1, d0 = lin_block1.6/.4/.2^T
// There are no biases (the chain is linear); the kernels point every layer at one zero block.
// ------------------------------------------------------------------------------------------------
struct PropBwdLayout {
    static constexpr int N_LAYERS = 4;
    static constexpr int NKG[4] = {2, 16, 16, 16};       // (head: its second K group is zero padding -- every cyclic stream holds an EVEN
    static constexpr int NFB[4] = {8, 8, 8, 8};          //  number of chunks, so that the barrier pattern of WeightStream is the same in every tile)
    static constexpr int START[4] = {0, 16, 144, 272};
    static constexpr int CHAIN_FRAGS = 400;              // what the fused parameter chain streams cyclically
    static constexpr int ENC_START = 400;                // + layers.0^T (rows = the 63 input columns, 2 blocks): the gradient w.r.t. the
    static constexpr int N_FRAGS = 432;                  //   encoded position; the density-gradient chain (RefNeRF.get_grad) streams all 432
    LAYOUT_HD static constexpr size_t stream_bytes(int prec) { return (size_t)N_FRAGS * (prec == NERF_AMD_BF16 ? 1024 : 2048); }
    LAYOUT_HD static constexpr size_t packed_bytes(int prec) { return stream_bytes(prec); }
};
struct MipBwdLayout {
    static constexpr int N_LAYERS = 8;
    static constexpr int NKG[8] = {2, 9, 16, 16, 16, 16, 16, 16};
    static constexpr int NFB[8] = {4, 8, 8, 8, 8, 8, 8, 8};
    static constexpr int START[8] = {0, 8, 80, 208, 336, 464, 592, 720};
    static constexpr int N_FRAGS = 848;
    static constexpr size_t FOLD_SCRATCH = (size_t)128 * 256 * 4;     // W_fold (fp32), rebuilt by every pack
    LAYOUT_HD static constexpr size_t stream_bytes(int prec) { return (size_t)N_FRAGS * (prec == NERF_AMD_BF16 ? 1024 : 2048); }
    LAYOUT_HD static constexpr size_t packed_bytes(int prec) { return stream_bytes(prec) + FOLD_SCRATCH; }
};
// Ref-NeRF backward: three fused chains (bwd_kernels.hip), each a contiguous cyclic stream of transposed layers in consumption order
// (rows x K groups; "e" = a side output in the reference's column order, written as fp32 rows):
//   DIR  (directional network, from the spec head down to the 167-wide input vector)
//      0 R   spec_rgb_head^T 256 x 2 (second K group: zero pad)   1-3 D7,D6,D5 dir_block2.{6,4,2}^T   4 D4h dir_block2.0[:, 167:]^T
//      5 D4a dir_block2.0[:, :167]^T 167 x 16 (e)                  6-8 D3,D2,D1 dir_block1.{6,4,2}^T   9 D0a dir_block1.0^T 167 x 16 (e)
//   SPA  (spatial network, parameter gradients: from [bottle-neck | heads] down to the first hidden layer)
//     10 H   [bottle_neck ; heads]^T 256 x 10 (K group 9: zero pad)   11-13 S7,S6,S5 spa_block2.{6,4,2}^T   14 S4h spa_block2.0[:, 63:]^T
//     15-17 S3,S2,S1 spa_block1.{6,4,2}^T
//   DEN  (spatial network, d density / d encoded position: RefNeRF.get_grad)
//     18 Hd  density row of the heads^T 256 x 2 (zero pad)   19-21 = 11-13   22 = 14   23 S4e spa_block2.0[:, :63]^T 63 x 16 (e)
//     24-26 = 15-17   27 S0e spa_block1.0^T 63 x 16 (e)
// Layers 19-22 and 24-26 are second copies of 11-17 (7 x 128 KiB in bf16): a chain's stream has to be contiguous.
struct RefBwdLayout {
    static constexpr int N_LAYERS = 28;
    static constexpr int NKG[28] = {2, 16, 16, 16, 16, 16, 16, 16, 16, 16,   10, 16, 16, 16, 16, 16, 16, 16,   2, 16, 16, 16, 16, 16, 16, 16, 16, 16};
    static constexpr int NFB[28] = {8, 8, 8, 8, 8, 6, 8, 8, 8, 6,   8, 8, 8, 8, 8, 8, 8, 8,   8, 8, 8, 8, 8, 2, 8, 8, 8, 2};
    static constexpr int START[28] = {0, 16, 144, 272, 400, 528, 624, 752, 880, 1008,
                                      1104, 1184, 1312, 1440, 1568, 1696, 1824, 1952,
                                      2080, 2096, 2224, 2352, 2480, 2608, 2640, 2768, 2896, 3024};
    static constexpr int DIR_START = 0, DIR_FRAGS = 1104, SPA_START = 1104, SPA_FRAGS = 976, DEN_START = 2080, DEN_FRAGS = 976;
    static constexpr int N_FRAGS = 3056;
    LAYOUT_HD static constexpr size_t stream_bytes(int prec) { return (size_t)N_FRAGS * (prec == NERF_AMD_BF16 ? 1024 : 2048); }
    LAYOUT_HD static constexpr size_t packed_bytes(int prec) { return stream_bytes(prec); }
};

// Dump slots (one slot = 16 K groups per 32-sample subtile, fragment order).  Activation dump of the training forward:
//   proposal 0..3 = layers.{0,2,4,6} outputs, 4 = [PE10 encoding: K groups 0..3];
//   MipNeRF  0..3 = lin_block1 outputs, 4..6 = lin_block2 outputs, 7 = rgb_layer.0 output (K groups 0..7),
//            8 = [PE10 encoding of the position: K groups 0..3 | PE4 encoding of the direction: K groups 4..5].
// Delta dump of the backward chain: slot L = delta of the layer whose activations sit in activation slot L; the head K group goes to
// K group 0 of slot 4 (proposal) / slot 8 (MipNeRF).
constexpr int PROP_DUMP_SLOTS = 5, MIP_DUMP_SLOTS = 9, REF_DUMP_SLOTS = 17;
// fp8 dumps (NERF_AMD_BF16_F8): a HIDDEN slot (proposal 0..3, MipNeRF 0..7; activations and deltas alike) holds per 32-sample subtile
//   8 data blocks of 1 KiB -- block fb, lane l: [K group 2 fb: 8 x e4m3 | K group 2 fb + 1: 8 x e4m3] -- followed by 1 KiB of scale
//   exponents -- lane l, byte kg = E (biased like an fp32 exponent): element value = e4m3 * 2^(E - 127).
// The slot keeps its `layer_stride` footprint (subtiles 9 KiB apart inside it); the encoding / head slot (proposal 4, MipNeRF 8) stays bf16.
constexpr int F8_SUB_BYTES = 9216, F8_SCALE_OFF = 8192;

// RefNeRF(10, 4, bottle_neck 128, hidden 256, output 256)  (ref_model.py:16-66), eval mode, use_srgb = False.
//   spatial: S0 63->256, S1-3, S4 319->256 (skip), S5-7;  H: [bottle_neck 128 rows | 11 head rows];
//   directional: D0 167->256, D1-3, D4 423->256 (skip), D5-7;  R: spec_rgb_head 256->3.
// Head rows of H's 5th feature block: 0-2 normal, 3 roughness, 4-6 diffuse, 7 density, 8-10 tint
// (rows 0-3 / 8-10 land in lane half 0, rows 4-7 in lane half 1 of the accumulator).
// The 167 directional inputs are [bottle_neck 128 | IDE real 19 | IDE imag 19 | n.d]; the 39 computed ones use
// 3 K groups: slot q of lane half h is IDE term q (real part for h=0, imaginary for h=1), slot 19 of half 0 is n.d.
struct RefLayout {
    static constexpr int N_LAYERS = 18;
    static constexpr int NKG[18] = {4, 16, 16, 16, 20, 16, 16, 16, 16, 11, 16, 16, 16, 27, 16, 16, 16, 16};
    static constexpr int NFB[18] = {8, 8, 8, 8, 8, 8, 8, 8, 5, 8, 8, 8, 8, 8, 8, 8, 8, 1};
    static constexpr int START[18] = {0, 32, 160, 288, 416, 576, 704, 832, 960, 1040, 1128, 1256, 1384, 1512, 1728, 1856, 1984, 2112};
    static constexpr int BIAS_OFF[18] = {0, 256, 512, 768, 1024, 1280, 1536, 1792, 2048, 2208, 2464, 2720, 2976, 3232, 3488, 3744, 4000, 4256};
    static constexpr int N_FRAGS = 2128;
    static constexpr int N_BIAS = 4288;
    static constexpr int N_IDE = 176;                 // 9 x 19 spherical-harmonic coefficients (ref_func.py:60-74), padded
    static constexpr int IDE_TERMS = 19;
    LAYOUT_HD static constexpr size_t stream_bytes(int prec) { return (size_t)N_FRAGS * (prec == NERF_AMD_BF16 ? 1024 : 2048); }
    LAYOUT_HD static constexpr size_t packed_bytes(int prec) { return stream_bytes(prec) + (N_BIAS + N_IDE) * 4; }
};
// column of dir_block{1,2}.0 that IDE slot (q, h) multiplies (relative to the start of the 39 computed inputs), or -1
LAYOUT_HD inline int ide_slot_column(int q, int h) {
    if (q < 19) return (h ? 19 : 0) + q;
    if (q == 19) return h ? -1 : 38;
    return -1;
}
