// Backward of the two MLPs as hand-written gfx950 kernels (second stage): no library GEMM is involved.
//
//   1. dgrad chain  (prop_bwd_kernel / mip_bwd_kernel): the forward machinery of mlp_core.h run on TRANSPOSED packed weights
//      (mlp_layout.h *BwdLayout).  delta stays in registers from the heads down to the first hidden layer exactly like the
//      activations do in the forward: D[input feature][sample] = W^T . delta with A = weight fragments streamed through the LDS
//      ring and B = delta in registers; the epilogue between two layers is the ReLU adjoint delta *= [y > 0], with y taken from
//      the training forward's dump.  The forward records ONE BIT per activation next to the dump (it ORs the sign tests of its
//      output registers into a per-wave LDS word table and writes 32 B per sample and layer); the chain LDS-DMAs a layer's 1 KiB of
//      bits per 256-sample subtile when the layer starts into a per-wave double buffer and expands them on registers (the dump's
//      fragment order IS the B-operand layout, so masking is element-wise) -- no VGPR-held loads inside the MFMA stream.  Every
//      layer's delta is written to HBM in the same fragment order (the "delta dump"): the operand of the weight gradients.
//      HBM traffic per sample and 256-wide layer: 32 B of mask bits + 512 B of delta writes (bf16).
//   2. wgrad (wgrad_kernel_bf16 / wgrad_kernel_f32): dW = delta^T . y contracts over SAMPLES, while both dumps hold samples along the
//      lanes (lane = sample, registers = features).  The transposition is done by the matrix cores themselves: used as the A operand
//      of a 32x32x16 MFMA against a constant 0/1 selection matrix, a fragment block comes out with lane = feature and registers =
//      samples -- exactly the A / B operand layout of the contraction over samples (any sample permutation is fine as long as
//      delta and y use the same one, and they do).  +12.5 % MFMAs, zero LDS traffic, no cross-wave exchange: every wave owns a
//      rectangle of at most 16 32x32 output blocks (256 accumulator registers), streams the K groups of delta and y it needs straight
//      from L2/HBM (16-byte coalesced loads, next subtile in flight while the current one is multiplied) and keeps its partial dW in
//      registers over its whole sample range.  Workgroup partials go to HBM once; wgrad_finalize_kernel sums them in a fixed order
//      (no atomics: a training run is bit-reproducible) straight into the reference's (out, in) layout.  Bias gradients are the row
//      sums of the transposed delta blocks, accumulated on the side.
//      HBM-bound by design: 1 KiB per sample and 256x256 layer against 131 kFLOP -> ~0.65 PFLOP/s at 5 TB/s.
//   3. mip_fold_grads_kernel: bottle_neck.0 is folded into rgb_layer.0 by the forward (mlp_layout.h), so their gradients are recovered
//      from the one data-dependent product G = dc^T . g6 by parameter-space algebra.
//   4. adam_kernel: Adam (torch.optim.Adam semantics, train.py:117-118) over all parameters of both networks in one launch.
//
// Reference semantics: what torch.autograd computes for addtional.py:88-96 and mip_model.py:41-60 (train.py:164-218).
#include <type_traits>

#include "mlp_core.h"

namespace {

using PB16 = PBF16W;

struct Dump {               // a fragment-ordered dump: slot l, subtile s, K group kg -> base + l*layer_stride + (s*16 + kg) * BREG_LDS
    char* base;
    unsigned long long layer_stride;
    const char* mask_base;  // the forward's ReLU bit masks: slot l, subtile s -> mask_base + l*mask_layer_stride + s*1024 (mlp_kernels.hip mask_or)
    unsigned long long mask_layer_stride;
};

// one 1 KiB global -> LDS DMA piece (16 B per lane, lane-linear); `lds_dst` must be wave-uniform (see WeightStream::issue for why asm)
DEVINL void glds_piece(const char* gptr, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gptr), "s"(lds_dst)
                 : "memory");
}

// delta *= [y > 0] on one B register group.  y is a post-ReLU activation: non-negative, +0 exactly where the unit was off.
DEVINL bf16x8 relu_mask(const bf16x8& d, const bf16x8& y) {
    // two packed 16-bit ops per dword: on = min(y, 1) (0 / 1 per element), d * on.  Written as asm: from the vector-builtin form
    // (__builtin_elementwise_min + a 16-bit multiply) hipcc recognises "y != 0 ? d : 0" and scalarises it into one compare, one select
    // and half a v_perm per ELEMENT -- 3x the instructions and the registers to match (the fused Ref-NeRF chains spilled 170-700
    // registers with it, none without).
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4 dv = __builtin_bit_cast(u32x4, d), yv = __builtin_bit_cast(u32x4, y);
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t on, out;
        const uint32_t yi = yv[i], di = dv[i];
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(on) : "v"(yi), "v"(0x00010001u));
        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(out) : "v"(di), "v"(on));
        r[i] = out;
    }
    return __builtin_bit_cast(bf16x8, r);
}
DEVINL f32x8 relu_mask(const f32x8& d, const f32x8& y) {
    f32x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = y[e] > 0.0f ? d[e] : 0.0f;
    return r;
}

// the same from the forward's bit masks: `w` = the lane's mask dword of K groups 4 (kg >> 2) .. + 3; element e of K group kg is bit
// 4 (kg & 3) + (e >> 1) + 16 (e & 1)
DEVINL bf16x8 relu_mask_bits(const bf16x8& d, uint32_t w, int kg) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
    u32x4 v = __builtin_bit_cast(u32x4, d);
    const uint32_t ws = w >> (4 * (kg & 3));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t sel = (ws >> i) & 0x00010001u, x = v[i];   // (scalar copies: __builtin_bit_cast of a vector-element expression misreads)
        v[i] = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, x) * __builtin_bit_cast(u16x2, sel)));
    }
    return __builtin_bit_cast(bf16x8, v);
}
DEVINL f32x8 relu_mask_bits(const f32x8& d, uint32_t w, int kg) {
    const uint32_t ws = w >> (4 * (kg & 3));
    f32x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = ((ws >> ((e >> 1) + 16 * (e & 1))) & 1u) ? d[e] : 0.0f;
    return r;
}

constexpr uint32_t BWD_LDS_ZERO = MLP_RING_BYTES;                    // 1 KiB of zeros: the "bias" of every chain layer (BWD_ZERO_BIAS = 0: read from here)
#ifndef BWD_ZERO_BIAS
#define BWD_ZERO_BIAS 1        /* the chains' accumulators start from the literal 0 (mlp_core.h dense<..., ZB>); 0 = the round-4 form, A/B */
#endif
constexpr uint32_t BWD_LDS_MASK = MLP_RING_BYTES + 1024;
template <class P> constexpr uint32_t bwd_pair_bytes() { return 4 * P::NT * P::BREG_LDS; }          // (2 blocks x 2 halves x NT tiles) mask groups
template <class P> constexpr uint32_t bwd_lds_total() { return BWD_LDS_MASK + P::NW * 2 * bwd_pair_bytes<P>(); }
template <class P> constexpr uint32_t bwd_lds_scale() { return bwd_lds_total<P>(); }                             // F8 chains: scale-exponent records
template <class P> constexpr uint32_t bwd_lds_total_f8() { return bwd_lds_total<P>() + P::NW * 2 * P::NT * 1024; }

// How many weight-ring pieces a wave is guaranteed to have issued AFTER the mask DMA of a feature-block pair of an NKG-step layer by
// the time the next pair starts: one chunk boundary per FPC fragments, LPW pieces per boundary.  VMEM loads return in order, so
// `s_waitcnt vmcnt(that many)` proves the masks have landed (stores in flight can only make the wait stricter, never weaker).
template <class P> constexpr int mask_wait_count(int nkg) { return ((MLP_CHUNK_BYTES / 1024) / P::NW) * ((2 * nkg) / P::FPC); }
template <int N> DEVINL void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Output functor of a layer of the fused chains (NKG_CUR K steps): convert, apply the ReLU adjoint from the forward's BIT masks of
// `layer` (1 KiB per subtile and layer, LDS-DMA'd when the layer starts into the buffer of the layer's parity -- consecutive chain
// layers have alternating slot parity, so the previous layer's pending pair still reads its own buffer), keep in registers, dump.
template <class P, int NKG_CUR, int NKG_PREV, bool F8 = false>
struct MaskedOut {
    typename P::BReg (&buf)[P::NT][16];
    Dump act, dlt;
    int layer;
    int64_t sub0;
    int lane;
    uint32_t mask_lds;          // this wave's two layer buffers of NT KiB each (wave-uniform byte offset)
    uint32_t scale_lds = 0;     // F8: this wave's two x NT scale-exponent records (flushed by f8_flush_layer)

    DEVINL void begin_group(int G) const {
        if (G == 0) {
#pragma unroll
            for (int t = 0; t < P::NT; ++t) {
                const char* src = act.mask_base + (size_t)layer * act.mask_layer_stride + (size_t)(sub0 + t) * 1024 + lane * 16;
                glds_piece(src, __builtin_amdgcn_readfirstlane(mask_lds + ((layer & 1) * P::NT + t) * 1024));
            }
        } else if (G == 1) {
            vm_wait<mask_wait_count<P>(NKG_CUR)>();          // pair 0's epilogue (the first reader) starts now
        }
    }
    DEVINL void operator()(int fb, int t, const f32x16& acc, int half) const {
        const int kg = 2 * fb + half;
        const typename P::BReg d = to_breg_half<P, false>(acc, half);
        const uint32_t w = *reinterpret_cast<const uint32_t*>(smem + mask_lds + ((layer & 1) * P::NT + t) * 1024 + lane * 16 + (kg >> 2) * 4);
        const typename P::BReg v = relu_mask_bits(d, w, kg);
        buf[t][kg] = v;
        if constexpr (F8) {                                   // delta dump in scaled e4m3 (the chain itself keeps bf16 in registers)
            const uint32_t E = f8_group_exponent<true>(v);
            f8_store_group(dlt.base + (size_t)layer * dlt.layer_stride + (size_t)(sub0 + t) * F8_SUB_BYTES, kg, lane, f8_encode_group(v, E), E,
                           scale_lds + ((layer & 1) * P::NT + t) * 1024);
        } else {
            P::store_global(dlt.base + (size_t)layer * dlt.layer_stride + ((size_t)(sub0 + t) * 16 + kg) * (size_t)P::BREG_LDS, lane, v);
        }
    }
};
// F8: delta slot `layer` of this wave's subtiles is complete (its last feature-block pair has been converted): write its scale exponents
template <class P>
DEVINL void f8_flush_layer(const Dump& dlt, uint32_t scale_lds, int layer, int64_t sub0, int lane) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < P::NT; ++t)
        f8_flush_scales(dlt.base + (size_t)layer * dlt.layer_stride + (size_t)(sub0 + t) * F8_SUB_BYTES, lane, scale_lds + ((layer & 1) * P::NT + t) * 1024);
    asm volatile("" ::: "memory");
}

template <class P>
DEVINL void bwd_prologue(WeightStream<P, MLP_NSLOT>& ws, const void* packed, int n_frags) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) reinterpret_cast<float*>(smem + BWD_LDS_ZERO)[i] = 0.0f;
    __syncthreads();
    ws.init(packed, n_frags / P::FPC);
}

// ================================================================================================
// ProposalNetwork dgrad chain: g_density (M) -> delta dump (slots 3..0, head in slot 4)
// ================================================================================================
template <class P, bool F8 = false>
__global__ __launch_bounds__(P::NW * 64) void prop_bwd_kernel(const void* __restrict__ packed, const float* __restrict__ g_density, int64_t M,
                                                              Dump act, Dump dlt) {
    using L = PropBwdLayout;
    using BReg = typename P::BReg;
    WeightStream<P, MLP_NSLOT> ws;
    bwd_prologue<P>(ws, packed, L::CHAIN_FRAGS);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform: lets address arithmetic on it run on the scalar unit)
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    const int64_t n_tiles = (M + TS - 1) / TS;
    const uint32_t mask_lds = __builtin_amdgcn_readfirstlane(BWD_LDS_MASK + wave * 2 * NT * 1024);
    const uint32_t scale_lds = bwd_lds_scale<P>() + wave * 2 * NT * 1024;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t sub0 = tile * (TS / 32) + wave * NT;
        BReg head[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t m = tile * TS + (wave * NT + t) * 32 + j;
            const float g = (h == 0 && m < M) ? g_density[m] : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) P::set(head[t], e, e == 0 ? g : 0.0f);          // slot feature 0 = lane half 0, element 0
            P::store_global(dlt.base + 4 * dlt.layer_stride + ((size_t)(sub0 + t) * 16) * (size_t)P::BREG_LDS, lane, head[t]);
        }
        BReg a[NT][16], b[NT][16], zero_kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) P::set(zero_kg, e, 0.0f);
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
        const MaskedOut<P, 2, 16, F8> O3{a, act, dlt, 3, sub0, lane, mask_lds, scale_lds};
        Deferred<P, 6, 2> d = dense<P, 2, 8, L::START[0], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, [&](int kg, int t) -> BReg { return kg == 0 ? head[t] : zero_kg; }, O3, NoPrev{});
        // d2, d1, d0: a -> b -> a -> b; the two a -> b layers share one code instance through the loop (same chunk parity)
        static_assert(L::START[1] % (2 * P::FPC) == L::START[3] % (2 * P::FPC), "chunk parity");
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const MaskedOut<P, 16, 2, F8> OB{b, act, dlt, 2 - 2 * r, sub0, lane, mask_lds, scale_lds};        // (follows the 2-step head layer in round 0)
            const MaskedOut<P, 16, 16, F8> OA_pend{a, act, dlt, 3 - 2 * r, sub0, lane, mask_lds, scale_lds};
            d = dense<P, 16, 8, L::START[1], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_A, OB, prev_of(d, OA_pend));
            if constexpr (F8) f8_flush_layer<P>(dlt, scale_lds, 3 - 2 * r, sub0, lane);      // (its last pair was converted during this layer)
            if (r == 0) {
                const MaskedOut<P, 16, 16, F8> OA{a, act, dlt, 1, sub0, lane, mask_lds, scale_lds};
                d = dense<P, 16, 8, L::START[2], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_B, OA, prev_of(d, OB));
                if constexpr (F8) f8_flush_layer<P>(dlt, scale_lds, 2, sub0, lane);
            }
        }
        const MaskedOut<P, 16, 16, F8> O0{b, act, dlt, 0, sub0, lane, mask_lds, scale_lds};
        vm_wait<mask_wait_count<P>(16)>();                   // the last pair's masks
        d.flush(O0);
        if constexpr (F8) f8_flush_layer<P>(dlt, scale_lds, 0, sub0, lane);
    }
    ws.drain();
}

// ================================================================================================
// MipNeRF dgrad chain: g_rgbo (M,4), rgbo (M,4) -> delta dump (slots 7..0, head in slot 8)
// ================================================================================================
template <class P, bool F8 = false>
__global__ __launch_bounds__(P::NW * 64) void mip_bwd_kernel(const void* __restrict__ packed, const float* __restrict__ g_rgbo,
                                                             const float* __restrict__ rgbo, int64_t M, Dump act, Dump dlt) {
    using L = MipBwdLayout;
    using BReg = typename P::BReg;
    constexpr int FPC = P::FPC;
    WeightStream<P, MLP_NSLOT> ws;
    bwd_prologue<P>(ws, packed, L::N_FRAGS);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform: lets address arithmetic on it run on the scalar unit)
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    const int64_t n_tiles = (M + TS - 1) / TS;
    const uint32_t mask_lds = __builtin_amdgcn_readfirstlane(BWD_LDS_MASK + wave * 2 * NT * 1024);
    const uint32_t scale_lds = bwd_lds_scale<P>() + wave * 2 * NT * 1024;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t sub0 = tile * (TS / 32) + wave * NT;
        BReg head[NT], zero_kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) P::set(zero_kg, e, 0.0f);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t m = tile * TS + (wave * NT + t) * 32 + j;
            f32x4 hv = {0.0f, 0.0f, 0.0f, 0.0f};
            if (h == 0 && m < M) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(g_rgbo + m * 4), o = *reinterpret_cast<const f32x4*>(rgbo + m * 4);
#pragma unroll
                for (int c = 0; c < 3; ++c) hv[c] = (g[c] * (1.0f - o[c])) * o[c];      // sigmoid adjoint (mip_model.py:60)
                hv[3] = g[3];                                                            // raw sigma: no activation in the network
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) P::set(head[t], e, e < 4 ? hv[e] : 0.0f);       // slot features 0..3 = lane half 0, elements 0..3
            P::store_global(dlt.base + 8 * dlt.layer_stride + ((size_t)(sub0 + t) * 16) * (size_t)P::BREG_LDS, lane, head[t]);
        }
        BReg a[NT][16], b[NT][16];
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
        // dc = W_rgb2^T dpre, masked by c (activation slot 7) -> b[.][0..7]
        const MaskedOut<P, 2, 16, F8> OC{b, act, dlt, 7, sub0, lane, mask_lds, scale_lds};
        const Deferred<P, 2, 2> dcp = dense<P, 2, 4, L::START[0], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO,
            [&](int kg, int t) -> BReg { return kg == 0 ? head[t] : zero_kg; }, OC, NoPrev{});
        // d6 = [W_fold^T | W_sigma^T] [dc | head], masked by g6 (slot 6) -> a
        const MaskedOut<P, 9, 2, F8> O6{a, act, dlt, 6, sub0, lane, mask_lds, scale_lds};
        Deferred<P, 6, 2> d = dense<P, 9, 8, L::START[1], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO,
            [&](int kg, int t) -> BReg { if (kg < 8) return b[t][kg < 8 ? kg : 0]; return head[t]; }, O6, prev_of(dcp, OC));
        if constexpr (F8) f8_flush_layer<P>(dlt, scale_lds, 7, sub0, lane);                  // (slot 7's last pair was converted during the d6 layer)
        // d5 .. d0: six 256 x 256 layers ping-ponging between the register buffers (a -> b -> a ...), two code instances
        static_assert(L::START[2] % (2 * FPC) == L::START[4] % (2 * FPC) && L::START[2] % (2 * FPC) == L::START[6] % (2 * FPC) &&
                      L::START[3] % (2 * FPC) == L::START[5] % (2 * FPC) && L::START[3] % (2 * FPC) == L::START[7] % (2 * FPC), "uniform layer loop");
#pragma unroll 1
        for (int r = 0; r < 3; ++r) {
            const MaskedOut<P, 16, 9, F8> OB{b, act, dlt, 5 - 2 * r, sub0, lane, mask_lds, scale_lds};       // (follows the 9-step layer in round 0)
            const MaskedOut<P, 16, 16, F8> OA_pend{a, act, dlt, 6 - 2 * r, sub0, lane, mask_lds, scale_lds}, OA{a, act, dlt, 4 - 2 * r, sub0, lane, mask_lds, scale_lds};
            d = dense<P, 16, 8, L::START[2], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_A, OB, prev_of(d, OA_pend));
            if constexpr (F8) f8_flush_layer<P>(dlt, scale_lds, 6 - 2 * r, sub0, lane);
            d = dense<P, 16, 8, L::START[3], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_B, OA, prev_of(d, OB));
            if constexpr (F8) f8_flush_layer<P>(dlt, scale_lds, 5 - 2 * r, sub0, lane);
        }
        const MaskedOut<P, 16, 16, F8> O0{a, act, dlt, 0, sub0, lane, mask_lds, scale_lds};
        vm_wait<mask_wait_count<P>(16)>();                   // the last pair's masks
        d.flush(O0);
        if constexpr (F8) f8_flush_layer<P>(dlt, scale_lds, 0, sub0, lane);
    }
    ws.drain();
}

// ================================================================================================
// Ref-NeRF's backward and the density-gradient chains.  (Rounds 1-2 ran them as single-layer launches --
// dgrad_layer_kernel, 1.5 KiB of HBM traffic per sample and layer; the fused chains below replaced them in round 3 and the A/B
// fallback left the sources in round 5: profiles/r03_refnerf_chains_ab.log.)
// ================================================================================================
// row-major fp32 side output of a chain layer: the gradient w.r.t. a layer INPUT that is not a hidden activation (the encoded
// position, Ref-NeRF's directional input vector), in the reference's column order; accumulate = += instead of =
// BF (round 6, bf16 precision only): the rows travel as bf16 -- same (M, ld) ELEMENT layout, half the bytes.  Used for the directional
// chain's (M, 192) gradient w.r.t. Ref-NeRF's input vector, which is written, read back + added to + rewritten, and read a third time
// (ref_heads_delta_kernel): 9.6 GB per 2^14-ray step as fp32, 2.2 % of the step (profiles/r06_dir_chain_rows_bf16_cost_probe.log).  Its
// first 128 columns become bf16 delta fragments anyway; the 39 IDE / n.d columns feed an fp32 element-wise stage whose outputs are bf16 deltas.
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
template <bool BF = false>
struct RowsOutT {                                            // accumulators 8 half .. 8 half + 7 of block fb = features 32 fb + 8 (2 half + q) + 4 h + 0..3
    float* rows; int ld; int accumulate; int64_t m0; int j, h; int64_t M;
    DEVINL void operator()(int fb, int t, const f32x16& acc, int half) const {
        const int64_t m = m0 + t * 32 + j;
        if (m >= M) return;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t at = m * ld + 32 * fb + 8 * (2 * half + q) + 4 * h;
            f32x4 v = {acc[8 * half + 4 * q], acc[8 * half + 4 * q + 1], acc[8 * half + 4 * q + 2], acc[8 * half + 4 * q + 3]};
            if constexpr (BF) {
                // The two lane halves hold interleaved 4-column groups (h = 0: +0..3 and +8..11, h = 1: +4..7 and +12..15).  One
                // v_permlane32_swap per value hands lane h the CONTIGUOUS columns +8 h .. +8 h + 7, so the 16 columns of (fb, half) go out as
                // one 16-byte load / store per lane instead of two 8-byte ones -- the chain pays per store INSTRUCTION, not per byte.
                if (q == 1) continue;
                f32x4 v1 = {acc[8 * half + 4], acc[8 * half + 5], acc[8 * half + 6], acc[8 * half + 7]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, v[i]), __builtin_bit_cast(uint32_t, v1[i]), false, false);
                    v[i] = __builtin_bit_cast(float, (uint32_t)r[0]); v1[i] = __builtin_bit_cast(float, (uint32_t)r[1]);
                }
                // after the swap: h = 0 holds [own group 0 | partner's group 0] = columns +0..7, h = 1 [partner's group 1 | own group 1] = +8..15
                bf16x8* p = reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(rows) + m * ld + 32 * fb + 16 * half + 8 * h);
                if (accumulate) {
                    const bf16x8 o = *p;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v[i] += (float)o[i]; v1[i] += (float)o[4 + i]; }
                }
                bf16x8 w;
#pragma unroll
                for (int i = 0; i < 4; ++i) { w[i] = (__bf16)v[i]; w[4 + i] = (__bf16)v1[i]; }
                *p = w;
            } else {
                f32x4* p = reinterpret_cast<f32x4*>(rows + at);
                if (accumulate) { const f32x4 o = *p; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
                *p = v;
            }
        }
    }
};
using RowsOut = RowsOutT<false>;


// Round 4: the Ref-NeRF training forward writes ReLU bit masks too (mlp_kernels.hip ref_kernel: one register per column tile instead of an
// LDS record), so the fused chains read 32 B per sample and layer like the MipNeRF chain instead of the 512 B of activations above:
// -13.6 GB (directional), -12.1 GB (spatial), -12.1 GB (density gradient) per 2^14-ray step.  Same protocol as MaskedOut: a layer's
// 1 KiB record per subtile is LDS-DMA'd when the layer starts into the buffer of the slot's parity (consecutive chain layers alternate),
// waited for when its second feature-block pair starts (the first reader is pair 0's deferred epilogue).
template <class P, int NKG_CUR, bool STORE, int parity>          // parity = slot & 1: which of the wave's two LDS buffers
struct BitMaskedOut {
    typename P::BReg (&buf)[P::NT][16];
    const char* mask;            // the slot's bit-mask records: subtile s -> mask + s * 1024
    char* dlt;                   // K group 0 of subtile 0 of the delta slot (STORE)
    int64_t sub0; int lane; uint32_t mask_lds;
    static constexpr size_t SUB = 16 * (size_t)P::BREG_LDS;
    DEVINL void begin_group(int G) const {
        if (G == 0) {
#pragma unroll
            for (int t = 0; t < P::NT; ++t)
                glds_piece(mask + (size_t)(sub0 + t) * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(mask_lds + (parity * P::NT + t) * 1024));
        } else if (G == 1) {
            vm_wait<mask_wait_count<P>(NKG_CUR)>();
        }
    }
    DEVINL void operator()(int fb, int t, const f32x16& acc, int half) const {
        const int kg = 2 * fb + half;
        const typename P::BReg d = to_breg_half<P, false>(acc, half);
        const uint32_t w = *reinterpret_cast<const uint32_t*>(smem + mask_lds + (parity * P::NT + t) * 1024 + lane * 16 + (kg >> 2) * 4);
        const typename P::BReg v = relu_mask_bits(d, w, kg);
        buf[t][kg] = v;
        if constexpr (STORE) P::store_global(dlt + (size_t)(sub0 + t) * SUB + (size_t)kg * P::BREG_LDS, lane, v);
        else pin(v);
    }
    static DEVINL void pin(const bf16x8& r) { asm volatile("" ::"v"(r)); }
    template <class T> static DEVINL void pin(const T& r) { asm volatile("" ::"v"(r.lo), "v"(r.hi)); }
};

// The chains' bodies are written as short loops over layer pairs like mip_bwd_kernel, not as ten layers of straight-line code: with
// one basic block per tile hipcc's scheduler and register allocator lose track of the pressure (170-1000 spilled registers per kernel
// against 0-50 in this form); layers that share a loop body must start at the same chunk phase (all layers here are multiples of
// 16 fragments, so they do).

// Nine-layer chains with two skip-layer side outputs -- one body for
//   !DEN  Ref-NeRF's directional network (parameter gradients): spec-head delta (K group 9 of delta slot 8, ref_spec_delta_kernel) ->
//         delta slots 16..9 and rows (M, 192) = dir_block2.0[:, :167]^T D4 + dir_block1.0^T D0, the gradient w.r.t. the 167-wide input vector;
//    DEN  Ref-NeRF's spatial network, d density / d (encoded position) (RefNeRF.get_grad): the density row -> activation slots 7..0,
//         nothing stored but rows (M, 64) = spa_block2.0[:, :63]^T S4 + spa_block1.0^T S0.
template <class P, bool DEN>
__global__ __launch_bounds__(P::NW * 64) void ref_chain9_kernel(const void* __restrict__ stream, int64_t M, const char* __restrict__ masks, unsigned long long mask_stride,
                                                                char* __restrict__ dlt, unsigned long long layer_stride, float* __restrict__ rows) {
    using L = RefBwdLayout;
    using BReg = typename P::BReg;
    using RowsT = RowsOutT<(!DEN && P::PREC == NERF_AMD_BF16)>;     // the directional chain's input-vector gradient travels as bf16 rows in bf16 precision
    constexpr int L0 = DEN ? 18 : 0, S0 = DEN ? L::DEN_START : L::DIR_START;      // first layer of the chain in the layer table, its stream start
    constexpr int TOP = DEN ? 7 : 16;                                              // activation / delta slot of the chain's first hidden layer
    constexpr int NFB_ROWS = DEN ? 2 : 6, LD = DEN ? 64 : 192;
    constexpr bool STORE = !DEN;
    WeightStream<P, MLP_NSLOT> ws;
    bwd_prologue<P>(ws, stream, DEN ? L::DEN_FRAGS : L::DIR_FRAGS);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    constexpr size_t SUB = 16 * (size_t)P::BREG_LDS;
    const int64_t n_tiles = (M + TS - 1) / TS;
    const uint32_t mask_lds = __builtin_amdgcn_readfirstlane(BWD_LDS_MASK + wave * 2 * bwd_pair_bytes<P>());
    auto A = [&](int slot) { return masks + (size_t)slot * mask_stride; };
    auto D = [&](int slot) { return STORE ? dlt + (size_t)slot * layer_stride : nullptr; };
    constexpr int W16 = mask_wait_count<P>(16);
    static_assert(L::START[L0 + 1] % (2 * P::FPC) == L::START[L0 + 3] % (2 * P::FPC) && L::START[L0 + 2] % (2 * P::FPC) == L::START[L0 + 4] % (2 * P::FPC) &&
                  L::START[L0 + 6] % (2 * P::FPC) == L::START[L0 + 8] % (2 * P::FPC) && S0 % (2 * P::FPC) == 0, "chunk phase of the shared loop bodies");
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t sub0 = tile * (TS / 32) + wave * NT;
        BReg head[NT], zero_kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) P::set(zero_kg, e, 0.0f);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (DEN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) P::set(head[t], e, (e == 0 && h == 0) ? 1.0f : 0.0f);          // the density row: slot feature 0
            } else {
                head[t] = P::load_global(dlt + 8 * (size_t)layer_stride + (size_t)(sub0 + t) * SUB + 9 * (size_t)P::BREG_LDS, lane);
            }
        }
        BReg a[NT][16], b[NT][16];
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
        const BitMaskedOut<P, 2, STORE, TOP & 1> OT{a, A(TOP), D(TOP), sub0, lane, mask_lds};
        Deferred<P, 6, 2> d = dense<P, 2, 8, L::START[L0] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, [&](int kg, int t) -> BReg { return kg == 0 ? head[t] : zero_kg; }, OT, NoPrev{});
        // TOP-1 .. TOP-4 (the fourth is the skip layer through its hidden columns): a -> b -> a -> b -> a
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const BitMaskedOut<P, 16, STORE, (TOP - 1) & 1> OB{b, A(TOP - 1 - 2 * r), D(TOP - 1 - 2 * r), sub0, lane, mask_lds};
            const BitMaskedOut<P, 16, STORE, TOP & 1> OA_pend{a, A(TOP - 2 * r), D(TOP - 2 * r), sub0, lane, mask_lds}, OA{a, A(TOP - 2 - 2 * r), D(TOP - 2 - 2 * r), sub0, lane, mask_lds};
            d = dense<P, 16, 8, L::START[L0 + 1] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_A, OB, prev_of(d, OA_pend));
            d = dense<P, 16, 8, L::START[L0 + 2] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_B, OA, prev_of(d, OB));
        }
        // the skip layer's side output from b (= TOP-3, complete); a's last pair (TOP-4) stays pending across it -- the row layer issues no
        // mask DMA, so the pair's activations stay where they are in LDS
        {
            const RowsT R1{rows, LD, 0, sub0 * 32, j, h, M};
            const auto r1 = dense<P, 16, NFB_ROWS, L::START[L0 + 5] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_B, R1, NoPrev{});
            r1.flush(R1);
        }
        // TOP-5, TOP-6, TOP-7: a -> b -> a -> b
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const BitMaskedOut<P, 16, STORE, (TOP - 5) & 1> OB{b, A(TOP - 5 - 2 * r), D(TOP - 5 - 2 * r), sub0, lane, mask_lds};
            const BitMaskedOut<P, 16, STORE, (TOP - 4) & 1> OA_pend{a, A(TOP - 4 - 2 * r), D(TOP - 4 - 2 * r), sub0, lane, mask_lds};
            d = dense<P, 16, 8, L::START[L0 + 6] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_A, OB, prev_of(d, OA_pend));
            if (r == 0) {
                const BitMaskedOut<P, 16, STORE, (TOP - 6) & 1> OA{a, A(TOP - 6), D(TOP - 6), sub0, lane, mask_lds};
                d = dense<P, 16, 8, L::START[L0 + 7] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_B, OA, prev_of(d, OB));
            }
        }
        vm_wait<W16>();                                      // (the row layer has no mask wait of its own)
        const BitMaskedOut<P, 16, STORE, (TOP - 7) & 1> OL{b, A(TOP - 7), D(TOP - 7), sub0, lane, mask_lds};
        const RowsT R2{rows, LD, 1, sub0 * 32, j, h, M};
        const auto r2 = dense<P, 16, NFB_ROWS, L::START[L0 + 9] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_B, R2, prev_of(d, OL));
        r2.flush(R2);
    }
    ws.drain();
}

// Ref-NeRF, spatial network (parameter gradients): [bottle-neck delta | head deltas] (K groups 0..8 of delta slot 8,
// ref_heads_delta_kernel) -> delta slots 7..0
template <class P>
__global__ __launch_bounds__(P::NW * 64) void ref_spa_bwd_kernel(const void* __restrict__ stream, int64_t M, const char* __restrict__ masks, unsigned long long mask_stride,
                                                                 char* __restrict__ dlt, unsigned long long layer_stride) {
    using L = RefBwdLayout;
    using BReg = typename P::BReg;
    constexpr int S0 = L::SPA_START;
    WeightStream<P, MLP_NSLOT> ws;
    bwd_prologue<P>(ws, stream, L::SPA_FRAGS);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    constexpr size_t SUB = 16 * (size_t)P::BREG_LDS;
    const int64_t n_tiles = (M + TS - 1) / TS;
    const uint32_t mask_lds = __builtin_amdgcn_readfirstlane(BWD_LDS_MASK + wave * 2 * bwd_pair_bytes<P>());
    auto A = [&](int slot) { return masks + (size_t)slot * mask_stride; };
    auto D = [&](int slot) { return dlt + (size_t)slot * layer_stride; };
    static_assert(L::START[11] % (2 * P::FPC) == L::START[13] % (2 * P::FPC) && L::START[11] % (2 * P::FPC) == L::START[15] % (2 * P::FPC) &&
                  L::START[11] % (2 * P::FPC) == L::START[17] % (2 * P::FPC) && L::START[12] % (2 * P::FPC) == L::START[14] % (2 * P::FPC) &&
                  L::START[12] % (2 * P::FPC) == L::START[16] % (2 * P::FPC) && S0 % (2 * P::FPC) == 0, "chunk phase of the shared loop bodies");
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t sub0 = tile * (TS / 32) + wave * NT;
        BReg x[NT][9], zero_kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) P::set(zero_kg, e, 0.0f);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 9; ++k) x[t][k] = P::load_global(D(8) + (size_t)(sub0 + t) * SUB + (size_t)k * P::BREG_LDS, lane);
        BReg a[NT][16], b[NT][16];
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
        const BitMaskedOut<P, 10, true, 1> O7{a, A(7), D(7), sub0, lane, mask_lds};
        Deferred<P, 6, 2> d = dense<P, 10, 8, L::START[10] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO,
            [&](int kg, int t) -> BReg { if (kg < 9) return x[t][kg < 9 ? kg : 0]; return zero_kg; }, O7, NoPrev{});
        // S6 .. S0: a -> b -> a ... -> b; round 3 runs the first layer of the body only
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const BitMaskedOut<P, 16, true, 0> OB{b, A(6 - 2 * r), D(6 - 2 * r), sub0, lane, mask_lds};
            const BitMaskedOut<P, 16, true, 1> OA_pend{a, A(7 - 2 * r), D(7 - 2 * r), sub0, lane, mask_lds};
            d = dense<P, 16, 8, L::START[11] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_A, OB, prev_of(d, OA_pend));
            if (r < 3) {
                const BitMaskedOut<P, 16, true, 1> OA{a, A(5 - 2 * r), D(5 - 2 * r), sub0, lane, mask_lds};
                d = dense<P, 16, 8, L::START[12] - S0, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_B, OA, prev_of(d, OB));
            }
        }
        const BitMaskedOut<P, 16, true, 0> O0{b, A(0), D(0), sub0, lane, mask_lds};
        vm_wait<mask_wait_count<P>(16)>();
        d.flush(O0);
    }
    ws.drain();
}

// the proposal network's d density / d (encoded position) (train.py:165-168, `prop_normal`): the head row -> activation slots 3..0,
// nothing stored but d_enc (M, 64) = layers.0^T d0
template <class P>
__global__ __launch_bounds__(P::NW * 64) void prop_density_chain_kernel(const void* __restrict__ stream, int64_t M, const char* __restrict__ masks,
                                                                        unsigned long long mask_stride, float* __restrict__ d_enc) {
    using L = PropBwdLayout;
    using BReg = typename P::BReg;
    WeightStream<P, MLP_NSLOT> ws;
    bwd_prologue<P>(ws, stream, L::N_FRAGS);
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NT = P::NT;
    constexpr int TS = P::NW * NT * 32;
    const int64_t n_tiles = (M + TS - 1) / TS;
    const uint32_t mask_lds = __builtin_amdgcn_readfirstlane(BWD_LDS_MASK + wave * 2 * bwd_pair_bytes<P>());
    auto A = [&](int slot) { return masks + (size_t)slot * mask_stride; };
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t sub0 = tile * (TS / 32) + wave * NT;
        BReg one_kg, zero_kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) { P::set(one_kg, e, (e == 0 && h == 0) ? 1.0f : 0.0f); P::set(zero_kg, e, 0.0f); }
        BReg a[NT][16], b[NT][16];
        auto IN_A = [&](int kg, int t) -> BReg { return a[t][kg]; };
        auto IN_B = [&](int kg, int t) -> BReg { return b[t][kg]; };
        const BitMaskedOut<P, 2, false, 1> O3{a, A(3), nullptr, sub0, lane, mask_lds};
        Deferred<P, 6, 2> d = dense<P, 2, 8, L::START[0], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, [&](int kg, int) -> BReg { return kg == 0 ? one_kg : zero_kg; }, O3, NoPrev{});
        // d2, d1, d0: a -> b -> a -> b; the two a -> b layers share one code instance through the loop (as in prop_bwd_kernel)
        static_assert(L::START[1] % (2 * P::FPC) == L::START[3] % (2 * P::FPC), "chunk phase");
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const BitMaskedOut<P, 16, false, 0> OB{b, A(2 - 2 * r), nullptr, sub0, lane, mask_lds};
            const BitMaskedOut<P, 16, false, 1> OA_pend{a, A(3 - 2 * r), nullptr, sub0, lane, mask_lds};
            d = dense<P, 16, 8, L::START[1], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_A, OB, prev_of(d, OA_pend));
            if (r == 0) {
                const BitMaskedOut<P, 16, false, 1> OA{a, A(1), nullptr, sub0, lane, mask_lds};
                d = dense<P, 16, 8, L::START[2], BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_B, OA, prev_of(d, OB));
            }
        }
        vm_wait<mask_wait_count<P>(16)>();
        const BitMaskedOut<P, 16, false, 0> O0{b, A(0), nullptr, sub0, lane, mask_lds};
        const RowsOut R{d_enc, 64, 0, sub0 * 32, j, h, M};
        const auto r = dense<P, 16, 2, L::ENC_START, BWD_ZERO_BIAS>(ws, BWD_LDS_ZERO, IN_B, R, prev_of(d, O0));
        r.flush(R);
    }
    ws.drain();
}

// ------------------------------------------------------------------------------------------------ element-wise stages of the chains
// where slot feature f < 16 of a one-K-group fragment lives: lane j + 32 ((f % 8) / 4), element 4 (f / 8) + f % 4   (the D map, kg = 0)
template <int ELEM>
DEVINL void store_slot_feature(char* block, int j, int f, float v) {
    const int lane = j + 32 * ((f & 7) >> 2), e = 4 * (f >> 3) + (f & 3);
    if (ELEM == 2) reinterpret_cast<__bf16*>(block)[lane * 8 + e] = (__bf16)v;
    else reinterpret_cast<float*>(block)[(e >> 2) * 256 + lane * 4 + (e & 3)] = v;
}
// all 16 features of sample j of a K group at once: the two lane records (j, h = 0 / 1) of the block as 16-byte pieces (bf16: two stores
// instead of sixteen 2-byte ones; fp32: four f32x4) -- feature f lives in lane half (f & 7) >> 2, element 4 (f >> 3) + (f & 3)
template <int ELEM>
DEVINL void store_slot_features16(char* block, int j, const float (&v)[16]) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int lane = j + 32 * hh;
        const float a0 = v[4 * hh], a1 = v[4 * hh + 1], a2 = v[4 * hh + 2], a3 = v[4 * hh + 3];
        const float b0 = v[8 + 4 * hh], b1 = v[8 + 4 * hh + 1], b2 = v[8 + 4 * hh + 2], b3 = v[8 + 4 * hh + 3];
        if constexpr (ELEM == 2) {
            const bf16x8 w = {(__bf16)a0, (__bf16)a1, (__bf16)a2, (__bf16)a3, (__bf16)b0, (__bf16)b1, (__bf16)b2, (__bf16)b3};
            *reinterpret_cast<bf16x8*>(block + lane * 16) = w;
        } else {
            const f32x4 lo = {a0, a1, a2, a3}, hi = {b0, b1, b2, b3};
            *reinterpret_cast<f32x4*>(block + lane * 16) = lo;
            *reinterpret_cast<f32x4*>(block + 1024 + lane * 16) = hi;
        }
    }
}
DEVINL float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

// gradient of the positional encoding (nerf_helper.py:38-48 with cat_origin): d_enc (M, ld) in the reference's column order
// [x y z | sin 2^0 xyz | cos 2^0 xyz | ...] -> dx_c = d[c] + sum_f 2^f (cos(2^f x_c) d_sin - sin(2^f x_c) d_cos), times scale[m]
__global__ __launch_bounds__(256) void pe_grad_kernel(const float* __restrict__ d_enc, int ld, const float* __restrict__ x, int x_stride, const float* __restrict__ scale, int scale_stride,
                               int64_t M, int L, float* __restrict__ out) {
    const int64_t total = M * 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / 3;
        const int c = (int)(i - m * 3);
        const float xv = x[m * x_stride + c];
        const float* d = d_enc + m * ld;
        float acc = d[c];
        for (int f = 0; f < L; ++f) {
            const float fr = (float)(1 << f), a = xv * fr;
            float sn, cs;
            sincos_quadrant(a, sn, cs);
            acc += fr * (cs * d[3 + 6 * f + c] - sn * d[3 + 6 * f + 3 + c]);
        }
        out[i] = scale ? acc * scale[m * scale_stride] : acc;
    }
}
// The same for positions that went through the Mip-NeRF 360 contraction before the encoding (mlp_kernels.hip contract_position; round 5):
// the encoding's derivative is taken at c(x) and pulled back through the contraction's Jacobian -- for r = |x| > 1, u = x / r,
// c(x) = (2 - 1/r) u and J = (2 - 1/r) / r (I - u u^T) + u u^T / r^2 (symmetric); J = I inside the unit ball.  One thread per sample.
__global__ __launch_bounds__(256) void pe_grad_contract_kernel(const float* __restrict__ d_enc, int ld, const float* __restrict__ x, int x_stride,
                                                               const float* __restrict__ scale, int scale_stride, int64_t M, int L, float* __restrict__ out) {
    for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
        const float x0 = x[m * x_stride], x1 = x[m * x_stride + 1], x2 = x[m * x_stride + 2];
        const float r = norm3(x0, x1, x2);
        const float k = r > 1.0f ? (2.0f - 1.0f / r) / r : 1.0f;
        const float cx[3] = {x0 * k, x1 * k, x2 * k};
        const float* d = d_enc + m * ld;
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = d[c];
            for (int f = 0; f < L; ++f) {
                const float fr = (float)(1 << f), a = cx[c] * fr;
                float sn, cs;
                sincos_quadrant(a, sn, cs);
                acc += fr * (cs * d[3 + 6 * f + c] - sn * d[3 + 6 * f + 3 + c]);
            }
            g[c] = acc;
        }
        if (r > 1.0f) {
            const float u0 = x0 / r, u1 = x1 / r, u2 = x2 / r;
            const float ug = (u0 * g[0] + u1 * g[1]) + u2 * g[2];
            const float t = ug * (1.0f / (r * r) - k);               // J g = k g + (1/r^2 - k) (u.g) u
            g[0] = k * g[0] + t * u0; g[1] = k * g[1] + t * u1; g[2] = k * g[2] + t * u2;
        }
        const float sc = scale ? scale[m * scale_stride] : 1.0f;
        out[m * 3] = g[0] * sc; out[m * 3 + 1] = g[1] * sc; out[m * 3 + 2] = g[2] * sc;
    }
}

// Ref-NeRF, stage 1 of the parameter backward: the spec head's delta.  rgb = f(sigmoid(spec) sigmoid(tint) + sigmoid(diffuse - c))
// (ref_model.py:98-105; f = identity, c = 0, or with use_srgb f = linear_to_srgb, c = log 3): d spec_raw = g_rgb f' sigmoid(tint) sigmoid'(spec),
// written as the head K group (slot features 0..2).
DEVINL float ref_rgb_slope(const float* ax, int c, int srgb) {            // f'(linear colour) of channel c from the saved pre-activations
    if (!srgb) return 1.0f;
    return srgb_slope(sigmoid_f(ax[11 + c]) * sigmoid_f(ax[8 + c]) + sigmoid_f(ax[4 + c] - SRGB_LOG3));
}
template <int ELEM>
__global__ __launch_bounds__(256) void ref_spec_delta_kernel(const float* __restrict__ g_out, int g_stride, const float* __restrict__ aux, int64_t M, int64_t Mpad,
                                      char* __restrict__ frag, unsigned long long sub_stride, int srgb) {
    // every element of the K group is written -- slot features 3..15 and the padding samples M .. Mpad-1 of the last tile as zeros -- so the
    // slot needs no memset (round 4 zeroed the whole 1.6 GB slot first: 0.24 ms per 2^14-ray step)
    for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < Mpad; m += (int64_t)gridDim.x * blockDim.x) {
        char* block = frag + (size_t)(m >> 5) * sub_stride;
        float v[3] = {0.0f, 0.0f, 0.0f};
        if (m < M) {
            const float* ax = aux + m * 16;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float sp = sigmoid_f(ax[11 + c]);
                v[c] = (g_out[m * g_stride + c] * ref_rgb_slope(ax, c, srgb)) * sigmoid_f(ax[8 + c]) * (sp * (1.0f - sp));
            }
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) store_slot_feature<ELEM>(block, (int)(m & 31), f, f < 3 ? v[f] : 0.0f);
    }
}

// Ref-NeRF, stage 2: everything between the directional network's input vector and the heads (ref_model.py:80-96 backwards).
// Per sample: d_allin (167: [bottle-neck 128 | IDE re 19 | IDE im 19 | n.d]) from the directional chain, the loss gradients w.r.t.
// the outputs (g_out (M,7) = [rgb 3, density 1, predicted normal 3]), the saved pre-activation heads (aux) and the view direction ->
//   delta of the 11 head rows [normal 0-2, roughness 3, diffuse 4-6, density 7, tint 8-10] (K group 8 of the slot) and
//   delta of the bottle-neck = d_allin[0:128] (K groups 0..7), both in fragment order for the heads' dgrad / wgrad.
// IDE backward (ref_func.py:76-108): out_t = (x + i y)^m P_t(z) exp(-sigma_l k), P_t(z) = sum_k mat[k][t] z^k, sigma_l = l (l + 1) / 2.
template <int ELEM>
__global__ __launch_bounds__(256) void ref_heads_delta_kernel(const float* __restrict__ g_out, int g_stride, const float* __restrict__ aux, const float* __restrict__ d_allin, int ld,
                                       const float* __restrict__ dirs, int dir_stride, const float* __restrict__ mat, int64_t M, int64_t Mpad,
                                       char* __restrict__ frag, unsigned long long sub_stride, int srgb) {
    constexpr int TM[19] = {0, 1, 0, 1, 2, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5, 6, 7, 8};
    constexpr int TL[19] = {1, 1, 2, 2, 2, 4, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8, 8, 8, 8};
    constexpr int BREG = ELEM == 2 ? 1024 : 2048;
    for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < Mpad; m += (int64_t)gridDim.x * blockDim.x) {
        if (m >= M) {                                        // padding samples of the last tile: zero deltas (the slot is not memset any more)
            char* sub = frag + (size_t)(m >> 5) * sub_stride;
            const float zero16[16] = {};
            store_slot_features16<ELEM>(sub + 8 * BREG, (int)(m & 31), zero16);
            continue;
        }
        const float* ax = aux + m * 16;
        // (bf16 precision: the directional chain wrote the rows as bf16, RowsOutT<true>; fp32 precision: fp32 rows)
        auto DA = [&](int c) -> float {
            if constexpr (ELEM == 2) return (float)(reinterpret_cast<const __bf16*>(d_allin)[m * ld + c]);
            else return d_allin[m * ld + c];
        };
        const float* g = g_out + m * g_stride;
        const float dx = dirs[m * dir_stride], dy = dirs[m * dir_stride + 1], dz = dirs[m * dir_stride + 2];
        // forward quantities
        const float n0x = ax[0], n0y = ax[1], n0z = ax[2];
        const float len = norm3(n0x, n0y, n0z), nn = len + 1e-7f;
        const float nx = -n0x / nn, ny = -n0y / nn, nz = -n0z / nn;
        const float dot = (dx * nx + dy * ny) + dz * nz;
        const float rx = dx - 2.0f * dot * nx, ry = dy - 2.0f * dot * ny, rz = dz - 2.0f * dot * nz;
        const float kinv = softplus_f(ax[3] - 1.0f);
        float zp[9], re[9], im[9];
        zp[0] = 1.0f; re[0] = 1.0f; im[0] = 0.0f;
#pragma unroll
        for (int k = 1; k < 9; ++k) { zp[k] = zp[k - 1] * rz; re[k] = re[k - 1] * rx - im[k - 1] * ry; im[k] = re[k - 1] * ry + im[k - 1] * rx; }
        const float att[4] = {expf(-1.0f * kinv), expf(-3.0f * kinv), expf(-10.0f * kinv), expf(-36.0f * kinv)};
        const float sig[4] = {1.0f, 3.0f, 10.0f, 36.0f};
        // IDE backward
        float d_re[9], d_im[9], d_rz = 0.0f, d_kinv = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { d_re[k] = 0.0f; d_im[k] = 0.0f; }
#pragma unroll
        for (int t = 0; t < 19; ++t) {
            const int mm = TM[t], l = TL[t];
            const int li = (l == 1) ? 0 : ((l == 2) ? 1 : ((l == 4) ? 2 : 3));
            float poly = 0.0f, dpoly = 0.0f;
#pragma unroll
            for (int k = 0; k <= l - mm; ++k) {
                poly = __builtin_fmaf(mat[k * 19 + t], zp[k], poly);
                if (k >= 1) dpoly = __builtin_fmaf((float)k * mat[k * 19 + t], zp[k - 1], dpoly);
            }
            const float gr = DA(128 + t), gi = DA(128 + 19 + t);
            const float A = gr * re[mm] + gi * im[mm];
            d_rz += A * att[li] * dpoly;
            d_kinv -= A * poly * sig[li] * att[li];
            d_re[mm] += gr * poly * att[li];
            d_im[mm] += gi * poly * att[li];
        }
        float d_rx = 0.0f, d_ry = 0.0f;
#pragma unroll
        for (int k = 1; k < 9; ++k) {                        // d (x + i y)^k / dx = k (x + i y)^(k-1),  d / dy = i k (x + i y)^(k-1)
            d_rx += (float)k * (d_re[k] * re[k - 1] + d_im[k] * im[k - 1]);
            d_ry += (float)k * (d_im[k] * re[k - 1] - d_re[k] * im[k - 1]);
        }
        // normal: gradient from the loss (predicted normal output), from n.d and from the reflection r = d - 2 (d.n) n
        const float g_nd = DA(166);
        const float rdotn = (d_rx * nx + d_ry * ny) + d_rz * nz;
        float dnx = g[4] + g_nd * dx - 2.0f * (rdotn * dx + dot * d_rx);
        float dny = g[5] + g_nd * dy - 2.0f * (rdotn * dy + dot * d_ry);
        float dnz = g[6] + g_nd * dz - 2.0f * (rdotn * dz + dot * d_rz);
        // n = -n0 / (|n0| + eps):  d n0 = -( dn / nn - n0 (n0 . dn) / (|n0| nn^2) )
        const float n0dn = (n0x * dnx + n0y * dny) + n0z * dnz;
        const float cden = n0dn / (fmaxf(len, 1e-30f) * nn * nn);
        float dh[11];
        dh[0] = -(dnx / nn - n0x * cden); dh[1] = -(dny / nn - n0y * cden); dh[2] = -(dnz / nn - n0z * cden);
        dh[3] = d_kinv * sigmoid_f(ax[3] - 1.0f);                                           // softplus'(v) = sigmoid(v)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sd = sigmoid_f(srgb ? ax[4 + c] - SRGB_LOG3 : ax[4 + c]), st = sigmoid_f(ax[8 + c]), sp = sigmoid_f(ax[11 + c]);
            const float gl = g[c] * ref_rgb_slope(ax, c, srgb);
            dh[4 + c] = gl * (sd * (1.0f - sd));
            dh[8 + c] = gl * sp * (st * (1.0f - st));
        }
        dh[7] = g[3];
        char* sub = frag + (size_t)(m >> 5) * sub_stride;
        const int j = (int)(m & 31);
        float out16[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) out16[f] = f < 11 ? dh[f] : 0.0f;
        store_slot_features16<ELEM>(sub + 8 * BREG, j, out16);
    }
    // delta of the bottle-neck = d_allin[:, 0:128] -> K groups 0..7 in fragment order.  Round 4: one work item per (subtile, K group, lane)
    // -- the lane's 8 features of its sample are two aligned float4 reads of the row and ONE 16-byte slot of the fragment block, so a wave
    // writes 1 KiB contiguously (it was 128 strided scalar reads and 128 two-byte scattered stores per sample inside the loop above:
    // 1.7 ms per 2^14-ray step).
    const int64_t n_items = (Mpad / 32) * 8 * 64;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), kg = (int)((i >> 6) & 7);
        const int64_t sb = i >> 9, m = sb * 32 + (lane & 31);
        char* blk = frag + (size_t)sb * sub_stride + (size_t)kg * BREG;
        if constexpr (ELEM == 2) {                           // bf16 rows -> the bf16 fragment block: a pure re-ordering
            bf16x4 lo4 = {(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f}, hi4 = lo4;
            if (m < M) {
                const __bf16* da = reinterpret_cast<const __bf16*>(d_allin) + m * ld + 16 * kg + 4 * (lane >> 5);
                lo4 = *reinterpret_cast<const bf16x4*>(da); hi4 = *reinterpret_cast<const bf16x4*>(da + 8);
            }
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = lo4[e]; v[4 + e] = hi4[e]; }
            *reinterpret_cast<bf16x8*>(blk + lane * 16) = v;
            continue;
        }
        f32x4 lo = {0.0f, 0.0f, 0.0f, 0.0f}, hi = lo;
        if (m < M) {
            const float* da = d_allin + m * ld + 16 * kg + 4 * (lane >> 5);
            lo = *reinterpret_cast<const f32x4*>(da); hi = *reinterpret_cast<const f32x4*>(da + 8);
        }
        if (ELEM == 2) {
        } else {
            *reinterpret_cast<f32x4*>(blk + lane * 16) = lo;
            *reinterpret_cast<f32x4*>(blk + 1024 + lane * 16) = hi;
        }
    }
}

// ================================================================================================
// wgrad:  dW (32 NOB x 32 NIB) = X^T . Y over all samples;  X = delta dump K groups, Y = activation / encoding dump K groups
// ================================================================================================
struct FragMat { const char* base; unsigned long long sub_stride; };      // K group kg of subtile s: base + s*sub_stride + kg*BREG bytes
struct WgradJob {
    FragMat x0; int kgx0;       // X = K groups [0, kgx0) of x0 followed by the K groups of x1
    FragMat x1;
    FragMat y;
    float* partial;             // [n_wg][32 NOB][32 NIB]
    float* bias_partial;        // [n_wg][32 NOB] row sums of X (nullptr = skip)
};
constexpr int WG_MAX_JOBS = 8;
struct WgradJobs { WgradJob j[WG_MAX_JOBS]; };

enum { Y_DMAP = 0, Y_PE10 = 1, Y_PE4 = 2, Y_IDE = 3 };

// slot (kg, h, e) of a Y operand -> feature (column of the reference weight matrix), or -1
template <int YKIND> DEVINL int y_slot_feature(int kg, int h, int e) {
    if (YKIND == Y_PE10) return pe_slot_column(8 * kg + e, h, 10);
    if (YKIND == Y_PE4) return pe_slot_column(8 * kg + e, h, 4);
    if (YKIND == Y_IDE) return ide_slot_column(8 * kg + e, h);
    return dmap_feature(kg, h, e);
}

// fp8 operands (NERF_AMD_BF16_F8 dumps, mlp_layout.h): K groups kg0 .. kg0 + NK - 1 (kg0 even, NK even) of one subtile as they come from
// memory -- NK / 2 blocks of two K groups + the lane's scale exponents -- and their decoding into bf16 B register groups
template <int NK>
struct F8Raw {
    f32x4 blk[NK / 2];
    uint32_t ex[(NK + 3) / 4];
    // `sub` = the subtile's slot base + lane * 16; exponent bytes of K groups (kg0 & ~3) .. are loaded as whole dwords
    DEVINL void load(const char* sub, int kg0) {
#pragma unroll
        for (int b = 0; b < NK / 2; ++b) blk[b] = *reinterpret_cast<const f32x4*>(sub + (size_t)((kg0 >> 1) + b) * 1024);
#pragma unroll
        for (int d = 0; d < (NK + 3) / 4; ++d) ex[d] = *reinterpret_cast<const uint32_t*>(sub + F8_SCALE_OFF + (kg0 & ~3) + 4 * d);
    }
    DEVINL void decode(int kg0, bf16x8* out) const {
        const int sh = kg0 & 3;                              // (non-zero only for NK = 2: the pair sits in the upper half of its dword)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const u32x4 w = __builtin_bit_cast(u32x4, blk[k >> 1]);
            const uint32_t lo = (k & 1) ? w[2] : w[0], hi = (k & 1) ? w[3] : w[1];
            const uint32_t E = (NK >= 4) ? ((ex[k >> 2] >> (8 * (k & 3))) & 0xffu) : ((ex[0] >> (8 * (k + sh))) & 0xffu);
            out[k] = f8_decode_group(lo, hi, E);
        }
    }
};

// KGX / KGY: K groups of X / Y;  WO: waves along the X (row) dimension, 4 / WO along Y
// XF8 / YF8: the X operand's first source (x0: a delta slot) / the Y operand (a hidden-activation slot) is an fp8 slot; x1 (the head K
// group) and the encoding operands are always bf16
template <int KGX, int KGY, int WO, int YKIND, bool XF8 = false, bool YF8 = false>
__global__ __launch_bounds__(256) void wgrad_kernel_bf16(WgradJobs jobs, int64_t n_sub) {
    static_assert(!YF8 || YKIND == Y_DMAP, "only hidden-activation slots are fp8");
    constexpr int NOB = (KGX + 1) / 2, NIB = (YKIND == Y_DMAP) ? (KGY + 1) / 2 : ((YKIND == Y_PE10 || YKIND == Y_IDE) ? 2 : 1);
    constexpr int WI = 4 / WO, OBW = NOB / WO, IBW = NIB / WI;
    static_assert(NOB % WO == 0 && NIB % WI == 0 && OBW * IBW <= 16, "wave tiling");
    // an odd K-group count leaves the last block half empty; the loads / MFMAs that skip the missing group must be decided at COMPILE
    // time (a runtime "load or not" makes hipcc branch around every load and drain vmcnt each time: the loads of a subtile would run
    // one L2 round trip after the other), so such a dimension is never split across waves
    static_assert((KGX % 2 == 0 || WO == 1) && (YKIND != Y_DMAP || KGY % 2 == 0 || WI == 1), "odd K-group counts need an unsplit dimension");
    constexpr int NXK = 2 * OBW;                                          // X K groups a wave loads per subtile
    constexpr int NYK = (YKIND == Y_DMAP) ? 2 * IBW : KGY;                // PE columns scatter over all K groups
    constexpr f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const WgradJob& J = jobs.j[blockIdx.y];
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform: lets address arithmetic on it run on the scalar unit)
    const int wo = (WO == 1) ? 0 : wave / WI, wi = (WI == 1) ? 0 : wave % WI;
    const int ob0 = wo * OBW, ib0 = wi * IBW;

    // constant 0/1 selection operands of the transposing MFMAs
    bf16x8 idx[2];                                                         // X (always the D map): K-group parity p, feature j of the block
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) idx[p][e] = (__bf16)((dmap_feature(p, h, e) == j) ? 1.0f : 0.0f);
    bf16x8 idy[(YKIND == Y_DMAP) ? 1 : NYK][(YKIND == Y_DMAP) ? 1 : IBW];
    if constexpr (YKIND != Y_DMAP) {
#pragma unroll
        for (int k = 0; k < NYK; ++k)
#pragma unroll
            for (int bi = 0; bi < IBW; ++bi)
#pragma unroll
                for (int e = 0; e < 8; ++e) idy[k][bi][e] = (__bf16)((y_slot_feature<YKIND>(k, h, e) == 32 * (ib0 + bi) + j) ? 1.0f : 0.0f);
    }

    f32x16 acc[OBW][IBW];
#pragma unroll
    for (int a = 0; a < OBW; ++a)
#pragma unroll
        for (int b = 0; b < IBW; ++b) acc[a][b] = zero16;
    float bsum[OBW];
#pragma unroll
    for (int a = 0; a < OBW; ++a) bsum[a] = 0.0f;

    const int64_t per = (n_sub + gridDim.x - 1) / gridDim.x;
    const int64_t s_begin = blockIdx.x * per, s_end = (s_begin + per < n_sub) ? s_begin + per : n_sub;

    auto x_ptr = [&](int64_t s, int kg) -> const char* {                   // kg = K group index inside the concatenated X
        return (kg < J.kgx0 ? J.x0.base + (size_t)s * J.x0.sub_stride + (size_t)kg * 1024
                            : J.x1.base + (size_t)s * J.x1.sub_stride + (size_t)(kg - J.kgx0) * 1024) + lane * 16;
    };
    auto load_x = [&](int64_t s, bf16x8 (&xs)[NXK]) {
#pragma unroll
        for (int k = 0; k < NXK; ++k) {
            if constexpr (KGX % 2 == 0) xs[k] = *reinterpret_cast<const bf16x8*>(x_ptr(s, 2 * ob0 + k));
            else if (k < KGX) xs[k] = *reinterpret_cast<const bf16x8*>(x_ptr(s, k));           // (ob0 == 0: k is a constant after unrolling)
        }
    };
    auto load_y = [&](int64_t s, bf16x8 (&ys)[NYK]) {
#pragma unroll
        for (int k = 0; k < NYK; ++k) {
            const char* base = J.y.base + (size_t)s * J.y.sub_stride + lane * 16;
            if constexpr (YKIND != Y_DMAP) ys[k] = *reinterpret_cast<const bf16x8*>(base + (size_t)k * 1024);
            else if constexpr (KGY % 2 == 0) ys[k] = *reinterpret_cast<const bf16x8*>(base + (size_t)(2 * ib0 + k) * 1024);
            else if (k < KGY) ys[k] = *reinterpret_cast<const bf16x8*>(base + (size_t)k * 1024);
        }
    };
    // fp8 operands of the plain (non-exchange) form: NXF / NYF K groups of the wave come from an fp8 slot (an odd KGX keeps its last K
    // group -- the bf16 head group of x1 -- apart)
    constexpr int NXF = XF8 ? ((KGX % 2 == 0) ? NXK : KGX - 1) : 0, NYF = YF8 ? NYK : 0;
    static_assert(!XF8 || NXF >= 2, "an fp8 X operand needs at least one K-group pair");
    static_assert(!YF8 || KGY % 2 == 0, "fp8 Y operands come in K-group pairs");
    struct RawX { F8Raw<(NXF > 0 ? NXF : 2)> f; bf16x8 tail; };
    auto load_x_raw = [&](int64_t s, RawX& r) {
        r.f.load(J.x0.base + (size_t)s * J.x0.sub_stride + lane * 16, (KGX % 2 == 0) ? 2 * ob0 : 0);
        if constexpr (KGX % 2 != 0) r.tail = *reinterpret_cast<const bf16x8*>(J.x1.base + (size_t)s * J.x1.sub_stride + lane * 16);
    };
    auto decode_x = [&](const RawX& r, bf16x8 (&xs)[NXK]) {
        r.f.decode((KGX % 2 == 0) ? 2 * ob0 : 0, xs);
        if constexpr (KGX % 2 != 0) xs[KGX - 1] = r.tail;
    };
    using RawY = F8Raw<(NYF > 0 ? NYF : 2)>;
    auto load_y_raw = [&](int64_t s, RawY& r) { r.load(J.y.base + (size_t)s * J.y.sub_stride + lane * 16, 2 * ib0); };
    auto cvt8 = [](const f32x16& v, int g) -> bf16x8 { return PBF16::from_acc<false>(v, 8 * g); };

    // one subtile: transpose the X and Y K groups on the matrix cores, multiply the transposed blocks
    auto multiply = [&](const bf16x8 (&xs)[NXK], const bf16x8 (&ys)[NYK]) {
        // X^T blocks of this wave: lane = row feature, 2 x 8 registers = the subtile's 32 samples
        bf16x8 xf[OBW][2];
#pragma unroll
        for (int a = 0; a < OBW; ++a) {
            f32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xs[2 * a], idx[0], zero16, 0, 0, 0);
            if (KGX % 2 == 0 || 2 * a + 1 < KGX) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xs[2 * a + 1], idx[1], t, 0, 0, 0);
            xf[a][0] = cvt8(t, 0); xf[a][1] = cvt8(t, 1);
            if (J.bias_partial != nullptr && wi == 0) {
                float r = 0.0f;
#pragma unroll
                for (int q = 0; q < 16; ++q) r += t[q];
                bsum[a] += r;
            }
        }
#pragma unroll
        for (int b = 0; b < IBW; ++b) {
            f32x16 t = zero16;
            if constexpr (YKIND == Y_DMAP) {
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ys[2 * b], idx[0], t, 0, 0, 0);
                if (KGY % 2 == 0 || 2 * b + 1 < KGY) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ys[2 * b + 1], idx[1], t, 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < NYK; ++k) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ys[k], idy[k][b], t, 0, 0, 0);
            }
            const bf16x8 yf0 = cvt8(t, 0), yf1 = cvt8(t, 1);
#pragma unroll
            for (int a = 0; a < OBW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[a][0], yf0, acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < OBW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[a][1], yf1, acc[a][b], 0, 0, 0);
        }
    };
    if constexpr (XF8 || YF8) {
        // plain form with fp8 operands: the next subtile travels as raw blocks (fewer registers in flight) and is decoded after the multiply
        bf16x8 xs[NXK], ys[NYK], yn[NYK], xn[NXK];
        RawX xr; RawY yr;
        auto fetch = [&](int64_t s) {
            if constexpr (XF8) load_x_raw(s, xr); else load_x(s, xn);
            if constexpr (YF8) load_y_raw(s, yr); else load_y(s, yn);
        };
        auto land = [&]() {
            if constexpr (XF8) decode_x(xr, xs); else {
#pragma unroll
                for (int k = 0; k < NXK; ++k) xs[k] = xn[k];
            }
            if constexpr (YF8) yr.decode(2 * ib0, ys); else {
#pragma unroll
                for (int k = 0; k < NYK; ++k) ys[k] = yn[k];
            }
        };
        if (s_begin < s_end) { fetch(s_begin); land(); }
        for (int64_t s = s_begin; s < s_end; ++s) {
            if constexpr (WO > 1 || WI > 1) __builtin_amdgcn_s_barrier();
            fetch((s + 1 < s_end) ? s + 1 : s);
            multiply(xs, ys);
            land();
        }
    } else {
    bf16x8 xs[NXK], ys[NYK], xn[NXK], yn[NYK];
    if (s_begin < s_end) { load_x(s_begin, xs); load_y(s_begin, ys); }
    for (int64_t s = s_begin; s < s_end; ++s) {
        // Waves that share K groups (same wo / same wi) start every subtile together, so that the second request of a K group is served
        // by L1 / L2 while the first is still in flight: FETCH_SIZE falls from 1.47x to 1.00x of the algorithmic bytes (PMC).  The kernel's
        // time does not change (it is bound by bytes in flight, not bandwidth) -- the barrier is there to not waste HBM reads.
        if constexpr (WO > 1 || WI > 1) __builtin_amdgcn_s_barrier();
        {   // next subtile in flight while this one is multiplied (the last iteration re-reads its own subtile: unconditional loads)
            const int64_t sn = (s + 1 < s_end) ? s + 1 : s;
            load_x(sn, xn); load_y(sn, yn);
        }
        multiply(xs, ys);
#pragma unroll
        for (int k = 0; k < NXK; ++k) xs[k] = xn[k];
#pragma unroll
        for (int k = 0; k < NYK; ++k) ys[k] = yn[k];
    }
    }
    // partial of this workgroup, row-major (32 NOB) x (32 NIB): register r of lane (j, h) = row 32 ob + (r&3) + 8 (r>>2) + 4 h, column 32 ib + j
    float* out = J.partial + (size_t)blockIdx.x * (32 * NOB) * (32 * NIB);
#pragma unroll
    for (int a = 0; a < OBW; ++a)
#pragma unroll
        for (int b = 0; b < IBW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (ob0 + a) + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)row * (32 * NIB) + 32 * (ib0 + b) + j] = acc[a][b][r];
            }
    if (J.bias_partial != nullptr && wi == 0) {
#pragma unroll
        for (int a = 0; a < OBW; ++a) {
            const float v = bsum[a] + __shfl_xor(bsum[a], 32, 64);
            if (h == 0) J.bias_partial[(size_t)blockIdx.x * (32 * NOB) + 32 * (ob0 + a) + j] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The 256 x 256 product on EIGHT waves (two per SIMD): dW (8 x 8 blocks of 32 x 32) = delta^T . y.
// The four-wave forms above run one wave per SIMD (256 accumulator registers each), and a subtile is a dependent chain -- loads land ->
// transposing MFMAs -> conversions -> LDS exchange -> barrier -> products -- that nothing overlaps: measured 53 % of the MFMA issue
// slots, and neither 44 % fewer bytes (fp8 dumps) nor 17 % fewer MFMAs (transposed exchange) moved its time by more than 3 %.  Here every
// wave owns a 2 x 4 block rectangle (128 accumulator registers), so TWO waves share a SIMD and the hardware interleaves one wave's
// conversions / LDS traffic / barrier wait with the other's MFMAs.  Per subtile and wave: load ONE X and ONE Y feature block (two K groups
// each; fp8: one 16-byte block + a scale dword), transpose them on the matrix cores (4 MFMAs), convert, row-sum the X block for the
// bias, park both in LDS; after the workgroup barrier read the other X block of the row pair and the other three Y blocks of the column
// quad (8 ds_read_b128) and issue the 16 products.  Every block is loaded and transposed exactly once per workgroup.
//   wave w = (wo = w >> 1, wi = w & 1): rows = X blocks 2 wo, 2 wo + 1; columns = Y blocks 4 wi .. 4 wi + 3;
//   it loads / transposes X block w (= 2 wo + wi) and Y block 4 wi + wo.
// ------------------------------------------------------------------------------------------------
// -DWGRAD256_PIPELINED=1: the subtile loop software-pipelined (transposition of s + 1 inside the LDS-read latency of s).  Parity-clean,
// spill-free (230 registers) and measured on one box at EXACTLY the un-pipelined time, with bf16 and with fp8 dumps alike (1.387 / 1.352 ms
// against 1.364 / 1.369 ms, profiles/r03_wgrad_pipelined_ab.log): the subtile's time is neither its bytes nor the length of one wave's
// dependent chain -- the workgroup barrier phase-locks all eight waves, so the LDS phase (96-128 KiB per subtile through a 128 B/clk port)
// and the MFMA phase (2 x 640 cycles per SIMD) of a subtile do not overlap ACROSS waves whatever one wave does inside its own stream.
template <bool F8>
__global__ __launch_bounds__(512) void wgrad256_kernel(WgradJobs jobs, int64_t n_sub) {
    constexpr f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    constexpr uint32_t STAGE = 32 * 1024;                                  // X blocks 0..7 (2 KiB each: two halves) | Y blocks 0..7
    const WgradJob& J = jobs.j[blockIdx.y];
    const int lane = lane_id(), h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wo = wave >> 1, wi = wave & 1;
    const int xb = wave, yb = 4 * wi + wo;                                 // the blocks this wave loads and transposes
    bf16x8 idx[2];                                                         // constant 0/1 selection operands of the transposing MFMAs
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) idx[p][e] = (__bf16)((dmap_feature(p, h, e) == j) ? 1.0f : 0.0f);
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = zero16;
    float bsum = 0.0f;
    const int64_t per = (n_sub + gridDim.x - 1) / gridDim.x;
    const int64_t s_begin = blockIdx.x * per, s_end = (s_begin + per < n_sub) ? s_begin + per : n_sub;
    auto cvt8 = [](const f32x16& v, int g) -> bf16x8 { return PBF16::from_acc<false>(v, 8 * g); };
    struct StageBf16 { bf16x8 x[2], y[2]; };
    struct StageF8 { F8Raw<2> x, y; };
    using Stage = typename std::conditional<F8, StageF8, StageBf16>::type;
    auto fetch = [&](int64_t s, Stage& q) {
        const char* xs = J.x0.base + (size_t)s * J.x0.sub_stride + lane * 16;
        const char* ys = J.y.base + (size_t)s * J.y.sub_stride + lane * 16;
        if constexpr (F8) { q.x.load(xs, 2 * xb); q.y.load(ys, 2 * yb); }
        else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                q.x[k] = *reinterpret_cast<const bf16x8*>(xs + (size_t)(2 * xb + k) * 1024);
                q.y[k] = *reinterpret_cast<const bf16x8*>(ys + (size_t)(2 * yb + k) * 1024);
            }
        }
    };
    auto clamp_s = [&](int64_t s) { return s < s_end ? s : s_end - 1; };
    Stage q0, q1;
    int buf = 0;
    if (s_begin < s_end) { fetch(s_begin, q0); fetch(clamp_s(s_begin + 1), q1); }
    auto body = [&](int64_t s, Stage& cur) {
        bf16x8 ox[2], oy[2];
        if constexpr (F8) { cur.x.decode(2 * xb, ox); cur.y.decode(2 * yb, oy); }
        else { ox[0] = cur.x[0]; ox[1] = cur.x[1]; oy[0] = cur.y[0]; oy[1] = cur.y[1]; }
        f32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ox[0], idx[0], zero16, 0, 0, 0);
        f32x16 u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oy[0], idx[0], zero16, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ox[1], idx[1], t, 0, 0, 0);
        u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oy[1], idx[1], u, 0, 0, 0);
        fetch(clamp_s(s + 2), cur);                                        // `cur` is consumed: refill it two subtiles ahead (past the end: re-read the last)
        bf16x8 xf[2][2], yf[4][2];                                         // local X block 0 = own, 1 = the pair's other; local Y block 0 = own, 1..3 = the others
        xf[0][0] = cvt8(t, 0); xf[0][1] = cvt8(t, 1);
        yf[0][0] = cvt8(u, 0); yf[0][1] = cvt8(u, 1);
        if (J.bias_partial != nullptr) {
            const float r0 = (t[0] + t[1]) + (t[2] + t[3]), r1 = (t[4] + t[5]) + (t[6] + t[7]), r2 = (t[8] + t[9]) + (t[10] + t[11]),
                        r3 = (t[12] + t[13]) + (t[14] + t[15]);
            bsum += (r0 + r1) + (r2 + r3);
        }
        const uint32_t st = (uint32_t)buf * STAGE + lane * 16;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            *reinterpret_cast<bf16x8*>(smem + st + (2 * xb + hf) * 1024) = xf[0][hf];
            *reinterpret_cast<bf16x8*>(smem + st + (16 + 2 * yb + hf) * 1024) = yf[0][hf];
        }
        __syncthreads();                                                   // all sixteen transposed blocks of s are in LDS (and stage buf^1 is free again)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            xf[1][hf] = *reinterpret_cast<const bf16x8*>(smem + st + (2 * (xb ^ 1) + hf) * 1024);
#pragma unroll
            for (int b = 1; b < 4; ++b)
                yf[b][hf] = *reinterpret_cast<const bf16x8*>(smem + st + (16 + 2 * (4 * wi + ((wo + b) & 3)) + hf) * 1024);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[a][hf], yf[b][hf], acc[a][b], 0, 0, 0);
        buf ^= 1;
    };
    for (int64_t s = s_begin; s < s_end;) {
        body(s, q0); if (++s >= s_end) break;
        body(s, q1); ++s;
    }
    // partial of this workgroup, row-major 256 x 256: local X block a -> block xb ^ a; local Y block b -> block 4 wi + ((wo + b) & 3)
    float* out = J.partial + (size_t)blockIdx.x * 256 * 256;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (xb ^ a) + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)row * 256 + 32 * (4 * wi + ((wo + b) & 3)) + j] = acc[a][b][r];
            }
    if (J.bias_partial != nullptr) {
        const float v = bsum + __shfl_xor(bsum, 32, 64);
        if (h == 0) J.bias_partial[(size_t)blockIdx.x * 256 + 32 * xb + j] = v;
    }
}

// fp32 twin (parity mode): v_mfma_f32_32x32x2_f32 contracts two samples per instruction, and with K = 2 the operands ARE single
// elements -- lane (feature i, k) reads dump element (sample 2 r + k, feature i) straight from the fragment-ordered dump (4-byte gathers
// served by L2); no transposition at all.  Exact fp32 products and sums.
template <int KGX, int KGY, int WO, int YKIND>
__global__ __launch_bounds__(256) void wgrad_kernel_f32(WgradJobs jobs, int64_t n_sub) {
    constexpr int NOB = (KGX + 1) / 2, NIB = (YKIND == Y_DMAP) ? (KGY + 1) / 2 : ((YKIND == Y_PE10 || YKIND == Y_IDE) ? 2 : 1);
    constexpr int WI = 4 / WO, OBW = NOB / WO, IBW = NIB / WI;
    static_assert(NOB % WO == 0 && NIB % WI == 0 && OBW * IBW <= 16, "wave tiling");
    constexpr f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const WgradJob& J = jobs.j[blockIdx.y];
    const int lane = lane_id(), hk = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (wave-uniform: lets address arithmetic on it run on the scalar unit)
    const int wo = wave / WI, wi = wave % WI;
    const int ob0 = wo * OBW, ib0 = wi * IBW;
    // byte offset of (feature, sample 0) inside a subtile's fp32 fragment block [kg][e>>2][lane = sample + 32 h][e&3]; -1 = no such feature
    auto slot_off = [](int kg, int hf, int e) -> int { return kg * 2048 + (e >> 2) * 1024 + (32 * hf) * 16 + (e & 3) * 4; };
    int xoff[OBW], yoff[IBW];
    bool xsec[OBW];
#pragma unroll
    for (int a = 0; a < OBW; ++a) {
        const int f = 32 * (ob0 + a) + j;                                  // D map: f = 16 kg + 8 (e>>2) + 4 h + (e&3)
        const int kg = f >> 4;
        xoff[a] = (kg < KGX) ? slot_off(kg < J.kgx0 ? kg : kg - J.kgx0, (f >> 2) & 1, ((f >> 3) & 1) * 4 + (f & 3)) : -1;
        xsec[a] = kg >= J.kgx0;
    }
#pragma unroll
    for (int b = 0; b < IBW; ++b) {
        const int f = 32 * (ib0 + b) + j;
        if constexpr (YKIND == Y_DMAP) {
            yoff[b] = ((f >> 4) < KGY) ? slot_off(f >> 4, (f >> 2) & 1, ((f >> 3) & 1) * 4 + (f & 3)) : -1;
        } else if constexpr (YKIND == Y_IDE) {
            int q = -1, hf = 0;                                            // inverse of ide_slot_column
            if (f < 19) q = f; else if (f < 38) { q = f - 19; hf = 1; } else if (f == 38) q = 19;
            yoff[b] = (q >= 0) ? slot_off(q >> 3, hf, q & 7) : -1;
        } else {
            constexpr int Lp = (YKIND == Y_PE10) ? 10 : 4;
            int q = -1, hf = 0;                                            // inverse of pe_slot_column
            if (f < 3) { q = 3 * Lp + (f == 1 ? 1 : 0); hf = (f == 2) ? 1 : 0; }
            else if (f < 3 + 6 * Lp) { const int t = f - 3; hf = (t % 6) / 3; q = 3 * (t / 6) + (t % 3); }
            yoff[b] = (q >= 0) ? slot_off(q >> 3, hf, q & 7) : -1;
        }
    }
    f32x16 acc[OBW][IBW];
#pragma unroll
    for (int a = 0; a < OBW; ++a)
#pragma unroll
        for (int b = 0; b < IBW; ++b) acc[a][b] = zero16;
    float bsum[OBW];
#pragma unroll
    for (int a = 0; a < OBW; ++a) bsum[a] = 0.0f;
    const int64_t per = (n_sub + gridDim.x - 1) / gridDim.x;
    const int64_t s_begin = blockIdx.x * per, s_end = (s_begin + per < n_sub) ? s_begin + per : n_sub;
    for (int64_t s = s_begin; s < s_end; ++s) {
        const char* x0 = J.x0.base + (size_t)s * J.x0.sub_stride;
        const char* x1 = J.x1.base ? J.x1.base + (size_t)s * J.x1.sub_stride : x0;
        const char* yb = J.y.base + (size_t)s * J.y.sub_stride;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int so = (2 * r + hk) * 16;                              // sample 2 r + k of the subtile
            float xa[OBW], yv[IBW];
#pragma unroll
            for (int a = 0; a < OBW; ++a) xa[a] = (xoff[a] >= 0) ? *reinterpret_cast<const float*>((xsec[a] ? x1 : x0) + xoff[a] + so) : 0.0f;
#pragma unroll
            for (int b = 0; b < IBW; ++b) yv[b] = (yoff[b] >= 0) ? *reinterpret_cast<const float*>(yb + yoff[b] + so) : 0.0f;
#pragma unroll
            for (int a = 0; a < OBW; ++a) {
                bsum[a] += xa[a];
#pragma unroll
                for (int b = 0; b < IBW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[a], yv[b], acc[a][b], 0, 0, 0);
            }
        }
    }
    float* out = J.partial + (size_t)blockIdx.x * (32 * NOB) * (32 * NIB);
    const int h = hk;
#pragma unroll
    for (int a = 0; a < OBW; ++a)
#pragma unroll
        for (int b = 0; b < IBW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (ob0 + a) + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)row * (32 * NIB) + 32 * (ib0 + b) + j] = acc[a][b][r];
            }
    if (J.bias_partial != nullptr && wi == 0) {
#pragma unroll
        for (int a = 0; a < OBW; ++a) {
            const float v = bsum[a] + __shfl_xor(bsum[a], 32, 64);
            if (h == 0) J.bias_partial[(size_t)blockIdx.x * (32 * NOB) + 32 * (ob0 + a) + j] = v;
        }
    }
}

// sum of the workgroup partials (fixed order) into the reference's layout: dst[row * ld + col0 + col], row < rows, col < cols; and
// the bias: bias_dst[row] for row in [brow0, brow0 + brows)
struct FinalizeJob {
    const float* partial; int prow, pcol;       // partial matrix shape (32 NOB, 32 NIB)
    float* dst; int ld, col0, row0, rows, cols; // rows [row0, row0 + rows) of the partial -> dst rows 0..rows-1
    const float* bias_partial; float* bias_dst; int bias_prow, brow0, brows;   // bias_prow = rows of the partial the bias sums belong to
    int n_wg;                                   // workgroup partials to sum
};
constexpr int FIN_MAX_JOBS = 40;
struct FinalizeJobs { FinalizeJob j[FIN_MAX_JOBS]; };

// 256 threads = 64 elements x 4 slices of the workgroup range; every thread sums its slice with four independent accumulators (the
// partials of one element are n_wg strided loads: one dependent chain would be a chain of L2 round trips), then the slices are
// combined through LDS in a fixed order -- the result does not depend on the launch geometry of anything but n_wg.
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(FinalizeJobs jobs) {
    __shared__ float red[4][64];
    const FinalizeJob& J = jobs.j[blockIdx.y];
    const int n_wg = J.n_wg;
    const int el = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int per = (n_wg + 3) / 4;
    const int w_lo = slice * per, w_hi = (w_lo + per < n_wg) ? w_lo + per : n_wg;
    auto slice_sum = [&](const float* p, size_t stride) -> float {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int w = w_lo;
        for (; w + 4 <= w_hi; w += 4) {
            a0 += p[(size_t)w * stride]; a1 += p[(size_t)(w + 1) * stride]; a2 += p[(size_t)(w + 2) * stride]; a3 += p[(size_t)(w + 3) * stride];
        }
        for (; w < w_hi; ++w) a0 += p[(size_t)w * stride];
        return (a0 + a1) + (a2 + a3);
    };
    const int total = J.rows * J.cols;
    for (int base = blockIdx.x * 64; base < total; base += gridDim.x * 64) {
        const int i = base + el;
        float s = 0.0f;
        int r = 0, c = 0;
        if (i < total) {
            r = i / J.cols; c = i - r * J.cols;
            s = slice_sum(J.partial + (size_t)(J.row0 + r) * J.pcol + c, (size_t)J.prow * J.pcol);
        }
        red[slice][el] = s;
        __syncthreads();
        if (slice == 0 && i < total) J.dst[(size_t)r * J.ld + J.col0 + c] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
        __syncthreads();
    }
    if (J.bias_dst != nullptr) {
        for (int base = blockIdx.x * 64; base < J.brows; base += gridDim.x * 64) {
            const int i = base + el;
            red[slice][el] = (i < J.brows) ? slice_sum(J.bias_partial + J.brow0 + i, (size_t)J.bias_prow) : 0.0f;
            __syncthreads();
            if (slice == 0 && i < J.brows) J.bias_dst[i] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
            __syncthreads();
        }
    }
}

// The forward folds bottle_neck.0 (Wb, bb) into rgb_layer.0 (W9 = [W9a 128x256 | W9d 128x27], b9):  pre_c = W9a (Wb g6 + bb) + W9d e + b9.
// With G = dc^T g6 (128 x 256) and s = sum_m dc = db9:
//   dW9a = dc^T (g6 Wb^T + 1 bb^T) = G Wb^T + s bb^T        dWb = W9a^T G        dbb = W9a^T s
// one thread per output element; 256-term dot products in fp32
__global__ void mip_fold_grads_kernel(const float* __restrict__ G, const float* __restrict__ s, const float* __restrict__ w9, const float* __restrict__ wb,
                                      const float* __restrict__ bb, float* __restrict__ dw9, float* __restrict__ dwb, float* __restrict__ dbb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 128 * 256) {                                                   // dW9[:, :256]
        const int r = i >> 8, c = i & 255;                                 // c = bottle-neck feature
        float a = 0.0f;
        for (int k = 0; k < 256; ++k) a = __builtin_fmaf(G[r * 256 + k], wb[c * 256 + k], a);
        dw9[r * 283 + c] = a + s[r] * bb[c];
    } else if (i < 128 * 256 + 256 * 256) {                                // dWb
        const int t = i - 128 * 256, r = t >> 8, c = t & 255;              // r = bottle-neck feature, c = g6 feature
        float a = 0.0f;
        for (int k = 0; k < 128; ++k) a = __builtin_fmaf(w9[k * 283 + r], G[k * 256 + c], a);
        dwb[r * 256 + c] = a;
    } else if (i < 128 * 256 + 256 * 256 + 256) {                          // dbb
        const int r = i - 128 * 256 - 256 * 256;
        float a = 0.0f;
        for (int k = 0; k < 128; ++k) a = __builtin_fmaf(w9[k * 283 + r], s[k], a);
        dbb[r] = a;
    }
}

// Adam over a table of tensors (torch.optim.Adam, no weight decay / amsgrad): in place on p, m, v.  `step` is a DEVICE scalar
// (incremented by the last workgroup... no: by a separate one-thread launch) so that a captured graph replays correctly.
struct AdamTensor { float* p; const float* g; float* m; float* v; long long n; };
constexpr int ADAM_MAX = 48;
struct AdamTable { AdamTensor t[ADAM_MAX]; };
__global__ void adam_kernel(AdamTable tab, const float* __restrict__ step_ptr, double lr, const double* __restrict__ lr_dev, double beta1, double beta2,
                            double eps, float grad_scale) {
    const AdamTensor& T = tab.t[blockIdx.y];
    if (lr_dev != nullptr) lr = lr_dev[0];                  // learning rate from device memory: a captured graph follows the schedule
    // scalars the way torch forms them (Python doubles, rounded to fp32 where they meet the tensors)
    const double step = (double)step_ptr[0];
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2), b2 = (float)beta2, epsf = (float)eps;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < T.n; i += (long long)gridDim.x * blockDim.x) {
        const float g = T.g[i] * grad_scale;
        const float m = T.m[i] + w1 * (g - T.m[i]);                        // torch: exp_avg.lerp_(grad, 1 - beta1)
        const float v = T.v[i] * b2 + w2 * (g * g);                        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        T.m[i] = m; T.v[i] = v;
        const float denom = sqrtf(v) / bc2_sqrt + epsf;
        T.p[i] = T.p[i] - step_size * (m / denom);                         // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
}
__global__ void adam_step_kernel(float* step) { step[0] += 1.0f; }

int bwd_grid(int64_t n_tiles) {
    const int n_cu = nerf_host::cu_count();
    return (int)(n_tiles < n_cu ? n_tiles : n_cu);
}

// The file is compiled as two translation units (Makefile: -DBWD_TU=1 / 2; 0 = everything in one): 2 holds the fused chains of
// Ref-NeRF's backward and of the density gradient -- whose straight-line ten-layer bodies do not go through hipcc's
// -amdgpu-mfma-vgpr-form pass -- behind bwd_launch_chain, 1 everything else.
#ifndef BWD_TU
#define BWD_TU 0
#endif
// TU 3 (round 4): the Ref-NeRF DIRECTIONAL chain alone, compiled WITHOUT -amdgpu-mfma-vgpr-form: with the bit-mask functor hipcc's
// AGPR-copy rewrite pass segfaults on this one kernel (AMDGPURewriteAGPRCopyMFMA, eliminateSpillsOfReassignedVGPRs); without the flag it
// compiles with 65 spilled registers (91 in round 3's activation-mask form with the flag).
#if BWD_TU == 3 || BWD_TU == 0
template <class P>
int launch_dir_chain_t(const char* stream, int64_t M, const char* masks, size_t ms, char* dlt, size_t ls, float* rows, hipStream_t st) {
    constexpr int TS = P::NW * P::NT * 32;
    const int64_t n_tiles = (M + TS - 1) / TS;
    if (n_tiles == 0) return 0;
    const size_t lds = bwd_lds_total<P>();
    if (int e = nerf_host::allow_dynamic_lds(reinterpret_cast<const void*>(ref_chain9_kernel<P, false>), lds)) return e;
    hipLaunchKernelGGL((ref_chain9_kernel<P, false>), dim3(bwd_grid(n_tiles)), dim3(P::NW * 64), lds, st, stream, M, masks, (unsigned long long)ms, dlt,
                       (unsigned long long)ls, rows);
    return (int)hipGetLastError();
}
}  // namespace
int bwd_launch_dir_chain(int precision, const char* stream, int64_t M, const char* masks, size_t ms, char* dlt, size_t ls, float* rows, hipStream_t st) {
    if (precision == NERF_AMD_BF16) return launch_dir_chain_t<PB16>(stream, M, masks, ms, dlt, ls, rows, st);
    return launch_dir_chain_t<PF32>(stream, M, masks, ms, dlt, ls, rows, st);
}
#if BWD_TU != 3
namespace {
#endif
#endif
#if BWD_TU == 2
}  // namespace
int bwd_launch_dir_chain(int precision, const char* stream, int64_t M, const char* masks, size_t ms, char* dlt, size_t ls, float* rows, hipStream_t st);
namespace {
#endif
#if BWD_TU != 1 && BWD_TU != 3
template <class P>
int launch_chain_t(int which, const char* stream, int64_t M, const char* act, char* dlt, size_t ls, size_t ms, float* rows, hipStream_t st) {
    // the forward's ReLU bit-mask records sit behind the activation slots of its dump (17 slots for Ref-NeRF, 5 for the proposal network)
    const char* masks = act + (size_t)(which == 3 ? PROP_DUMP_SLOTS : REF_DUMP_SLOTS) * ls;
    constexpr int TS = P::NW * P::NT * 32;
    const int64_t n_tiles = (M + TS - 1) / TS;
    if (n_tiles == 0) return 0;
    const size_t lds = bwd_lds_total<P>();
    const dim3 grid(bwd_grid(n_tiles)), block(P::NW * 64);
    switch (which) {
        case 0:                                                // (its own translation unit: see bwd_launch_dir_chain)
            return bwd_launch_dir_chain(P::PREC, stream, M, masks, ms, dlt, ls, rows, st);
        case 1:
            if (int e = nerf_host::allow_dynamic_lds(reinterpret_cast<const void*>(ref_spa_bwd_kernel<P>), lds)) return e;
            hipLaunchKernelGGL((ref_spa_bwd_kernel<P>), grid, block, lds, st, stream, M, masks, (unsigned long long)ms, dlt, (unsigned long long)ls);
            break;
        case 2:
            if (int e = nerf_host::allow_dynamic_lds(reinterpret_cast<const void*>(ref_chain9_kernel<P, true>), lds)) return e;
            hipLaunchKernelGGL((ref_chain9_kernel<P, true>), grid, block, lds, st, stream, M, masks, (unsigned long long)ms, (char*)nullptr, (unsigned long long)ls, rows);
            break;
        default:
            if (int e = nerf_host::allow_dynamic_lds(reinterpret_cast<const void*>(prop_density_chain_kernel<P>), lds)) return e;
            hipLaunchKernelGGL((prop_density_chain_kernel<P>), grid, block, lds, st, stream, M, masks, (unsigned long long)ms, rows);
    }
    return (int)hipGetLastError();
}
// which: 0 Ref-NeRF directional, 1 Ref-NeRF spatial, 2 Ref-NeRF density gradient, 3 proposal density gradient; `start_frag`: where the
// chain's stream begins in the blob
}  // namespace
size_t mlp_train_layer_stride(int precision, int64_t M);
size_t mlp_train_mask_stride(int precision, int64_t M);
int bwd_launch_chain(int which, int precision, const void* blob, int start_frag, int64_t M, const void* act, void* dlt, float* rows, hipStream_t st) {
    const size_t fb = precision == NERF_AMD_BF16 ? 1024 : 2048;
    const char* stream = reinterpret_cast<const char*>(blob) + (size_t)start_frag * fb;
    const size_t ls = mlp_train_layer_stride(precision, M), ms = mlp_train_mask_stride(precision, M);
    // the SPATIAL chain of Ref-NeRF on the 8-wave x 32-sample tile: like the training forwards, its delta stores cost issue slots that a second
    // wave per SIMD fills -- 3.61 -> 3.31 ms per 2^14-ray step, same box, alternated twice (profiles/r04_ref_chains_8wave_ab.log; the
    // directional chain, with 65 spilled registers, got slower on it and stays on the wide tile).  -DREF_SPA_CHAIN_WIDE = the A side.
    if (precision == NERF_AMD_BF16 && which == 1) return launch_chain_t<PBF16>(which, stream, M, reinterpret_cast<const char*>(act), reinterpret_cast<char*>(dlt), ls, ms, rows, st);
    if (precision == NERF_AMD_BF16) return launch_chain_t<PB16>(which, stream, M, reinterpret_cast<const char*>(act), reinterpret_cast<char*>(dlt), ls, ms, rows, st);
    return launch_chain_t<PF32>(which, stream, M, reinterpret_cast<const char*>(act), reinterpret_cast<char*>(dlt), ls, ms, rows, st);
}
#endif  // BWD_TU != 1 && != 3
#if BWD_TU != 2 && BWD_TU != 3
#if BWD_TU == 1
}  // namespace
int bwd_launch_chain(int which, int precision, const void* blob, int start_frag, int64_t M, const void* act, void* dlt, float* rows, hipStream_t st);
#endif
namespace {

template <class P, bool F8 = false>
int launch_prop_bwd(const void* packed, const float* g, int64_t M, Dump act, Dump dlt, hipStream_t st) {
    constexpr int TS = P::NW * P::NT * 32;
    const int64_t n_tiles = (M + TS - 1) / TS;
    if (n_tiles == 0) return 0;
    const size_t lds = F8 ? bwd_lds_total_f8<P>() : bwd_lds_total<P>();
    if (int e = nerf_host::allow_dynamic_lds(reinterpret_cast<const void*>(prop_bwd_kernel<P, F8>), lds)) return e;
    hipLaunchKernelGGL((prop_bwd_kernel<P, F8>), dim3(bwd_grid(n_tiles)), dim3(P::NW * 64), lds, st, packed, g, M, act, dlt);
    return (int)hipGetLastError();
}
template <class P, bool F8 = false>
int launch_mip_bwd(const void* packed, const float* g, const float* rgbo, int64_t M, Dump act, Dump dlt, hipStream_t st) {
    constexpr int TS = P::NW * P::NT * 32;
    const int64_t n_tiles = (M + TS - 1) / TS;
    if (n_tiles == 0) return 0;
    const size_t lds = F8 ? bwd_lds_total_f8<P>() : bwd_lds_total<P>();
    if (int e = nerf_host::allow_dynamic_lds(reinterpret_cast<const void*>(mip_bwd_kernel<P, F8>), lds)) return e;
    hipLaunchKernelGGL((mip_bwd_kernel<P, F8>), dim3(bwd_grid(n_tiles)), dim3(P::NW * 64), lds, st, packed, g, rgbo, M, act, dlt);
    return (int)hipGetLastError();
}

template <bool F8>
int launch_wgrad256(const WgradJobs& jobs, int n_jobs, int n_wg, int64_t n_sub, hipStream_t st) {
    if (int e = nerf_host::allow_dynamic_lds(reinterpret_cast<const void*>(wgrad256_kernel<F8>), 65536)) return e;
    hipLaunchKernelGGL((wgrad256_kernel<F8>), dim3(n_wg, n_jobs), dim3(512), 65536, st, jobs, n_sub);
    return (int)hipGetLastError();
}
template <int KGX, int KGY, int WO, int YKIND>
int launch_wgrad(int precision, const WgradJobs& jobs, int n_jobs, int n_wg, int64_t n_sub, hipStream_t st) {
    if (precision == NERF_AMD_BF16) hipLaunchKernelGGL((wgrad_kernel_bf16<KGX, KGY, WO, YKIND>), dim3(n_wg, n_jobs), dim3(256), 0, st, jobs, n_sub);
    else hipLaunchKernelGGL((wgrad_kernel_f32<KGX, KGY, WO, YKIND>), dim3(n_wg, n_jobs), dim3(256), 0, st, jobs, n_sub);
    return (int)hipGetLastError();
}
// fp8 dump operands (NERF_AMD_BF16_F8): XF8 / YF8 say which operand comes from an fp8 slot
template <int KGX, int KGY, int WO, int YKIND, bool XF8, bool YF8>
int launch_wgrad_f8(const WgradJobs& jobs, int n_jobs, int n_wg, int64_t n_sub, hipStream_t st) {
    hipLaunchKernelGGL((wgrad_kernel_bf16<KGX, KGY, WO, YKIND, XF8, YF8>), dim3(n_wg, n_jobs), dim3(256), 0, st, jobs, n_sub);
    return (int)hipGetLastError();
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host-visible launchers (capi.hip)
size_t mlp_train_layer_stride(int precision, int64_t M);
size_t mlp_train_mask_stride(int precision, int64_t M);

int bwd_launch_prop_chain(const void* packed_bwd, int precision, const float* g_density, int64_t M, const void* act_dump, void* delta_dump,
                          hipStream_t st) {
    const unsigned long long ls = mlp_train_layer_stride(precision, M), ms = mlp_train_mask_stride(precision, M);
    const Dump act{const_cast<char*>(reinterpret_cast<const char*>(act_dump)), ls, reinterpret_cast<const char*>(act_dump) + (size_t)PROP_DUMP_SLOTS * ls, ms},
               dlt{reinterpret_cast<char*>(delta_dump), ls, nullptr, 0ull};
    if (precision == NERF_AMD_BF16_F8) return launch_prop_bwd<PB16, true>(packed_bwd, g_density, M, act, dlt, st);   // (the chain reads only the mask bits of `act`)
    if (precision == NERF_AMD_BF16) return launch_prop_bwd<PB16>(packed_bwd, g_density, M, act, dlt, st);
    return launch_prop_bwd<PF32>(packed_bwd, g_density, M, act, dlt, st);
}
int bwd_launch_mip_chain(const void* packed_bwd, int precision, const float* g_rgbo, const float* rgbo, int64_t M, const void* act_dump,
                         void* delta_dump, hipStream_t st) {
    const unsigned long long ls = mlp_train_layer_stride(precision, M), ms = mlp_train_mask_stride(precision, M);
    const Dump act{const_cast<char*>(reinterpret_cast<const char*>(act_dump)), ls, reinterpret_cast<const char*>(act_dump) + (size_t)MIP_DUMP_SLOTS * ls, ms},
               dlt{reinterpret_cast<char*>(delta_dump), ls, nullptr, 0ull};
    if (precision == NERF_AMD_BF16_F8) return launch_mip_bwd<PB16, true>(packed_bwd, g_rgbo, rgbo, M, act, dlt, st);
    if (precision == NERF_AMD_BF16) return launch_mip_bwd<PB16>(packed_bwd, g_rgbo, rgbo, M, act, dlt, st);
    return launch_mip_bwd<PF32>(packed_bwd, g_rgbo, rgbo, M, act, dlt, st);
}

// ------------------------------------------------------------------------------------------------ weight gradients: orchestration
namespace {
struct Product { const char* x0; int kgx0; const char* x1; const char* y; float* partial; float* bias_partial; };

// shape codes: 0 = 256 x 256 (D map), 1 = 256 x PE10, 2 = (128 + head) x 256, 3 = head x 128, 4 = 128 x PE4, 5 = head x 256
// f8 (NERF_AMD_BF16_F8 dumps of the proposal / MipNeRF networks): the hidden delta / activation slots among the operands are fp8 slots
// (subtiles F8_SUB_BYTES apart); which operand of a shape is one follows from what the shape multiplies (see the callers)
int run_wgrad(int shape, int precision, const Product* prods, int n, int n_wg, int64_t n_sub, hipStream_t st, bool f8 = false) {
    if (n < 1 || n > WG_MAX_JOBS) return (int)hipErrorInvalidValue;
    const size_t breg = precision == NERF_AMD_BF16 ? 1024 : 2048;
    const bool xf8 = f8 && shape != 3 && shape != 5, yf8 = f8 && (shape == 0 || shape == 2 || shape == 3 || shape == 5);
    WgradJobs jobs = {};
    for (int i = 0; i < n; ++i) {
        WgradJob& J = jobs.j[i];
        J.x0 = FragMat{prods[i].x0, xf8 ? (size_t)F8_SUB_BYTES : 16 * breg};
        J.kgx0 = prods[i].kgx0;
        J.x1 = FragMat{prods[i].x1, 16 * breg};
        J.y = FragMat{prods[i].y, yf8 ? (size_t)F8_SUB_BYTES : 16 * breg};
        J.partial = prods[i].partial; J.bias_partial = prods[i].bias_partial;
    }
    if (f8) {
        if (precision != NERF_AMD_BF16) return (int)hipErrorInvalidValue;
        switch (shape) {
            case 0: return launch_wgrad256<true>(jobs, n, n_wg, n_sub, st);
            case 1: return launch_wgrad_f8<16, 4, 4, Y_PE10, true, false>(jobs, n, n_wg, n_sub, st);
            case 2: return launch_wgrad_f8<9, 16, 1, Y_DMAP, true, true>(jobs, n, n_wg, n_sub, st);
            case 3: return launch_wgrad_f8<1, 8, 1, Y_DMAP, false, true>(jobs, n, n_wg, n_sub, st);
            case 4: return launch_wgrad_f8<8, 2, 4, Y_PE4, true, false>(jobs, n, n_wg, n_sub, st);
            case 5: return launch_wgrad_f8<1, 16, 1, Y_DMAP, false, true>(jobs, n, n_wg, n_sub, st);
        }
        return (int)hipErrorInvalidValue;
    }
    switch (shape) {
        case 0: if (precision == NERF_AMD_BF16) return launch_wgrad256<false>(jobs, n, n_wg, n_sub, st);      // 4 x 2 waves of 2 x 4 blocks
                hipLaunchKernelGGL((wgrad_kernel_f32<16, 16, 2, Y_DMAP>), dim3(n_wg, n), dim3(256), 0, st, jobs, n_sub);   // fp32: 2 x 2 waves of 4 x 4 blocks
                return (int)hipGetLastError();
        case 1: return launch_wgrad<16, 4, 4, Y_PE10>(precision, jobs, n, n_wg, n_sub, st);      // 4 x 1 waves of 2 x 2 blocks
        case 2: return launch_wgrad<9, 16, 1, Y_DMAP>(precision, jobs, n, n_wg, n_sub, st);      // NOB 5 x NIB 8: 1 x 4 waves of 5 x 2 blocks
        case 3: return launch_wgrad<1, 8, 1, Y_DMAP>(precision, jobs, n, n_wg, n_sub, st);       // NOB 1 x NIB 4
        case 4: return launch_wgrad<8, 2, 4, Y_PE4>(precision, jobs, n, n_wg, n_sub, st);        // NOB 4 x NIB 1
        case 5: return launch_wgrad<1, 16, 1, Y_DMAP>(precision, jobs, n, n_wg, n_sub, st);      // NOB 1 x NIB 8
        case 6: return launch_wgrad<16, 8, 2, Y_DMAP>(precision, jobs, n, n_wg, n_sub, st);      // 256 x 128 (Ref-NeRF: delta x bottle-neck)
        case 7: return launch_wgrad<16, 3, 4, Y_IDE>(precision, jobs, n, n_wg, n_sub, st);       // 256 x [IDE 38 | n.d]
    }
    return (int)hipErrorInvalidValue;
}
int run_finalize(const FinalizeJob* f, int n, hipStream_t st) {
    if (n < 1 || n > FIN_MAX_JOBS) return (int)hipErrorInvalidValue;
    FinalizeJobs jobs = {};
    for (int i = 0; i < n; ++i) jobs.j[i] = f[i];
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(128, n), dim3(256), 0, st, jobs);
    return (int)hipGetLastError();
}
// workgroups per product of a launch that batches `n_jobs` products: the big-accumulator shapes (512 registers: one workgroup per CU)
// get one round of the chip, the light shapes two
int wgrad_workgroups(int64_t n_sub, int n_jobs, bool heavy) {
    const int64_t total = (int64_t)nerf_host::cu_count() * (heavy ? 1 : 2);
    int64_t n = total / n_jobs;
    if (n > n_sub) n = n_sub;
    return (int)(n < 1 ? 1 : n);
}
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
struct Carver {
    char* p;
    float* take(size_t floats) { float* r = reinterpret_cast<float*>(p); p += align256(floats * 4); return r; }
};
}  // namespace

int64_t bwd_n_sub(int precision, int64_t M) {
    const int64_t ts = (precision != NERF_AMD_F32) ? (int64_t)PB16::NW * PB16::NT * 32 : (int64_t)PF32::NW * PF32::NT * 32;
    return ((M + ts - 1) / ts) * (ts / 32);
}

// workspace: per-product workgroup partials (+ G for the un-fold)
size_t bwd_wgrad_workspace_bytes(int net, int precision, int64_t M) {
    const int64_t n_sub = bwd_n_sub(precision, M);
    if (net == NERF_AMD_NET_PROPOSAL) {
        const size_t w0 = wgrad_workgroups(n_sub, 3, true), w1 = wgrad_workgroups(n_sub, 1, false);
        return 3 * (align256(w0 * 256 * 256 * 4) + align256(w0 * 256 * 4)) + align256(w1 * 256 * 64 * 4) + align256(w1 * 256 * 4) +
               align256(w1 * 32 * 256 * 4) + align256(w1 * 32 * 4) + 256;
    }
    if (net == NERF_AMD_NET_MIP) {
        const size_t w0 = wgrad_workgroups(n_sub, 6, true), w1 = wgrad_workgroups(n_sub, 2, false), w2 = wgrad_workgroups(n_sub, 1, true),
                     w3 = wgrad_workgroups(n_sub, 1, false);
        return 6 * (align256(w0 * 256 * 256 * 4) + align256(w0 * 256 * 4)) + 2 * align256(w1 * 256 * 64 * 4) + align256(w1 * 256 * 4) +
               align256(w2 * 160 * 256 * 4) + align256(w2 * 160 * 4) + align256(w3 * 32 * 128 * 4) + align256(w3 * 128 * 32 * 4) +
               align256((size_t)128 * 256 * 4) + 256;
    }
    return 0;
}

// ProposalNetwork: d_w / d_b = gradients of layers.{0,2,4,6,8} (addtional.py:67-71), written in the reference's (out, in) layout
int bwd_prop_weight_grads(int precision, int64_t M, const void* act_dump, const void* delta_dump, float* const* d_w, float* const* d_b,
                          void* workspace, hipStream_t st) {
    const bool f8 = precision == NERF_AMD_BF16_F8;          // hidden slots of both dumps in scaled e4m3 (arithmetic: bf16)
    if (f8) precision = NERF_AMD_BF16;
    const int64_t n_sub = bwd_n_sub(precision, M);
    if (n_sub == 0) return 0;
    const size_t breg = precision == NERF_AMD_BF16 ? 1024 : 2048, ls = mlp_train_layer_stride(precision, M);
    const char* act = reinterpret_cast<const char*>(act_dump);
    const char* dlt = reinterpret_cast<const char*>(delta_dump);
    auto A = [&](int slot, int kg0 = 0) { return act + (size_t)slot * ls + (size_t)kg0 * breg; };
    auto D = [&](int slot, int kg0 = 0) { return dlt + (size_t)slot * ls + (size_t)kg0 * breg; };
    const int w0 = wgrad_workgroups(n_sub, 3, true), w1 = wgrad_workgroups(n_sub, 1, false);
    Carver ws{reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255)};
    float *pw[3], *pb[3];
    for (int i = 0; i < 3; ++i) { pw[i] = ws.take((size_t)w0 * 256 * 256); pb[i] = ws.take((size_t)w0 * 256); }
    float* pe = ws.take((size_t)w1 * 256 * 64); float* pe_b = ws.take((size_t)w1 * 256);
    float* ph = ws.take((size_t)w1 * 32 * 256); float* ph_b = ws.take((size_t)w1 * 32);
    // layers.{2,4,6}: delta_L^T y_{L-1}
    Product p0[3];
    for (int i = 0; i < 3; ++i) p0[i] = Product{D(i + 1), 16, nullptr, A(i), pw[i], pb[i]};
    if (int e = run_wgrad(0, precision, p0, 3, w0, n_sub, st, f8)) return e;
    const Product p1{D(0), 16, nullptr, A(4), pe, pe_b};                   // layers.0: delta_0^T [x | PE10(x)]
    if (int e = run_wgrad(1, precision, &p1, 1, w1, n_sub, st, f8)) return e;
    const Product p5{D(4), 1, nullptr, A(3), ph, ph_b};                    // layers.8: g^T y_3 (slot feature 0 of the head K group)
    if (int e = run_wgrad(5, precision, &p5, 1, w1, n_sub, st, f8)) return e;
    FinalizeJob f[5];
    for (int i = 0; i < 3; ++i) f[i] = FinalizeJob{pw[i], 256, 256, d_w[i + 1], 256, 0, 0, 256, 256, pb[i], d_b[i + 1], 256, 0, 256, w0};
    f[3] = FinalizeJob{pe, 256, 64, d_w[0], 63, 0, 0, 256, 63, pe_b, d_b[0], 256, 0, 256, w1};
    f[4] = FinalizeJob{ph, 32, 256, d_w[4], 256, 0, 0, 1, 256, ph_b, d_b[4], 32, 0, 1, w1};
    return run_finalize(f, 5, st);
}

// MipNeRF: tensors in _linear_layers() order (0..3 lin_block1, 4..6 lin_block2, 7 bottle_neck.0, 8 opacity_head.0, 9, 10 rgb_layer.{0,2})
int bwd_mip_weight_grads(int precision, int64_t M, const void* act_dump, const void* delta_dump, const float* const* w, const float* const* b,
                         float* const* d_w, float* const* d_b, void* workspace, hipStream_t st) {
    const bool f8 = precision == NERF_AMD_BF16_F8;          // hidden slots of both dumps in scaled e4m3 (arithmetic: bf16)
    if (f8) precision = NERF_AMD_BF16;
    const int64_t n_sub = bwd_n_sub(precision, M);
    if (n_sub == 0) return 0;
    const size_t breg = precision == NERF_AMD_BF16 ? 1024 : 2048, ls = mlp_train_layer_stride(precision, M);
    const char* act = reinterpret_cast<const char*>(act_dump);
    const char* dlt = reinterpret_cast<const char*>(delta_dump);
    auto A = [&](int slot, int kg0 = 0) { return act + (size_t)slot * ls + (size_t)kg0 * breg; };
    auto D = [&](int slot, int kg0 = 0) { return dlt + (size_t)slot * ls + (size_t)kg0 * breg; };
    const int w0 = wgrad_workgroups(n_sub, 6, true), w1 = wgrad_workgroups(n_sub, 2, false), w2 = wgrad_workgroups(n_sub, 1, true),
              w3 = wgrad_workgroups(n_sub, 1, false);
    Carver ws{reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255)};
    float *pw[6], *pb[6];
    for (int i = 0; i < 6; ++i) { pw[i] = ws.take((size_t)w0 * 256 * 256); pb[i] = ws.take((size_t)w0 * 256); }
    float* pe0 = ws.take((size_t)w1 * 256 * 64); float* pe4 = ws.take((size_t)w1 * 256 * 64); float* pe0_b = ws.take((size_t)w1 * 256);
    float* pg = ws.take((size_t)w2 * 160 * 256); float* pg_b = ws.take((size_t)w2 * 160);
    float* phc = ws.take((size_t)w3 * 32 * 128);
    float* pcd = ws.take((size_t)w3 * 128 * 32);
    float* G = ws.take((size_t)128 * 256);
    // the six 256 x 256 products in one launch: layer L = delta_L^T y_{L-1} for L = 1, 2, 3, 5, 6 and the hidden columns of the skip layer 4
    const int Ls[6] = {1, 2, 3, 4, 5, 6};
    Product p0[6];
    for (int i = 0; i < 6; ++i) p0[i] = Product{D(Ls[i]), 16, nullptr, A(Ls[i] - 1), pw[i], pb[i]};
    if (int e = run_wgrad(0, precision, p0, 6, w0, n_sub, st, f8)) return e;
    // lin_block1.0 and the encoding columns of lin_block2.0: delta^T [x | PE10(x)]
    const Product p1[2] = {Product{D(0), 16, nullptr, A(8), pe0, pe0_b}, Product{D(4), 16, nullptr, A(8), pe4, nullptr}};
    if (int e = run_wgrad(1, precision, p1, 2, w1, n_sub, st, f8)) return e;
    // heads: [dc | dpre dsigma]^T g6 -> G (rows 0..127), rows 128..130 unused, row 131 = opacity_head.0
    const Product p2{D(7), 8, D(8), A(6), pg, pg_b};
    if (int e = run_wgrad(2, precision, &p2, 1, w2, n_sub, st, f8)) return e;
    const Product p3{D(8), 1, nullptr, A(7), phc, nullptr};                // rgb_layer.2: dpre^T c
    if (int e = run_wgrad(3, precision, &p3, 1, w3, n_sub, st, f8)) return e;
    const Product p4{D(7), 8, nullptr, A(8, 4), pcd, nullptr};             // rgb_layer.0's direction columns: dc^T [d | PE4(d)]
    if (int e = run_wgrad(4, precision, &p4, 1, w3, n_sub, st, f8)) return e;
    FinalizeJob f[13];
    int n = 0;
    for (int i = 0; i < 6; ++i) {
        const int L = Ls[i];
        if (L == 4) f[n++] = FinalizeJob{pw[i], 256, 256, d_w[4], 319, 63, 0, 256, 256, pb[i], d_b[4], 256, 0, 256, w0};
        else f[n++] = FinalizeJob{pw[i], 256, 256, d_w[L], 256, 0, 0, 256, 256, pb[i], d_b[L], 256, 0, 256, w0};
    }
    f[n++] = FinalizeJob{pe4, 256, 64, d_w[4], 319, 0, 0, 256, 63, nullptr, nullptr, 0, 0, 0, w1};
    f[n++] = FinalizeJob{pe0, 256, 64, d_w[0], 63, 0, 0, 256, 63, pe0_b, d_b[0], 256, 0, 256, w1};
    f[n++] = FinalizeJob{pg, 160, 256, G, 256, 0, 0, 128, 256, pg_b, d_b[9], 160, 0, 128, w2};            // G and db9 = sum dc
    f[n++] = FinalizeJob{pg, 160, 256, d_w[8], 256, 0, 131, 1, 256, pg_b, d_b[8], 160, 131, 1, w2};       // opacity_head.0
    f[n++] = FinalizeJob{phc, 32, 128, d_w[10], 128, 0, 0, 3, 128, nullptr, nullptr, 0, 0, 0, w3};        // rgb_layer.2
    f[n++] = FinalizeJob{pg, 160, 256, G, 256, 0, 0, 0, 256, pg_b, d_b[10], 160, 128, 3, w2};             // its bias: sum dpre (no matrix rows)
    f[n++] = FinalizeJob{pcd, 128, 32, d_w[9], 283, 256, 0, 128, 27, nullptr, nullptr, 0, 0, 0, w3};
    if (int e = run_finalize(f, n, st)) return e;
    // bottle_neck.0 and the folded columns of rgb_layer.0 from G (mip_fold_grads_kernel)
    const int total = 128 * 256 + 256 * 256 + 256;
    hipLaunchKernelGGL(mip_fold_grads_kernel, dim3((total + 255) / 256), dim3(256), 0, st, G, d_b[9], w[9], w[7], b[7], d_w[9], d_w[7], d_b[7]);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ Ref-NeRF backward: host side
namespace {
struct ChainCtx {                     // addressing of fragment dumps: slot l, K group kg
    int precision; int64_t M; size_t breg, ls, sub;
    ChainCtx(int prec, int64_t m) : precision(prec), M(m), breg(prec == NERF_AMD_BF16 ? 1024 : 2048), ls(mlp_train_layer_stride(prec, m)), sub(16 * breg) {}
    const char* at(const void* dump, int slot, int kg = 0) const { return reinterpret_cast<const char*>(dump) + (size_t)slot * ls + (size_t)kg * breg; }
    char* at(void* dump, int slot, int kg = 0) const { return reinterpret_cast<char*>(dump) + (size_t)slot * ls + (size_t)kg * breg; }
};
int blocks_1d(int64_t work) { int64_t b = (work + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

// d density / d position, times scale[m] (RefNeRF.get_grad, ref_model.py:119-125; train.py:165-168,178) for the proposal network
// (net 0: activation slots 0..3) and Ref-NeRF's spatial network (net 2: slots 0..7): a dgrad-only chain from the density row down to
// the encoded position, then the encoding's derivative.  workspace: two delta buffers of one slot each + d_enc rows (M, 64) fp32.
size_t bwd_density_grad_workspace_bytes(int precision, int64_t M) {
    return align256((size_t)M * 64 * 4) + 256;
}
int bwd_density_grad(int net, const void* blob, int precision, int64_t M, const void* act, const float* x, int x_stride, const float* scale, int scale_stride,
                     float* out, void* workspace, hipStream_t st, int contract) {
    if (M == 0) return 0;
    const ChainCtx c(precision, M);
    Carver ws{reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255)};
    float* denc = ws.take((size_t)M * 64);
    if (net == NERF_AMD_NET_PROPOSAL) { if (int e = bwd_launch_chain(3, precision, blob, 0, M, act, nullptr, denc, st)) return e; }
    else if (int e = bwd_launch_chain(2, precision, blob, RefBwdLayout::DEN_START, M, act, nullptr, denc, st)) return e;
    if (contract) hipLaunchKernelGGL(pe_grad_contract_kernel, dim3(blocks_1d(M)), dim3(256), 0, st, denc, 64, x, x_stride, scale, scale_stride, M, 10, out);
    else hipLaunchKernelGGL(pe_grad_kernel, dim3(blocks_1d(M * 3)), dim3(256), 0, st, denc, 64, x, x_stride, scale, scale_stride, M, 10, out);
    return (int)hipGetLastError();
}

// Ref-NeRF parameter backward (what autograd computes for ref_model.py:68-106 inside train.py:176-199).
// workspace: delta dump (REF_DUMP_SLOTS slots) | d_allin rows (M,192) | wgrad partials
size_t bwd_ref_workspace_bytes(int precision, int64_t M) {
    const int64_t n_sub = bwd_n_sub(precision, M);
    const size_t w7 = wgrad_workgroups(n_sub, 7, true), w2 = wgrad_workgroups(n_sub, 2, false), w1h = wgrad_workgroups(n_sub, 1, true),
                 w1 = wgrad_workgroups(n_sub, 1, false);
    size_t b = align256((size_t)REF_DUMP_SLOTS * mlp_train_layer_stride(precision, M)) + align256((size_t)M * 192 * 4);
    b += 14 * (align256(w7 * 256 * 256 * 4) + align256(w7 * 256 * 4));                       // 2 x 7 hidden products
    b += 2 * align256(w2 * 256 * 64 * 4) + align256(w2 * 256 * 4);                           // encoding columns
    b += 2 * (align256(w2 * 256 * 128 * 4) + align256(w2 * 256 * 4)) + 2 * align256(w2 * 256 * 64 * 4);   // x bottle-neck, x IDE
    b += align256(w1h * 160 * 256 * 4) + align256(w1h * 160 * 4) + align256(w1 * 32 * 256 * 4) + align256(w1 * 32 * 4);
    return b + 512;
}
// w: the 20 tensors of nerf_amd_pack_weights(NET_REF) (only the ide_table, index 19, is read here).  d_w / d_b (20 each): 0..7 spatial,
// 8 bottle_neck, 9 norm_col_tint_head (9 rows), 10 rho_tau_head (2 rows), 11..18 directional, 19 spec_rgb_head.0
int bwd_ref_backward(const void* blob, int precision, int64_t M, const void* act, const float* aux, const float* dirs, int dir_stride,
                     const float* g_out, int g_stride, const float* ide_table, float* const* d_w, float* const* d_b, void* workspace, int flags, hipStream_t st) {
    if (M == 0) return 0;
    using L = RefBwdLayout;
    const int srgb = (flags & NERF_AMD_REF_SRGB) ? 1 : 0;
    const ChainCtx c(precision, M);
    const int64_t n_sub = bwd_n_sub(precision, M);
    Carver ws{reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255)};
    char* dlt = reinterpret_cast<char*>(ws.take((size_t)REF_DUMP_SLOTS * c.ls / 4));
    float* dallin = ws.take((size_t)M * 192);                // (M, 192): the 6 feature blocks of the 167-wide input vector
    const int elem = precision == NERF_AMD_BF16 ? 2 : 4;
    auto A = [&](int slot, int kg = 0) { return c.at(act, slot, kg); };
    auto D = [&](int slot, int kg = 0) { return c.at((void*)dlt, slot, kg); };
    // slot 8 of the delta dump: K groups 0..7 delta of the bottle-neck, 8 the head rows, 9 the spec head -- each written COMPLETELY by the two
    // element-wise kernels below (unused features and the padding samples up to the tile boundary as zeros): no memset
    const int64_t Mpad = (int64_t)(c.ls / c.sub) * 32;
    // stage 1: spec head delta, then the directional network backwards
    if (elem == 2) hipLaunchKernelGGL(ref_spec_delta_kernel<2>, dim3(blocks_1d(Mpad)), dim3(256), 0, st, g_out, g_stride, aux, M, Mpad, D(8, 9), (unsigned long long)c.sub, srgb);
    else hipLaunchKernelGGL(ref_spec_delta_kernel<4>, dim3(blocks_1d(Mpad)), dim3(256), 0, st, g_out, g_stride, aux, M, Mpad, D(8, 9), (unsigned long long)c.sub, srgb);
    if (int e = bwd_launch_chain(0, precision, blob, L::DIR_START, M, act, dlt, dallin, st)) return e;                            // D7 .. D0 + the input-vector columns
    // stage 2: IDE / reflection / normal / head activations backwards -> delta of the heads and of the bottle-neck
    if (elem == 2) hipLaunchKernelGGL(ref_heads_delta_kernel<2>, dim3(blocks_1d(Mpad)), dim3(256), 0, st, g_out, g_stride, aux, dallin, 192, dirs, dir_stride, ide_table, M, Mpad,
                                      D(8), (unsigned long long)c.sub, srgb);
    else hipLaunchKernelGGL(ref_heads_delta_kernel<4>, dim3(blocks_1d(Mpad)), dim3(256), 0, st, g_out, g_stride, aux, dallin, 192, dirs, dir_stride, ide_table, M, Mpad, D(8),
                            (unsigned long long)c.sub, srgb);
    // stage 3: the spatial network backwards
    if (int e = bwd_launch_chain(1, precision, blob, L::SPA_START, M, act, dlt, nullptr, st)) return e;                           // S7 .. S0
    // weight gradients
    const int w7 = wgrad_workgroups(n_sub, 7, true), w2 = wgrad_workgroups(n_sub, 2, false), w1h = wgrad_workgroups(n_sub, 1, true),
              w1 = wgrad_workgroups(n_sub, 1, false);
    float *pS[7], *pSb[7], *pD[7], *pDb[7];
    for (int i = 0; i < 7; ++i) { pS[i] = ws.take((size_t)w7 * 65536); pSb[i] = ws.take((size_t)w7 * 256); }
    for (int i = 0; i < 7; ++i) { pD[i] = ws.take((size_t)w7 * 65536); pDb[i] = ws.take((size_t)w7 * 256); }
    float* pe0 = ws.take((size_t)w2 * 256 * 64); float* pe4 = ws.take((size_t)w2 * 256 * 64); float* pe0b = ws.take((size_t)w2 * 256);
    float *pb0 = ws.take((size_t)w2 * 256 * 128), *pb0b = ws.take((size_t)w2 * 256), *pb4 = ws.take((size_t)w2 * 256 * 128), *pb4b = ws.take((size_t)w2 * 256);
    float *pi0 = ws.take((size_t)w2 * 256 * 64), *pi4 = ws.take((size_t)w2 * 256 * 64);
    float *ph = ws.take((size_t)w1h * 160 * 256), *phb = ws.take((size_t)w1h * 160), *psp = ws.take((size_t)w1 * 32 * 256), *pspb = ws.take((size_t)w1 * 32);
    const int SL[7] = {1, 2, 3, 4, 5, 6, 7};                 // spatial layer l: delta slot l x activation slot l-1 (layer 4: its hidden columns)
    Product pj[7];
    for (int i = 0; i < 7; ++i) pj[i] = Product{D(SL[i]), 16, nullptr, A(SL[i] - 1), pS[i], pSb[i]};
    if (int e = run_wgrad(0, precision, pj, 7, w7, n_sub, st)) return e;
    for (int i = 0; i < 7; ++i) pj[i] = Product{D(9 + SL[i]), 16, nullptr, A(9 + SL[i] - 1), pD[i], pDb[i]};                  // directional layers 1..7
    if (int e = run_wgrad(0, precision, pj, 7, w7, n_sub, st)) return e;
    const Product pe[2] = {Product{D(0), 16, nullptr, A(8, 11), pe0, pe0b}, Product{D(4), 16, nullptr, A(8, 11), pe4, nullptr}};
    if (int e = run_wgrad(1, precision, pe, 2, w2, n_sub, st)) return e;
    const Product pb[2] = {Product{D(9), 16, nullptr, A(8, 0), pb0, pb0b}, Product{D(13), 16, nullptr, A(8, 0), pb4, pb4b}};  // x bottle-neck
    if (int e = run_wgrad(6, precision, pb, 2, w2, n_sub, st)) return e;
    const Product pi[2] = {Product{D(9), 16, nullptr, A(8, 8), pi0, nullptr}, Product{D(13), 16, nullptr, A(8, 8), pi4, nullptr}};   // x [IDE | n.d]
    if (int e = run_wgrad(7, precision, pi, 2, w2, n_sub, st)) return e;
    const Product phd{D(8), 9, nullptr, A(7), ph, phb};                                                                      // [bottle-neck | heads]^T S7
    if (int e = run_wgrad(2, precision, &phd, 1, w1h, n_sub, st)) return e;
    const Product pspec{D(8, 9), 1, nullptr, A(16), psp, pspb};                                                              // spec head
    if (int e = run_wgrad(5, precision, &pspec, 1, w1, n_sub, st)) return e;
    FinalizeJob f[40];
    int n = 0;
    for (int i = 0; i < 7; ++i) {                            // spatial
        const int l = SL[i];
        if (l == 4) f[n++] = FinalizeJob{pS[i], 256, 256, d_w[4], 319, 63, 0, 256, 256, pSb[i], d_b[4], 256, 0, 256, w7};
        else f[n++] = FinalizeJob{pS[i], 256, 256, d_w[l], 256, 0, 0, 256, 256, pSb[i], d_b[l], 256, 0, 256, w7};
    }
    f[n++] = FinalizeJob{pe0, 256, 64, d_w[0], 63, 0, 0, 256, 63, pe0b, d_b[0], 256, 0, 256, w2};
    f[n++] = FinalizeJob{pe4, 256, 64, d_w[4], 319, 0, 0, 256, 63, nullptr, nullptr, 0, 0, 0, w2};
    for (int i = 0; i < 7; ++i) {                            // directional: tensors 11..18 = dir layers 0..7
        const int l = SL[i];
        if (l == 4) f[n++] = FinalizeJob{pD[i], 256, 256, d_w[15], 423, 167, 0, 256, 256, pDb[i], d_b[15], 256, 0, 256, w7};
        else f[n++] = FinalizeJob{pD[i], 256, 256, d_w[11 + l], 256, 0, 0, 256, 256, pDb[i], d_b[11 + l], 256, 0, 256, w7};
    }
    f[n++] = FinalizeJob{pb0, 256, 128, d_w[11], 167, 0, 0, 256, 128, pb0b, d_b[11], 256, 0, 256, w2};
    f[n++] = FinalizeJob{pi0, 256, 64, d_w[11], 167, 128, 0, 256, 39, nullptr, nullptr, 0, 0, 0, w2};
    f[n++] = FinalizeJob{pb4, 256, 128, d_w[15], 423, 0, 0, 256, 128, nullptr, nullptr, 0, 0, 0, w2};
    f[n++] = FinalizeJob{pi4, 256, 64, d_w[15], 423, 128, 0, 256, 39, nullptr, nullptr, 0, 0, 0, w2};
    if (int e = run_finalize(f, n, st)) return e;
    n = 0;
    f[n++] = FinalizeJob{ph, 160, 256, d_w[8], 256, 0, 0, 128, 256, phb, d_b[8], 160, 0, 128, w1h};                          // bottle_neck
    f[n++] = FinalizeJob{ph, 160, 256, d_w[9], 256, 0, 128, 3, 256, phb, d_b[9], 160, 128, 3, w1h};                          // norm_col_tint rows 0-2 (normal)
    f[n++] = FinalizeJob{ph, 160, 256, d_w[9] + 3 * 256, 256, 0, 132, 3, 256, phb, d_b[9] + 3, 160, 132, 3, w1h};            //   rows 3-5 (diffuse)
    f[n++] = FinalizeJob{ph, 160, 256, d_w[9] + 6 * 256, 256, 0, 136, 3, 256, phb, d_b[9] + 6, 160, 136, 3, w1h};            //   rows 6-8 (tint)
    f[n++] = FinalizeJob{ph, 160, 256, d_w[10], 256, 0, 131, 1, 256, phb, d_b[10], 160, 131, 1, w1h};                        // rho_tau row 0 (roughness)
    f[n++] = FinalizeJob{ph, 160, 256, d_w[10] + 256, 256, 0, 135, 1, 256, phb, d_b[10] + 1, 160, 135, 1, w1h};              //   row 1 (density)
    f[n++] = FinalizeJob{psp, 32, 256, d_w[19], 256, 0, 0, 3, 256, pspb, d_b[19], 32, 0, 3, w1};                             // spec_rgb_head.0
    return run_finalize(f, n, st);
}

int bwd_launch_adam(float* const* p, const float* const* g, float* const* m, float* const* v, const long long* n, int count, float* step, double lr,
                    const double* lr_dev, double beta1, double beta2, double eps, float grad_scale, hipStream_t st) {
    hipLaunchKernelGGL(adam_step_kernel, dim3(1), dim3(1), 0, st, step);
    for (int base = 0; base < count; base += ADAM_MAX) {
        AdamTable tab = {};
        const int c = (count - base < ADAM_MAX) ? count - base : ADAM_MAX;
        for (int i = 0; i < c; ++i) tab.t[i] = AdamTensor{p[base + i], g[base + i], m[base + i], v[base + i], n[base + i]};
        hipLaunchKernelGGL(adam_kernel, dim3(32, c), dim3(256), 0, st, tab, step, lr, lr_dev, beta1, beta2, eps, grad_scale);
    }
    return (int)hipGetLastError();
}
#endif  // BWD_TU != 2 && != 3
