// Shared machinery of the fused MLP kernels (forward: mlp_kernels.hip, backward chain: bwd_kernels.hip): precision policies, the
// L2 -> LDS weight stream with its chunk protocol, and `dense<>` -- one layer for a wavefront's NT x 32 samples with deferred epilogues.
// Design notes: the header comment of mlp_kernels.hip and DESIGN.md section 3.
#pragma once
#include "device_common.h"
#include "host_common.h"
#include "mlp_layout.h"

extern __shared__ __attribute__((aligned(16))) char smem[];

// (The round-1..4 timing ablations -- builds without epilogues / encodings / barriers / LDS re-reads / ring refills / dump stores /
// mask bits -- are retired: their results are in DESIGN.md section 3.2 and profiles/r0[1-4]_*; the switches left the sources in round 5.)
// MLP_CLOCKPROBE: workgroup 0 overwrites output record 0 with (shader cycles, 100 MHz ticks) -- scripts/gpu_clockprobe.sh
namespace {


// ------------------------------------------------------------------------------------------------
// precision policies
// ------------------------------------------------------------------------------------------------
struct PBF16 {
    using BReg = bf16x8;                       // one 16-feature K group of the B operand (4 VGPRs)
    static constexpr int PREC = NERF_AMD_BF16;
    static constexpr int NW = MLP_NW_BF16;     // wavefronts per workgroup
    static constexpr int NT = 1;               // 32-sample MFMA column tiles per wavefront
    static constexpr int FRAG_BYTES = 1024;    // one A fragment: 32 rows x 16 k, bf16
    static constexpr int FPC = MLP_CHUNK_BYTES / FRAG_BYTES;
    using AReg = bf16x8;                       // one A fragment per lane (4 VGPRs)
    static constexpr int DEPTH = 4;               // A fragments prefetched LDS -> VGPR ahead of their MFMA
    static DEVINL AReg load_a(uint32_t frag_addr) { return *reinterpret_cast<const bf16x8*>(smem + frag_addr); }
    static DEVINL f32x16 mma(const AReg& a, const BReg& b, f32x16 acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    // chain position: 0 first (C = bias in VGPRs), 1 middle, 2 last, 4 first with C = 0
    template <int POS>
    static DEVINL f32x16 mma_pos(const AReg& a, const BReg& b, f32x16 acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    static DEVINL void set(BReg& r, int e, float v) { r[e] = (__bf16)v; }
    // 8 accumulators -> one B register group.  ReLU is applied AFTER the bf16 conversion as a packed signed-int16
    // max with 0 (negative floats have the sign bit set): 8 v_cvt_pk + 4 v_pk_max_i16 instead of 8 v_max + ... per group
    template <bool RELU>
    static DEVINL BReg from_acc(const f32x16& acc, int off) {
        // pairwise vector conversion: one v_cvt_pk_bf16_f32 per two values (element-wise casts compile to a single-lane
        // convert each plus a v_perm_b32 to merge the halves -- three instructions instead of one)
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        BReg r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 v = {acc[off + 2 * e], acc[off + 2 * e + 1]};
            const bf16x2 p = __builtin_convertvector(v, bf16x2);
            r[2 * e] = p[0]; r[2 * e + 1] = p[1];
        }
        if (RELU) {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 v = __builtin_bit_cast(s16x8, r);
            const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            v = __builtin_elementwise_max(v, zero);
            r = __builtin_bit_cast(BReg, v);
        }
        return r;
    }
    static constexpr bool FAST_PE = true;      // octaves by angle doubling (error << bf16 ulp)
    // per-lane LDS stash of one B register group (lane-linear 16-byte slots: conflict-free)
    static constexpr int BREG_LDS = 1024;
    static DEVINL void stash(uint32_t addr, const BReg& r) { *reinterpret_cast<bf16x8*>(smem + addr) = r; }
    static DEVINL BReg unstash(uint32_t addr) { return *reinterpret_cast<const bf16x8*>(smem + addr); }
    // training dump: one B register group of a subtile = a 64 x BREG_LDS/64-byte block in fragment order (lane-linear, coalesced)
    // The training dumps are written once and read a whole pass later, while the weight stream lives in the same L2: non-temporal
    // stores (same-box A/B at 16 384 rays: training forwards -6 %, dgrad chains -3...-6 %, the weight-gradient kernels that read the
    // dumps next -2.5 %; -0.38 ms per step).
    static DEVINL void store_global(char* block, int lane, const BReg& r) {
        __builtin_nontemporal_store(__builtin_bit_cast(f32x4, r), reinterpret_cast<f32x4*>(block + lane * 16));
    }
    static DEVINL BReg load_global(const char* block, int lane) { return *reinterpret_cast<const bf16x8*>(block + lane * 16); }
};

// bf16, wide tile: 4 wavefronts x 64 samples.  Every A fragment read from LDS feeds TWO MFMAs (one per 32-sample column
// tile), which halves the LDS->VGPR traffic per flop -- the limiter of the 32-sample tile.
// One wavefront per SIMD with the 512-register budget; the activations of both column tiles stay in registers.
struct PBF16W : PBF16 {
    static constexpr int NW = 4;
    static constexpr int NT = 2;
};

// bf16, narrow networks (hidden width 128: PropLayout128): 4 wavefronts x 96 samples.  Half the K groups per layer leave room for THREE
// column tiles per wave (2 x 3 x 8 activation register groups = 192 VGPRs, 96 accumulators: 480 registers with everything else,
// spill-free), so every A fragment read from LDS feeds three MFMAs and a chunk of the weight ring lasts 24 MFMAs instead of 16.
// (Four column tiles crash hipcc's AGPR-copy rewrite pass under -amdgpu-mfma-vgpr-form and spill 57 registers without it.)
struct PBF16N : PBF16 {
    static constexpr int NW = 4;
    static constexpr int NT = 3;
};

struct PF32 {
    using BReg = f32x8;                        // 8 VGPRs per 16-feature K group
    static constexpr int PREC = NERF_AMD_F32;
    static constexpr int NW = MLP_NW_F32;
    static constexpr int NT = 1;
    static constexpr int FRAG_BYTES = 2048;    // [2 halves][64 lanes][4 floats]
    static constexpr int FPC = MLP_CHUNK_BYTES / FRAG_BYTES;
    struct AReg { f32x4 lo, hi; };
    static constexpr int DEPTH = 2;
    static DEVINL AReg load_a(uint32_t frag_addr) {
        AReg a;
        a.lo = *reinterpret_cast<const f32x4*>(smem + frag_addr);
        a.hi = *reinterpret_cast<const f32x4*>(smem + frag_addr + 1024);
        return a;
    }
    static DEVINL f32x16 mma(const AReg& a, const BReg& b, f32x16 acc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[e], b[e], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[e], b[4 + e], acc, 0, 0, 0);
        return acc;
    }
    template <int POS>
    static DEVINL f32x16 mma_pos(const AReg& a, const BReg& b, f32x16 acc) { return mma(a, b, acc); }
    static DEVINL void set(BReg& r, int e, float v) { r[e] = v; }
    template <bool RELU>
    static DEVINL BReg from_acc(const f32x16& acc, int off) {
        BReg r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = RELU ? fmaxf(acc[off + e], 0.0f) : acc[off + e];
        return r;
    }
    static constexpr bool FAST_PE = false;     // parity mode: every octave through the exact range reduction
    static constexpr int BREG_LDS = 2048;
    static DEVINL void stash(uint32_t addr, const BReg& r) {
        f32x4 lo = {r[0], r[1], r[2], r[3]}, hi = {r[4], r[5], r[6], r[7]};
        *reinterpret_cast<f32x4*>(smem + addr) = lo;
        *reinterpret_cast<f32x4*>(smem + addr + 1024) = hi;
    }
    static DEVINL BReg unstash(uint32_t addr) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + addr), hi = *reinterpret_cast<const f32x4*>(smem + addr + 1024);
        BReg r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return r;
    }
    static DEVINL void store_global(char* block, int lane, const BReg& r) {
        f32x4 lo = {r[0], r[1], r[2], r[3]}, hi = {r[4], r[5], r[6], r[7]};
        __builtin_nontemporal_store(lo, reinterpret_cast<f32x4*>(block + lane * 16));
        __builtin_nontemporal_store(hi, reinterpret_cast<f32x4*>(block + 1024 + lane * 16));
    }
    static DEVINL BReg load_global(const char* block, int lane) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(block + lane * 16), hi = *reinterpret_cast<const f32x4*>(block + 1024 + lane * 16);
        BReg r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return r;
    }
};

// ------------------------------------------------------------------------------------------------
// fp8 training dumps (mlp_layout.h F8_SUB_BYTES): scaled e4m3 <-> bf16 B register groups.
// gfx950's scaled conversions divide by / multiply with a float scale (only powers of two are used: exact); an out-of-range value
// encodes as NaN (no saturation), so the scale is taken from the group's largest magnitude: 2^(e_max - 7) maps it into [128, 256) <= 448.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

// biased scale exponent E of one B register group; SIGNED = false for post-ReLU activations (no sign bits to clear)
template <bool SIGNED>
DEVINL uint32_t f8_group_exponent(const bf16x8& v) {
    const u32x4 d = __builtin_bit_cast(u32x4, v);
    uint32_t a = d[0], b = d[1], c = d[2], e = d[3];
    if (SIGNED) { a &= 0x7fff7fffu; b &= 0x7fff7fffu; c &= 0x7fff7fffu; e &= 0x7fff7fffu; }
    const u16x2 m01 = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));      // bf16 magnitudes order like uint16
    const u16x2 m23 = __builtin_elementwise_max(__builtin_bit_cast(u16x2, c), __builtin_bit_cast(u16x2, e));
    const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(m01, m23));
    const uint32_t top = (m & 0xffffu) > (m >> 16) ? (m & 0xffffu) : (m >> 16);
    const uint32_t ex = top >> 7;                                     // the bf16 exponent field of the largest magnitude
    return ex > 8u ? ex - 7u : 1u;
}
DEVINL u32x2 f8_encode_group(const bf16x8& v, uint32_t E) {
    const float scale = __builtin_bit_cast(float, E << 23);
    const u32x4 d = __builtin_bit_cast(u32x4, v);
    s16x2 lo = {0, 0}, hi = {0, 0};
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2, (uint32_t)d[0]), scale, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2, (uint32_t)d[1]), scale, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2, (uint32_t)d[2]), scale, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2, (uint32_t)d[3]), scale, true);
    u32x2 r = {__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
    return r;
}
DEVINL bf16x8 f8_decode_group(uint32_t lo, uint32_t hi, uint32_t E) {
    const float scale = __builtin_bit_cast(float, E << 23);
    const bf16x2 p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, scale, false), p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, scale, true);
    const bf16x2 p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, scale, false), p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, scale, true);
    bf16x8 r = {p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
    return r;
}
// write one K group of a subtile's fp8 slot: 8 bytes of data to global memory (non-temporal like the bf16 dumps), the scale exponent
// into the wave's LDS record `scale_rec` (lane-major, 16 bytes per lane), which f8_flush_scales writes out once the slot's 16 groups are in
DEVINL void f8_store_group(char* sub_base, int kg, int lane, const u32x2& v, uint32_t E, uint32_t scale_rec) {
    u32x2* dst = reinterpret_cast<u32x2*>(sub_base + (kg >> 1) * 1024 + lane * 16 + (kg & 1) * 8);
    __builtin_nontemporal_store(v, dst);
    *reinterpret_cast<unsigned char*>(smem + scale_rec + lane * 16 + kg) = (unsigned char)E;
}
DEVINL void f8_flush_scales(char* sub_base, int lane, uint32_t scale_rec) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(smem + scale_rec + lane * 16);
    *reinterpret_cast<f32x4*>(sub_base + F8_SCALE_OFF + lane * 16) = v;
}

// ------------------------------------------------------------------------------------------------
// weight stream: L2 -> LDS ring, consumed in lock step by all wavefronts of the workgroup
// ------------------------------------------------------------------------------------------------
// The counted wait at a chunk boundary stays valid in the TRAINING kernels, which also issue dump stores: VMEM loads return in order among
// themselves, so `vmcnt(N)` proves that all but the last N issued loads have landed; stores in flight only add to the counter (a
// stricter wait, never a weaker one).  Retired experiments, logs under profiles/: a vmcnt(0) "safe" stream (round 2), M0 declared
// clobbered + power-of-two slot wrap (r03_lean_ring_ab.log: 169 fewer SALU per layer, no time), a store-aware wait ladder
// (r04_store_aware_wait_ab.log: the waves do not wait for their stores), explicit sched_group_barrier interleaves.
template <class P, int NSLOT = MLP_NSLOT>
struct WeightStream {
    static constexpr int LPW = (MLP_CHUNK_BYTES / 1024) / P::NW;     // 1 KiB glds pieces per wave per chunk
    const char* src;        // packed stream + this lane's offset inside a chunk
    uint32_t n_chunks;      // chunks in one pass over the network
    uint32_t load_idx;      // next chunk of the stream to fetch (wraps)
    uint32_t load_slot;     // ring slot it goes to
    uint32_t cur;           // LDS byte offset of the chunk the register prefetch reads from (+ lane*16)
    uint32_t cur_slot;
    uint32_t wave_lds;      // wave-uniform LDS offset of this wave's pieces inside a slot
    typename P::AReg q[P::DEPTH];   // A fragments f .. f+DEPTH-1 already in registers (f = next fragment to multiply)
    static DEVINL void dummy_sink(const bf16x8& d) { asm volatile("" ::"v"(d)); }
    template <class T> static DEVINL void dummy_sink(const T& d) { asm volatile("" ::"v"(d.lo), "v"(d.hi)); }
    DEVINL void issue() {
        const char* g = src + (size_t)load_idx * MLP_CHUNK_BYTES;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(load_slot * MLP_CHUNK_BYTES + wave_lds);
        // global -> LDS DMA (16 B/lane, lane-linear).  Issued through inline asm on purpose: hipcc's waitcnt pass treats
        // the builtin form as a "flat" access that may touch LDS and from then on turns EVERY s_waitcnt lgkmcnt(N) of
        // the A-fragment prefetch into lgkmcnt(0) -- which serialises the LDS pipeline (measured: 59% -> MFMA-bound).
        // The asm is invisible to that pass; completion is tracked by our own counted vmcnt in boundary().
        // (Round 2 A/B: a scalar-base form -- `global_load_lds_dwordx4 v_lane16, s[base] offset:1024`, M0 written once per chunk and
        // not restored -- removes 370 of 520 s_mov, all 64-bit VALU adds and 136 of 234 s_cselect from the fine kernel's code and
        // changes its time by < 0.5 %, inside the box-to-box noise: issue slots of the ring bookkeeping are not what bounds the
        // kernel.  It also faulted intermittently (an SGPR hazard between v_readfirstlane and the VMEM scalar base that the compiler
        // does not pad inside inline asm), so it was dropped.)
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(g + i * 1024), "s"(dst + i * 1024)
                         : "memory");
        }
        load_idx = (load_idx + 1 == n_chunks) ? 0u : load_idx + 1;
        load_slot = (load_slot + 1 == NSLOT) ? 0u : load_slot + 1;
    }
    // Synchronisation protocol (all code is branch-free; the only conditional instruction is the s_barrier itself):
    //   * chunk boundary i = the moment a wave's register prefetch enters chunk i.  At EVERY boundary a wave waits
    //     until at most 3 of its own chunk pieces are in flight and then issues its piece of one more chunk.
    //   * "early" waves (first half of the workgroup) execute s_barrier at even boundaries, "late" waves (second
    //     half = the other wavefront of each SIMD) at odd ones, so barrier b pairs early@2b with late@2b+1: the two
    //     wavefronts of a SIMD run one chunk (FPC MFMAs = half a 256-wide feature block) apart, and between two
    //     barriers (2 chunks) each has slack to overlap its VALU epilogue with the partner's MFMA run.
    //   * invariants after barrier b: chunks <= 2b+2 are completely in LDS (every wave waited for its pieces);
    //     every wave holds chunks <= 2b-1 in registers, so those ring slots may be refilled.  Early waves issue chunk
    //     i+NSLOT-2 at boundary i, late waves chunk i+NSLOT-3: both groups issue the same chunk within the same barrier
    //     interval, and NSLOT-5 chunks per wave stay in flight across every wait.
    //   * with a single wave group (NW <= 4: no late waves) the barrier falls on even boundaries only, so only those need the
    //     wait, and NSLOT-4 chunks may stay in flight: issued before the wait at boundary 2b are chunks <= 2b+NSLOT-3, needed
    //     complete are chunks <= 2b+1 (read until barrier b+1).
    static constexpr bool TWO_GROUPS = P::NW > 4;
    static constexpr int INFLIGHT = (TWO_GROUPS ? NSLOT - 5 : NSLOT - 4) * LPW;
    uint32_t late;

    DEVINL void init(const void* packed, uint32_t nchunks) {
        static_assert(NSLOT >= 6, "the protocol needs at least 6 ring slots");
        const int lane = lane_id();
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        src = reinterpret_cast<const char*>(packed) + (size_t)wave * LPW * 1024 + lane * 16;
        wave_lds = wave * LPW * 1024;
        n_chunks = nchunks;
        load_idx = 0; load_slot = 0;
        cur_slot = 0;
        cur = lane * 16;
        late = __builtin_amdgcn_readfirstlane((P::NW > 4 && wave >= P::NW / 2) ? 1 : 0);
#pragma unroll
        for (int i = 0; i < NSLOT - 2; ++i) issue();                         // chunks 0 .. NSLOT-3
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");   // my pieces of chunks 0..2 have landed ...
        __builtin_amdgcn_s_barrier();                                        // ... and everybody else's
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < P::DEPTH; ++i) q[i] = P::load_a(cur + i * P::FRAG_BYTES);
        if (!late) {                                                         // boundary 0 of the early waves
            __builtin_amdgcn_s_barrier();                                    // barrier 0 (late waves: at their boundary 1)
            asm volatile("" ::: "memory");
            issue();                                                         // chunk NSLOT-2
        }
    }
    // Fragment F of the stream (compile-time index, F mod DEPTH == queue slot): hand out its registers and
    // start the LDS read of fragment F+DEPTH into the same slot.  The boundary work therefore runs DEPTH
    // fragments BEFORE the first MFMA that needs the new chunk: the MFMA pipe keeps draining the register queue.
    template <int F>
    DEVINL typename P::AReg next() {
        const typename P::AReg a = q[F % P::DEPTH];
        constexpr int G = F + P::DEPTH;
        if (G % P::FPC == 0) {
            cur_slot = (cur_slot + 1 == NSLOT) ? 0u : cur_slot + 1;
            cur = cur_slot * MLP_CHUNK_BYTES + lane_id() * 16;
            boundary<(G / P::FPC) & 1>();
        }
        q[F % P::DEPTH] = P::load_a(cur + (G % P::FPC) * P::FRAG_BYTES);
        return a;
    }
    template <int PARITY>
    DEVINL void boundary() {
        if constexpr (!TWO_GROUPS && PARITY != 0) {          // single group, odd boundary: nothing to wait for, no barrier
            issue();
            return;
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        // s_barrier only when this boundary's parity is mine; the branch lives inside the asm so that the compiler
        // sees straight-line code (a C++ `if` here splits every feature block into many basic blocks and spills)
        asm volatile("s_cmp_lg_u32 %0, %1\n\ts_cbranch_scc1 .Lnobar%=\n\ts_barrier\n.Lnobar%=:" ::"s"(__builtin_amdgcn_readfirstlane(late)), "n"(PARITY) : "memory", "scc");
        issue();
    }
    // End of kernel: the early waves ran one barrier more (barrier 0 at init); the late waves supply its partner here.
    DEVINL void drain() {
        if (late) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
};

// fetch and drop N fragments of the stream (padding that keeps a cyclic stream at an even number of chunks): the ring protocol
// (boundaries, barriers, refills) advances exactly as if they had been multiplied
template <class P, int F0, int N, class WS, int I = 0>
DEVINL void skip_frags(WS& ws) {
    if constexpr (I < N) {
        const typename P::AReg a = ws.template next<F0 + I>();
        WS::dummy_sink(a);
        skip_frags<P, F0, N, WS, I + 1>(ws);
    }
}

// ------------------------------------------------------------------------------------------------
// One dense layer for this wavefront's NT x 32 samples:  out[fb] = act(W[fb] . in + bias[fb]).
//   NKG    K groups (16 input features each) consumed;  in(kg, t) returns the B registers of group kg, column tile t
//   NFB    32-row output feature blocks;  out(fb, t, acc, half) converts accumulators 8*half .. 8*half+7 of block fb
//          (= K group 2*fb + half of the next layer) for column tile t
//   START  fragment index of the layer inside the stream modulo FPC (chunk phase)
// ------------------------------------------------------------------------------------------------
// Feature blocks are processed two at a time with their MFMAs alternating between independent accumulators.
// Why: on gfx950 any instruction issued between two MFMAs that chain through the SAME accumulator (here: the
// ds_read of the next A fragment) costs ~+43 cycles on the dependent MFMA (MI355X_MICROARCH.md, per-instruction
// constants).  Alternating accumulators puts a full MFMA between dependent ones.  The stream stores a pair's fragments
// interleaved (kg-major), so the B registers of K group kg are fetched once and feed both blocks.
// A single trailing block (odd NFB) splits K over two accumulators instead and adds them at the end.
//
// Deferred epilogues (software pipelining by hand): the conversion of a block pair's accumulators (bias is already in,
// so: fp32 -> bf16, ReLU) is NOT done when the pair's MFMA chain ends but sliced into the first K steps of the NEXT
// pair -- of the same layer, or of the next layer, whose first K groups never read the last pair's features (kg 12..15
// are consumed at K steps >= 12).  The VALU work then sits between MFMAs that do not depend on it instead of in one
// clump during which the MFMA pipe idles (one wave per SIMD cannot hide it behind a partner).
template <int S> struct IC { static constexpr int value = S; };

// accumulators 8*half .. 8*half+7 of a feature block -> the B register group they form for the next layer
template <class P, bool RELU>
DEVINL typename P::BReg to_breg_half(const f32x16& acc, int half) {
    return P::template from_acc<RELU>(acc, 8 * half);
}

// the not-yet-converted accumulators of NB (1 or 2) feature blocks starting at block FB0
template <class P, int FB0, int NB>
struct Deferred {
    static constexpr int NSL = NB * 2 * P::NT;                   // slices: (block, half, tile)
    f32x16 acc[NB][P::NT];
    template <int S, class OutF>
    DEVINL void emit(OutF& out) const {
        constexpr int blk = S / (2 * P::NT), half = (S % (2 * P::NT)) / P::NT, t = S % P::NT;
        out(FB0 + blk, t, acc[blk][t], half);
    }
    template <class OutF, int S = 0>
    DEVINL void flush(OutF&& out) const {
        if constexpr (S < NSL) { emit<S>(out); flush<OutF, S + 1>(static_cast<OutF&&>(out)); }
    }
};
struct NoPrev {
    static constexpr int NSL = 0;
    template <int S> DEVINL void emit() const {}
};
template <class D, class OutF>
struct PrevOf {
    static constexpr int NSL = D::NSL;
    const D& d;
    OutF& out;
    template <int S> DEVINL void emit() const { d.template emit<S>(out); }
};
template <class D, class OutF> DEVINL PrevOf<D, OutF> prev_of(const D& d, OutF& out) { return PrevOf<D, OutF>{d, out}; }
// slices of the pending epilogue that belong to K step KG of an NKG-step chain: everything is out within the first half
template <int NKG, int KG, class Prev, int I = 0>
DEVINL void emit_step(const Prev& prev) {
    constexpr int STEPS = (NKG / 2 > 0) ? NKG / 2 : 1;
    constexpr int PER = (Prev::NSL + STEPS - 1) / STEPS;
    if constexpr (I < PER && KG * PER + I < Prev::NSL) {
        prev.template emit<KG * PER + I>();
        emit_step<NKG, KG, Prev, I + 1>(prev);
    }
}

// bias of one 32-row feature block in accumulator layout (acc[r] <- bias[(r&3) + 8(r>>2) + 4h]); it enters the chain
// as the C operand of the block's first MFMA, so the epilogue needs no adds
DEVINL f32x16 load_bias(uint32_t addr) {
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + addr + 32 * q);
        v[4 * q] = b4[0]; v[4 * q + 1] = b4[1]; v[4 * q + 2] = b4[2]; v[4 * q + 3] = b4[3];
    }
    return v;
}
template <class P, int NKG, int FRAG0, int KG, bool MORE, class WS, class InF, class Prev>
DEVINL void pair_k(WS& ws, f32x16 (&acc0)[P::NT], f32x16 (&acc1)[P::NT], f32x16 (&nb)[2], uint32_t next_bias, InF& in, const Prev& prev) {
    if constexpr (KG < NKG) {
        typename P::BReg b[P::NT];
#pragma unroll
        for (int t = 0; t < P::NT; ++t) b[t] = in(KG, t);
        const typename P::AReg a0 = ws.template next<FRAG0 + 2 * KG>();
        constexpr int POS = (KG == 0) ? 0 : ((KG == NKG - 1) ? 2 : 1);
#pragma unroll
        for (int t = 0; t < P::NT; ++t) acc0[t] = P::template mma_pos<POS>(a0, b[t], acc0[t]);
        const typename P::AReg a1 = ws.template next<FRAG0 + 2 * KG + 1>();
#pragma unroll
        for (int t = 0; t < P::NT; ++t) acc1[t] = P::template mma_pos<POS>(a1, b[t], acc1[t]);
        emit_step<NKG, KG>(prev);
        if constexpr (MORE && KG == (NKG > 3 ? NKG - 3 : 0)) {  // next group's bias: read late (short live range), the
            nb[0] = load_bias(next_bias);                         // latency is covered by the last MFMAs of this pair
            nb[1] = load_bias(next_bias + 128);
        }
        pair_k<P, NKG, FRAG0, KG + 1, MORE>(ws, acc0, acc1, nb, next_bias, in, prev);
    }
}
template <class P, int NKG, int FRAG0, int KG, class WS, class InF, class Prev>
DEVINL void single_k(WS& ws, f32x16 (&acc0)[P::NT], f32x16 (&acc1)[P::NT], InF& in, const Prev& prev) {
    if constexpr (KG < NKG) {
        const typename P::AReg a = ws.template next<FRAG0 + KG>();
        constexpr int LAST_EVEN = ((NKG - 1) / 2) * 2, LAST_ODD = (NKG % 2 == 0) ? NKG - 1 : NKG - 2;
#pragma unroll
        for (int t = 0; t < P::NT; ++t) {
            if constexpr (KG % 2 == 0) acc0[t] = P::template mma_pos<(KG == 0) ? 0 : ((KG == LAST_EVEN) ? 2 : 1)>(a, in(KG, t), acc0[t]);
            else acc1[t] = P::template mma_pos<(KG == 1) ? 4 : ((KG == LAST_ODD) ? 2 : 1)>(a, in(KG, t), acc1[t]);
        }
        emit_step<NKG, KG>(prev);
        single_k<P, NKG, FRAG0, KG + 1>(ws, acc0, acc1, in, prev);
    }
}
// Runs feature-block groups G, G+1, ... of the layer; `prev` is the pending epilogue that the FIRST K steps of group G
// work off.  Returns the layer's last group, unconverted.
// optional hook of an output functor: out.begin_group(G) runs when the MFMAs of feature-block group G start (the backward chain issues
// the LDS-DMA of the group's ReLU masks there, one whole group before its deferred epilogue needs them)
template <class T, class = void> struct has_group_hook { static constexpr bool value = false; };
template <class T> struct has_group_hook<T, decltype(void(static_cast<T*>(nullptr)->begin_group(0)))> { static constexpr bool value = true; };

// ZB (round 5): a layer WITHOUT a bias -- every layer of the backward chains (W^T delta).  The accumulators start from the literal zero
// (the MFMA's C operand becomes the inline constant 0) instead of from a "bias" of zeros read from LDS: no 2 x 16 bias registers per
// pending feature-block pair (the chains ran at 503-512 registers and spilled 16-65 of them), no 8 ds_read_b128 and 32 v_mov per pair.
template <class P, int NKG, int NFB, int START, int G, bool ZB, class WS, class InF, class OutF, class Prev>
DEVINL auto dense_group(WS& ws, uint32_t bias_lane, f32x16 (&cb)[2], InF& in, OutF& out, const Prev& prev) {
    static_assert(2 * G < NFB, "group index");
    constexpr int FRAG0 = START + 2 * G * NKG;
    constexpr f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (has_group_hook<OutF>::value) out.begin_group(G);
    if constexpr (2 * G + 1 < NFB) {
        Deferred<P, 2 * G, 2> d;
#pragma unroll
        for (int t = 0; t < P::NT; ++t) {
            if constexpr (ZB) { d.acc[0][t] = zero; d.acc[1][t] = zero; }
            else { d.acc[0][t] = cb[0]; d.acc[1][t] = cb[1]; }
        }
        f32x16 nb[2];
        constexpr bool MORE = 2 * (G + 1) < NFB;
        pair_k<P, NKG, FRAG0, 0, MORE && !ZB>(ws, d.acc[0], d.acc[1], nb, bias_lane + 256 * (G + 1), in, prev);
        if constexpr (MORE) return dense_group<P, NKG, NFB, START, G + 1, ZB>(ws, bias_lane, nb, in, out, prev_of(d, out));
        else return d;
    } else {
        Deferred<P, 2 * G, 1> d;
        f32x16 acc1[P::NT];
#pragma unroll
        for (int t = 0; t < P::NT; ++t) { if constexpr (ZB) d.acc[0][t] = zero; else d.acc[0][t] = cb[0]; acc1[t] = zero; }
        single_k<P, NKG, FRAG0, 0>(ws, d.acc[0], acc1, in, prev);
#pragma unroll
        for (int t = 0; t < P::NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) d.acc[0][t][r] += acc1[t][r];
        return d;
    }
}
template <class P, int NKG, int NFB, int START, bool ZB = false, class WS, class InF, class OutF, class Prev>
DEVINL auto dense(WS& ws, uint32_t bias_lds, InF&& in, OutF&& out, const Prev& prev) {
    static_assert(START % P::DEPTH == 0 && (NKG * NFB) % P::DEPTH == 0, "layers must start on a prefetch-queue boundary");
    const uint32_t bias_lane = bias_lds + 16 * (lane_id() >> 5);
    f32x16 cb[2];
    if constexpr (!ZB) {
        cb[0] = load_bias(bias_lane);
        if constexpr (NFB > 1) cb[1] = load_bias(bias_lane + 128);
    }
    return dense_group<P, NKG, NFB, START, 0, ZB>(ws, bias_lane, cb, in, out, prev);
}

}  // namespace
