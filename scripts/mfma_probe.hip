// What does one wave per SIMD sustain on gfx950 for the instruction mix of the fused MLP kernels' dense loop?  
// One workgroup of 4 waves per CU (160 KiB of LDS requested so that nothing else is co-resident), every wave runs the same loop of
// v_mfma_f32_32x32x16_bf16 on FOUR independent accumulators (the wide tile: two feature blocks x two column tiles) and, per variant,
// the companions the real kernel issues per MFMA:
//   0  MFMAs only, A and B from VGPRs that never change
//   1  + the A fragment of every second MFMA comes from LDS (ds_read_b128, prefetched 4 fragments = 8 MFMAs ahead, counted lgkmcnt)
//   2  + 2 VALU per MFMA (v_cvt_pk_bf16_f32 + v_pk_max_i16 on registers no MFMA touches)
//   3  + 1.25 SALU per MFMA (s_add / s_cmp / s_cselect chains)
//   4  + one 1 KiB LDS-DMA piece (global_load_lds_dwordx4) per 8 MFMAs and a workgroup barrier per 32 MFMAs
//   5  variant 0 with B operands in AGPRs
//   6  variant 0 with TWO accumulators (the narrow tile's chains)
// Output: shader cycles (s_memtime) per MFMA of wave 0 of workgroup 0, and the wall-clock rate of the whole chip.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_probe.hip -o scripts/mfma_probe.bin && scripts/mfma_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
extern __shared__ __attribute__((aligned(16))) char smem[];

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_AB(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define VALU2(x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1\n\tv_pk_max_i16 %0, %0, 0" : "=&v"(x) : "v"(y))
#define SALU(s) asm volatile("s_add_i32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 8\n\ts_cselect_b32 %0, %0, 0\n\ts_add_i32 %0, %0, 3\n\ts_lshl_b32 %0, %0, 1" : "+s"(s) : : "scc")

template <int V>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, int iters, unsigned long long* __restrict__ cycles, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    u32x4 a0 = {0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, a1 = a0, a2 = a0, a3 = a0, b0 = a0, b1 = a0;
    const uint32_t lds_addr = lane * 16;
    uint32_t x0 = 0, x1 = 0;
    float y0 = 1.0f + lane;
    int s0 = 0;
    const char* g = src + (size_t)wave * 2048 + lane * 16;
    if (V >= 1 && V <= 4) { DSREAD(a0, lds_addr, 0); DSREAD(a1, lds_addr, 1024); DSREAD(a2, lds_addr, 2048); DSREAD(a3, lds_addr, 3072); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {                                  // 8 x (4 A fragments x 2 MFMAs... ) = 32 MFMAs per iteration
            // fragment 0
            if (V >= 1 && V <= 4) LGKM(3);
            if (V == 5) { MFMA_AB(acc0, a0, b0); MFMA_AB(acc1, a0, b1); }
            else if (V == 6) { MFMA(acc0, a0, b0); MFMA(acc1, a0, b1); }
            else { MFMA(acc0, a0, b0); MFMA(acc1, a0, b1); }
            if (V >= 1 && V <= 4) DSREAD(a0, lds_addr, 4096);
            if (V >= 2 && V <= 4) { VALU2(x0, y0); VALU2(x1, y0); VALU2(x0, y0); VALU2(x1, y0); }
            if (V >= 3 && V <= 4) SALU(s0);
            // fragment 1
            if (V >= 1 && V <= 4) LGKM(3);
            if (V == 5) { MFMA_AB(acc2, a1, b0); MFMA_AB(acc3, a1, b1); }
            else if (V == 6) { MFMA(acc0, a1, b0); MFMA(acc1, a1, b1); }
            else { MFMA(acc2, a1, b0); MFMA(acc3, a1, b1); }
            if (V >= 1 && V <= 4) DSREAD(a1, lds_addr, 5120);
            if (V >= 2 && V <= 4) { VALU2(x0, y0); VALU2(x1, y0); VALU2(x0, y0); VALU2(x1, y0); }
            if (V >= 3 && V <= 4) SALU(s0);
            if (V == 4 && (k & 1) == 0) {                              // one DMA piece per 8 MFMAs
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g + (size_t)((it * 8 + k) & 63) * 8192), "s"(65536 + wave * 2048 + (k & 6) * 512) : "memory");
            }
            // fragments 2, 3 (the other accumulator pair is reused: four chains in all)
            if (V >= 1 && V <= 4) LGKM(3);
            if (V == 5) { MFMA_AB(acc0, a2, b0); MFMA_AB(acc1, a2, b1); }
            else { MFMA(acc0, a2, b0); MFMA(acc1, a2, b1); }
            if (V >= 1 && V <= 4) DSREAD(a2, lds_addr, 6144);
            if (V >= 2 && V <= 4) { VALU2(x0, y0); VALU2(x1, y0); VALU2(x0, y0); VALU2(x1, y0); }
            if (V >= 3 && V <= 4) SALU(s0);
            if (V >= 1 && V <= 4) LGKM(3);
            if (V == 5) { MFMA_AB(acc2, a3, b0); MFMA_AB(acc3, a3, b1); }
            else if (V == 6) { MFMA(acc0, a3, b0); MFMA(acc1, a3, b1); }
            else { MFMA(acc2, a3, b0); MFMA(acc3, a3, b1); }
            if (V >= 1 && V <= 4) DSREAD(a3, lds_addr, 7168);
            if (V >= 2 && V <= 4) { VALU2(x0, y0); VALU2(x1, y0); VALU2(x0, y0); VALU2(x1, y0); }
            if (V >= 3 && V <= 4) SALU(s0);
            if (V == 4 && k == 7) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); __builtin_amdgcn_s_barrier(); }     // (64 MFMAs per barrier: 8 fragments x 2 x 4)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    r += __builtin_bit_cast(float, a0[0] ^ a1[1] ^ a2[2] ^ a3[3]) * 0.0f + (float)(x0 ^ x1) * 0.0f + (float)s0 * 0.0f;
    if (r == 123.456f) sink[threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = t1 - t0;
}

// second family: four chains, constant operands, NF fillers of one KIND placed after EVERY MFMA
//   KIND 0 v_cvt_pk_bf16_f32 (independent registers)   1 v_mov_b32   2 s_add_i32   3 s_nop 0   4 v_pk_max_i16   5 v_fma_f32
template <int NF, int KIND>
__global__ __launch_bounds__(256) void probe_even(int iters, unsigned long long* __restrict__ cycles, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    u32x4 a0 = {0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b0 = a0;
    uint32_t x0 = 0, x1 = 1, x2 = 2, x3 = 3;
    float y0 = 1.0f + lane, f0 = 0.5f;
    int s0 = 0;
#define FILL1(x)                                                                                                  \
    if (KIND == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(x) : "v"(y0));                              \
    else if (KIND == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y0));                                      \
    else if (KIND == 2) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0) : : "scc");                                 \
    else if (KIND == 3) asm volatile("s_nop 0");                                                                   \
    else if (KIND == 4) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(x) : "v"(x1));                               \
    else asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(f0) : "v"(y0));
#define FILL()                                                                                                     \
    { if (NF > 0) { FILL1(x0) } if (NF > 1) { FILL1(x2) } if (NF > 2) { FILL1(x3) } if (NF > 3) { FILL1(x0) } if (NF > 4) { FILL1(x2) } if (NF > 5) { FILL1(x3) } \
      if (NF > 6) { FILL1(x0) } if (NF > 7) { FILL1(x2) } }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            MFMA(acc0, a0, b0); FILL();
            MFMA(acc1, a0, b0); FILL();
            MFMA(acc2, a0, b0); FILL();
            MFMA(acc3, a0, b0); FILL();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = f0 * 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    r += (float)(x0 ^ x2 ^ x3) * 0.0f + (float)s0 * 0.0f;
    if (r == 123.456f) sink[threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = t1 - t0;
}
template <int NF, int KIND>
void run_even(const char* kind, unsigned long long* cyc, float* sink) {
    const int iters = 4000, mfma_per_iter = 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_even<NF, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_even<NF, KIND>), dim3(256), dim3(256), 160 * 1024, 0, 200, cyc, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_even<NF, KIND>), dim3(256), dim3(256), 160 * 1024, 0, iters, cyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double n = (double)iters * mfma_per_iter;
    printf("even  %d x %-22s after every MFMA: %6.2f cycles/MFMA   %7.1f TFLOP/s   clock %.2f GHz\n", NF, kind, (double)c / n,
           256.0 * 4.0 * n * 32768.0 / (ms * 1e-3) * 1e-12, (double)c / (ms * 1e-3) * 1e-9);
}

template <int V>
void run(const char* name, const char* src, unsigned long long* cyc, float* sink) {
    const int iters = 4000, mfma_per_iter = 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256), 160 * 1024, 0, src, 200, cyc, sink);       // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256), 160 * 1024, 0, src, iters, cyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double n = (double)iters * mfma_per_iter;
    const double tflops = 256.0 * 4.0 * n * 32768.0 / (ms * 1e-3) * 1e-12;
    printf("variant %d  %-58s %6.2f cycles/MFMA   %7.1f TFLOP/s chip-wide   clock %.2f GHz\n", V, name, (double)c / n, tflops, (double)c / (ms * 1e-3) * 1e-9);
}

int main() {
    char* src; unsigned long long* cyc; float* sink;
    hipMalloc(&src, 64 * 8192 + 65536); hipMemset(src, 0, 64 * 8192 + 65536);
    hipMalloc(&cyc, 8); hipMalloc(&sink, 4096);
    run<0>("MFMA only (4 chains)", src, cyc, sink);
    run<6>("MFMA only (2 chains)", src, cyc, sink);
    run<5>("MFMA only, B operands in AGPRs", src, cyc, sink);
    run<1>("+ A fragments from LDS (1 ds_read_b128 per 2 MFMAs)", src, cyc, sink);
    run<2>("+ 2 VALU per MFMA", src, cyc, sink);
    run<3>("+ 1.25 SALU per MFMA", src, cyc, sink);
    run<4>("+ LDS-DMA piece per 8 MFMAs, barrier per 64", src, cyc, sink);
    run_even<1, 0>("v_cvt_pk_bf16_f32", cyc, sink); run_even<2, 0>("v_cvt_pk_bf16_f32", cyc, sink); run_even<3, 0>("v_cvt_pk_bf16_f32", cyc, sink);
    run_even<4, 0>("v_cvt_pk_bf16_f32", cyc, sink); run_even<6, 0>("v_cvt_pk_bf16_f32", cyc, sink); run_even<8, 0>("v_cvt_pk_bf16_f32", cyc, sink);
    run_even<2, 1>("v_mov_b32", cyc, sink); run_even<4, 1>("v_mov_b32", cyc, sink); run_even<8, 1>("v_mov_b32", cyc, sink);
    run_even<4, 4>("v_pk_max_i16", cyc, sink); run_even<4, 5>("v_fma_f32", cyc, sink);
    run_even<2, 2>("s_add_i32", cyc, sink); run_even<4, 2>("s_add_i32", cyc, sink); run_even<8, 2>("s_add_i32", cyc, sink);
    run_even<4, 3>("s_nop 0", cyc, sink); run_even<8, 3>("s_nop 0", cyc, sink);
    return 0;
}
