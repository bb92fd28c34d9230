// probe of the gfx950 scaled fp8 conversions used by the fp8 training dumps: direction of the scale, rounding, saturation
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const float* in, float scale, float* out) {
    const int i = threadIdx.x;
    const float a = in[2 * i], b = in[2 * i + 1];
    s16x2 q = {0, 0};
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q, a, b, scale, false);
    const int p = __builtin_bit_cast(int, q);
    const bf16x2 h = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(p, scale, false);
    const f32x2 r = __builtin_amdgcn_cvt_pk_f32_fp8(p, false);
    bf16x2 src = {(__bf16)a, (__bf16)b};
    s16x2 q2 = {0, 0};
    q2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(q2, src, scale, false);
    out[8 * i] = a; out[8 * i + 1] = b; out[8 * i + 2] = (float)h[0]; out[8 * i + 3] = (float)h[1]; out[8 * i + 4] = r[0]; out[8 * i + 5] = r[1];
    out[8 * i + 6] = (float)(p & 0xffff); out[8 * i + 7] = (float)(__builtin_bit_cast(int, q2) & 0xffff);
}
int main() {
    float h[16] = {1.0f, -3.0f, 0.07f, 100.0f, 448.0f, 500.0f, 1e-3f, 0.0f, 17.3f, 0.4f, 250.0f, -0.011f, 3e4f, 2e-5f, 5.0f, 6.5f};
    float *d, *o, out[64];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(out));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (float scale : {1.0f, 4.0f, 0.25f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, scale, o);
        hipMemcpy(out, o, sizeof(out), hipMemcpyDeviceToHost);
        printf("scale %g\n", scale);
        for (int i = 0; i < 8; ++i)
            printf("  in (%g, %g) -> scaled-decode (%g, %g)  raw-decode (%g, %g)  bits f32-src %04x bf16-src %04x\n", out[8 * i], out[8 * i + 1], out[8 * i + 2], out[8 * i + 3],
                   out[8 * i + 4], out[8 * i + 5], (int)out[8 * i + 6], (int)out[8 * i + 7]);
    }
    return 0;
}
