// Developer probe: semantics of v_cvt_scalef32_pk_fp8_f16 / v_cvt_pk_fp8_f32 on gfx950 (scale direction, overflow, MODE.FP16_OVFL).
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/ablate/cvt_fp8_probe.hip -o /tmp/cvt_fp8_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float scale, int ovfl) {
    int i = threadIdx.x;
    if (ovfl) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // MODE.FP16_OVFL = 1
    h2 v = {(_Float16)in[2 * i], (_Float16)in[2 * i + 1]};
    s2 old = {0, 0};
    s2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, v, scale, false);
    out[i] = (unsigned)(unsigned short)r[0] | ((unsigned)(unsigned short)r[1] << 16);
    // reference path
    float a = in[2*i] , b = in[2*i+1];
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    out[32 + i] = (unsigned)w;
}
int main() {
    float h[16] = {1.0f, 3.0f, 500.f, 3000.f, 60000.f, -60000.f, 1e-3f, 0.3f, 448.f, 449.f, 464.f, 480.f, 1792.f, 2000.f, -0.01f, 7.3f};
    float* d; unsigned* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    float scales[4] = {1.0f, 4.0f, 0.25f, 4.0f};
    for (int s = 0; s < 4; ++s) {
        hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o, scales[s], s == 3);
        unsigned r[64]; hipMemcpy(r, o, 64 * 4, hipMemcpyDeviceToHost);
        printf("scale %g%s:", scales[s], s == 3 ? " FP16_OVFL=1" : "");
        for (int i = 0; i < 8; ++i) printf(" [%g,%g]->%02x,%02x (plain f32 cvt %02x,%02x)", h[2*i], h[2*i+1], r[i] & 0xff, (r[i] >> 8) & 0xff, r[32+i] & 0xff, (r[32+i]>>8)&0xff);
        printf("\n");
    }
    return 0;
}
