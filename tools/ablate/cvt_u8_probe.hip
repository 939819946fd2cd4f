// rounding / saturation of v_cvt_pk_u8_f32 on gfx950 (hipcc --offload-arch=gfx950 -O3 cvt_u8_probe.hip -o cvt_u8_probe.bin)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0);
}
int main() {
    const float h[] = {0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 127.5f, 254.6f, 255.4f, 255.5f, 300.f, -0.4f, -3.f, 1e30f, -1e30f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* d; unsigned* o; unsigned r[32];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(r, o, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%g -> %u\n", h[i], r[i]);
    return 0;
}
