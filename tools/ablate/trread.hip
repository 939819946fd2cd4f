// Decode ds_read_b64_tr_b16: every LDS halfword holds its own index; each lane supplies its own byte address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const int* addr_in, unsigned short* out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)(lds) + (unsigned)addr_in[threadIdx.x];
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    int *d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<int> addr(64);
        for (int l = 0; l < 64; ++l) {
            if (mode == 0) addr[l] = l * 8;                       // lane l -> its own consecutive 8-byte chunk
            if (mode == 1) addr[l] = (l & 15) * 64 + (l >> 4) * 8; // 16-lane group g: rows (l&15) of 32 halfwords, chunk g
            if (mode == 2) addr[l] = ((l & 15) >> 2) * 128 + (l & 3) * 8 + (l >> 4) * 1024;  // 4 rows x 4 chunks per group
        }
        hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        std::vector<unsigned short> out(256);
        hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("mode %d (addresses in halfwords: lane0=%d lane1=%d lane4=%d lane16=%d)\n", mode, addr[0] / 2, addr[1] / 2, addr[4] / 2, addr[16] / 2);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3], (l % 4 == 3) ? "\n" : " |");
    }
    return 0;
}
