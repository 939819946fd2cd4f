// Developer tool: throughput of the global -> LDS staging paths on one CU (L2-resident source), per wave-instruction.
//   MODE 0: buffer_load_dwordx4 ... lds   1: global_load_lds_dwordx4   2: global_load_dwordx4 -> VGPR -> ds_write_b128
//   MODE 3: global_load_dwordx4 -> VGPR only   4: ds_read_b128 only (24 per iteration)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void dma_kernel(const unsigned char* __restrict__ src, unsigned long long* __restrict__ out, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char* base = src + (size_t)(blockIdx.x & 31) * 131072;    // 4 MB footprint: L2 resident
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 131072, 0x00020000);
    int vo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vo[i] = ((wave * 8 + i) * 1024 + lane * 16) & 65535;
    uint4 acc = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int st = (it & 1) * 65536;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + st + (wave * 8 + i) * 1024 % 65536), 16, vo[i], (it & 1) * 65536, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (it & 1) * 65536 + vo[i]),
                                                 (lds_ptr_t)(lds + st + (wave * 8 + i) * 1024 % 65536), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (MODE == 2 || MODE == 3) {
            uint4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4*>(base + (it & 1) * 65536 + vo[i]);
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(lds + st + ((wave * 8 + i) * 1024 % 65536) + lane * 16) = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const uint4 v = *reinterpret_cast<const uint4*>(lds + st + ((wave * 24 + i) * 1024 % 65536) + (((lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4)) & 1023));
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) out[blockIdx.x * NW + wave] = t1 - t0;
    if (acc.x == 0x12345 && sink) sink[tid] = (float)(acc.x + acc.y + acc.z + acc.w + lds[tid]);
}

template <int MODE, int NW>
static void run(const char* name, const unsigned char* src, unsigned long long* dout, float* sink, int iters) {
    CHECK(hipFuncSetAttribute((const void*)dma_kernel<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    dma_kernel<MODE, NW><<<256, NW * 64, 131072>>>(src, dout, 10, sink);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    dma_kernel<MODE, NW><<<256, NW * 64, 131072>>>(src, dout, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(256 * NW);
    CHECK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto c : h) avg += (double)c;
    avg /= h.size();
    const int per_it = MODE == 4 ? 24 : 8;
    const double bytes_cu = (double)NW * per_it * 1024.0 * iters;
    printf("%-44s waves/CU %d: %7.1f cycles per wave-instr (wave clock), %6.1f B/clk/CU, kernel %.3f ms -> %.2f TB/s chip\n", name, NW,
           avg / ((double)iters * per_it), bytes_cu / avg, ms, bytes_cu * 256 / ms / 1e9);
}

int main() {
    unsigned char* src; unsigned long long* dout; float* sink;
    CHECK(hipMalloc(&src, 8 << 20)); CHECK(hipMemset(src, 1, 8 << 20));
    CHECK(hipMalloc(&dout, 256 * 8 * 8)); CHECK(hipMalloc(&sink, 4096));
    const int iters = 2000;
    run<0, 8>("buffer_load_dwordx4 lds", src, dout, sink, iters);
    run<0, 4>("buffer_load_dwordx4 lds", src, dout, sink, iters);
    run<1, 8>("global_load_lds_dwordx4", src, dout, sink, iters);
    run<1, 4>("global_load_lds_dwordx4", src, dout, sink, iters);
    run<2, 8>("global_load_dwordx4 + ds_write_b128", src, dout, sink, iters);
    run<2, 4>("global_load_dwordx4 + ds_write_b128", src, dout, sink, iters);
    run<3, 8>("global_load_dwordx4 -> VGPR", src, dout, sink, iters);
    run<3, 4>("global_load_dwordx4 -> VGPR", src, dout, sink, iters);
    run<4, 8>("ds_read_b128 (swizzled fragment pattern)", src, dout, sink, iters);
    run<4, 4>("ds_read_b128 (swizzled fragment pattern)", src, dout, sink, iters);
    return 0;
}
