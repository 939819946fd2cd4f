// Developer tool (round 6, review item 3): what a slab-free rel-pos backward would pay for its cross-workgroup reductions.
//
// The key-stationary decomposition the review names (dK / dV in registers, dQ and the positional-table gradient through fp32 atomics, no
// [B H, Tpad, Tpad] dS^T / P^T slabs) replaces 4.0 GB of plain slab traffic per layer (2 x 0.8 GB written by the dQ kernel, 1.6 GB read by
// the dK / dV stream, 0.8 GB by the dP kernel) by device-scope `global_atomic_add_f32` traffic: with 64-key workgroups and 128-query tiles
// every (clip, head) adds 16 key blocks x 1000 queries x 64 floats into dQ [B, T, 768] and the same count of retired band rows into
// dPos [2T, 768] -- 2 x 403 M atomic lanes = 2 x 1.6 GB per layer at B = 32.  This program issues exactly those two address streams
// (no arithmetic, no loads: the floor of the reduction alone) next to a plain-store pass over the same addresses.
//   build:  hipcc --offload-arch=gfx950 -O3 tools/ablate/atomic_bench.hip -o /tmp/atomic_bench      run: /tmp/atomic_bench [B]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int T = 1000, H = 12, DM = 768, KB = 64, QT = 128, NW = 8;

// MODE 0: plain stores   1: atomics into dQ [B, T, 768] (rows of this (clip, head), every key block adds the same rows)
// MODE 2: atomics into dPos [2048, 768] (band rows retired per tile: every clip and key block of a head adds into the head's 2T rows)
// MODE 3: MODE 1 with the 16 key blocks of a (clip, head) reduced first (one add per element: the lower bound if the keys were ONE block)
template <int MODE>
__global__ __launch_bounds__(64 * NW) void red_kernel(float* __restrict__ dq, float* __restrict__ dpos, int nkb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kb = blockIdx.x % nkb, bh = blockIdx.x / nkb, b = bh / H, h = bh - b * H;
    const float v = 1.0f + lane;
    for (int q0 = 0; q0 < T; q0 += QT) {
        // a wave's 16 query rows (dQ) / 16 of the 128 band rows this tile retires (dPos); a row = 64 consecutive floats = one 256-byte line pair
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + 16 * wave + r;
            if (q >= T) break;
            if (MODE == 0) dq[((size_t)b * T + q) * DM + h * 64 + lane] = v;
            else if (MODE == 1 || MODE == 3) unsafeAtomicAdd(&dq[((size_t)b * T + q) * DM + h * 64 + lane], v);
            else {
                const int rho = KB * kb - q + T - 1 + 63;                    // the newest band row this (key block, query row) retires, 0 .. 2T
                unsafeAtomicAdd(&dpos[(size_t)(rho < 0 ? 0 : rho) * DM + h * 64 + lane], v);
            }
        }
    }
}

template <int MODE> static float run(float* dq, float* dpos, int B, int nkb, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(red_kernel<MODE>, dim3(B * H * nkb), dim3(64 * NW), 0, 0, dq, dpos, nkb);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(red_kernel<MODE>, dim3(B * H * nkb), dim3(64 * NW), 0, 0, dq, dpos, nkb);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int nkb = (T + KB - 1) / KB;
    float *dq, *dpos;
    CHECK(hipMalloc(&dq, (size_t)B * T * DM * 4));
    CHECK(hipMalloc(&dpos, (size_t)2048 * DM * 4));
    CHECK(hipMemset(dq, 0, (size_t)B * T * DM * 4));
    CHECK(hipMemset(dpos, 0, (size_t)2048 * DM * 4));
    const double lanes = (double)B * H * nkb * T * 64, gb = lanes * 4 / 1e9;
    const float t0 = run<0>(dq, dpos, B, nkb, 5), t1 = run<1>(dq, dpos, B, nkb, 3), t2 = run<2>(dq, dpos, B, nkb, 3), t3 = run<3>(dq, dpos, B, 1, 5);
    printf("B = %d, %d key blocks of %d, %d-query tiles: %.0f M lanes = %.2f GB of payload per stream and layer\n", B, nkb, KB, QT, lanes / 1e6, gb);
    printf("  plain stores, same addresses            %8.1f us   %7.1f GB/s\n", t0 * 1e3, gb / t0 * 1e3);
    printf("  atomics -> dQ   [B, T, 768]             %8.1f us   %7.1f GB/s   %6.1f G atomic lanes/s\n", t1 * 1e3, gb / t1 * 1e3, lanes / t1 / 1e6);
    printf("  atomics -> dPos [2T, 768] (hot rows)    %8.1f us   %7.1f GB/s   %6.1f G atomic lanes/s\n", t2 * 1e3, gb / t2 * 1e3, lanes / t2 / 1e6);
    printf("  atomics -> dQ, ONE key block per (b, h) %8.1f us   %7.1f GB/s  (payload %.2f GB)\n", t3 * 1e3, gb / nkb / t3 * 1e3, gb / nkb);
    printf("  today (profiles/r4_relpos_kernels.txt): rel-pos backward of one layer 1.75 ms in all, of which ~1.0 ms is the 4.0 GB of slab traffic\n");
    return 0;
}
