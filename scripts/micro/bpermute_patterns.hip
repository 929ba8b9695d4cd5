// Microbenchmark (development): cost of ds_bpermute_b32 by ADDRESS PATTERN, four waves of a CU at once (one per SIMD) -- does the
// crossbar pay for two destination lanes of a 32-lane group asking for source lanes 32 apart (the same "bank")?  G1 of the persistent
// forward rearranges its loaded B rows with  paddr = 4 (4 (lane & 15) + (lane >> 4))  : destination lanes n and n + 8 of a half-wave
// ask for source lanes 4 n + kg and 4 n + kg + 32.
//   build: hipcc --offload-arch=gfx950 -O3 -o bpermute_patterns bpermute_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(256) void rate(int iters, int *out, long long *cyc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int v[8];
    for (int r = 0; r < 8; ++r) v[r] = lane * 8 + r;
    int src;
    if (KIND == 0) src = lane;                                                    // identity
    else if (KIND == 1) src = 4 * (lane & 15) + (lane >> 4);                      // G1's: sources 32 apart within a half-wave
    else if (KIND == 2) src = 4 * (lane & 15) + ((lane >> 4) ^ ((lane & 8) ? 2 : 0));   // chunk order flipped for rows 8..15
    else if (KIND == 3) src = lane ^ 32;                                          // swap the halves
    else src = (lane * 2) & 63;                                                   // pairs of destinations share a source (broadcast)
    const int paddr = src * 4;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_ds_bpermute(paddr, v[k]);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int r = 0; r < 8; ++r) s += v[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main()
{
    int *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 2000;
    const char *names[5] = {"identity", "G1 (4 n + kg: sources 32 apart in a half-wave)", "G1 with the chunk order flipped for rows 8-15",
                            "lane ^ 32", "2 lane mod 64 (shared sources)"};
    for (int kind = 0; kind < 5; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, iters, out, cyc);
            if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, iters, out, cyc);
            if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, iters, out, cyc);
            if (kind == 3) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(256), 0, 0, iters, out, cyc);
            if (kind == 4) hipLaunchKernelGGL(rate<4>, dim3(256), dim3(256), 0, 0, iters, out, cyc);
            (void)hipDeviceSynchronize();
        }
        long long h[1024];
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 1024; ++i) m += h[i];
        printf("%-52s %.2f memtime ticks per ds_bpermute_b32 (4 waves per CU, one per SIMD)\n", names[kind], m / 1024 / iters / 8);
    }
    return 0;
}
