// Microbenchmark (development): how many VALU instructions of the SAME wave hide in the shadow of one MFMA, and does
// s_setprio on a co-resident VALU wave buy it issue slots?   build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV>   // NV VALU fmas after every MFMA, one wave per SIMD
__global__ __launch_bounds__(256) void same_wave(int iters, float *out, long long *cyc)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a = lane * 0.001f, b = lane * 0.002f;
    f16x8 ha, hb;
    for (int r = 0; r < 8; ++r) { ha[r] = (_Float16)(lane * 0.01f + r); hb[r] = (_Float16)(r * 0.5f); }
    float x[16];
    for (int r = 0; r < 16; ++r) x[r] = lane * 0.1f + r;
    const float c = 1.0001f, e = 0.0001f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (KIND == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < NV; ++r) x[r % 16] = fmaf(x[r % 16], c, e);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KIND == 0) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < NV; ++r) x[r % 16] = fmaf(x[r % 16], c, e);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + x[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// two waves per SIMD: waves 0-3 MFMA only, waves 4-7 VALU only with priority `prio`
template <int KIND>
__global__ __launch_bounds__(512) void two_waves(int iters, int prio, float *out, long long *cyc)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        f32x16 acc0, acc1;
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        float a = lane * 0.001f, b = lane * 0.002f;
        f16x8 ha, hb;
        for (int r = 0; r < 8; ++r) { ha[r] = (_Float16)(lane * 0.01f + r); hb[r] = (_Float16)(r * 0.5f); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (KIND == 0) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc1, 0, 0, 0);
                }
            }
        }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        float x[8];
        for (int r = 0; r < 8; ++r) x[r] = lane * 0.1f + r;
        const float c = 1.0001f, e = 0.0001f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 8; ++r) x[r] = fmaf(x[r], c, e);
        }
        float s = 0.f;
        for (int r = 0; r < 8; ++r) s += x[r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int NV>
void run_same(const char *name, int iters, float *out, long long *cyc)
{
    long long h[1024];
    same_wave<KIND, NV><<<256, 256>>>(iters, out, cyc);
    hipDeviceSynchronize();
    same_wave<KIND, NV><<<256, 256>>>(iters, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 1024; ++i) m += h[i];
    printf("%s same wave, %2d VALU per MFMA: %.1f cycles per MFMA\n", name, NV, m / 1024 / iters / 16);
}
template <int KIND>
void run_two(const char *name, int iters, int prio, float *out, long long *cyc)
{
    long long h[2048];
    two_waves<KIND><<<256, 512>>>(iters, prio, out, cyc);
    hipDeviceSynchronize();
    two_waves<KIND><<<256, 512>>>(iters, prio, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int i = 0; i < 256; ++i) { for (int w = 0; w < 4; ++w) m += h[i * 8 + w]; for (int w = 4; w < 8; ++w) v += h[i * 8 + w]; }
    printf("%s two waves, VALU wave prio %d: matrix wave %.1f cycles per MFMA, vector wave %.2f cycles per FMA\n", name, prio,
           m / 1024 / iters / 16, v / 1024 / iters / 32);
}
int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 5000;
    run_same<0, 0>("f32", iters, out, cyc); run_same<0, 4>("f32", iters, out, cyc); run_same<0, 8>("f32", iters, out, cyc);
    run_same<0, 12>("f32", iters, out, cyc); run_same<0, 16>("f32", iters, out, cyc); run_same<0, 24>("f32", iters, out, cyc);
    run_same<1, 0>("f16", iters, out, cyc); run_same<1, 2>("f16", iters, out, cyc); run_same<1, 4>("f16", iters, out, cyc);
    run_same<1, 6>("f16", iters, out, cyc); run_same<1, 8>("f16", iters, out, cyc); run_same<1, 12>("f16", iters, out, cyc);
    for (int prio : {0, 1, 3}) { run_two<0>("f32", iters, prio, out, cyc); run_two<1>("f16", iters, prio, out, cyc); }
    return 0;
}
