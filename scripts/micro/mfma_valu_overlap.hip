// Microbenchmark (development): does a VALU-only wave slow down when an MFMA-only wave runs on the same SIMD?
// Block = 8 waves; waves 0-3 ("matrix") and 4-7 ("vector") land pairwise on the same SIMDs.
//   mode bit0: matrix waves run MFMAs, bit1: vector waves run VALU FMAs.   kind: 0 = f32 32x32x2, 1 = f16 32x32x16, 2 = bf16 32x32x16
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k(int mode, int iters, float *out, long long *cyc)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc0, acc1;
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            float a = lane * 0.001f, b = lane * 0.002f;
            f16x8 ha, hb; bf16x8 ba, bb;
            for (int r = 0; r < 8; ++r) { ha[r] = (_Float16)(lane * 0.01f + r); hb[r] = (_Float16)(r * 0.5f); ba[r] = (__bf16)(lane * 0.01f + r); bb[r] = (__bf16)(r * 0.5f); }
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if constexpr (KIND == 0) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
                    } else if constexpr (KIND == 1) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc1, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ba, acc1, 0, 0, 0);
                    }
                }
            }
            float s = 0.f;
            for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    } else {
        if (mode & 2) {
            float x[8];
            for (int r = 0; r < 8; ++r) x[r] = lane * 0.1f + r;
            const float c = 1.0001f, e = 0.0001f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[r] = fmaf(x[r], c, e);   // 32 independent-ish FMAs per iteration
            }
            float s = 0.f;
            for (int r = 0; r < 8; ++r) s += x[r];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char *name, int iters)
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    long long h[256 * 8];
    for (int mode = 1; mode <= 3; ++mode) {
        k<KIND><<<256, 512>>>(mode, iters, out, cyc);   // warm
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        k<KIND><<<256, 512>>>(mode, iters, out, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int i = 0; i < 256; ++i) { for (int w = 0; w < 4; ++w) m += h[i * 8 + w]; for (int w = 4; w < 8; ++w) v += h[i * 8 + w]; }
        printf("%-5s mode %d (%s%s): %.3f ms   matrix-wave cycles/iter %.1f (per MFMA %.1f)   vector-wave cycles/iter %.1f (per FMA %.2f)\n", name, mode,
               (mode & 1) ? "MFMA " : "", (mode & 2) ? "VALU" : "", ms, m / 1024 / iters, m / 1024 / iters / 16, v / 1024 / iters, v / 1024 / iters / 32);
    }
}
int main()
{
    const int iters = 20000;
    run<0>("f32", iters);
    run<1>("f16", iters);
    run<2>("bf16", iters);
    return 0;
}
