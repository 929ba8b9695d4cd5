// Microbenchmark (development): issue cost per instruction, one wave per SIMD (s_memtime cycles / instruction), of
//   the fp16 MFMA shapes, ds_bpermute_b32, ds_read_b128, and -- four waves of a CU at once -- the same LDS ops,
// to tell whether the 16x16x32 MFMA runs at the rate of the 32x32x16 one and what the LDS crossbar sustains per CU.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_rates mfma_lds_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void rate(int iters, float *out, long long *cyc)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
    __syncthreads();
    f16x8 ha, hb;
    for (int r = 0; r < 8; ++r) { ha[r] = (_Float16)(lane * 0.01f + r); hb[r] = (_Float16)(r * 0.5f); }
    f32x16 c32[4];
    f32x4 c16[8];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) c32[k][r] = 0.f;
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) c16[k][r] = 0.f;
    int v[8];
    for (int r = 0; r < 8; ++r) v[r] = lane + r;
    f32x4 rd[8];
    for (int r = 0; r < 8; ++r) rd[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int paddr = (4 * (lane & 15) + (lane >> 4)) * 4;
    const float *rp = lds + wave * 2048 + (lane & 15) * 260 / 4 * 4 + (lane >> 4) * 4;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {          // 8 independent 16x16x32 f16
#pragma unroll
            for (int k = 0; k < 8; ++k) c16[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c16[k], 0, 0, 0);
        } else if constexpr (KIND == 1) {   // 4 independent 32x32x16 f16 (x2 = the same number of MACs as KIND 0 x 4)
#pragma unroll
            for (int k = 0; k < 8; ++k) c32[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c32[k & 3], 0, 0, 0);
        } else if constexpr (KIND == 2) {   // 8 ds_bpermute
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_ds_bpermute(paddr, v[k]);
        } else if constexpr (KIND == 3) {   // 8 ds_read_b128, fragment pattern
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(rp + k * 16);
                rd[k] += t;
            }
        } else {                            // 2 back-to-back DEPENDENT 16x16x32 (latency)
#pragma unroll
            for (int k = 0; k < 8; ++k) c16[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c16[0], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += c32[k][r];
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) s += c16[k][r] + rd[k][r];
    for (int r = 0; r < 8; ++r) s += v[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 2000;
    const char *names[5] = {"v_mfma_f32_16x16x32_f16 (independent)", "v_mfma_f32_32x32x16_f16 (independent)", "ds_bpermute_b32",
                            "ds_read_b128 (fragment pattern)", "v_mfma_f32_16x16x32_f16 (dependent chain)"};
    for (int kind = 0; kind < 5; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 32768, 0, iters, out, cyc);
            if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 32768, 0, iters, out, cyc);
            if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 32768, 0, iters, out, cyc);
            if (kind == 3) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(256), 32768, 0, iters, out, cyc);
            if (kind == 4) hipLaunchKernelGGL(rate<4>, dim3(256), dim3(256), 32768, 0, iters, out, cyc);
            hipDeviceSynchronize();
        }
        long long h[1024];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 1024; ++i) m += h[i];
        // s_memtime ticks at 100 MHz on this part: convert with the ratio measured against the shader clock elsewhere? report raw per-instruction ticks
        printf("%-44s %.2f memtime ticks per instruction (4 waves per CU, one per SIMD, 8 per iteration)\n", names[kind], m / 1024 / iters / 8);
    }
    return 0;
}
