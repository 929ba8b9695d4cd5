// Development: an attempt at a minimal reproducer of the fault of scripts/dev/README.md -- packed-fp32 VALU arithmetic
// (v_pk_add_f32 / v_pk_mul_f32 with op_sel) in a wave whose SIMD is shared with waves running a split-fp16 GEMM phase
// (buffer loads, ds_bpermute, v_cvt_pk_f16_f32, v_mfma_f32_16x16x32_f16).  Blocks of four waves alternate between a
// "GEMM" phase and a "tap" phase, odd and even blocks out of step, three blocks per CU; the tap phase computes bilinear
// tap coordinates once with <2 x float> arithmetic and once with scalar arithmetic fenced off from the vectoriser, and counts
// lanes whose results differ bitwise (they are the same IEEE operations in the same order: any difference is a fault).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pk_f32_corun.hip -o /tmp/pk_f32_corun && /tmp/pk_f32_corun
// Result (MI355X, round 3): 0 differing lanes in every run -- the fault is NOT reproduced by this reduction; it needs more of the
// real kernel's context (register pressure / spills, the LDS traffic of the row-set lookups, the exact instruction order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

extern __shared__ float s_dyn[];

__device__ __forceinline__ float fence(float v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__global__ __launch_bounds__(256, 3) void corun(const float *__restrict__ src, const float *__restrict__ locs, int tiles, int flag,
                                                 unsigned *__restrict__ bad, float *__restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x2 *s_nxy = reinterpret_cast<f32x2 *>(s_dyn);                 // [32][64]
    float *s_stage = s_dyn + 32 * 64 * 2;                            // GEMM-phase A stage
    for (int i = threadIdx.x; i < 32 * 64; i += 256) s_nxy[i] = f32x2{locs[(blockIdx.x * 2048 + i) * 2 % (1 << 20)], locs[((blockIdx.x * 2048 + i) * 2 + 1) % (1 << 20)]};
    for (int i = threadIdx.x; i < 8448; i += 256) s_stage[i] = (float)(i & 255) * 0.01f;
    __syncthreads();
    const f32x2 half_a = {32.f * 0.5f - 0.5f, 32.f * 0.5f - 0.5f}, half_b = {32.f, 32.f};
    f32x4 acc[4] = {};
    unsigned errs[4] = {0, 0, 0, 0};
    for (int t = 0; t < tiles; ++t) {
        const bool gemm_first = (blockIdx.x + t) & 1;
        for (int ph = 0; ph < 2; ++ph) {
            if ((ph == 0) == gemm_first) {
                // ---- "GEMM" phase: quad-contiguous loads, ds_bpermute to operand order, split to fp16, MFMAs -----------------
                const int paddr = (4 * (lane & 15) + (lane >> 4)) * 4;
                for (int ks = 0; ks < 32; ++ks) {
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(src + ((size_t)((blockIdx.x * 37 + t * 11 + ks * 4 + wave) & 4095) * 256 + lane * 4));
                    float v[4];
                    for (int e = 0; e < 4; ++e) v[e] = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(x[e]))) * 0.5f;
                    f16x8 hi, lo;
                    for (int e = 0; e < 4; e += 2) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const f16x2 h = __builtin_convertvector(f2{v[e], v[e + 1]}, f16x2);
                        hi[e] = h[0]; hi[e + 1] = h[1]; hi[e + 4] = h[0]; hi[e + 5] = h[1];
                        lo[e] = (_Float16)(v[e] - (float)h[0]); lo[e + 1] = (_Float16)(v[e + 1] - (float)h[1]);
                        lo[e + 4] = lo[e]; lo[e + 5] = lo[e + 1];
                    }
                    const f16x8 a = *reinterpret_cast<const f16x8 *>(reinterpret_cast<const char *>(s_stage) + (lane & 15) * 528 + (lane >> 4) * 16 + (ks & 7) * 64);
                    for (int g = 0; g < 4; ++g) {
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, hi, acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, lo, acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, a, acc[g], 0, 0, 0);
                    }
                }
            } else {
                // ---- "tap" phase: the same arithmetic packed and scalar ---------------------------------------------------------
                for (int rep = 0; rep < 24; ++rep) {
                    for (int j = 0; j < 8; ++j) {
                        const f32x2 n = s_nxy[(wave * 8 + j) * 64 + lane];
                        // packed: what the SLP vectoriser makes of taps_of / unnormalize
                        const f32x2 n1 = n + 1.0f;
                        const f32x2 pa = half_a * n1.yx;
                        const f32x2 pb = half_b * n1.yx - 0.5f;
                        const f32x2 p = flag ? pb : pa;
                        const f32x2 fl = {__builtin_floorf(p.x), __builtin_floorf(p.y)};
                        const f32x2 fr = p - fl;
                        const f32x2 om = 1.0f - fr;
                        const float w00 = om.x * om.y, w11 = fr.x * fr.y;
                        // scalar: the same operations, kept apart
                        const float sx1 = fence(n.x) + 1.0f, sy1 = fence(n.y) + 1.0f;
                        const float qa0 = fence(half_a.x) * sy1, qa1 = fence(half_a.y) * sx1;
                        const float qb0 = fence(fence(half_b.x) * sy1) - 0.5f, qb1 = fence(fence(half_b.y) * sx1) - 0.5f;
                        const float q0 = flag ? qb0 : qa0, q1 = flag ? qb1 : qa1;
                        const float g0 = __builtin_floorf(q0), g1 = __builtin_floorf(q1);
                        const float r0 = fence(q0 - g0), r1 = fence(q1 - g1);
                        const float o0 = fence(1.0f - r0), o1 = fence(1.0f - r1);
                        const float z00 = o0 * o1, z11 = r0 * r1;
                        const bool ne = __float_as_uint(w00) != __float_as_uint(z00) || __float_as_uint(w11) != __float_as_uint(z11) ||
                                        __float_as_uint(fl.x) != __float_as_uint(g0) || __float_as_uint(fl.y) != __float_as_uint(g1);
                        errs[lane >> 4 & 3] += ne ? 1u : 0u;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int q = 0; q < 4; ++q)
        if (errs[q]) atomicAdd(&bad[q], errs[q]);
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) sink[threadIdx.x] = acc[0][0];
}

int main()
{
    const size_t nsrc = (size_t)4096 * 256, nloc = 1 << 20;
    std::vector<float> hs(nsrc), hl(nloc);
    srand(3);
    for (auto &v : hs) v = (float)(rand() % 2000) * 0.001f;
    for (auto &v : hl) v = (float)(rand() % 20000) * 0.0001f - 1.0f;
    float *src, *locs, *sink;
    unsigned *bad;
    hipMalloc(&src, nsrc * 4);
    hipMalloc(&locs, nloc * 4);
    hipMalloc(&sink, 1024);
    hipMalloc(&bad, 16);
    hipMemcpy(src, hs.data(), nsrc * 4, hipMemcpyHostToDevice);
    hipMemcpy(locs, hl.data(), nloc * 4, hipMemcpyHostToDevice);
    const size_t lds = 32 * 64 * 8 + 8448 * 4 + 2048;     // ~52 KB: three blocks per CU, like the kernel that showed the fault
    hipFuncSetAttribute((const void *)corun, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int flag = 0; flag < 2; ++flag) {
        hipMemset(bad, 0, 16);
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(corun, dim3(256 * 3 * 4), dim3(256), lds, 0, src, locs, 6, flag, bad, sink);
        unsigned h[4];
        hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        printf("flag %d: lanes whose packed and scalar results differ, by quarter of the wave (0-15, 16-31, 32-47, 48-63): %u %u %u %u  (%s)\n",
               flag, h[0], h[1], h[2], h[3], hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
