// Development: do co-resident waves keep private (scratch) memory apart at 3 blocks x 4 waves per CU?
// Every thread fills a dynamically indexed private array (forced to scratch), idles, and checks it.
//   hipcc --offload-arch=gfx950 -O2 scratch_overlap.hip -o /tmp/scratch_overlap && /tmp/scratch_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float s_dyn[];
__global__ __launch_bounds__(256, 3) void probe(const int *perm, int *bad, int rounds)
{
    int a[96];
    const int id = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < 96; ++i) a[perm[i]] = id * 131 + i * 7;      // perm: runtime permutation -> scratch, not registers
    s_dyn[threadIdx.x] = (float)id;
    __syncthreads();
    int errs = 0;
    for (int r = 0; r < rounds; ++r) {
        __builtin_amdgcn_s_sleep(20);
        for (int i = 0; i < 96; ++i) errs += a[perm[95 - i]] != id * 131 + (95 - i) * 7;
        for (int i = 0; i < 96; ++i) a[perm[i]] = id * 131 + i * 7;
        __syncthreads();
    }
    if (errs) atomicAdd(bad, errs);
}
int main()
{
    int h_perm[96];
    for (int i = 0; i < 96; ++i) h_perm[i] = (i * 37) % 96;
    int *perm, *bad;
    hipMalloc(&perm, sizeof h_perm);
    hipMalloc(&bad, 4);
    hipMemcpy(perm, h_perm, sizeof h_perm, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 4);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 53056);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(probe, dim3(8192), dim3(256), 53056, 0, perm, bad, 50);
    int h_bad = -1;
    hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void *)probe);
    printf("scratch bytes per thread %zu, registers %d; mismatches %d\n", fa.localSizeBytes, fa.numRegs, h_bad);
    return 0;
}
