// Microbenchmark (development): request rate of float atomics onto pseudo-random 128-byte runs of a 256 MB array, by memory scope /
// cache-policy bits -- is there a cheaper (XCD-local) form than the device-scope one the tiled backward uses?  Also checks the SUM:
// every element must end up with the number of additions it received whatever the scope (all CUs of all XCDs add to the same array).
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_scopes atomic_scopes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(8))) void *rsrc_t;

template <int MODE>
__global__ __launch_bounds__(256) void k(float *dst, unsigned rows, int iters)
{
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    unsigned state = wave * 2654435761u + 12345u;
    __amdgpu_buffer_rsrc_t buf = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(rows * 1024u), 0x00020000);
    for (int i = 0; i < iters; ++i) {
        state = state * 1664525u + 1013904223u;
        const unsigned row = ((state >> 8) + (unsigned)lh * 4u) % rows;
        const unsigned off = row * 1024u + (unsigned)((i & 7) * 32 + li) * 4u;
        float *p = dst + off / 4;
        if (MODE == 0) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(1.0f, buf, (int)off, 0, 0);
        else if (MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 3) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (MODE == 4) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else if (MODE == 5) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(1.0f, buf, (int)off, 0, 16);   // aux bit 4: sc1
        else if (MODE == 6) { float v = *p; *p = v + 1.0f; }      // (plain read-modify-write: WRONG sums, the rate of the traffic alone)
        // 64-bit operands: is the rate one OPERATION or one dword per clock and channel?  (lane -> 8 bytes: a half-wave covers
        // 256 bytes of the row; the sum check counts the low dword only)
        else if (MODE == 7) {
            unsigned long long *q = (unsigned long long *)(dst + (row * 1024u + (unsigned)((i & 3) * 64 + li * 2) * 4u) / 4);
            __hip_atomic_fetch_add(q, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 8) {
            double *q = (double *)(dst + (row * 1024u + (unsigned)((i & 3) * 64 + li * 2) * 4u) / 4);
            __hip_atomic_fetch_add(q, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 9) {   // one 256-byte run per wave instead of two 128-byte runs
            const unsigned row1 = (state >> 8) % rows;
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(1.0f, buf, (int)(row1 * 1024u + (unsigned)((i & 3) * 64 + lane) * 4u), 0, 0);
        } else if (MODE == 10) {  // integer add, 32 bit
            __hip_atomic_fetch_add((unsigned *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MODE>
void run(const char *name, float *dst, unsigned rows, int blocks, int iters)
{
    hipMemset(dst, 0, (size_t)rows * 1024);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(a);
        k<MODE><<<blocks, 256>>>(dst, rows, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r && ms < best) best = ms;
    }
    std::vector<float> h((size_t)rows * 256);
    hipMemcpy(h.data(), dst, h.size() * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (float v : h) sum += v;
    const double want = 4.0 * blocks * 4.0 * iters * 64.0;
    const double n = (double)blocks * 4 * iters;
    const double bytes = (MODE == 7 || MODE == 8) ? 512 : 256;
    if (MODE >= 7 && MODE != 9) {
        printf("%-44s %.3f ms  %.2f G wave-instr/s  %.0f GB/s\n", name, best, n / best / 1e6, n * bytes / best / 1e6);
        return;
    }
    printf("%-44s %.3f ms  %.2f G wave-instr/s  %.0f GB/s   sum %.0f of %.0f %s\n", name, best, n / best / 1e6, n * 256 / best / 1e6, sum, want,
           sum == want ? "(exact)" : "(LOST UPDATES)");
}

int main(int argc, char **argv)
{
    const unsigned rows = argc > 1 ? (unsigned)atoi(argv[1]) : 1u << 18;      // (1 KB each; default 256 MB)
    printf("array: %u rows of 1 KB = %.1f MB\n", rows, rows / 1024.0);
    float *dst; hipMalloc(&dst, (size_t)rows * 1024);
    const int blocks = argc > 2 ? atoi(argv[2]) : 2048, iters = argc > 3 ? atoi(argv[3]) : 256;      // (blocks of four waves)
    printf("blocks %d x 4 waves, %d atomics per wave\n", blocks, iters);
    run<0>("buffer_atomic_fadd aux 0 (the backward's)", dst, rows, blocks, iters);
    run<5>("buffer_atomic_fadd aux sc1", dst, rows, blocks, iters);
    run<4>("__hip_atomic_fetch_add wavefront scope", dst, rows, blocks, iters);
    run<1>("__hip_atomic_fetch_add workgroup scope", dst, rows, blocks, iters);
    run<2>("__hip_atomic_fetch_add agent scope", dst, rows, blocks, iters);
    run<3>("__hip_atomic_fetch_add system scope", dst, rows, blocks, iters);
    run<6>("plain load + store (no atomic)", dst, rows, blocks, iters);
    run<9>("buffer_atomic_fadd, one 256-byte run / wave", dst, rows, blocks, iters);
    run<10>("atomic add u32 agent scope", dst, rows, blocks, iters);
    run<7>("atomic add u64 agent scope (8 B / lane)", dst, rows, blocks, iters);
    run<8>("atomic add f64 agent scope (8 B / lane)", dst, rows, blocks, iters);
    return 0;
}
