// Development: does a VALU write to a register an MFMA has just been issued to (write-after-write) survive, and after how many
// wait states -- alone on the SIMD, and with other waves' MFMAs contending for the same matrix pipe?
//
// Background (scripts/dev/README.md): the one-block-per-tile forward with split-fp16 GEMMs returned wrong attention in lanes
// 48-63 of single pixels, only with several blocks per CU, and only in builds where the compiler placed a (packed) VALU result
// in registers that had been MFMA accumulators a dozen instructions earlier.  Hypothesis: the MFMA's LAST pass writes lanes
// 48-63 of its last destination registers after the VALU instruction did -- i.e. the number of wait states the compiler leaves
// between an MFMA and a VALU write to its destination is enough when the MFMA starts at issue, and not enough when the matrix
// pipe is still busy with another wave's MFMA.
//
// One wave-loop iteration, in ONE asm block on fixed registers:
//     acc = 0;  v_mfma acc += A . B  (A = B = 1.0: every element of the product is K_total)
//     K x s_nop 0                                  (K wait states)
//     v_mov_b32 <last register of acc>, MARKER      (the write under test; PK = 1: v_pk_mul_f32 on the last PAIR)
//     4 x s_nop 15                                  (the MFMA is certainly done)
//     out = <last register of acc>
// A lane whose `out` is not MARKER was overwritten by the MFMA's late write.  Reported per quarter of the wave.
//   hipcc --offload-arch=gfx950 -O3 mfma_waw_probe.hip -o mfma_waw_probe && ./mfma_waw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// INSN 0: v_mfma_f32_32x32x16_f16 (acc v[32:47], 8 passes); 1: v_mfma_f32_16x16x32_f16 (acc v[32:35], 4 passes);
//      2: v_mfma_f32_32x32x2_f32 (acc v[32:47], 16 passes)
#define PROBE_BODY(KNOPS, MFMA, LAST, WRITE)                                                                        \
    asm volatile(                                                                                                    \
        "v_mov_b32 v48, %1\n v_mov_b32 v49, %1\n v_mov_b32 v50, %1\n v_mov_b32 v51, %1\n"                          \
        "v_mov_b32 v52, %1\n v_mov_b32 v53, %1\n v_mov_b32 v54, %1\n v_mov_b32 v55, %1\n"                          \
        "v_mov_b32 v56, %2\n v_mov_b32 v57, %2\n"                                                                  \
        "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n"                              \
        "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"                              \
        "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"                              \
        "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"                              \
        "s_nop 15\n s_nop 15\n"                                                                                      \
        MFMA "\n"                                                                                                    \
        ".rept " #KNOPS "\n s_nop 0\n .endr\n"                                                                       \
        WRITE "\n"                                                                                                   \
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"                                        \
        "v_mov_b32 %0, " LAST "\n"                                                                                   \
        : "=v"(got)                                                                                                  \
        : "v"(ones), "v"(marker)                                                                                     \
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", \
          "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57")

template <int INSN, int K, int PK>
__global__ void probe(unsigned *bad, int iters)
{
    const int lane = threadIdx.x & 63;
    const unsigned ones = INSN == 2 ? 0x3f800000u : 0x3c003c00u;   // 1.0f, or two fp16 ones
    const float marker = 12345.f;
    unsigned cnt = 0;
    for (int it = 0; it < iters; ++it) {
        float got;
        if constexpr (INSN == 0 && !PK) {
            switch (K) {
#define C_(KK) case KK: PROBE_BODY(KK, "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]", "v47", "v_mov_b32 v47, %2"); break;
                C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16) C_(18) C_(20) C_(24)
#undef C_
            }
        } else if constexpr (INSN == 0 && PK) {
            switch (K) {
#define C_(KK) case KK: PROBE_BODY(KK, "v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]", "v47", "v_pk_mul_f32 v[46:47], v[56:57], 1.0 op_sel_hi:[1,0]"); break;
                C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16) C_(18) C_(20) C_(24)
#undef C_
            }
        } else if constexpr (INSN == 1) {
            switch (K) {
#define C_(KK) case KK: PROBE_BODY(KK, "v_mfma_f32_16x16x32_f16 v[32:35], v[48:51], v[52:55], v[32:35]", "v35", "v_mov_b32 v35, %2"); break;
                C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16) C_(18) C_(20) C_(24)
#undef C_
            }
        } else {
            switch (K) {
#define C_(KK) case KK: PROBE_BODY(KK, "v_mfma_f32_32x32x2_f32 v[32:47], v48, v52, v[32:47]", "v47", "v_mov_b32 v47, %2"); break;
                C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16) C_(18) C_(20) C_(24)
#undef C_
            }
        }
        cnt += got != marker;
    }
    if (cnt) atomicAdd(&bad[lane >> 4], cnt);
}

// Second probe: a PACKED fp32 VALU instruction behind a burst of MFMAs, then a plain VALU read of its result after K wait
// states (read-after-write).  If the packed instruction is ordered behind the MFMAs in flight and the plain VALU pipe does not
// wait for it, the read returns the register's OLD value.
#define PK_BODY(KNOPS, NMFMA, PKOP)                                                                                  \
    asm volatile(                                                                                                    \
        "v_mov_b32 v48, %1\n v_mov_b32 v49, %1\n v_mov_b32 v50, %1\n v_mov_b32 v51, %1\n"                          \
        "v_mov_b32 v52, %1\n v_mov_b32 v53, %1\n v_mov_b32 v54, %1\n v_mov_b32 v55, %1\n"                          \
        "v_mov_b32 v56, %2\n v_mov_b32 v57, %2\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n"                            \
        "v_mov_b32 v60, %3\n v_mov_b32 v61, %3\n"                                                                  \
        "s_nop 15\n s_nop 15\n"                                                                                      \
        ".rept " #NMFMA "\n v_mfma_f32_32x32x16_f16 v[32:47], v[48:51], v[52:55], v[32:47]\n .endr\n"               \
        PKOP "\n"                                                                                                    \
        ".rept " #KNOPS "\n s_nop 0\n .endr\n"                                                                       \
        "v_mov_b32 %0, v61\n"                                                                                        \
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"                  \
        : "=v"(got)                                                                                                  \
        : "v"(ones), "v"(marker), "v"(oldv)                                                                          \
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", \
          "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61")

template <int OP, int NM, int K>
__global__ void probe_pk(unsigned *bad, int iters)
{
    const int lane = threadIdx.x & 63;
    const unsigned ones = 0x3c003c00u;
    const float marker = 12345.f, oldv = -7.f;
    unsigned cnt = 0;
    for (int it = 0; it < iters; ++it) {
        float got;
#define P_(KK, NN)                                                                                                   \
    if constexpr (K == KK && NM == NN) {                                                                             \
        if constexpr (OP == 0) PK_BODY(KK, NN, "v_pk_mul_f32 v[60:61], v[56:57], 1.0 op_sel_hi:[1,0]");             \
        else if constexpr (OP == 1) PK_BODY(KK, NN, "v_pk_add_f32 v[60:61], v[56:57], v[58:59]");                    \
        else PK_BODY(KK, NN, "v_mul_f32 v61, v57, 1.0");                                                             \
    }
        P_(0, 0) P_(1, 0) P_(2, 0) P_(4, 0) P_(8, 0) P_(0, 1) P_(1, 1) P_(2, 1) P_(4, 1) P_(8, 1) P_(16, 1) P_(0, 4) P_(1, 4) P_(2, 4)
        P_(4, 4) P_(8, 4) P_(16, 4) P_(32, 4)
#undef P_
        cnt += got != marker;
    }
    if (cnt) atomicAdd(&bad[lane >> 4], cnt);
}

template <int OP, int NM, int K>
void run_pk(unsigned *d_bad)
{
    unsigned h[2][4];
    for (int mode = 0; mode < 2; ++mode) {
        CHECK(hipMemset(d_bad, 0, 16));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_pk<OP, NM, K>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024));
        if (mode == 0) hipLaunchKernelGGL((probe_pk<OP, NM, K>), dim3(256), dim3(256), 150 * 1024, 0, d_bad, 2000);
        else hipLaunchKernelGGL((probe_pk<OP, NM, K>), dim3(512), dim3(1024), 0, 0, d_bad, 2000);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h[mode], d_bad, 16, hipMemcpyDeviceToHost));
    }
    printf("   %-12s %d MFMAs ahead, K = %2d | %9u %9u %9u %9u | %9u %9u %9u %9u\n",
           OP == 0 ? "v_pk_mul_f32" : OP == 1 ? "v_pk_add_f32" : "v_mul_f32", NM, K, h[0][0], h[0][1], h[0][2], h[0][3], h[1][0],
           h[1][1], h[1][2], h[1][3]);
}

template <int OP>
void sweep_pk(unsigned *d_bad)
{
    run_pk<OP, 0, 0>(d_bad); run_pk<OP, 0, 1>(d_bad); run_pk<OP, 0, 2>(d_bad); run_pk<OP, 0, 4>(d_bad); run_pk<OP, 0, 8>(d_bad);
    run_pk<OP, 1, 0>(d_bad); run_pk<OP, 1, 1>(d_bad); run_pk<OP, 1, 2>(d_bad); run_pk<OP, 1, 4>(d_bad); run_pk<OP, 1, 8>(d_bad);
    run_pk<OP, 1, 16>(d_bad);
    run_pk<OP, 4, 0>(d_bad); run_pk<OP, 4, 1>(d_bad); run_pk<OP, 4, 2>(d_bad); run_pk<OP, 4, 4>(d_bad); run_pk<OP, 4, 8>(d_bad);
    run_pk<OP, 4, 16>(d_bad); run_pk<OP, 4, 32>(d_bad);
}

// Third probe: an LDS store whose data registers are overwritten by a VALU instruction K wait states after the store was
// ISSUED, with NQ 16-byte LDS reads queued in front of it (write-after-read on the store's data).  The store must carry the
// OLD values whatever the queue holds.
#define DS_BODY(KNOPS, NQ, WRITE)                                                                                    \
    asm volatile(                                                                                                    \
        "v_mov_b32 v56, %3\n v_mov_b32 v57, %3\n v_mov_b32 v60, %2\n v_mov_b32 v61, %2\n"                          \
        "s_nop 7\n"                                                                                                  \
        ".rept " #NQ "\n ds_read_b128 v[64:67], %1 offset:2048\n .endr\n"                                           \
        "ds_write2_b32 %1, v60, v61 offset1:1\n"                                                                    \
        ".rept " #KNOPS "\n s_nop 0\n .endr\n"                                                                       \
        WRITE "\n"                                                                                                   \
        "s_waitcnt lgkmcnt(0)\n"                                                                                     \
        "ds_read2_b32 v[62:63], %1 offset1:1\n"                                                                     \
        "s_waitcnt lgkmcnt(0)\n"                                                                                     \
        "v_max_f32 %0, v62, v63\n"                                                                                   \
        : "=v"(got)                                                                                                  \
        : "v"(addr), "v"(oldv), "v"(marker)                                                                          \
        : "memory", "v56", "v57", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67")

template <int OP, int NQ, int K>
__global__ void probe_ds(unsigned *bad, int iters)
{
    extern __shared__ float s_probe[];
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)(threadIdx.x * 8);      // 8 bytes per lane, 8 KB per 1024 threads (+ the 2 KB read window)
    const float marker = 12345.f, oldv = -7.f;              // (max(old, old) = old; any lane holding the marker shows)
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_probe[i] = 0.f;
    __syncthreads();
    unsigned cnt = 0;
    for (int it = 0; it < iters; ++it) {
        float got;
#define D_(KK, QQ)                                                                                                   \
    if constexpr (K == KK && NQ == QQ) {                                                                             \
        if constexpr (OP == 0) DS_BODY(KK, QQ, "v_mov_b32 v60, %3\n v_mov_b32 v61, %3");                             \
        else DS_BODY(KK, QQ, "v_pk_mul_f32 v[60:61], v[56:57], 1.0 op_sel_hi:[1,0]");                                \
    }
        D_(0, 0) D_(1, 0) D_(2, 0) D_(4, 0) D_(0, 8) D_(1, 8) D_(2, 8) D_(4, 8) D_(8, 8) D_(0, 24) D_(1, 24) D_(2, 24) D_(4, 24) D_(8, 24)
#undef D_
        cnt += got != oldv;
    }
    if (cnt) atomicAdd(&bad[lane >> 4], cnt);
}

template <int OP, int NQ, int K>
void run_ds(unsigned *d_bad)
{
    unsigned h[2][4];
    for (int mode = 0; mode < 2; ++mode) {
        CHECK(hipMemset(d_bad, 0, 16));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_ds<OP, NQ, K>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024));
        if (mode == 0) hipLaunchKernelGGL((probe_ds<OP, NQ, K>), dim3(256), dim3(256), 150 * 1024, 0, d_bad, 2000);
        else hipLaunchKernelGGL((probe_ds<OP, NQ, K>), dim3(512), dim3(1024), 32 * 1024, 0, d_bad, 2000);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h[mode], d_bad, 16, hipMemcpyDeviceToHost));
    }
    printf("   %-12s %2d reads queued, K = %2d | %9u %9u %9u %9u | %9u %9u %9u %9u\n", OP == 0 ? "v_mov_b32 x2" : "v_pk_mul_f32", NQ, K,
           h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]);
}

template <int OP>
void sweep_ds(unsigned *d_bad)
{
    run_ds<OP, 0, 0>(d_bad); run_ds<OP, 0, 1>(d_bad); run_ds<OP, 0, 2>(d_bad); run_ds<OP, 0, 4>(d_bad);
    run_ds<OP, 8, 0>(d_bad); run_ds<OP, 8, 1>(d_bad); run_ds<OP, 8, 2>(d_bad); run_ds<OP, 8, 4>(d_bad); run_ds<OP, 8, 8>(d_bad);
    run_ds<OP, 24, 0>(d_bad); run_ds<OP, 24, 1>(d_bad); run_ds<OP, 24, 2>(d_bad); run_ds<OP, 24, 4>(d_bad); run_ds<OP, 24, 8>(d_bad);
}

template <int INSN, int PK>
void sweep(const char *name, unsigned *d_bad)
{
    // (a) one wave per SIMD: 256 threads, one block per CU (64 KB of LDS each would also do; the grid is just 256 blocks
    //     of 4 waves -- the dispatcher may still stack them, so the LDS request pins one block per CU)
    // (b) four waves per SIMD: 1024 threads per block, two blocks per CU
    printf("%s\n   K (wait states)  | alone: bad lanes by quarter            | 8 waves per SIMD: bad lanes by quarter\n", name);
#define RUN_(KK)                                                                                                     \
    {                                                                                                                \
        unsigned h[2][4];                                                                                            \
        for (int mode = 0; mode < 2; ++mode) {                                                                       \
            CHECK(hipMemset(d_bad, 0, 16));                                                                          \
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<INSN, KK, PK>),                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));                      \
            if (mode == 0) hipLaunchKernelGGL((probe<INSN, KK, PK>), dim3(256), dim3(256), 150 * 1024, 0, d_bad, 2000);  \
            else hipLaunchKernelGGL((probe<INSN, KK, PK>), dim3(512), dim3(1024), 0, 0, d_bad, 2000);                \
            CHECK(hipDeviceSynchronize());                                                                           \
            CHECK(hipMemcpy(h[mode], d_bad, 16, hipMemcpyDeviceToHost));                                             \
        }                                                                                                            \
        printf("   %2d               | %9u %9u %9u %9u | %9u %9u %9u %9u\n", KK, h[0][0], h[0][1], h[0][2], h[0][3],  \
               h[1][0], h[1][1], h[1][2], h[1][3]);                                                                  \
    }
    RUN_(0) RUN_(1) RUN_(2) RUN_(3) RUN_(4) RUN_(5) RUN_(6) RUN_(7) RUN_(8) RUN_(9) RUN_(10) RUN_(11) RUN_(12) RUN_(13) RUN_(14)
    RUN_(15) RUN_(16) RUN_(18) RUN_(20) RUN_(24)
#undef RUN_
}

int main()
{
    unsigned *d_bad;
    CHECK(hipMalloc(&d_bad, 16));
    if (getenv("WAW_ONLY_PK") == nullptr && getenv("WAW_ONLY_DS") == nullptr) {
    sweep<0, 0>("v_mfma_f32_32x32x16_f16, then v_mov_b32 into its last accumulator register", d_bad);
    sweep<0, 1>("v_mfma_f32_32x32x16_f16, then v_pk_mul_f32 into its last accumulator PAIR", d_bad);
    sweep<1, 0>("v_mfma_f32_16x16x32_f16, then v_mov_b32 into its last accumulator register", d_bad);
    sweep<2, 0>("v_mfma_f32_32x32x2_f32, then v_mov_b32 into its last accumulator register", d_bad);
    }
    printf("a VALU instruction behind MFMAs, its result read by v_mov_b32 after K wait states: lanes that read the OLD value\n"
           "                                          | alone, by quarter                       | 8 waves per SIMD, by quarter\n");
    if (getenv("WAW_ONLY_DS") == nullptr) {
    sweep_pk<0>(d_bad);
    sweep_pk<1>(d_bad);
    sweep_pk<2>(d_bad);
    }
    printf("ds_write2_b32 of two registers, overwritten K wait states after the store was issued: lanes whose store carried the NEW value\n"
           "                                          | alone, by quarter                       | 16 waves per CU x 2 blocks, by quarter\n");
    sweep_ds<0>(d_bad);
    sweep_ds<1>(d_bad);
    return 0;
}
