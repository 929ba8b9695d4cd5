// Wave-wide reductions on DPP / permlane swaps (device code; included inside the translation units' anonymous namespace).
#pragma once
// ---- wave-wide all-reduces on DPP / permlane swaps (no LDS round trips) ------------------------
template <class Op>
__device__ __forceinline__ float wave_all(float v, Op op)
{
    v = op(v, dpp<0x128>(v));  // row_ror:8
    v = op(v, dpp<0x124>(v));  // row_ror:4
    v = op(v, dpp<0x122>(v));  // row_ror:2
    v = op(v, dpp<0x121>(v));  // row_ror:1   -> every lane of a 16-lane row holds the row's result
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));  // rows 0|1 and 2|3 combined
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return op(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float wave_all_sum(float v) { return wave_all(v, [](float a, float b) { return a + b; }); }

// max / min over the wave, FOUR independent values at a time.  fmaxf() makes the compiler quiet both inputs
// first (v_max x,x,x), keep the DPP move separate and pad every dependent step with s_nop -- ~45 instructions
// per reduction.  Spelled out and interleaved four wide, each butterfly step is one DPP-modified instruction
// and the three instructions of the other chains between two dependent steps cover the VALU-write ->
// DPP / permlane-read hazards (2 wait states), so the block needs a single s_nop at its head.
// (No NaN can reach here that the reference would not propagate as well.)
#define ET_STEP4(OPC, CTRL)                                                  \
    OPC "_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"            \
    OPC "_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"            \
    OPC "_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"            \
    OPC "_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define ET_SWAP4(OPC, SWAP)                                                                              \
    "v_mov_b32 %4, %0\n\tv_mov_b32 %5, %1\n\tv_mov_b32 %6, %2\n\tv_mov_b32 %7, %3\n\t"                  \
    SWAP " %0, %4\n\t" SWAP " %1, %5\n\t" SWAP " %2, %6\n\t" SWAP " %3, %7\n\t"                         \
    OPC " %0, %0, %4\n\t" OPC " %1, %1, %5\n\t" OPC " %2, %2, %6\n\t" OPC " %3, %3, %7\n\t"
#define ET_WAVE_ALL4(NAME, OPC)                                                                          \
    __device__ __forceinline__ void NAME(float (&v)[4])                                                  \
    {                                                                                                    \
        float t0, t1, t2, t3;                                                                            \
        asm("s_nop 1\n\t" ET_STEP4(OPC, "row_ror:8") ET_STEP4(OPC, "row_ror:4") ET_STEP4(OPC, "row_ror:2") \
            ET_STEP4(OPC, "row_ror:1") ET_SWAP4(OPC, "v_permlane16_swap_b32") ET_SWAP4(OPC, "v_permlane32_swap_b32") \
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));  \
    }
ET_WAVE_ALL4(wave_all_max4, "v_max_f32")
ET_WAVE_ALL4(wave_all_min4, "v_min_f32")
ET_WAVE_ALL4(wave_all_sum4, "v_add_f32")
#undef ET_WAVE_ALL4
#undef ET_SWAP4
#undef ET_STEP4
__device__ __forceinline__ float wave_all_max(float v)
{
    float q[4] = {v, v, v, v};
    wave_all_max4(q);
    return q[0];
}
__device__ __forceinline__ float wave_all_min(float v)
{
    float q[4] = {v, v, v, v};
    wave_all_min4(q);
    return q[0];
}
