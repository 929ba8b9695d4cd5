// Split-fp16 products (device code; included inside the translation units' anonymous namespace): an fp32 value times a
// power-of-two scale is written as hi + lo in fp16 -- hi its top 11 significant bits (exact), lo the fp16 of the exact
// remainder -- and a product a b as hi lo' + lo hi' + hi hi' on the fp16 matrix cores with fp32 accumulation: the dropped
// lo lo' term is 2^-22 relative, i.e. fp32-GEMM-level error at a fifth of the fp32 MFMA time.
#pragma once
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Two scaled fp32 values -> packed fp16 hi (top 11 significant bits: conversion with round-toward-zero, exact) and
// packed fp16 lo (fp16 of the exact remainder v - hi, taken straight off the packed hi by v_fma_mix_f32: hi * -1 + v).
// 2.5 instructions per value with the scale multiply, against 3 for mask / subtract / two conversions.
__device__ __forceinline__ void split_f16_pair(float v0, float v1, unsigned &hi, unsigned &lo)
{
    typedef __fp16 h16x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, (h16x2_t)__builtin_amdgcn_cvt_pkrtz(v0, v1));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(v1));
    lo = __builtin_bit_cast(unsigned, f16x2_t{(_Float16)l0, (_Float16)l1});
}

// The same split with both halves of a value kept in ONE dword, ( hi | lo << 16 ): hi = RN16(v) (v_cvt_pk_f16_f32, round
// to nearest: |v - hi| <= 2^-12 |v|), lo = RN16(v - hi) (the remainder is exact in fp32), so hi + lo carries >= 22
// significant bits of v while |lo| stays above fp16's subnormal range, and 2^-25 of the scale's range below it.  With the
// other operand held the same way, ( hi', lo' ),
//     mfma(a, b) + mfma(a, rot16(b))  =  sum_k  (hi'_k hi_k + lo'_k lo_k) + (hi'_k lo_k + lo'_k hi_k)  =  sum_k a_k b_k
// -- all four cross terms from TWO fp16 MFMAs, and one v_alignbit per operand dword instead of a conversion.
// 2.5 instructions per value (the second conversion re-derives hi: same instruction, same rounding, same bits).
__device__ __forceinline__ void split_hilo_pair(float v0, float v1, unsigned &d0, unsigned &d1)
{
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v0, v1}, f16x2_t));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h2), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h2), "v"(v1));
    d0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v0, l0}, f16x2_t));
    d1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v1, l1}, f16x2_t));
}

// eight scaled fp32 values -> fp16 hi and lo; amax tracks the largest magnitude seen (the overflow guard)
template <bool GUARD>
__device__ __forceinline__ void split_f16x8(const float (&v)[8], f16x8 &hi, f16x8 &lo, float &amax)
{
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hh, ll;
        split_f16_pair(v[2 * i], v[2 * i + 1], hh, ll);
        h[i] = hh;
        l[i] = ll;
        if (GUARD) amax = fmaxf(amax, fmaxf(fabsf(v[2 * i]), fabsf(v[2 * i + 1])));
    }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

// The split of eight values for the second GEMM's source rows, on the register pairs the loads and the lane trade leave them in
// (v[i] = (row i of the even channel, row i of the odd channel)): the scale multiply stays packed on those pairs.
// Paired any other way (e.g. rows i, i + 1 of one channel as a packed multiply) every k-step copies its freshly loaded
// registers into the other arrangement right behind the loads -- i.e. waits for them.
__device__ __forceinline__ void split_f16x8_pairs(const f32x2 (&v)[8], f16x8 &ehi, f16x8 &elo, f16x8 &ohi, f16x8 &olo)
{
    u32x4 eh, el, oh, ol;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hh, ll;
        split_f16_pair(v[2 * i][0], v[2 * i + 1][0], hh, ll);
        eh[i] = hh;
        el[i] = ll;
        split_f16_pair(v[2 * i][1], v[2 * i + 1][1], hh, ll);
        oh[i] = hh;
        ol[i] = ll;
    }
    ehi = __builtin_bit_cast(f16x8, eh);
    elo = __builtin_bit_cast(f16x8, el);
    ohi = __builtin_bit_cast(f16x8, oh);
    olo = __builtin_bit_cast(f16x8, ol);
}
// power of two that puts a magnitude mx into [2^10, 2^11) (fp16 keeps a factor 32 of head-room above it), and its
// inverse; mx = 0 / denormal / huge are clamped to a finite pair
__device__ __forceinline__ void pow2_scale_of(float mx, float &sc, float &inv)
{
    int E = (int)((__float_as_uint(mx) >> 23) & 255u);
    E = min(max(E, 12), 250);
    sc = __uint_as_float((unsigned)(264 - E) << 23);      // 2^(137 - E)
    inv = __uint_as_float((unsigned)(E - 10) << 23);      // 2^(E - 137)
}

// the same scale with its exponent capped to [-60, 60]: products of two such scales (and their inverses) stay finite
__device__ __forceinline__ float pow2_scale_capped(float mx)
{
    int E = (int)((__float_as_uint(mx) >> 23) & 255u);
    E = min(max(E, 77), 197);
    return __uint_as_float((unsigned)(264 - E) << 23);    // 2^(137 - E)
}

// a scaled value must stay below this for the split (fp16 max 65504): the guard of kernels that scale by an ESTIMATE
constexpr float kF16GuardLimit = 32768.f;
