// MI355X (gfx950 / CDNA4) kernels of the Epipolar Transformer hot path and the
// C ABI declared in include/epipolar_amd.h.  Written for wave64 only.
//
// Work decomposition (DESIGN.md "Kernels"):
//   * one wavefront owns one reference pixel at a time; lanes <-> channels
//     (float4 per lane, so C = 256 is exactly one 1 KiB coalesced row per tap);
//   * the K samples of the pixel's epipolar segment are set up with
//     lanes <-> samples (bit-faithful float32 geometry, epipolar_geometry.h) and
//     broadcast back sample by sample with v_readlane;
//   * taps are held in a 2x2 parity-addressed, tag-checked register cache so a
//     source row is fetched once per pixel, not once per sample that touches it;
//   * samples are processed in batches of 8: eight partial dot products are
//     summed across the wave with a transposing butterfly (v_permlane32_swap,
//     v_permlane16_swap, DPP), the masked soft-max is folded in online
//     (running max + rescale) so the K x C sampled strip never exists anywhere;
//   * the attention tile of 16 consecutive pixels is staged in LDS and written
//     as 64-byte rows of the reference's (N,K,H,W) `depth` layout;
//   * blockIdx is remapped so each XCD walks whole pairs (its L2 keeps the
//     4 MiB source map of the pair it is working on).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>

#include "epipolar_amd.h"
#include "epipolar_geometry.h"

namespace {

// ----------------------------------------------------------------------------
// error reporting
// ----------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

// ----------------------------------------------------------------------------
// constants
// ----------------------------------------------------------------------------
constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kPixPerWave = 4;
constexpr int kPixPerBlock = kWavesPerBlock * kPixPerWave;  // 16 consecutive pixels
constexpr int kXcds = 8;

struct FwdParams {
    EtLayerDesc d;
    const float *xs, *ys, *steps, *cam;
    const float *fref, *fsrc;
    float *out, *attn, *corr;
    const float *res_bias;
    float *res_base;
    int blocks_per_pair;
    int total_blocks;
    int interleave;
    int ablate;  // profiling only: 1 = issue no tap loads after the first sample, 2 = every tap reads row 0
};

struct BwdParams {
    EtLayerDesc d;
    const float *xs, *ys, *steps, *cam;
    const float *fref, *fsrc, *gout;
    float *gref, *gsrc;
    int blocks_per_pair;
    int total_blocks;
    // gather-form backward (workspace given): per-(pixel, source row) coefficient entries
    int cap;            // entry slots per reference pixel (4 * K)
    int *ent_u;         // [N*HW*cap] source pixel index of the entry
    float *ent_a;       // [N*HW*cap] alpha = sum_k w_ku a_k      (value path,      OTHER_GRAD 'other2')
    float *ent_b;       // [N*HW*cap] beta  = sum_k w_ku ds_k     (similarity path, OTHER_GRAD 'other1')
    int *ent_count;     // [N*HW]     entries emitted by each reference pixel
    int *row_count;     // [N*HW]     entries received by each source pixel (zeroed per call)
};

// ----------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------
// Bijective XCD-aware remap: hardware places block b on XCD b % 8; give each
// XCD a contiguous chunk of the logical grid (guide T1, bijective form).
__device__ __forceinline__ int xcd_remap(int b, int nwg)
{
    const int q = nwg / kXcds, r = nwg % kXcds;
    const int xcd = b % kXcds;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + b / kXcds;
}

__device__ __forceinline__ float lane_bcast(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// Lanes whose `bit` is clear keep a, the others keep b; each adds the value its
// partner (lane ^ bit) did not keep.  Building block of the transposing sum.
__device__ __forceinline__ float xstep_safe(float a, float b, int lane, int bit)
{
    const bool hi = (lane & bit) != 0;
    const float keep = hi ? b : a;
    const float send = hi ? a : b;
    return keep + __shfl_xor(send, bit);
}

// Sum eight per-lane partials over the 64 lanes at once.  Result: the 8-lane
// group g = lane >> 3 holds (replicated) the total of partial j = bitrev3(g),
// i.e. j = ((lane >> 5) & 1) | ((lane >> 4) & 1) << 1 | ((lane >> 3) & 1) << 2.
template <bool FAST>
__device__ __forceinline__ float reduce8(const float (&p)[8], int lane)
{
    if constexpr (FAST) {
        float q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // v_permlane32_swap: a' = [a.lo, b.lo], b' = [a.hi, b.hi]
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[2 * i]), __float_as_uint(p[2 * i + 1]),
                                                      false, false);
            q[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        float t[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // v_permlane16_swap: odd rows of a <-> even rows of b
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(q[2 * i]), __float_as_uint(q[2 * i + 1]),
                                                      false, false);
            t[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        const bool hi = (lane & 8) != 0;
        const float keep = hi ? t[1] : t[0];
        const float send = hi ? t[0] : t[1];
        float u = keep + dpp<0x128>(send);  // row_ror:8  == lane ^ 8 inside a row of 16
        u += dpp<0xB1>(u);                  // quad_perm [1,0,3,2]  (lane ^ 1)
        u += dpp<0x4E>(u);                  // quad_perm [2,3,0,1]  (lane ^ 2)
        u += dpp<0x141>(u);                 // row_half_mirror      (7 - lane inside 8)
        return u;
    } else {
        const float q0 = xstep_safe(p[0], p[1], lane, 32);
        const float q1 = xstep_safe(p[2], p[3], lane, 32);
        const float q2 = xstep_safe(p[4], p[5], lane, 32);
        const float q3 = xstep_safe(p[6], p[7], lane, 32);
        const float t0 = xstep_safe(q0, q1, lane, 16);
        const float t1 = xstep_safe(q2, q3, lane, 16);
        float u = xstep_safe(t0, t1, lane, 8);
        u += __shfl_xor(u, 4);
        u += __shfl_xor(u, 2);
        u += __shfl_xor(u, 1);
        return u;
    }
}

// Four-partial variant: the 16-lane row r = lane >> 4 holds the total of partial
// j = ((lane >> 5) & 1) | ((lane >> 4) & 1) << 1.
template <bool FAST>
__device__ __forceinline__ float reduce4(const float (&p)[4], int lane)
{
    if constexpr (FAST) {
        float q[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[2 * i]), __float_as_uint(p[2 * i + 1]),
                                                      false, false);
            q[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(q[0]), __float_as_uint(q[1]), false, false);
        float u = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        u += dpp<0x128>(u);  // row_ror:8
        u += dpp<0x124>(u);  // row_ror:4
        u += dpp<0x122>(u);  // row_ror:2
        u += dpp<0x121>(u);  // row_ror:1
        return u;
    } else {
        const float q0 = xstep_safe(p[0], p[1], lane, 32);
        const float q1 = xstep_safe(p[2], p[3], lane, 32);
        float u = xstep_safe(q0, q1, lane, 16);
        u += __shfl_xor(u, 8);
        u += __shfl_xor(u, 4);
        u += __shfl_xor(u, 2);
        u += __shfl_xor(u, 1);
        return u;
    }
}

template <int BATCH, bool FAST>
__device__ __forceinline__ float reduce_batch(const float (&p)[BATCH], int lane)
{
    static_assert(BATCH == 4 || BATCH == 8, "batch of 4 or 8 samples");
    if constexpr (BATCH == 8) return reduce8<FAST>(p, lane);
    else return reduce4<FAST>(p, lane);
}

// lane that holds batch sample j after reduce_batch, and the sample a lane holds
template <int BATCH>
__host__ __device__ constexpr int lane_of_sample(int j)
{
    return BATCH == 8 ? 32 * (j & 1) + 16 * ((j >> 1) & 1) + 8 * ((j >> 2) & 1) : 32 * (j & 1) + 16 * ((j >> 1) & 1);
}

template <int BATCH>
__device__ __forceinline__ int sample_of_lane(int lane)
{
    const int j = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1);
    return BATCH == 8 ? (j | (((lane >> 3) & 1) << 2)) : j;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// max over the lane groups of a batch (values already uniform inside a group)
template <int BATCH>
__device__ __forceinline__ float group_max(float v)
{
    if constexpr (BATCH == 8) v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Buffer addressing (guide T8/T20): the per-pair map is one raw buffer resource
// held in SGPRs; a tap row is selected by the SCALAR byte offset (soffset) and
// the lane's channel group by a constant 32-bit VGPR offset, so a tap fetch
// costs no vector address arithmetic at all.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ float4 buf_load_f4(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float buf_load_f1(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

__device__ __forceinline__ float4 f4_fma(float s, const float4 &a, const float4 &c)
{
    return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

__device__ __forceinline__ float4 f4_mul(float s, const float4 &a)
{
    return make_float4(s * a.x, s * a.y, s * a.z, s * a.w);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// a . b through packed fp32: (a.xy * b.xy), fma with (a.zw, b.zw), one add
__device__ __forceinline__ float f4_dot(const float4 &a, const float4 &b)
{
    const f32x2 lo = f32x2{a.x, a.y} * f32x2{b.x, b.y};
    const f32x2 t = __builtin_elementwise_fma(f32x2{a.z, a.w}, f32x2{b.z, b.w}, lo);
    return t.x + t.y;
}

__device__ __forceinline__ float f4_dot_acc(const float4 &a, const float4 &b, float acc)
{
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    return fmaf(a.w, b.w, acc);
}

// ----------------------------------------------------------------------------
// per-pixel sample table, lanes <-> samples
// ----------------------------------------------------------------------------
// Everything a sample needs inside the channel loop is precomputed here, 64
// samples at a time, so that the loop itself spends one v_readlane + a handful
// of scalar bit tests per sample on bookkeeping:
//   off[r]  byte offset of the source row routed to tap register r (-1: none)
//   w[r]    its bilinear weight (0 for out-of-image taps and for k >= K)
//   need    bit r set when register r does not already hold that row, i.e.
//           the previous sample's tap r was a different row
template <int KPL>
struct SampleTable {
    int off[KPL][4];
    float w[KPL][4];
    int need[KPL];
    float nx[KPL], ny[KPL];
};

template <int KPL, bool CACHE>
__device__ __forceinline__ void build_sample_table(const EtLayerDesc &d, const et::Segment &seg,
                                                   const float *__restrict__ steps, int lane, int row_bytes,
                                                   SampleTable<KPL> &t)
{
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
        const int k = s * kWave + lane;
        const bool in = k < d.K;
        const et::SampleSetup su = et::sample_setup(d, seg, in ? steps[k] : 0.f);
        t.nx[s] = su.nx;
        t.ny[s] = su.ny;
        int need = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int off = (in && su.tap[r] >= 0) ? su.tap[r] * row_bytes : -1;
            t.off[s][r] = off;
            t.w[s][r] = in ? su.weight[r] : 0.f;
            int prev = __shfl_up(off, 1);
            if (lane == 0) prev = -1;
            if (s > 0) {
                const int carry = __shfl(t.off[s > 0 ? s - 1 : 0][r], kWave - 1);
                if (lane == 0) prev = carry;
            }
            if (off >= 0 && (!CACHE || off != prev)) need |= 1 << r;
        }
        t.need[s] = need;
    }
}

// ----------------------------------------------------------------------------
// forward: fused epipolar sample + dot + masked softmax + weighted sum
// ----------------------------------------------------------------------------
// CPL: float4 channel groups per lane (C <= 256*CPL); KPL: samples per lane
// (K <= 64*KPL); BATCH: samples per cross-lane reduction; FAST: permlane/DPP
// reductions; CACHE: 2x2 tap register cache; MINW: waves per SIMD the register
// allocator must leave room for (__launch_bounds__ second argument); RAGGED: K is
// not a multiple of BATCH, so batches carry per-sample validity logic.
template <int CPL, int KPL, int BATCH, bool FAST, bool CACHE, int MINW, bool RAGGED>
__global__ __launch_bounds__(kWave *kWavesPerBlock, MINW) void epipolar_fwd_kernel(const FwdParams p)
{
    // dynamic LDS: [K][16] attention tile (only when attn is requested), then one
    // [KPL*64] float4 table of bilinear weights per wave
    extern __shared__ float s_dyn[];
    const EtLayerDesc &d = p.d;
    const int H = d.H, W = d.W, C = d.C, K = d.K;
    const int HW = H * W;
    const int nvec = C >> 2;  // float4 groups per pixel row

    const int vb = xcd_remap(blockIdx.x, p.total_blocks);
    const int n = vb / p.blocks_per_pair;
    const int pb = vb - n * p.blocks_per_pair;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pix_base = pb * kPixPerBlock;
    float *s_attn = s_dyn;
    float4 *s_wt = reinterpret_cast<float4 *>(s_dyn + (p.attn ? K * kPixPerBlock : 0)) + wave * (KPL * kWave);

    const float *cam = p.cam + (size_t)n * ET_CAM_STRIDE;
    const __amdgpu_buffer_rsrc_t src = make_rsrc(p.fsrc + (size_t)n * HW * C, (unsigned)HW * C * 4u);
    const int row_bytes = C * 4;
    const float neg_inf = -__builtin_huge_valf();
    // Lanes beyond C/4 (only when C < 256*CPL) re-read the last channel group:
    // their reference features are zeroed and their results never stored, so
    // the hot loop carries no per-lane predicate and stays on scalar branches.
    int voff[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) voff[c] = min(lane + c * kWave, nvec - 1) * 16;

    for (int pp = 0; pp < kPixPerWave; ++pp) {
        // consecutive pixels either per wave (0..3 | 4..7 | ..) or interleaved across the block's
        // waves (wave w takes pixels w, w+4, ..), which keeps the four waves on neighbouring
        // epipolar lines at the same time (shared rows hit in the CU's L1)
        const int slot_in_block = p.interleave ? pp * kWavesPerBlock + wave : wave * kPixPerWave + pp;
        const int pix = pix_base + slot_in_block;
        if (pix >= HW) break;  // wave-uniform
        const int h = pix / W, w = pix - h * W;

        // ---- lanes <-> samples: geometry and tap table ----------------------
        const et::Segment seg = et::epipolar_segment(d, cam, p.xs[w], p.ys[h]);
        SampleTable<KPL> tb;
        build_sample_table<KPL, CACHE>(d, seg, p.steps, lane, row_bytes, tb);
        if (p.ablate) {  // roofline ablations (results are wrong by construction)
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                if (p.ablate == 1 && (s > 0 || lane > 0)) tb.need[s] = 0;
                if (p.ablate == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) tb.off[s][r] = 0;
                }
            }
        }
        float v_sim[KPL];
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            v_sim[s] = neg_inf;
            // a sample's four weights come back as ONE broadcast ds_read_b128 (same address in
            // every lane) instead of four v_readlane
            s_wt[s * kWave + lane] = make_float4(tb.w[s][0], tb.w[s][1], tb.w[s][2], tb.w[s][3]);
        }
        __builtin_amdgcn_wave_barrier();

        // ---- lanes <-> channels -------------------------------------------
        float4 f1[CPL], acc[CPL], R[4][CPL];
        const float4 *ref = reinterpret_cast<const float4 *>(p.fref + ((size_t)n * HW + pix) * C);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int v = lane + c * kWave;
            f1[c] = (v < nvec) ? ref[v] : f4_zero();
            acc[c] = f4_zero();
#pragma unroll
            for (int r = 0; r < 4; ++r) R[r][c] = f4_zero();
        }
        if (p.res_base) {
            // additive term of the residual fusion, while the reference row is still in registers
            float4 *b4 = reinterpret_cast<float4 *>(p.res_base + ((size_t)n * HW + pix) * C);
            const float4 *bias4 = reinterpret_cast<const float4 *>(p.res_bias);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int v = lane + c * kWave;
                if (v < nvec) {
                    float4 r = f1[c];
                    if (bias4) {
                        const float4 bb = bias4[v];
                        r = make_float4(r.x + bb.x, r.y + bb.y, r.z + bb.z, r.w + bb.w);
                    }
                    b4[v] = r;
                }
            }
        }
        float m_run = neg_inf;

#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int kcount = min(kWave, K - s * kWave);  // samples held in this slot (uniform)
            for (int kb = 0; kb < kcount; kb += BATCH) {
                float4 S[BATCH][CPL];
                float part[BATCH];
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    const int kk = kb + j;
                    if (RAGGED) {
                        part[j] = 0.f;
#pragma unroll
                        for (int c = 0; c < CPL; ++c) S[j][c] = f4_zero();
                    }
                    // (the branch is kept even when it is always taken: per-sample basic blocks stop the
                    //  compiler from renaming the tap registers across samples, which costs ~100 VGPRs)
                    if (kk < kcount) {  // wave-uniform
                        const int need = __builtin_amdgcn_readlane(tb.need[s], kk);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (need & (1 << r)) {
                                const int off = __builtin_amdgcn_readlane(tb.off[s][r], kk);
#pragma unroll
                                for (int c = 0; c < CPL; ++c) R[r][c] = buf_load_f4(src, voff[c], off);
                            }
                        }
                        const float4 wv = s_wt[s * kWave + kk];
                        const float w0 = wv.x, w1 = wv.y, w2 = wv.z, w3 = wv.w;
#pragma unroll
                        for (int c = 0; c < CPL; ++c) {
                            float4 sv = f4_mul(w0, R[0][c]);
                            sv = f4_fma(w1, R[1][c], sv);
                            sv = f4_fma(w2, R[2][c], sv);
                            sv = f4_fma(w3, R[3][c], sv);
                            S[j][c] = sv;
                            part[j] = (c == 0) ? f4_dot(sv, f1[c]) : part[j] + f4_dot(sv, f1[c]);
                        }
                    }
                }
                // eight dot products -> 8-lane groups
                const float u = reduce_batch<BATCH, FAST>(part, lane);
                const bool jvalid = !RAGGED || (kb + sample_of_lane<BATCH>(lane)) < kcount;
                float sv = (u == 0.f) ? -1e10f : u;  // epipolar.py:298
                float e;
                if (d.softmax_enabled) {
                    sv = sv * d.softmax_scale;  // epipolar.py:306
                    const float bm = group_max<BATCH>(jvalid ? sv : neg_inf);
                    const float m_new = fmaxf(m_run, bm);
                    // accumulator weights only need ~1e-6 relative accuracy (the returned attention is
                    // recomputed with expf in the final pass): hardware exp2
                    const float alpha = __expf(m_run - m_new);
                    e = jvalid ? __expf(sv - m_new) : 0.f;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) acc[c] = f4_mul(alpha, acc[c]);
                    m_run = m_new;
                } else {
                    sv = sv / (float)K;  // epipolar.py:311
                    e = jvalid ? sv : 0.f;
                }
                // keep the logit of sample (kb + j) in lane (kb + j) for the final pass
                const float mine = __shfl(sv, lane_of_sample<BATCH>(lane & (BATCH - 1)));
                if ((lane / BATCH) == (kb / BATCH)) v_sim[s] = mine;
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    const float ej = lane_bcast(e, lane_of_sample<BATCH>(j));
#pragma unroll
                    for (int c = 0; c < CPL; ++c) acc[c] = f4_fma(ej, S[j][c], acc[c]);
                }
            }
        }

        // ---- lanes <-> samples: exact soft-max, arg-max, outputs ------------
        float a[KPL];
        float denom = 1.f;
        if (d.softmax_enabled) {
            float lsum = 0.f;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                const int k = s * kWave + lane;
                a[s] = (k < K) ? expf(v_sim[s] - m_run) : 0.f;  // m_run == max_k logit
                lsum += a[s];
            }
            denom = wave_sum(lsum);
#pragma unroll
            for (int s = 0; s < KPL; ++s) a[s] = a[s] / denom;
        } else {
#pragma unroll
            for (int s = 0; s < KPL; ++s) a[s] = v_sim[s];
        }
        // first maximum over k (torch.argmax), value then lowest index
        float bestv = neg_inf;
        int besti = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int k = s * kWave + lane;
            if (k < K && (a[s] > bestv)) {
                bestv = a[s];
                besti = k;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(bestv, m);
            const int oi = __shfl_xor(besti, m);
            if (ov > bestv || (ov == bestv && oi < besti)) {
                bestv = ov;
                besti = oi;
            }
        }
        if (p.corr) {
            float bx = 0.f, by = 0.f;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                const float tx = __shfl(tb.nx[s], besti & (kWave - 1));
                const float ty = __shfl(tb.ny[s], besti & (kWave - 1));
                if ((besti >> 6) == s) {
                    bx = tx;
                    by = ty;
                }
            }
            if (lane == 0) {
                float *o = p.corr + ((size_t)n * HW + pix) * 2;
                o[0] = et::de_normalize(d, bx, W);
                o[1] = et::de_normalize(d, by, H);
            }
        }
        if (p.attn) {
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                const int k = s * kWave + lane;
                if (k < K) s_attn[k * kPixPerBlock + slot_in_block] = a[s];
            }
        }
        float4 *o4 = reinterpret_cast<float4 *>(p.out + ((size_t)n * HW + pix) * C);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int v = lane + c * kWave;
            float4 r = acc[c];
            if (d.softmax_enabled) r = make_float4(r.x / denom, r.y / denom, r.z / denom, r.w / denom);
            if (v < nvec) o4[v] = r;
        }
    }

    if (p.attn) {
        __syncthreads();
        // (N,K,H,W): for a fixed k the 16 pixels of this block are contiguous
        const int npix = min(kPixPerBlock, HW - pix_base);
        float *dst = p.attn + (size_t)n * K * HW + pix_base;
        if (npix == kPixPerBlock && (HW & 3) == 0) {
            for (int t = threadIdx.x; t < K * 4; t += blockDim.x) {
                const int k = t >> 2, q = t & 3;
                const float4 v = *reinterpret_cast<const float4 *>(&s_attn[k * kPixPerBlock + q * 4]);
                *reinterpret_cast<float4 *>(dst + (size_t)k * HW + q * 4) = v;
            }
        } else {
            for (int t = threadIdx.x; t < K * kPixPerBlock; t += blockDim.x) {
                const int k = t / kPixPerBlock, i = t % kPixPerBlock;
                if (i < npix) dst[(size_t)k * HW + i] = s_attn[k * kPixPerBlock + i];
            }
        }
    }
}

// ----------------------------------------------------------------------------
// forward, several pixels per wave ("multi"): PPW neighbouring pixels advance in
// lockstep, LPP = 64 / PPW lanes each, CQ float4 channel groups per lane.
// ----------------------------------------------------------------------------
// Why: in the one-pixel-per-wave kernel only ~13 of ~36 VALU instructions per sample
// are arithmetic; the rest is per-sample bookkeeping (lane broadcasts, 64-lane
// reductions, soft-max updates) that here is issued once per step for PPW pixels:
// the per-step record (tap offsets, weights, need bits) is read from LDS by each
// lane group, tap loads are exec-masked per group, the dot product reduces over
// LPP lanes only, and the online soft-max runs one sample per step (no batching).
// Requires C == 4 * LPP * CQ exactly (C = 256: PPW 2 / CQ 2 or PPW 4 / CQ 4).
template <int PPW, int CQ, int KPL, bool PIPE>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void epipolar_fwd_multi_kernel(const FwdParams p)
{
    constexpr int LPP = kWave / PPW;
    constexpr int ROUNDS = kPixPerWave / PPW;
    constexpr int KP = KPL * kWave;  // padded samples per pixel
    extern __shared__ float s_dyn[];
    const EtLayerDesc &d = p.d;
    const int H = d.H, W = d.W, C = d.C, K = d.K;
    const int HW = H * W;

    const int vb = xcd_remap(blockIdx.x, p.total_blocks);
    const int n = vb / p.blocks_per_pair;
    const int pb = vb - n * p.blocks_per_pair;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pix_base = pb * kPixPerBlock;
    const int g = lane / LPP, li = lane % LPP;

    // LDS carve-up: attention tile, then per wave: offsets, weights, need bits, logits, segments
    float *s_attn = s_dyn;
    char *wbase = reinterpret_cast<char *>(s_dyn + (p.attn ? K * kPixPerBlock : 0)) +
                  (size_t)wave * (PPW * KP * 40 + PPW * 16);
    int4 *s_off = reinterpret_cast<int4 *>(wbase);                       // [PPW][KP]
    float4 *s_wt = reinterpret_cast<float4 *>(wbase + PPW * KP * 16);    // [PPW][KP]
    int *s_need = reinterpret_cast<int *>(wbase + PPW * KP * 32);        // [PPW][KP]
    float *s_sim = reinterpret_cast<float *>(wbase + PPW * KP * 36);     // [PPW][KP]
    float4 *s_seg = reinterpret_cast<float4 *>(wbase + PPW * KP * 40);   // [PPW]

    const float *cam = p.cam + (size_t)n * ET_CAM_STRIDE;
    const __amdgpu_buffer_rsrc_t src = make_rsrc(p.fsrc + (size_t)n * HW * C, (unsigned)HW * C * 4u);
    const int row_bytes = C * 4;
    const float neg_inf = -__builtin_huge_valf();
    const int lane_off = li * 16;  // + c * LPP * 16 as the instruction's immediate offset

    for (int round = 0; round < ROUNDS; ++round) {
        const int first = pix_base + wave * kPixPerWave + round * PPW;  // first pixel of this wave's group
        if (first >= HW) break;                                         // wave-uniform
        // ---- the PPW epipolar segments at once: lane group g evaluates pixel g's ------------
        {
            const int pixg = min(first + g, HW - 1);
            const int hg = pixg / W, wg = pixg - hg * W;
            const et::Segment sg = et::epipolar_segment(d, cam, p.xs[wg], p.ys[hg]);
            if (li == 0) s_seg[g] = make_float4(sg.sx, sg.sy, sg.vx, sg.vy);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lanes <-> samples, one pixel after the other: records into LDS ----
#pragma unroll
        for (int gg = 0; gg < PPW; ++gg) {
            const bool live = first + gg < HW;
            const float4 sq = s_seg[gg];
            et::Segment seg;
            seg.sx = sq.x; seg.sy = sq.y; seg.vx = sq.z; seg.vy = sq.w;
            SampleTable<KPL> tb;
            build_sample_table<KPL, true>(d, seg, p.steps, lane, row_bytes, tb);
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                const int k = s * kWave + lane;
                s_off[gg * KP + k] = make_int4(tb.off[s][0], tb.off[s][1], tb.off[s][2], tb.off[s][3]);
                s_wt[gg * KP + k] = make_float4(tb.w[s][0], tb.w[s][1], tb.w[s][2], tb.w[s][3]);
                s_need[gg * KP + k] = live ? tb.need[s] : 0;
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- lanes <-> (pixel, channels) ------------------------------------------
        const int mypix = first + g;
        const bool mylive = mypix < HW;
        float4 f1[CQ], acc[CQ], R[4][CQ];
        const float4 *ref = reinterpret_cast<const float4 *>(p.fref + ((size_t)n * HW + (mylive ? mypix : 0)) * C);
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            f1[c] = mylive ? ref[c * LPP + li] : f4_zero();
            acc[c] = f4_zero();
#pragma unroll
            for (int r = 0; r < 4; ++r) R[r][c] = f4_zero();
        }
        if (p.res_base && mylive) {
            float4 *b4 = reinterpret_cast<float4 *>(p.res_base + ((size_t)n * HW + mypix) * C);
            const float4 *bias4 = reinterpret_cast<const float4 *>(p.res_bias);
#pragma unroll
            for (int c = 0; c < CQ; ++c) {
                float4 r = f1[c];
                if (bias4) {
                    const float4 bb = bias4[c * LPP + li];
                    r = make_float4(r.x + bb.x, r.y + bb.y, r.z + bb.z, r.w + bb.w);
                }
                b4[c * LPP + li] = r;
            }
        }
        float m_run = neg_inf;
        const int rec0 = g * KP;

        // request (exec-masked per lane group) the tap rows step k does not have yet
        auto issue_step = [&](int k) {
            const int need = (p.ablate == 1 && k > 0) ? 0 : s_need[rec0 + k];
            const int4 off = s_off[rec0 + k];
            if (need & 1) {
#pragma unroll
                for (int c = 0; c < CQ; ++c) R[0][c] = buf_load_f4(src, off.x + lane_off, c * LPP * 16);
            }
            if (need & 2) {
#pragma unroll
                for (int c = 0; c < CQ; ++c) R[1][c] = buf_load_f4(src, off.y + lane_off, c * LPP * 16);
            }
            if (need & 4) {
#pragma unroll
                for (int c = 0; c < CQ; ++c) R[2][c] = buf_load_f4(src, off.z + lane_off, c * LPP * 16);
            }
            if (need & 8) {
#pragma unroll
                for (int c = 0; c < CQ; ++c) R[3][c] = buf_load_f4(src, off.w + lane_off, c * LPP * 16);
            }
        };
        if (PIPE) issue_step(0);
        for (int k = 0; k < K; ++k) {
            if (!PIPE) issue_step(k);
            const float4 wv = s_wt[rec0 + k];
            float4 S[CQ];
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < CQ; ++c) {
                float4 sv = f4_mul(wv.x, R[0][c]);
                sv = f4_fma(wv.y, R[1][c], sv);
                sv = f4_fma(wv.z, R[2][c], sv);
                sv = f4_fma(wv.w, R[3][c], sv);
                S[c] = sv;
            }
            if (PIPE) {
                // the next step's rows fly during this step's dot product, reduction and soft-max update
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < K) issue_step(k + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < CQ; ++c) part = (c == 0) ? f4_dot(S[c], f1[c]) : part + f4_dot(S[c], f1[c]);
            // all-reduce over the LPP lanes of the pixel
            part += dpp<0x128>(part);  // row_ror:8
            part += dpp<0x124>(part);  // row_ror:4
            part += dpp<0x122>(part);  // row_ror:2
            part += dpp<0x121>(part);  // row_ror:1
            if (LPP == 32) part += __shfl_xor(part, 16);
            float sv = (part == 0.f) ? -1e10f : part;  // epipolar.py:298
            float e;
            if (d.softmax_enabled) {
                sv = sv * d.softmax_scale;  // epipolar.py:306
                const float m_new = fmaxf(m_run, sv);
                // rescale every step (alpha == 1 when the maximum did not move): a voted lazy rescale
                // costs more in branch + register copies than these 2*CQ packed multiplies
                const float alpha = __expf(m_run - m_new);
#pragma unroll
                for (int c = 0; c < CQ; ++c) acc[c] = f4_mul(alpha, acc[c]);
                m_run = m_new;
                e = __expf(sv - m_new);
            } else {
                sv = sv / (float)K;  // epipolar.py:311
                e = sv;
            }
            s_sim[rec0 + k] = sv;
#pragma unroll
            for (int c = 0; c < CQ; ++c) acc[c] = f4_fma(e, S[c], acc[c]);
        }
        __builtin_amdgcn_wave_barrier();

        // ---- lanes <-> samples per pixel: exact soft-max, arg-max, outputs ----------
        float my_scale = 1.f;
#pragma unroll
        for (int gg = 0; gg < PPW; ++gg) {
            const int pix = first + gg;
            if (pix >= HW) break;  // wave-uniform
            const float m_g = lane_bcast(m_run, gg * LPP);
            float a[KPL];
            float denom = 1.f;
            if (d.softmax_enabled) {
                float lsum = 0.f;
#pragma unroll
                for (int s = 0; s < KPL; ++s) {
                    const int k = s * kWave + lane;
                    a[s] = (k < K) ? expf(s_sim[gg * KP + k] - m_g) : 0.f;
                    lsum += a[s];
                }
                denom = wave_sum(lsum);
#pragma unroll
                for (int s = 0; s < KPL; ++s) a[s] = a[s] / denom;
                if (g == gg) my_scale = 1.f / denom;
            } else {
#pragma unroll
                for (int s = 0; s < KPL; ++s) a[s] = (s * kWave + lane < K) ? s_sim[gg * KP + s * kWave + lane] : 0.f;
            }
            float bestv = neg_inf;
            int besti = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                const int k = s * kWave + lane;
                if (k < K && (a[s] > bestv)) {
                    bestv = a[s];
                    besti = k;
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const float ov = __shfl_xor(bestv, m);
                const int oi = __shfl_xor(besti, m);
                if (ov > bestv || (ov == bestv && oi < besti)) {
                    bestv = ov;
                    besti = oi;
                }
            }
            if (p.corr && lane == 0) {
                const float4 sg = s_seg[gg];
                et::Segment seg;
                seg.sx = sg.x; seg.sy = sg.y; seg.vx = sg.z; seg.vy = sg.w;
                const et::SampleSetup su = et::sample_setup(d, seg, p.steps[besti]);
                float *o = p.corr + ((size_t)n * HW + pix) * 2;
                o[0] = et::de_normalize(d, su.nx, W);
                o[1] = et::de_normalize(d, su.ny, H);
            }
            if (p.attn) {
                const int slot_in_block = wave * kPixPerWave + round * PPW + gg;
#pragma unroll
                for (int s = 0; s < KPL; ++s) {
                    const int k = s * kWave + lane;
                    if (k < K) s_attn[k * kPixPerBlock + slot_in_block] = a[s];
                }
            }
        }
        if (mylive) {
            float4 *o4 = reinterpret_cast<float4 *>(p.out + ((size_t)n * HW + mypix) * C);
#pragma unroll
            for (int c = 0; c < CQ; ++c) o4[c * LPP + li] = f4_mul(my_scale, acc[c]);
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (p.attn) {
        __syncthreads();
        const int npix = min(kPixPerBlock, HW - pix_base);
        float *dst = p.attn + (size_t)n * K * HW + pix_base;
        if (npix == kPixPerBlock && (HW & 3) == 0) {
            for (int t = threadIdx.x; t < K * 4; t += blockDim.x) {
                const int k = t >> 2, q = t & 3;
                const float4 v = *reinterpret_cast<const float4 *>(&s_attn[k * kPixPerBlock + q * 4]);
                *reinterpret_cast<float4 *>(dst + (size_t)k * HW + q * 4) = v;
            }
        } else {
            for (int t = threadIdx.x; t < K * kPixPerBlock; t += blockDim.x) {
                const int k = t / kPixPerBlock, i = t % kPixPerBlock;
                if (i < npix) dst[(size_t)k * HW + i] = s_attn[k * kPixPerBlock + i];
            }
        }
    }
}

// ----------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------
// Same walk, twice.  Pass A recomputes the logits and da_k = g . S_k; the
// soft-max gradient is formed with lanes <-> samples; pass B accumulates
// d(feat_ref) and scatters d(feat_src) through a gradient twin of the tap
// cache: a tap's gradient row is flushed with float atomics only when the tap
// is evicted, i.e. once per (pixel, source row) instead of once per sample.
// Channel mapping here is lane + 64*i (dword-strided) so that each atomic
// instruction of a flush covers 256 contiguous bytes.
template <int CPD /*dwords per lane: C <= 64*CPD*/, int KPL, bool FAST>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void epipolar_bwd_kernel(const BwdParams p)
{
    // Float-atomic scatter form (no workspace needed); the default is the gather form below.
    const EtLayerDesc &d = p.d;
    const int H = d.H, W = d.W, C = d.C, K = d.K;
    const int HW = H * W;

    const int vb = xcd_remap(blockIdx.x, p.total_blocks);
    const int n = vb / p.blocks_per_pair;
    const int pb = vb - n * p.blocks_per_pair;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pix_base = pb * kPixPerBlock;

    const float *cam = p.cam + (size_t)n * ET_CAM_STRIDE;
    const __amdgpu_buffer_rsrc_t src = make_rsrc(p.fsrc + (size_t)n * HW * C, (unsigned)HW * C * 4u);
    const __amdgpu_buffer_rsrc_t gsrc = make_rsrc(p.gsrc + (size_t)n * HW * C, (unsigned)HW * C * 4u);
    const int row_bytes = C * 4;
    const float neg_inf = -__builtin_huge_valf();
    // lanes beyond C (only when C < 64*CPD) alias the last channel: they read
    // it again, contribute zero to every dot product and add 0.0f in flushes.
    int voff[CPD];
    bool live[CPD];
#pragma unroll
    for (int c = 0; c < CPD; ++c) {
        live[c] = (lane + c * kWave) < C;
        voff[c] = min(lane + c * kWave, C - 1) * 4;
    }

    for (int pp = 0; pp < kPixPerWave; ++pp) {
        const int pix = pix_base + wave * kPixPerWave + pp;
        if (pix >= HW) break;
        const int h = pix / W, w = pix - h * W;

        const et::Segment seg = et::epipolar_segment(d, cam, p.xs[w], p.ys[h]);
        SampleTable<KPL> tb;
        build_sample_table<KPL, true>(d, seg, p.steps, lane, row_bytes, tb);
        float v_logit[KPL], v_da[KPL];
        bool v_masked[KPL];
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            v_logit[s] = neg_inf;
            v_da[s] = 0.f;
            v_masked[s] = false;
        }

        float f1[CPD], g[CPD], R[4][CPD];
        const float *ref = p.fref + ((size_t)n * HW + pix) * C;
        const float *go = p.gout + ((size_t)n * HW + pix) * C;
#pragma unroll
        for (int c = 0; c < CPD; ++c) {
            const int ch = lane + c * kWave;
            f1[c] = live[c] ? ref[ch] : 0.f;
            g[c] = live[c] ? go[ch] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) R[r][c] = 0.f;
        }

        // ---------------- pass A: logits and da ---------------------------------
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int kcount = min(kWave, K - s * kWave);
            for (int kb = 0; kb < kcount; kb += 8) {
                float p1[8], p2[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = kb + j;
                    p1[j] = 0.f;
                    p2[j] = 0.f;
                    if (kk < kcount) {
                        const int need = __builtin_amdgcn_readlane(tb.need[s], kk);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (need & (1 << r)) {
                                const int off = __builtin_amdgcn_readlane(tb.off[s][r], kk);
#pragma unroll
                                for (int c = 0; c < CPD; ++c) R[r][c] = buf_load_f1(src, voff[c], off);
                            }
                        }
                        const float w0 = lane_bcast(tb.w[s][0], kk), w1 = lane_bcast(tb.w[s][1], kk);
                        const float w2 = lane_bcast(tb.w[s][2], kk), w3 = lane_bcast(tb.w[s][3], kk);
#pragma unroll
                        for (int c = 0; c < CPD; ++c) {
                            float sv = w0 * R[0][c];
                            sv = fmaf(w1, R[1][c], sv);
                            sv = fmaf(w2, R[2][c], sv);
                            sv = fmaf(w3, R[3][c], sv);
                            p1[j] = fmaf(sv, f1[c], p1[j]);
                            p2[j] = fmaf(sv, g[c], p2[j]);
                        }
                    }
                }
                const float u1 = reduce8<FAST>(p1, lane);
                const float u2 = reduce8<FAST>(p2, lane);
                const bool masked = (u1 == 0.f);
                float sv = masked ? -1e10f : u1;
                sv = d.softmax_enabled ? sv * d.softmax_scale : sv / (float)K;
                const int srcl = lane_of_sample<8>(lane & 7);
                const float mine_l = __shfl(sv, srcl);
                const float mine_d = __shfl(u2, srcl);
                const int mine_m = __shfl((int)masked, srcl);
                if ((lane >> 3) == (kb >> 3)) {
                    v_logit[s] = mine_l;
                    v_da[s] = mine_d;
                    v_masked[s] = mine_m != 0;
                }
            }
        }

        // ---------------- soft-max gradient, lanes <-> samples -----------------
        float v_a[KPL], v_ds[KPL];
        if (d.softmax_enabled) {
            float mx = neg_inf;
#pragma unroll
            for (int s = 0; s < KPL; ++s) mx = fmaxf(mx, (s * kWave + lane < K) ? v_logit[s] : neg_inf);
            mx = wave_max(mx);
            float lsum = 0.f;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                v_a[s] = (s * kWave + lane < K) ? expf(v_logit[s] - mx) : 0.f;
                lsum += v_a[s];
            }
            const float denom = wave_sum(lsum);
            float dsum = 0.f;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                v_a[s] = v_a[s] / denom;
                dsum = fmaf(v_a[s], v_da[s], dsum);
            }
            const float dot = wave_sum(dsum);
#pragma unroll
            for (int s = 0; s < KPL; ++s)
                v_ds[s] = v_masked[s] ? 0.f : d.softmax_scale * v_a[s] * (v_da[s] - dot);
        } else {
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                const bool in = s * kWave + lane < K;
                v_a[s] = in ? v_logit[s] : 0.f;  // already sim / K
                v_ds[s] = (in && !v_masked[s]) ? v_da[s] / (float)K : 0.f;
            }
        }

        // ---------------- pass B: d(feat_ref) and scatter of d(feat_src) --------
        float d1[CPD], G[4][CPD];
#pragma unroll
        for (int c = 0; c < CPD; ++c) {
            d1[c] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                G[r][c] = 0.f;
                R[r][c] = 0.f;
            }
        }
        int tag[4] = {-1, -1, -1, -1};  // byte offset of the row whose gradient G[r] holds (scalar)

#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int kcount = min(kWave, K - s * kWave);
            for (int kk = 0; kk < kcount; ++kk) {
                const int need = __builtin_amdgcn_readlane(tb.need[s], kk);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (need & (1 << r)) {
                        const int off = __builtin_amdgcn_readlane(tb.off[s][r], kk);
                        if (tag[r] >= 0) {
#pragma unroll
                            for (int c = 0; c < CPD; ++c) {
                                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(G[r][c], gsrc, voff[c], tag[r], 0);
                                G[r][c] = 0.f;
                            }
                        }
#pragma unroll
                        for (int c = 0; c < CPD; ++c) R[r][c] = buf_load_f1(src, voff[c], off);
                        tag[r] = off;
                    }
                }
                const float w0 = lane_bcast(tb.w[s][0], kk), w1 = lane_bcast(tb.w[s][1], kk);
                const float w2 = lane_bcast(tb.w[s][2], kk), w3 = lane_bcast(tb.w[s][3], kk);
                const float ak = lane_bcast(v_a[s], kk), dsk = lane_bcast(v_ds[s], kk);
                // OTHER_GRAD (epipolar.py:141-153): which of the two uses of feat_src carry gradient
                const float ak_src = (d.src_grad_mask & 2) ? ak : 0.f;
                const float dsk_src = (d.src_grad_mask & 1) ? dsk : 0.f;
#pragma unroll
                for (int c = 0; c < CPD; ++c) {
                    float sv = w0 * R[0][c];
                    sv = fmaf(w1, R[1][c], sv);
                    sv = fmaf(w2, R[2][c], sv);
                    sv = fmaf(w3, R[3][c], sv);
                    d1[c] = fmaf(dsk, sv, d1[c]);
                    const float dS = fmaf(dsk_src, f1[c], ak_src * g[c]);
                    G[0][c] = fmaf(w0, dS, G[0][c]);
                    G[1][c] = fmaf(w1, dS, G[1][c]);
                    G[2][c] = fmaf(w2, dS, G[2][c]);
                    G[3][c] = fmaf(w3, dS, G[3][c]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (tag[r] >= 0) {
#pragma unroll
                for (int c = 0; c < CPD; ++c)
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(G[r][c], gsrc, voff[c], tag[r], 0);
            }
        }
        float *gr = p.gref + ((size_t)n * HW + pix) * C;
#pragma unroll
        for (int c = 0; c < CPD; ++c) {
            const int ch = lane + c * kWave;
            if (live[c]) gr[ch] = d1[c];
        }
    }
}

// ----------------------------------------------------------------------------
// backward, emit form with the forward's channel mapping (float4 per lane):
// d(feat_ref) + per-(pixel, source row) coefficients for the gather pass
// ----------------------------------------------------------------------------
template <int CPL, int KPL, bool FAST>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void epipolar_bwd_emit_kernel(const BwdParams p)
{
    extern __shared__ float s_dyn[];  // per wave: [KPL*64] float4 weights, then [cap] u, [cap] alpha, [cap] beta
    const EtLayerDesc &d = p.d;
    const int H = d.H, W = d.W, C = d.C, K = d.K;
    const int HW = H * W;
    const int nvec = C >> 2;

    const int vb = xcd_remap(blockIdx.x, p.total_blocks);
    const int n = vb / p.blocks_per_pair;
    const int pb = vb - n * p.blocks_per_pair;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pix_base = pb * kPixPerBlock;
    char *wbase = reinterpret_cast<char *>(s_dyn) + (size_t)wave * (KPL * kWave * 16 + 3 * p.cap * 4);
    float4 *s_wt = reinterpret_cast<float4 *>(wbase);
    int *s_eu = reinterpret_cast<int *>(wbase + KPL * kWave * 16);
    float *s_ea = reinterpret_cast<float *>(s_eu) + p.cap;
    float *s_eb = s_ea + p.cap;

    const float *cam = p.cam + (size_t)n * ET_CAM_STRIDE;
    const __amdgpu_buffer_rsrc_t src = make_rsrc(p.fsrc + (size_t)n * HW * C, (unsigned)HW * C * 4u);
    const int row_bytes = C * 4;
    const float neg_inf = -__builtin_huge_valf();
    int voff[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) voff[c] = min(lane + c * kWave, nvec - 1) * 16;

    for (int pp = 0; pp < kPixPerWave; ++pp) {
        const int pix = pix_base + pp * kWavesPerBlock + wave;  // waves of a block on neighbouring pixels
        if (pix >= HW) continue;
        const int h = pix / W, w = pix - h * W;
        const et::Segment seg = et::epipolar_segment(d, cam, p.xs[w], p.ys[h]);
        SampleTable<KPL> tb;
        build_sample_table<KPL, true>(d, seg, p.steps, lane, row_bytes, tb);
        float v_logit[KPL], v_da[KPL];
        bool v_masked[KPL];
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            v_logit[s] = neg_inf;
            v_da[s] = 0.f;
            v_masked[s] = false;
            s_wt[s * kWave + lane] = make_float4(tb.w[s][0], tb.w[s][1], tb.w[s][2], tb.w[s][3]);
        }
        __builtin_amdgcn_wave_barrier();

        float4 f1[CPL], g[CPL], R[4][CPL];
        const float4 *ref = reinterpret_cast<const float4 *>(p.fref + ((size_t)n * HW + pix) * C);
        const float4 *go = reinterpret_cast<const float4 *>(p.gout + ((size_t)n * HW + pix) * C);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int v = lane + c * kWave;
            f1[c] = (v < nvec) ? ref[v] : f4_zero();
            g[c] = (v < nvec) ? go[v] : f4_zero();
#pragma unroll
            for (int r = 0; r < 4; ++r) R[r][c] = f4_zero();
        }

        // ---------------- pass A: logits and da -----------------------------------
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int kcount = min(kWave, K - s * kWave);
            for (int kb = 0; kb < kcount; kb += 8) {
                float p1[8], p2[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = kb + j;
                    p1[j] = 0.f;
                    p2[j] = 0.f;
                    if (kk < kcount) {
                        const int need = __builtin_amdgcn_readlane(tb.need[s], kk);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (need & (1 << r)) {
                                const int off = __builtin_amdgcn_readlane(tb.off[s][r], kk);
#pragma unroll
                                for (int c = 0; c < CPL; ++c) R[r][c] = buf_load_f4(src, voff[c], off);
                            }
                        }
                        const float4 wv = s_wt[s * kWave + kk];
#pragma unroll
                        for (int c = 0; c < CPL; ++c) {
                            float4 sv = f4_mul(wv.x, R[0][c]);
                            sv = f4_fma(wv.y, R[1][c], sv);
                            sv = f4_fma(wv.z, R[2][c], sv);
                            sv = f4_fma(wv.w, R[3][c], sv);
                            p1[j] += f4_dot(sv, f1[c]);
                            p2[j] += f4_dot(sv, g[c]);
                        }
                    }
                }
                const float u1 = reduce8<FAST>(p1, lane);
                const float u2 = reduce8<FAST>(p2, lane);
                const bool masked = (u1 == 0.f);
                float sv = masked ? -1e10f : u1;
                sv = d.softmax_enabled ? sv * d.softmax_scale : sv / (float)K;
                const int srcl = lane_of_sample<8>(lane & 7);
                const float mine_l = __shfl(sv, srcl);
                const float mine_d = __shfl(u2, srcl);
                const int mine_m = __shfl((int)masked, srcl);
                if ((lane >> 3) == (kb >> 3)) {
                    v_logit[s] = mine_l;
                    v_da[s] = mine_d;
                    v_masked[s] = mine_m != 0;
                }
            }
        }

        // ---------------- soft-max gradient, lanes <-> samples -----------------------
        float v_a[KPL], v_ds[KPL];
        if (d.softmax_enabled) {
            float mx = neg_inf;
#pragma unroll
            for (int s = 0; s < KPL; ++s) mx = fmaxf(mx, (s * kWave + lane < K) ? v_logit[s] : neg_inf);
            mx = wave_max(mx);
            float lsum = 0.f;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                v_a[s] = (s * kWave + lane < K) ? expf(v_logit[s] - mx) : 0.f;
                lsum += v_a[s];
            }
            const float denom = wave_sum(lsum);
            float dsum = 0.f;
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                v_a[s] = v_a[s] / denom;
                dsum = fmaf(v_a[s], v_da[s], dsum);
            }
            const float dot = wave_sum(dsum);
#pragma unroll
            for (int s = 0; s < KPL; ++s)
                v_ds[s] = v_masked[s] ? 0.f : d.softmax_scale * v_a[s] * (v_da[s] - dot);
        } else {
#pragma unroll
            for (int s = 0; s < KPL; ++s) {
                const bool in = s * kWave + lane < K;
                v_a[s] = in ? v_logit[s] : 0.f;
                v_ds[s] = (in && !v_masked[s]) ? v_da[s] / (float)K : 0.f;
            }
        }
        // ---------------- pass B: d(feat_ref) and the coefficient entries ----------------
        float4 d1[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            d1[c] = f4_zero();
#pragma unroll
            for (int r = 0; r < 4; ++r) R[r][c] = f4_zero();
        }
        int tag[4] = {-1, -1, -1, -1};
        float ea[4] = {0.f, 0.f, 0.f, 0.f}, eb[4] = {0.f, 0.f, 0.f, 0.f};
        int ecount = 0;
        auto emit = [&](int r) {  // wave-uniform call
            if (lane == 0) {
                s_eu[ecount] = tag[r] / row_bytes;
                s_ea[ecount] = ea[r];
                s_eb[ecount] = eb[r];
            }
            ++ecount;
            ea[r] = 0.f;
            eb[r] = 0.f;
        };
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
            const int kcount = min(kWave, K - s * kWave);
            for (int kk = 0; kk < kcount; ++kk) {
                const int need = __builtin_amdgcn_readlane(tb.need[s], kk);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (need & (1 << r)) {
                        const int off = __builtin_amdgcn_readlane(tb.off[s][r], kk);
                        if (tag[r] >= 0) emit(r);
#pragma unroll
                        for (int c = 0; c < CPL; ++c) R[r][c] = buf_load_f4(src, voff[c], off);
                        tag[r] = off;
                    }
                }
                const float4 wv = s_wt[s * kWave + kk];
                const float ak = lane_bcast(v_a[s], kk), dsk = lane_bcast(v_ds[s], kk);
                // OTHER_GRAD (epipolar.py:141-153): which of the two uses of feat_src carry gradient
                const float ak_src = (d.src_grad_mask & 2) ? ak : 0.f;
                const float dsk_src = (d.src_grad_mask & 1) ? dsk : 0.f;
                ea[0] = fmaf(wv.x, ak_src, ea[0]); eb[0] = fmaf(wv.x, dsk_src, eb[0]);
                ea[1] = fmaf(wv.y, ak_src, ea[1]); eb[1] = fmaf(wv.y, dsk_src, eb[1]);
                ea[2] = fmaf(wv.z, ak_src, ea[2]); eb[2] = fmaf(wv.z, dsk_src, eb[2]);
                ea[3] = fmaf(wv.w, ak_src, ea[3]); eb[3] = fmaf(wv.w, dsk_src, eb[3]);
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    float4 sv = f4_mul(wv.x, R[0][c]);
                    sv = f4_fma(wv.y, R[1][c], sv);
                    sv = f4_fma(wv.z, R[2][c], sv);
                    sv = f4_fma(wv.w, R[3][c], sv);
                    d1[c] = f4_fma(dsk, sv, d1[c]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (tag[r] >= 0) emit(r);
        __builtin_amdgcn_wave_barrier();
        const size_t ebase = ((size_t)n * HW + pix) * p.cap;
        for (int i = lane; i < ecount; i += kWave) {
            const int u = s_eu[i];
            p.ent_u[ebase + i] = u;
            p.ent_a[ebase + i] = s_ea[i];
            p.ent_b[ebase + i] = s_eb[i];
            atomicAdd(&p.row_count[(size_t)n * HW + u], 1);
        }
        if (lane == 0) p.ent_count[(size_t)n * HW + pix] = ecount;
        float4 *gr = reinterpret_cast<float4 *>(p.gref + ((size_t)n * HW + pix) * C);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int v = lane + c * kWave;
            if (v < nvec) gr[v] = d1[c];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ----------------------------------------------------------------------------
// gather-form backward of d(feat_src): scan, bucket, gather
// ----------------------------------------------------------------------------
// Per pair: exclusive prefix sum of row_count[HW] -> row_base[HW] (one block per pair).
__global__ __launch_bounds__(256) void bwd_scan_kernel(int HW, const int *row_count, int *row_base, int *row_cursor)
{
    __shared__ int s_part[256];
    const int n = blockIdx.x, t = threadIdx.x;
    const int *cnt = row_count + (size_t)n * HW;
    int *base = row_base + (size_t)n * HW;
    int *cur = row_cursor + (size_t)n * HW;
    const int per = (HW + 255) / 256;
    const int lo = t * per, hi = min(HW, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    s_part[t] = sum;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) {
            const int v = s_part[i];
            s_part[i] = run;
            run += v;
        }
    }
    __syncthreads();
    int run = s_part[t];
    for (int i = lo; i < hi; ++i) {
        base[i] = run;
        cur[i] = 0;
        run += cnt[i];
    }
}

// One wave per reference pixel: move its entries to their source rows' segments.
__global__ __launch_bounds__(256) void bwd_bucket_kernel(int HW, int cap, int total_rows, const int *ent_count,
                                                          const int *ent_u, const float *ent_a, const float *ent_b,
                                                          const int *row_base, int *row_cursor, int4 *csr)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    for (int gp = wave_global; gp < total_rows; gp += nwaves) {  // global reference pixel n*HW + p
        const int cnt = ent_count[gp];
        const int n = gp / HW;
        const int pidx = gp - n * HW;
        const size_t ebase = (size_t)gp * cap;
        const size_t pair_rows = (size_t)n * HW;
        for (int slot = lane; slot < cnt; slot += kWave) {
            const int u = ent_u[ebase + slot];
            const size_t gu = pair_rows + u;
            const int pos = row_base[gu] + atomicAdd(&row_cursor[gu], 1);
            const size_t o = pair_rows * cap + pos;     // per-pair CSR region of HW*cap slots
            // one 16-byte record per entry (a single scattered store): {ordering key, alpha, beta, -}
            csr[o] = make_int4(pidx * cap + slot, __float_as_int(ent_a[ebase + slot]),
                               __float_as_int(ent_b[ebase + slot]), 0);
        }
    }
}

// One wave per source pixel u: d feat_src[u] = sum_e alpha_e * g[p_e] + beta_e * f[p_e].
// Entries are ordered by reference pixel index first (bitonic sort in LDS) so the float32 sum has a
// fixed order: the result is bit-reproducible, unlike the atomic scatter.
template <int CPL>
__global__ __launch_bounds__(256) void epipolar_bwd_gather_kernel(int HW, int C, int cap, int total_rows, int mask,
                                                                   const int *row_count, const int *row_base,
                                                                   const int4 *csr, const float *fref,
                                                                   const float *gout, float *gsrc, int max_sort)
{
    extern __shared__ int s_sort[];  // per wave: max_sort keys + max_sort CSR indices
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gu = xcd_remap(blockIdx.x, gridDim.x) * kWavesPerBlock + wave;  // global source pixel n*HW + u
    if (gu >= total_rows) return;
    const int n = gu / HW;
    const int cnt = row_count[gu];
    const size_t seg = (size_t)n * HW * cap + row_base[gu];
    const int nvec = C >> 2;
    float4 acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = f4_zero();
    int voff[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) voff[c] = min(lane + c * kWave, nvec - 1);

    // order of summation: ascending (reference pixel, emission slot) -- a key that does not depend on
    // the arrival order of the bucket pass.  Keys are unique, so an entry's place in that order is the
    // number of smaller keys: a rank sort (every lane compares its keys against all keys, read as LDS
    // broadcasts, no dependent passes) is cheaper here than a bitonic network (n ~ 140).
    int *keys = s_sort + wave * 2 * max_sort;
    int *vals = keys + max_sort;
    const bool sorted = cnt <= max_sort;
    if (sorted) {
        const int cnt4 = (cnt + 3) & ~3;
        for (int i = lane; i < cnt4; i += kWave) keys[i] = (i < cnt) ? csr[seg + i].x : 0x7fffffff;
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < cnt; base += kWave) {
            const int mine = (base + lane < cnt) ? keys[base + lane] : 0x7fffffff;
            int rank = 0;
            for (int j = 0; j < cnt4; j += 4) {
                const int4 kq = *reinterpret_cast<const int4 *>(&keys[j]);  // same address in every lane
                rank += (kq.x < mine) + (kq.y < mine) + (kq.z < mine) + (kq.w < mine);
            }
            if (base + lane < cnt) vals[rank] = base + lane;
        }
        __builtin_amdgcn_wave_barrier();
    }
    const __amdgpu_buffer_rsrc_t G4 = make_rsrc(gout + (size_t)n * HW * C, (unsigned)HW * C * 4u);
    const __amdgpu_buffer_rsrc_t F4 = make_rsrc(fref + (size_t)n * HW * C, (unsigned)HW * C * 4u);
    const int row_bytes = C * 4;
    for (int e0 = 0; e0 < cnt; e0 += kWave) {
        const int m = min(kWave, cnt - e0);
        int idx = 0, pp = 0;
        float aa = 0.f, bb = 0.f;
        if (lane < m) {
            idx = sorted ? vals[e0 + lane] : (e0 + lane);
            const int4 rec = csr[seg + idx];
            pp = rec.x / cap;
            aa = __int_as_float(rec.y);
            bb = __int_as_float(rec.z);
        }
        for (int j = 0; j < m; ++j) {
            const int pj = __builtin_amdgcn_readlane(pp, j) * row_bytes;  // scalar row offset
            const float aj = lane_bcast(aa, j), bj = lane_bcast(bb, j);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                if (mask & 2) acc[c] = f4_fma(aj, buf_load_f4(G4, voff[c] * 16, pj), acc[c]);
                if (mask & 1) acc[c] = f4_fma(bj, buf_load_f4(F4, voff[c] * 16, pj), acc[c]);
            }
        }
    }
    float4 *o4 = reinterpret_cast<float4 *>(gsrc) + (size_t)gu * nvec;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int v = lane + c * kWave;
        if (v < nvec) o4[v] = acc[c];
    }
}

// ----------------------------------------------------------------------------
// sample_locs (debug / VIS.EPIPOLAR_LINE / parity gate on the geometry)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_locs_kernel(const EtLayerDesc d, const float *xs, const float *ys,
                                                           const float *steps, const float *cam, float *locs)
{
    const int HW = d.H * d.W;
    const size_t total = (size_t)d.N * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int pix = (int)(i - (size_t)n * HW);
        const int h = pix / d.W, w = pix - h * d.W;
        const et::Segment seg = et::epipolar_segment(d, cam + (size_t)n * ET_CAM_STRIDE, xs[w], ys[h]);
        for (int k = 0; k < d.K; ++k) {
            const et::SampleSetup su = et::sample_setup(d, seg, steps[k]);
            float2 *o = reinterpret_cast<float2 *>(locs) + ((size_t)k * d.N + n) * HW + pix;
            *o = make_float2(su.nx, su.ny);
        }
    }
}

// ----------------------------------------------------------------------------
// residual epilogue: x = feat + out + (y*scale + shift)    (HBM-bound stream)
// ----------------------------------------------------------------------------
template <bool HAS_Y>
__global__ __launch_bounds__(256) void residual_epilogue_kernel(size_t nvec_total, int nvec_c, const float4 *feat,
                                                                 const float4 *out, const float4 *y,
                                                                 const float4 *scale, const float4 *shift,
                                                                 float4 *finalout, float4 *x)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec_total;
         i += (size_t)gridDim.x * blockDim.x) {
        const float4 o = out[i];
        float4 fo = o;
        if constexpr (HAS_Y) {
            const int c = (int)(i % (size_t)nvec_c);
            const float4 yy = y[i], sc = scale[c], sh = shift[c];
            // bn(z(out)) + out  (epipolar.py:250-253): affine first, then the residual add
            fo.x = fmaf(yy.x, sc.x, sh.x) + o.x;
            fo.y = fmaf(yy.y, sc.y, sh.y) + o.y;
            fo.z = fmaf(yy.z, sc.z, sh.z) + o.z;
            fo.w = fmaf(yy.w, sc.w, sh.w) + o.w;
        }
        if (finalout) finalout[i] = fo;
        if (x) {
            const float4 f = feat[i];
            x[i] = make_float4(fo.x + f.x, fo.y + f.y, fo.z + f.z, fo.w + f.w);  // resnet.py:388
        }
    }
}

// ----------------------------------------------------------------------------
// NCHW <-> NHWC (per image: [C][HW] <-> [HW][C] transpose through LDS)
// ----------------------------------------------------------------------------
constexpr int kTile = 64;
// src is [rows][cols] row-major, dst is [cols][rows]; batch stride rows*cols
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const float *src, float *dst)
{
    __shared__ float tile[kTile][kTile + 1];
    const size_t img = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * kTile, r0 = blockIdx.y * kTile;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < kTile; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        if (r < rows && c < cols) tile[i][tx] = src[img + (size_t)r * cols + c];
    }
    __syncthreads();
    for (int i = ty; i < kTile; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) dst[img + (size_t)c * rows + r] = tile[tx][i];
    }
}

// ----------------------------------------------------------------------------
// host-side dispatch
// ----------------------------------------------------------------------------
int validate(const EtLayerDesc *d)
{
    if (!d) return fail("desc is NULL");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0) return fail("bad shape N=%d H=%d W=%d", d->N, d->H, d->W);
    if (d->C <= 0 || (d->C & 3)) return fail("C=%d must be a positive multiple of 4", d->C);
    if (d->C > 512) return fail("C=%d > 512 not supported", d->C);
    if (d->K < 2 || d->K > 256) return fail("K=%d outside [2, 256]", d->K);
    if ((long long)d->H * d->W * d->C * 4 >= (1LL << 31)) return fail("one feature map must stay below 2 GiB");
    if (!(d->downsample > 0.f) || !(d->image_resize > 0.f) || !(d->predict_resize > 0.f))
        return fail("downsample / resize factors must be positive");
    return 0;
}

template <int CPL, int KPL>
void launch_fwd(const FwdParams &p, int variant, dim3 grid, size_t lds, hipStream_t st)
{
    const bool safe = variant & ET_VARIANT_SAFE_REDUCE, nocache = variant & ET_VARIANT_NO_TAP_CACHE;
    const bool b4 = variant & ET_VARIANT_BATCH4;
    const int occ = (variant & ET_VARIANT_OCC6) ? 6 : (variant & ET_VARIANT_OCC5) ? 5 : 1;
    const bool ragged = (p.d.K % 8) != 0;   // K % 4 == 0 but % 8 != 0 also takes the ragged build (fewer variants)
#define ET_FWD(B, F, Cc, W)                                                                                  \
    do {                                                                                                     \
        if (ragged)                                                                                          \
            hipLaunchKernelGGL((epipolar_fwd_kernel<CPL, KPL, B, F, Cc, W, true>), grid, dim3(256), lds, st, p);  \
        else                                                                                                 \
            hipLaunchKernelGGL((epipolar_fwd_kernel<CPL, KPL, B, F, Cc, W, false>), grid, dim3(256), lds, st, p); \
    } while (0)
    if (safe || nocache) {
        // ablation / fallback variants, default register budget
        if (safe && !nocache) { if (b4) ET_FWD(4, false, true, 1); else ET_FWD(8, false, true, 1); }
        else if (!safe && nocache) ET_FWD(8, true, false, 1);
        else ET_FWD(8, false, false, 1);
    } else if (b4) {
        if (occ == 6) ET_FWD(4, true, true, 6);
        else if (occ == 5) ET_FWD(4, true, true, 5);
        else ET_FWD(4, true, true, 1);
    } else {
        if (occ == 6) ET_FWD(8, true, true, 6);
        else if (occ == 5) ET_FWD(8, true, true, 5);
        else ET_FWD(8, true, true, 1);
    }
#undef ET_FWD
}

template <int CPD, int KPL>
void launch_bwd(const BwdParams &p, int variant, dim3 grid, hipStream_t st)
{
    if (variant & ET_VARIANT_SAFE_REDUCE)
        hipLaunchKernelGGL((epipolar_bwd_kernel<CPD, KPL, false>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((epipolar_bwd_kernel<CPD, KPL, true>), grid, dim3(256), 0, st, p);
}

}  // namespace

// ============================================================================
// C ABI
// ============================================================================
extern "C" {

int et_abi_version(void) { return ET_ABI_VERSION; }

const char *et_last_error(void) { return g_err; }

int et_sample_locs(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                   const float *cam, float *sample_locs, void *stream)
{
    if (int e = validate(desc)) return e;
    if (!xs || !ys || !steps || !cam || !sample_locs) return fail("et_sample_locs: NULL pointer");
    const size_t total = (size_t)desc->N * desc->H * desc->W;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(sample_locs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *desc, xs, ys, steps,
                       cam, sample_locs);
    return check_launch("et_sample_locs");
}

int et_epipolar_forward(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                        const float *cam, const float *feat_ref, const float *feat_src, float *out,
                        float *attn, float *corr_pos, const float *res_bias, float *res_base, void *stream)
{
    if (int e = validate(desc)) return e;
    if (!xs || !ys || !steps || !cam || !feat_ref || !feat_src || !out)
        return fail("et_epipolar_forward: NULL pointer");
    if (res_bias && !res_base) return fail("et_epipolar_forward: res_bias given without res_base");
    FwdParams p;
    p.d = *desc;
    p.xs = xs; p.ys = ys; p.steps = steps; p.cam = cam;
    p.fref = feat_ref; p.fsrc = feat_src;
    p.out = out; p.attn = attn; p.corr = corr_pos;
    p.res_bias = res_bias; p.res_base = res_base;
    const int HW = desc->H * desc->W;
    p.blocks_per_pair = (HW + kPixPerBlock - 1) / kPixPerBlock;
    const long long total = (long long)p.blocks_per_pair * desc->N;
    if (total > 0x7fffffffLL) return fail("grid too large");
    p.total_blocks = (int)total;
    p.interleave = (desc->variant & ET_VARIANT_PIXEL_INTERLEAVE) ? 1 : 0;
    p.ablate = (desc->variant & ET_VARIANT_ABLATE_NO_LOADS) ? 1 : (desc->variant & ET_VARIANT_ABLATE_ONE_ROW) ? 2 : 0;
    const dim3 grid((unsigned)total);
    const int kpl_ = (desc->K + 63) / 64;
    const size_t lds = (attn ? (size_t)desc->K * kPixPerBlock * sizeof(float) : 0) +
                       (size_t)kWavesPerBlock * kpl_ * kWave * 4 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int cpl = (desc->C + 255) / 256, kpl = (desc->K + 63) / 64;
    // variant 0 = the tuned default (measured on MI355X, profiles/): for the 256-channel head with K <= 64
    // four pixels per wave in lockstep; otherwise one pixel per wave, batches of 4 samples, <= 96 VGPRs (5 waves/SIMD),
    // waves of a block interleaved over neighbouring pixels
    int v = desc->variant;
    if ((v & ~(ET_VARIANT_ABLATE_NO_LOADS | ET_VARIANT_ABLATE_ONE_ROW)) == 0)
        v |= (desc->C == 256 && kpl == 1) ? ET_VARIANT_MULTI4   // K > 64: its LDS records cut occupancy (measured 1.8x slower)
                                          : (ET_VARIANT_BATCH4 | ET_VARIANT_OCC5 | ET_VARIANT_PIXEL_INTERLEAVE);
    if (v & ET_VARIANT_BASELINE)
        v &= ~(ET_VARIANT_BATCH4 | ET_VARIANT_OCC5 | ET_VARIANT_OCC6 | ET_VARIANT_PIXEL_INTERLEAVE |
               ET_VARIANT_MULTI2 | ET_VARIANT_MULTI4);
    p.ablate = (v & ET_VARIANT_ABLATE_NO_LOADS) ? 1 : (v & ET_VARIANT_ABLATE_ONE_ROW) ? 2 : 0;
    p.interleave = (v & ET_VARIANT_PIXEL_INTERLEAVE) ? 1 : 0;
    if ((v & (ET_VARIANT_MULTI2 | ET_VARIANT_MULTI4)) && desc->C == 256 && kpl <= 2) {
        // several pixels per wave; per-wave LDS: PPW * KP * 40 + PPW * 16 bytes
        const int ppw = (v & ET_VARIANT_MULTI4) ? 4 : 2;
        const size_t lds_m = (attn ? (size_t)desc->K * kPixPerBlock * sizeof(float) : 0) +
                             (size_t)kWavesPerBlock * (ppw * kpl * kWave * 40 + ppw * 16);
        const bool pipe = v & ET_VARIANT_PIPELINE;
#define ET_MULTI(P, Q, KK)                                                                                     \
    do {                                                                                                       \
        if (pipe) hipLaunchKernelGGL((epipolar_fwd_multi_kernel<P, Q, KK, true>), grid, dim3(256), lds_m, st, p);  \
        else hipLaunchKernelGGL((epipolar_fwd_multi_kernel<P, Q, KK, false>), grid, dim3(256), lds_m, st, p);      \
    } while (0)
        if (ppw == 4) { if (kpl == 1) ET_MULTI(4, 4, 1); else ET_MULTI(4, 4, 2); }
        else { if (kpl == 1) ET_MULTI(2, 2, 1); else ET_MULTI(2, 2, 2); }
#undef ET_MULTI
        return check_launch("et_epipolar_forward(multi)");
    }
    if (cpl == 1) {
        if (kpl == 1) launch_fwd<1, 1>(p, v, grid, lds, st);
        else if (kpl == 2) launch_fwd<1, 2>(p, v, grid, lds, st);
        else launch_fwd<1, 4>(p, v, grid, lds, st);
    } else {
        if (kpl == 1) launch_fwd<2, 1>(p, v, grid, lds, st);
        else if (kpl == 2) launch_fwd<2, 2>(p, v, grid, lds, st);
        else launch_fwd<2, 4>(p, v, grid, lds, st);
    }
    return check_launch("et_epipolar_forward");
}

size_t et_epipolar_backward_workspace_bytes(const EtLayerDesc *desc)
{
    if (validate(desc)) return 0;
    const size_t rows = (size_t)desc->N * desc->H * desc->W;
    const size_t cap = 4u * (size_t)desc->K;
    return rows * cap * (3u * 4u + 16u) + rows * 4u * 4u + 256u;
}

int et_epipolar_backward(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                         const float *cam, const float *feat_ref, const float *feat_src,
                         const float *grad_out, float *grad_ref, float *grad_src, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    if (int e = validate(desc)) return e;
    if (!xs || !ys || !steps || !cam || !feat_ref || !feat_src || !grad_out || !grad_ref || !grad_src)
        return fail("et_epipolar_backward: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const int HW = desc->H * desc->W;
    const size_t rows = (size_t)desc->N * HW;
    const size_t bytes = rows * desc->C * sizeof(float);
    const bool gather = workspace != nullptr && !(desc->variant & ET_VARIANT_BWD_ATOMIC) && desc->src_grad_mask != 0;
    if (gather && workspace_bytes < et_epipolar_backward_workspace_bytes(desc))
        return fail("et_epipolar_backward: workspace of %zu bytes is smaller than the %zu required", workspace_bytes,
                    et_epipolar_backward_workspace_bytes(desc));
    if (gather && ((long long)HW * 4 * desc->K >= (1LL << 31)))
        return fail("et_epipolar_backward: H*W*4K must stay below 2^31 for the gather-form backward");
    BwdParams p;
    std::memset(&p, 0, sizeof(p));
    p.d = *desc;
    p.xs = xs; p.ys = ys; p.steps = steps; p.cam = cam;
    p.fref = feat_ref; p.fsrc = feat_src; p.gout = grad_out;
    p.gref = grad_ref; p.gsrc = grad_src;
    p.blocks_per_pair = (HW + kPixPerBlock - 1) / kPixPerBlock;
    const long long total = (long long)p.blocks_per_pair * desc->N;
    if (total > 0x7fffffffLL) return fail("grid too large");
    p.total_blocks = (int)total;
    int *row_base = nullptr, *row_cursor = nullptr;
    int4 *csr = nullptr;
    if (gather) {
        // carve the workspace: 3 pixel-major entry arrays, 3 row-major (CSR) arrays, 4 per-row int arrays
        p.cap = 4 * desc->K;
        const size_t slots = rows * p.cap;
        char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
        p.ent_u = reinterpret_cast<int *>(w);        w += slots * 4;
        p.ent_a = reinterpret_cast<float *>(w);      w += slots * 4;
        p.ent_b = reinterpret_cast<float *>(w);      w += slots * 4;
        csr = reinterpret_cast<int4 *>(w);           w += slots * 16;
        p.ent_count = reinterpret_cast<int *>(w);    w += rows * 4;
        p.row_count = reinterpret_cast<int *>(w);    w += rows * 4;
        row_base = reinterpret_cast<int *>(w);       w += rows * 4;
        row_cursor = reinterpret_cast<int *>(w);
        hipError_t me = hipMemsetAsync(p.row_count, 0, rows * 4, st);
        if (me != hipSuccess) return fail("hipMemsetAsync(row_count): %s", hipGetErrorString(me));
    } else {
        hipError_t me = hipMemsetAsync(grad_src, 0, bytes, st);
        if (me != hipSuccess) return fail("hipMemsetAsync(grad_src): %s", hipGetErrorString(me));
    }
    const dim3 grid((unsigned)total);
    const int cpd = (desc->C + 63) / 64, kpl = (desc->K + 63) / 64;
    const int v = desc->variant;
    if (gather) {
        const size_t lds_e = (size_t)kWavesPerBlock * (kpl * kWave * 16 + 3 * p.cap * 4);
        const bool safe = v & ET_VARIANT_SAFE_REDUCE;
#define ET_EMIT(CPLv, KPLv)                                                                                        \
    do {                                                                                                           \
        if (safe) hipLaunchKernelGGL((epipolar_bwd_emit_kernel<CPLv, KPLv, false>), grid, dim3(256), lds_e, st, p); \
        else hipLaunchKernelGGL((epipolar_bwd_emit_kernel<CPLv, KPLv, true>), grid, dim3(256), lds_e, st, p);       \
    } while (0)
        if (desc->C <= 256) { if (kpl == 1) ET_EMIT(1, 1); else if (kpl == 2) ET_EMIT(1, 2); else ET_EMIT(1, 4); }
        else { if (kpl == 1) ET_EMIT(2, 1); else if (kpl == 2) ET_EMIT(2, 2); else ET_EMIT(2, 4); }
#undef ET_EMIT
    } else {
#define ET_BWD_CASE(CPD)                                   \
    if (kpl == 1) launch_bwd<CPD, 1>(p, v, grid, st);      \
    else if (kpl == 2) launch_bwd<CPD, 2>(p, v, grid, st); \
    else launch_bwd<CPD, 4>(p, v, grid, st);
    if (cpd <= 1) { ET_BWD_CASE(1) }
    else if (cpd <= 2) { ET_BWD_CASE(2) }
    else if (cpd <= 4) { ET_BWD_CASE(4) }
    else { ET_BWD_CASE(8) }
#undef ET_BWD_CASE
    }
    if (int e = check_launch("et_epipolar_backward")) return e;
    if (gather) {
        hipLaunchKernelGGL(bwd_scan_kernel, dim3(desc->N), dim3(256), 0, st, HW, p.row_count, row_base, row_cursor);
        const unsigned bblocks = (unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock < 16384
                                            ? (rows + kWavesPerBlock - 1) / kWavesPerBlock : 16384);
        hipLaunchKernelGGL(bwd_bucket_kernel, dim3(bblocks), dim3(256), 0, st, HW, p.cap, (int)rows, p.ent_count,
                           p.ent_u, p.ent_a, p.ent_b, row_base, row_cursor, csr);
        // entries per source pixel ordered in LDS (beyond that, or with ET_VARIANT_BWD_UNSORTED: arrival order)
        const int max_sort = (desc->variant & ET_VARIANT_BWD_UNSORTED) ? 0 : 1024;
        const unsigned gblocks = (unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock);
        const size_t lds = (size_t)kWavesPerBlock * 2 * (max_sort ? max_sort : 1) * sizeof(int);
        if (desc->C <= 256)
            hipLaunchKernelGGL((epipolar_bwd_gather_kernel<1>), dim3(gblocks), dim3(256), lds, st, HW, desc->C, p.cap,
                               (int)rows, desc->src_grad_mask, p.row_count, row_base, csr, feat_ref, grad_out, grad_src,
                               max_sort);
        else
            hipLaunchKernelGGL((epipolar_bwd_gather_kernel<2>), dim3(gblocks), dim3(256), lds, st, HW, desc->C, p.cap,
                               (int)rows, desc->src_grad_mask, p.row_count, row_base, csr, feat_ref, grad_out, grad_src,
                               max_sort);
        if (int e = check_launch("et_epipolar_backward(gather)")) return e;
    }
    return 0;
}

int et_residual_epilogue(int64_t num_pixels, int32_t C, const float *feat, const float *out, const float *y,
                         const float *scale, const float *shift, float *finalout, float *x, void *stream)
{
    if (num_pixels <= 0 || C <= 0 || (C & 3)) return fail("et_residual_epilogue: bad sizes");
    if (!out || (!finalout && !x)) return fail("et_residual_epilogue: NULL pointer");
    if (x && !feat) return fail("et_residual_epilogue: x requested without feat");
    if (y && (!scale || !shift)) return fail("et_residual_epilogue: y given without scale/shift");
    const size_t nvec = (size_t)num_pixels * (C >> 2);
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = (hipStream_t)stream;
    if (y)
        hipLaunchKernelGGL(residual_epilogue_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, nvec, C >> 2,
                           (const float4 *)feat, (const float4 *)out, (const float4 *)y, (const float4 *)scale,
                           (const float4 *)shift, (float4 *)finalout, (float4 *)x);
    else
        hipLaunchKernelGGL(residual_epilogue_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, nvec, C >> 2,
                           (const float4 *)feat, (const float4 *)out, (const float4 *)nullptr,
                           (const float4 *)nullptr, (const float4 *)nullptr, (float4 *)finalout, (float4 *)x);
    return check_launch("et_residual_epilogue");
}

static int launch_transpose(int batch, int rows, int cols, const float *src, float *dst, void *stream,
                            const char *what)
{
    if (batch <= 0 || rows <= 0 || cols <= 0 || !src || !dst) return fail("%s: bad arguments", what);
    if (batch > 65535) return fail("%s: batch %d > 65535", what, batch);
    dim3 grid((cols + kTile - 1) / kTile, (rows + kTile - 1) / kTile, batch);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, rows, cols, src, dst);
    return check_launch(what);
}

int et_nchw_to_nhwc(int32_t N, int32_t C, int32_t H, int32_t W, const float *src, float *dst, void *stream)
{
    return launch_transpose(N, C, H * W, src, dst, stream, "et_nchw_to_nhwc");
}

int et_nhwc_to_nchw(int32_t N, int32_t C, int32_t H, int32_t W, const float *src, float *dst, void *stream)
{
    return launch_transpose(N, H * W, C, src, dst, stream, "et_nhwc_to_nchw");
}

int et_debug_host_sample_setup(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                               const float *cam, int32_t h, int32_t w, int32_t *taps, float *weights,
                               float *locs)
{
    if (int e = validate(desc)) return e;
    if (h < 0 || h >= desc->H || w < 0 || w >= desc->W) return fail("pixel out of range");
    const et::Segment seg = et::epipolar_segment(*desc, cam, xs[w], ys[h]);
    for (int k = 0; k < desc->K; ++k) {
        const et::SampleSetup su = et::sample_setup(*desc, seg, steps[k]);
        for (int r = 0; r < 4; ++r) {
            taps[k * 4 + r] = su.tap[r];
            weights[k * 4 + r] = su.weight[r];
        }
        locs[k * 2] = su.nx;
        locs[k * 2 + 1] = su.ny;
    }
    return 0;
}

}  // extern "C"
