// MI355X (gfx950 / CDNA4) kernels of the Epipolar Transformer hot path and the
// C ABI declared in include/epipolar_amd.h.  Written for wave64 only.
//
// Layout of this translation unit:
//   epipolar_geometry.h    bit-faithful float32 geometry (segment, sample set-up), host+device
//   this file              shared device helpers (cross-lane reductions, buffer addressing),
//                          host-side dispatch and the extern "C" entry points
//   kernels_forward_tile.inc   C == 256 head, forward: reference pixels ordered by epipolar line, 32 per tile, two fp32
//                          GEMMs per tile on the matrix cores with the resampling / soft-max between them
//   kernels_forward.inc    fused forward, any shape: one pixel per wave (epipolar_fwd_kernel) and four
//                          pixels per wave in lockstep (epipolar_fwd_multi_kernel)
//   kernels_backward_tile.inc  C == 256 head, backward in the same tile form (five GEMMs per tile, d(feat_src)
//                          accumulated across tiles with float atomics)
//   kernels_backward.inc   backward, any shape: coefficient emission + scan / bucket / ordered gather
//                          (no float atomics, bit-reproducible), and the float-atomic scatter fallback
//   kernels_misc.inc       sample_locs, residual epilogue, NCHW <-> NHWC
//
// Common ideas:
//   * lanes <-> samples for the per-pixel geometry, lanes <-> channels for the arithmetic;
//   * a 2x2 parity-addressed tap register cache: a source row is fetched once per pixel,
//     not once per sample that touches it;
//   * cross-lane sums with v_permlane32_swap / v_permlane16_swap / DPP, online masked soft-max:
//     the K x C sampled strip never exists anywhere;
//   * raw buffer resources with scalar row offsets (no vector address arithmetic);
//   * the attention tile of 16 consecutive pixels is staged in LDS and written as 64-byte
//     rows of the reference's (N,K,H,W) `depth` layout;
//   * blockIdx is remapped so each XCD walks whole pairs (its L2 keeps the 4 MiB source map
//     of the pair it is working on).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <type_traits>

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
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

__device__ __forceinline__ f32x2 buf_load_f2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
}

__device__ __forceinline__ float4 f4_fma(float s, const float4 &a, const float4 &c)
{
    return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

__device__ __forceinline__ float4 f4_mul(float s, const float4 &a)
{
    return make_float4(s * a.x, s * a.y, s * a.z, s * a.w);
}

// a . b through packed fp32: (a.xy * b.xy), fma with (a.zw, b.zw), one add
__device__ __forceinline__ float f4_dot(const float4 &a, const float4 &b)
{
    const f32x2 lo = f32x2{a.x, a.y} * f32x2{b.x, b.y};
    const f32x2 t = __builtin_elementwise_fma(f32x2{a.z, a.w}, f32x2{b.z, b.w}, lo);
    return t.x + t.y;
}

// The kernels themselves (same translation unit and anonymous namespace):
#include "kernels_forward.inc"   // SampleTable, epipolar_fwd_kernel, epipolar_fwd_multi_kernel
#include "kernels_forward_tile.inc"  // tile_order_kernel, epipolar_fwd_tile_kernel (MFMA formulation)
#include "kernels_backward.inc"
#include "kernels_backward_tile.inc" // epipolar_bwd_tile_kernel (MFMA formulation, float-atomic accumulation of d(feat_src))  // epipolar_bwd_kernel, epipolar_bwd_emit_kernel, bwd_scan/bucket, epipolar_bwd_gather_kernel
#include "kernels_misc.inc"      // sample_locs_kernel, residual_epilogue_kernel, transpose_kernel

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
    int v = desc->variant & ~(ET_VARIANT_NO_TILE | ET_VARIANT_TILE_SPLIT);
    if ((v & ~(ET_VARIANT_ABLATE_NO_LOADS | ET_VARIANT_ABLATE_ONE_ROW)) == 0)
        v |= (desc->C == 256 && kpl == 1) ? ET_VARIANT_MULTI4   // K > 64: its LDS records cut occupancy (measured 1.8x slower)
                                          : (ET_VARIANT_BATCH4 | ET_VARIANT_OCC5 | ET_VARIANT_PIXEL_INTERLEAVE);
    if (v & ET_VARIANT_BASELINE)
        v &= ~(ET_VARIANT_BATCH4 | ET_VARIANT_OCC5 | ET_VARIANT_OCC6 | ET_VARIANT_PIXEL_INTERLEAVE |
               ET_VARIANT_MULTI2 | ET_VARIANT_MULTI4);
    p.ablate = (v & ET_VARIANT_ABLATE_NO_LOADS) ? 1 : (v & ET_VARIANT_ABLATE_ONE_ROW) ? 2 : 0;
    p.interleave = (v & ET_VARIANT_PIXEL_INTERLEAVE) ? 1 : 0;
    if ((v & (ET_VARIANT_MULTI2 | ET_VARIANT_MULTI4)) && desc->C == 256 && kpl <= 2) {
        // several pixels per wave; per-wave LDS: PPW * KP * 32 + PPW * 16 bytes
        const int ppw = (v & ET_VARIANT_MULTI4) ? 4 : 2;
        const size_t lds_m = (attn ? (size_t)desc->K * kPixPerBlock * sizeof(float) : 0) +
                             (size_t)kWavesPerBlock * (ppw * kpl * kWave * 32 + ppw * 16);
        const bool occ4 = v & ET_VARIANT_OCC5;   // multi kernels: compile for 4 waves per SIMD (128 VGPRs)
        const bool pipe = v & ET_VARIANT_PIPELINE;
#define ET_MULTI(P, Q, KK)                                                                                     \
    do {                                                                                                       \
        if (pipe) hipLaunchKernelGGL((epipolar_fwd_multi_kernel<P, Q, KK, true, 1>), grid, dim3(256), lds_m, st, p);   \
        else if (occ4) hipLaunchKernelGGL((epipolar_fwd_multi_kernel<P, Q, KK, false, 4>), grid, dim3(256), lds_m, st, p); \
        else hipLaunchKernelGGL((epipolar_fwd_multi_kernel<P, Q, KK, false, 1>), grid, dim3(256), lds_m, st, p);    \
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

// The MFMA tile path applies to the 256-channel head when one reference pixel alone can never
// overflow the tile's row array: a pixel's K samples touch at most 4K source pixels, and a line
// through a W x H map at most 4 per column (or per row, whichever way it runs), i.e. 4 max(W, H).
static int *g_tile_stats = nullptr;  // tuning hooks, see et_debug_tile_stats / et_debug_tile_ablate
static int g_tile_ablate_fwd = 0, g_tile_ablate_bwd = 0;

// one pixel's K samples touch at most 4K source pixels, and a line through a W x H map at most 4 per column (or per
// row, whichever way it runs), i.e. 4 max(W, H)
static int tile_rows_per_pixel(const EtLayerDesc *d)
{
    const int longest = d->W > d->H ? d->W : d->H;
    return (d->K < longest) ? 4 * d->K : 4 * longest;
}
// rows per tile the kernel is instantiated with: 256 up to 64 x 64 maps, 384 beyond (longer lines), 512 when a
// single pixel may need more than that
static int tile_rows(const EtLayerDesc *d)
{
    if (tile_rows_per_pixel(d) > kTileRowsLarge) return kTileRowsHuge;
    return (d->W > 64 || d->H > 64) ? kTileRowsLarge : kTileRowsSmall;
}
static int tile_rows_cap(const EtLayerDesc *d) { return (d->variant & ET_VARIANT_TILE_SPLIT) ? 64 : tile_rows(d); }

static bool tile_eligible(const EtLayerDesc *d)
{
    if (d->C != 256 || d->K > 256) return false;
    const long long hw = (long long)d->H * d->W;
    if (hw > 16384) return false;  // bitonic sort of one pair's pixels lives in LDS
    return tile_rows_per_pixel(d) <= tile_rows_cap(d);
}

int et_debug_tile_stats(int32_t *device_buffer)
{
    g_tile_stats = device_buffer;
    return 0;
}

int et_debug_tile_ablate(int32_t forward_bits, int32_t backward_bits)
{
    g_tile_ablate_fwd = forward_bits;
    g_tile_ablate_bwd = backward_bits;
    return 0;
}

size_t et_epipolar_forward_workspace_bytes(const EtLayerDesc *desc)
{
    if (validate(desc) || !tile_eligible(desc)) return 0;
    const size_t tiles = ((size_t)desc->H * desc->W + kTilePix - 1) / kTilePix;
    return (size_t)desc->N * tiles * kTilePix * sizeof(int) + 256u;
}

int et_epipolar_forward_tiled(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                              const float *cam, const float *feat_ref, const float *feat_src, float *out,
                              float *attn, float *corr_pos, const float *res_bias, float *res_base,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    if (int e = validate(desc)) return e;
    if (!xs || !ys || !steps || !cam || !feat_ref || !feat_src || !out)
        return fail("et_epipolar_forward_tiled: NULL pointer");
    if (res_bias && !res_base) return fail("et_epipolar_forward_tiled: res_bias given without res_base");
    if (!tile_eligible(desc))
        return fail("et_epipolar_forward_tiled: needs C == 256, H*W <= 16384 and 4 min(K, max(W,H)) <= %d "
                    "(got C=%d H=%d W=%d K=%d); use et_epipolar_forward", tile_rows_cap(desc), desc->C, desc->H, desc->W, desc->K);
    const size_t need = et_epipolar_forward_workspace_bytes(desc);
    if (!workspace || workspace_bytes < need)
        return fail("et_epipolar_forward_tiled: workspace of %zu bytes is smaller than the %zu required",
                    workspace ? workspace_bytes : (size_t)0, need);
    hipStream_t st = (hipStream_t)stream;
    const int HW = desc->H * desc->W;
    TileParams tp;
    FwdParams &p = tp.f;
    p.d = *desc;
    p.xs = xs; p.ys = ys; p.steps = steps; p.cam = cam;
    p.fref = feat_ref; p.fsrc = feat_src;
    p.out = out; p.attn = attn; p.corr = corr_pos;
    p.res_bias = res_bias; p.res_base = res_base;
    p.interleave = 0; p.ablate = 0;
    tp.tiles_per_pair = (HW + kTilePix - 1) / kTilePix;
    p.blocks_per_pair = tp.tiles_per_pair;
    const long long total = (long long)tp.tiles_per_pair * desc->N;
    if (total > 0x7fffffffLL) return fail("grid too large");
    p.total_blocks = (int)total;
    tp.hw_words = (HW + 31) / 32;
    tp.stats = g_tile_stats;
    tp.ablate = g_tile_ablate_fwd;
    tp.rows_cap = tile_rows_cap(desc);
    int *perm = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    tp.perm = perm;
    // 1. order every pair's reference pixels by their epipolar line
    int n2 = 64;
    while (n2 < HW) n2 <<= 1;
    const size_t lds_sort = (size_t)n2 * sizeof(unsigned long long);
    if (lds_sort > 48 * 1024) {
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(tile_order_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort);
        if (ae != hipSuccess) return fail("hipFuncSetAttribute(tile_order_kernel): %s", hipGetErrorString(ae));
    }
    if (!(tp.ablate & 32))
    hipLaunchKernelGGL(tile_order_kernel, dim3(desc->N), dim3(1024), lds_sort, st, *desc, xs, ys, cam, n2,
                       tp.tiles_per_pair * kTilePix, perm);
    if (tp.ablate & 16) return 0;
    if (int e = check_launch("et_epipolar_forward_tiled(order)")) return e;
    // 2. one block per tile
    const int kpl = (desc->K + 63) / 64;
    const int rows = tile_rows(desc);
    const size_t lds = (size_t)(tile_array_floats(rows) + rows + kTilePix + 4 + kTilePix * 4) * 4 +
                       (size_t)tp.hw_words * 8 + (kpl == 1 ? (size_t)kTilePix * kWave * 8 : 0);
#define ET_TILE(KK, RR)                                                                                          \
    do {                                                                                                         \
        if (lds > 48 * 1024) {                                                                                   \
            hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(epipolar_fwd_tile_kernel<KK, RR>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
            if (ae != hipSuccess) return fail("hipFuncSetAttribute(tile kernel): %s", hipGetErrorString(ae));    \
        }                                                                                                        \
        hipLaunchKernelGGL((epipolar_fwd_tile_kernel<KK, RR>), dim3((unsigned)total), dim3(256), lds, st, tp);   \
    } while (0)
    if (rows == kTileRowsSmall) {
        if (kpl == 1) ET_TILE(1, kTileRowsSmall);
        else if (kpl == 2) ET_TILE(2, kTileRowsSmall);
        else ET_TILE(4, kTileRowsSmall);
    } else if (rows == kTileRowsLarge) {
        if (kpl == 1) ET_TILE(1, kTileRowsLarge);
        else if (kpl == 2) ET_TILE(2, kTileRowsLarge);
        else ET_TILE(4, kTileRowsLarge);
    } else {
        if (kpl == 2) ET_TILE(2, kTileRowsHuge);   // (512 rows per pixel need K > 96)
        else ET_TILE(4, kTileRowsHuge);
    }
#undef ET_TILE
    return check_launch("et_epipolar_forward_tiled");
}

size_t et_epipolar_backward_tiled_workspace_bytes(const EtLayerDesc *desc)
{
    if (validate(desc) || !tile_eligible(desc) || desc->K > 64) return 0;
    return et_epipolar_forward_workspace_bytes(desc);
}

int et_epipolar_backward_tiled(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                               const float *cam, const float *feat_ref, const float *feat_src,
                               const float *grad_out, float *grad_ref, float *grad_src, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    if (int e = validate(desc)) return e;
    if (!xs || !ys || !steps || !cam || !feat_ref || !feat_src || !grad_out || !grad_ref || !grad_src)
        return fail("et_epipolar_backward_tiled: NULL pointer");
    const size_t need = et_epipolar_backward_tiled_workspace_bytes(desc);
    if (need == 0)
        return fail("et_epipolar_backward_tiled: needs C == 256, K <= 64, H*W <= 16384 and 4 min(K, max(W,H)) <= %d "
                    "(got C=%d H=%d W=%d K=%d); use et_epipolar_backward", tile_rows_cap(desc), desc->C, desc->H,
                    desc->W, desc->K);
    if (!workspace || workspace_bytes < need)
        return fail("et_epipolar_backward_tiled: workspace of %zu bytes is smaller than the %zu required",
                    workspace ? workspace_bytes : (size_t)0, need);
    hipStream_t st = (hipStream_t)stream;
    const int HW = desc->H * desc->W;
    BwdTileParams tp;
    std::memset(&tp, 0, sizeof(tp));
    BwdParams &p = tp.b;
    p.d = *desc;
    p.xs = xs; p.ys = ys; p.steps = steps; p.cam = cam;
    p.fref = feat_ref; p.fsrc = feat_src; p.gout = grad_out;
    p.gref = grad_ref; p.gsrc = grad_src;
    tp.tiles_per_pair = (HW + kTilePix - 1) / kTilePix;
    p.blocks_per_pair = tp.tiles_per_pair;
    const long long total = (long long)tp.tiles_per_pair * desc->N;
    if (total > 0x7fffffffLL) return fail("grid too large");
    p.total_blocks = (int)total;
    tp.hw_words = (HW + 31) / 32;
    tp.rows_cap = tile_rows_cap(desc);
    tp.ablate = g_tile_ablate_bwd;
    int *perm = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    tp.perm = perm;
    hipError_t me = hipMemsetAsync(grad_src, 0, (size_t)desc->N * HW * desc->C * sizeof(float), st);
    if (me != hipSuccess) return fail("hipMemsetAsync(grad_src): %s", hipGetErrorString(me));
    int n2 = 64;
    while (n2 < HW) n2 <<= 1;
    const size_t lds_sort = (size_t)n2 * sizeof(unsigned long long);
    if (lds_sort > 48 * 1024) {
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(tile_order_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort);
        if (ae != hipSuccess) return fail("hipFuncSetAttribute(tile_order_kernel): %s", hipGetErrorString(ae));
    }
    hipLaunchKernelGGL(tile_order_kernel, dim3(desc->N), dim3(1024), lds_sort, st, *desc, xs, ys, cam, n2,
                       tp.tiles_per_pair * kTilePix, perm);
    if (int e = check_launch("et_epipolar_backward_tiled(order)")) return e;
    const int rows = tile_rows(desc);
    const size_t lds = (size_t)(tile_array_floats(rows) + rows + kTilePix + 4 + kTilePix * 4) * 4 +
                       (size_t)tp.hw_words * 8 + (size_t)kTilePix * kWave * 8;
#define ET_BTILE(RR)                                                                                            \
    do {                                                                                                        \
        if (lds > 48 * 1024) {                                                                                  \
            hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(epipolar_bwd_tile_kernel<RR>),   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
            if (ae != hipSuccess) return fail("hipFuncSetAttribute(bwd tile kernel): %s", hipGetErrorString(ae)); \
        }                                                                                                       \
        hipLaunchKernelGGL((epipolar_bwd_tile_kernel<RR>), dim3((unsigned)total), dim3(256), lds, st, tp);      \
    } while (0)
    if (rows == kTileRowsSmall) ET_BTILE(kTileRowsSmall);
    else ET_BTILE(kTileRowsLarge);
#undef ET_BTILE
    return check_launch("et_epipolar_backward_tiled");
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
