// Shared by every translation unit of libepipolar_amd.so (MI355X / gfx950, wave64 only): error reporting, launch
// parameter blocks, and the device helpers (cross-lane reductions, buffer addressing) the kernels are built from.
//
// Translation units (compiled in parallel by epipolar_transformers_amd/build.py, linked into one library):
//   et_forward.hip        kernels_forward.inc        fused forward, any shape: one pixel per wave (epipolar_fwd_kernel)
//                                                    and four pixels per wave in lockstep (epipolar_fwd_multi_kernel)
//   et_forward_tile.hip   kernels_forward_tile.inc   C == 256 head, forward: reference pixels ordered by epipolar line,
//                         kernels_forward_tile_ws.inc 32 per tile, two fp32 GEMMs per tile on the matrix cores with the
//                                                    resampling / soft-max between them; persistent warp-specialised form
//   et_backward.hip       kernels_backward.inc       backward, any shape: coefficient emission + scan / bucket / ordered
//                                                    gather (no float atomics, bit-reproducible), float-atomic fallback
//   et_backward_tile.hip  kernels_backward_tile.inc  C == 256 head, backward in the same tile form
//   et_misc.hip           kernels_misc.inc           sample_locs, residual epilogue, NCHW <-> NHWC, ABI version / errors
//   epipolar_geometry.h                              bit-faithful float32 geometry (segment, sample set-up), host+device
//
// Common ideas:
//   * lanes <-> samples for the per-pixel geometry, lanes <-> channels for the arithmetic;
//   * cross-lane sums with v_permlane32_swap / v_permlane16_swap / DPP, masked soft-max without ever storing the
//     K x C sampled strip;
//   * raw buffer resources with scalar row offsets (no vector address arithmetic);
//   * blockIdx / tile lists are laid out so each XCD walks whole pairs (its L2 keeps the 4 MiB source map of the
//     pair it is working on).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "epipolar_amd.h"
#include "epipolar_geometry.h"

// thread-local error text behind et_last_error() (defined in et_misc.hip)
extern thread_local char et_g_err[512];

namespace {

// ----------------------------------------------------------------------------
// error reporting
// ----------------------------------------------------------------------------

int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(et_g_err, sizeof(et_g_err), fmt, ap);
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
// uint32 words of the packed weight of the residual GEMM (residual_gemm_pack_kernel): [k-step][n-block][hi|lo][lane][16 B];
// the float behind them is 1 / (the power-of-two scale the weight was split under).  Read by residual_gemm_kernel and by the
// fused forward (third GEMM of the persistent kernel).
constexpr int kRgPackedWords = 16 * 8 * 2 * 64 * 4;

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

// streaming variant (read once: marked non-temporal so it does not push the re-used source rows out of L2)
__device__ __forceinline__ float4 buf_load_f4_nt(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float buf_load_f1(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

__device__ __forceinline__ float buf_load_f1_nt(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2));
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

// ----------------------------------------------------------------------------
// host-side dispatch
// ----------------------------------------------------------------------------
// compute units of a device (the persistent kernels launch one block per CU): queried once per device and process
int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    return dev;
}
int device_cus(int dev)
{
    static int cached[64];   // zero-initialised; relaxed atomics: every thread would store the same value
    int v = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        __atomic_store_n(&cached[dev], v, __ATOMIC_RELAXED);
    }
    return v;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (call site, device) and process instead of on every launch:
// `slot` is a static int[64] owned by the call site, holding the largest size already granted on each device.
template <class Kernel>
int grant_lds(Kernel kernel, size_t bytes, int dev, int *slot, const char *name)
{
    if (bytes <= 48 * 1024) return 0;
    if ((size_t)__atomic_load_n(&slot[dev], __ATOMIC_RELAXED) >= bytes) return 0;
    hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (ae != hipSuccess) return fail("hipFuncSetAttribute(%s, %zu bytes): %s", name, bytes, hipGetErrorString(ae));
    __atomic_store_n(&slot[dev], (int)bytes, __ATOMIC_RELAXED);
    return 0;
}
#define ET_GRANT_LDS(KERNEL, BYTES, DEV)                                            \
    do {                                                                            \
        static int granted_[64];                                                    \
        if (int e_ = grant_lds((KERNEL), (BYTES), (DEV), granted_, #KERNEL)) return e_; \
    } while (0)

int validate(const EtLayerDesc *d)
{
    if (!d) return fail("desc is NULL");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0) return fail("bad shape N=%d H=%d W=%d", d->N, d->H, d->W);
    if (d->C <= 0 || (d->C & 3)) return fail("C=%d must be a positive multiple of 4", d->C);
    if (d->C > 512) return fail("C=%d > 512 not supported", d->C);
    if (d->K < 2 || d->K > 256) return fail("K=%d outside [2, 256]", d->K);
    if ((long long)d->H * d->W * d->C * 4 >= (1LL << 31)) return fail("one feature map must stay below 2 GiB");
#ifndef ET_DEV_ABLATE
    if (d->variant & (64 | 128)) return fail("variant bits 64 / 128 (roofline ablations, wrong results) exist in -DET_DEV_ABLATE builds only");
#endif
    if (!(d->downsample > 0.f) || !(d->image_resize > 0.f) || !(d->predict_resize > 0.f))
        return fail("downsample / resize factors must be positive");
    return 0;
}

}  // namespace

// Internal to the library (hidden: not part of the C ABI), defined in et_residual_gemm.hip: x = feat + bias + out . Wf^T for the
// pixel rows of a device-side LIST of tiles -- the fused forward's left-over tiles (et_forward_tile.hip).  Returns the status.
__attribute__((visibility("hidden"))) int et_internal_residual_rows_list(const int *perm, const int *tile_list, const int *tile_count,
                                                                         int tiles_per_pair, int HW, long long total_tiles,
                                                                         const float *out, const float *feat, const unsigned *packed,
                                                                         const float *bias, float *x, hipStream_t st);
