// libepipolar_amd.so: the eval-mode residual fusion as one GEMM (C = 256) -- et_residual_gemm_pack, et_residual_gemm.
#include "et_common.h"

#ifndef ET_WGRAD_FP32
#define ET_WGRAD_FP32 0
#endif

namespace {
#include "et_wave_reduce.h"
#include "kernels_residual_gemm.inc"
}  // namespace

int et_internal_residual_rows_list(const int *perm, const int *tile_list, const int *tile_count, int tiles_per_pair, int HW,
                                   long long total_tiles, const float *out, const float *feat, const unsigned *packed,
                                   const float *bias, float *x, hipStream_t st)
{
    const int dev = current_device();
    auto kern = residual_gemm_kernel<true, false, false, true>;
    ET_GRANT_LDS(kern, kRgLdsBytes, dev);
    // two blocks per compute unit -- what is resident at once -- walk the list, two tiles per trip
    const long long want = (total_tiles + 1) / 2, cap = 2LL * device_cus(dev);
    hipLaunchKernelGGL(kern, dim3((unsigned)(want < cap ? want : cap)), dim3(256), kRgLdsBytes, st, out, feat, packed, bias, x, 0LL,
                       (float *)nullptr, (const float *)nullptr, 1, perm, tile_list, tile_count, tiles_per_pair, HW);
    return check_launch("et_epipolar_forward_fused(list rows)");
}

extern "C" {

size_t et_residual_gemm_packed_bytes(void) { return (size_t)kRgPackedWords * 4 + 256; }

int et_residual_gemm_pack(const float *wf, void *packed, void *stream)
{
    if (!wf || !packed) return fail("et_residual_gemm_pack: NULL pointer");
    if (reinterpret_cast<uintptr_t>(packed) & 15) return fail("et_residual_gemm_pack: packed buffer must be 16-byte aligned");
    hipLaunchKernelGGL(residual_gemm_pack_kernel, dim3(kRgPackBlocks), dim3(1024), 0, (hipStream_t)stream, wf,
                       reinterpret_cast<unsigned *>(packed));
    return check_launch("et_residual_gemm_pack");
}

int et_residual_gemm(int64_t num_pixels, int32_t C, const float *out, const float *feat, const void *packed,
                     const float *bias, float *x, void *stream)
{
    if (C != 256) return fail("et_residual_gemm: C = %d (the kernel is written for the 256-channel head)", C);
    if (num_pixels <= 0) return fail("et_residual_gemm: bad sizes");
    if (!out || !packed || !bias || !x) return fail("et_residual_gemm: NULL pointer");
    if (reinterpret_cast<uintptr_t>(packed) & 15) return fail("et_residual_gemm: packed buffer must be 16-byte aligned");
    const long long blocks = (num_pixels + kRgRows - 1) / kRgRows;
    if (blocks > 0x7fffffffLL) return fail("et_residual_gemm: too many rows");
    hipStream_t st = (hipStream_t)stream;
    const int dev = current_device();
    if (feat) ET_GRANT_LDS(residual_gemm_kernel<true>, kRgLdsBytes, dev);
    else ET_GRANT_LDS(residual_gemm_kernel<false>, kRgLdsBytes, dev);
    if (feat)
        hipLaunchKernelGGL(residual_gemm_kernel<true>, dim3((unsigned)blocks), dim3(256), kRgLdsBytes, st, out, feat,
                           reinterpret_cast<const unsigned *>(packed), bias, x, (long long)num_pixels);
    else
        hipLaunchKernelGGL(residual_gemm_kernel<false>, dim3((unsigned)blocks), dim3(256), kRgLdsBytes, st, out, feat,
                           reinterpret_cast<const unsigned *>(packed), bias, x, (long long)num_pixels);
    return check_launch("et_residual_gemm");
}

size_t et_z_batch_stats_workspace_bytes(int64_t num_pixels)
{
    if (num_pixels <= 0) return 0;
    return (size_t)((num_pixels + kRgRows - 1) / kRgRows) * 512 * sizeof(float);
}

// First pass of the training-mode epilogue: y = out . Wz^T + bz (written: the batch norm's input) and its per-channel batch
// mean / biased variance over all num_pixels rows (BN.py:59-82 with training = True).  `packed_wz`: et_residual_gemm_pack of the
// raw 256 x 256 z weight.
int et_z_batch_stats(int64_t num_pixels, int32_t C, const float *out, const void *packed_wz, const float *z_bias, float *y,
                     float *mean, float *var, void *workspace, size_t workspace_bytes, void *stream)
{
    if (C != 256) return fail("et_z_batch_stats: C = %d (the kernel is written for the 256-channel head)", C);
    if (num_pixels <= 1) return fail("et_z_batch_stats: batch statistics need more than one row (got %lld)", (long long)num_pixels);
    if (!out || !packed_wz || !z_bias || !y || !mean || !var || !workspace) return fail("et_z_batch_stats: NULL pointer");
    if (reinterpret_cast<uintptr_t>(packed_wz) & 15) return fail("et_z_batch_stats: packed buffer must be 16-byte aligned");
    if (workspace_bytes < et_z_batch_stats_workspace_bytes(num_pixels))
        return fail("et_z_batch_stats: workspace of %zu bytes is smaller than the %zu required", workspace_bytes,
                    et_z_batch_stats_workspace_bytes(num_pixels));
    const long long blocks = (num_pixels + kRgRows - 1) / kRgRows;
    if (blocks > 0x7fffffffLL) return fail("et_z_batch_stats: too many rows");
    hipStream_t st = (hipStream_t)stream;
    const int dev = current_device();
    auto kern = residual_gemm_kernel<false, true>;
    ET_GRANT_LDS(kern, kRgLdsBytes, dev);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), kRgLdsBytes, st, out, (const float *)nullptr,
                       reinterpret_cast<const unsigned *>(packed_wz), z_bias, y, (long long)num_pixels,
                       reinterpret_cast<float *>(workspace), static_cast<const float *>(nullptr), 1, (const int *)nullptr,
                       (const int *)nullptr, (const int *)nullptr, 0, 0);
    if (int e = check_launch("et_z_batch_stats(gemm)")) return e;
    hipLaunchKernelGGL(z_stats_finish_kernel, dim3(256), dim3(256), 0, st, reinterpret_cast<const float *>(workspace), blocks,
                       (long long)num_pixels, z_bias, mean, var);
    return check_launch("et_z_batch_stats(finish)");
}

size_t et_z_backward_workspace_bytes(int64_t num_pixels)
{
    if (num_pixels <= 0) return 0;
    return (size_t)((num_pixels + kZbRows - 1) / kZbRows) * 512 * sizeof(float) + 1024 * sizeof(float);
}

// Backward of the training-mode epilogue x = bn(z(out)) [+ out] (+ feat) w.r.t. `out` and the batch norm's affine parameters,
// from g = d loss / d x and the saved y = z(out), mean, invstd (et_z_batch_stats):
//   grad_gamma, grad_beta (256) written;  grad_y (num_pixels, 256) written -- the batch norm's input gradient, from which the
//   caller forms d Wz = grad_y^T out and d bz = sum grad_y with library calls;  grad_out (num_pixels, 256) written:
//   grad_y . Wz [+ g when zresidual].  `packed_wzt` = et_residual_gemm_pack of the TRANSPOSED z weight.
int et_z_backward(int64_t num_pixels, int32_t C, const float *g, const float *y, const float *mean, const float *invstd,
                  const float *gamma, const void *packed_wzt, int32_t zresidual, float *grad_out, float *grad_y, float *grad_gamma,
                  float *grad_beta, void *workspace, size_t workspace_bytes, void *stream)
{
    if (C != 256) return fail("et_z_backward: C = %d (the kernel is written for the 256-channel head)", C);
    if (num_pixels <= 1) return fail("et_z_backward: bad sizes");
    if (!g || !y || !mean || !invstd || !gamma || !packed_wzt || !grad_out || !grad_y || !grad_gamma || !grad_beta || !workspace)
        return fail("et_z_backward: NULL pointer");
    if (reinterpret_cast<uintptr_t>(packed_wzt) & 15) return fail("et_z_backward: packed buffer must be 16-byte aligned");
    if (workspace_bytes < et_z_backward_workspace_bytes(num_pixels))
        return fail("et_z_backward: workspace of %zu bytes is smaller than the %zu required", workspace_bytes,
                    et_z_backward_workspace_bytes(num_pixels));
    hipStream_t st = (hipStream_t)stream;
    const int dev = current_device();
    const long long sblocks = (num_pixels + kZbRows - 1) / kZbRows, gblocks = (num_pixels + kRgRows - 1) / kRgRows;
    if (gblocks > 0x7fffffffLL) return fail("et_z_backward: too many rows");
    float *partial = reinterpret_cast<float *>(workspace);
    float *coef = partial + (size_t)sblocks * 512;
    hipLaunchKernelGGL(z_bwd_sums_kernel, dim3((unsigned)sblocks), dim3(256), 0, st, g, y, mean, (long long)num_pixels, partial);
    if (int e = check_launch("et_z_backward(sums)")) return e;
    hipLaunchKernelGGL(z_bwd_finish_kernel, dim3(256), dim3(256), 0, st, partial, sblocks, (long long)num_pixels, gamma, mean, invstd,
                       coef, grad_gamma, grad_beta);
    if (int e = check_launch("et_z_backward(finish)")) return e;
    auto kern = residual_gemm_kernel<true, false, true>;
    ET_GRANT_LDS(kern, kRgLdsBytes, dev);
    hipLaunchKernelGGL(kern, dim3((unsigned)gblocks), dim3(256), kRgLdsBytes, st, y, g, reinterpret_cast<const unsigned *>(packed_wzt),
                       static_cast<const float *>(nullptr), grad_out, (long long)num_pixels, grad_y, coef, (int)(zresidual != 0),
                       (const int *)nullptr, (const int *)nullptr, (const int *)nullptr, 0, 0);
    return check_launch("et_z_backward(gemm)");
}

// The weight gradient of the training-mode epilogue: grad_w (256 out x 256 in) = grad_y^T . out, grad_b (256) = sum_rows grad_y,
// exact fp32 MFMAs, per-block partials summed in block order (no atomics).
static int wgrad_blocks(int dev, long long rows)
{
    const long long steps = (rows + kWgRows - 1) / kWgRows;
    const int cus = device_cus(dev);
    return (int)(steps < cus ? steps : cus);
}

size_t et_z_wgrad_workspace_bytes(int64_t num_pixels)
{
    if (num_pixels <= 0) return 0;
    return (size_t)wgrad_blocks(current_device(), (long long)num_pixels) * (65536 + 256) * sizeof(float);   // (one partial result per block)
}

int et_z_wgrad(int64_t num_pixels, int32_t C, const float *grad_y, const float *out, float *grad_w, float *grad_b, void *workspace,
               size_t workspace_bytes, void *stream)
{
    if (C != 256) return fail("et_z_wgrad: C = %d (the kernel is written for the 256-channel head)", C);
    if (num_pixels <= 0) return fail("et_z_wgrad: bad sizes");
    if (!grad_y || !out || !grad_w || !grad_b || !workspace) return fail("et_z_wgrad: NULL pointer");
    if (workspace_bytes < et_z_wgrad_workspace_bytes(num_pixels))
        return fail("et_z_wgrad: workspace of %zu bytes is smaller than the %zu required", workspace_bytes,
                    et_z_wgrad_workspace_bytes(num_pixels));
    if ((long long)num_pixels * 1024 >= (1LL << 32) * 256) return fail("et_z_wgrad: too many rows");
    hipStream_t st = (hipStream_t)stream;
    const int dev = current_device();
    const int blocks = wgrad_blocks(dev, (long long)num_pixels);
    long long per = (((long long)num_pixels + blocks - 1) / blocks + kWgRows - 1) / kWgRows * kWgRows;
    // (the kernel forms byte offsets into a block's share as 32-bit ints: (k * 16 + row) * 1024 + chunk)
    if ((per + kWgRows) * 1024 >= (1LL << 31)) return fail("et_z_wgrad: a block's share of the rows must stay below 2 GiB");
    float *pw = reinterpret_cast<float *>(workspace), *pb = pw + (size_t)blocks * 65536;
#if ET_WGRAD_FP32      // (development: the exact-fp32 MFMA form of rounds 5-6)
    ET_GRANT_LDS(z_wgrad_kernel, kWgLdsBytes, dev);
    hipLaunchKernelGGL(z_wgrad_kernel, dim3((unsigned)blocks), dim3(512), kWgLdsBytes, st, grad_y, out, (long long)num_pixels, per, pw, pb);
#else
    ET_GRANT_LDS(z_wgrad_bf16x3_kernel, kWgbLdsBytes, dev);
    hipLaunchKernelGGL(z_wgrad_bf16x3_kernel, dim3((unsigned)blocks), dim3(512), kWgbLdsBytes, st, grad_y, out, (long long)num_pixels, per,
                       pw, pb);
#endif
    if (int e = check_launch("et_z_wgrad")) return e;
    hipLaunchKernelGGL(z_wgrad_finish_kernel, dim3(257), dim3(256), 0, st, pw, pb, blocks, grad_w, grad_b);
    return check_launch("et_z_wgrad(finish)");
}

}  // extern "C"
