// libepipolar_amd.so: ABI version / error text, sample_locs, residual epilogue, layout converters, host test hook.
#include "et_common.h"

thread_local char et_g_err[512] = "";

namespace {
#include "kernels_misc.inc"      // sample_locs_kernel, residual_epilogue_kernel, transpose_kernel
}  // namespace

extern "C" {

int et_abi_version(void) { return ET_ABI_VERSION; }

const char *et_last_error(void) { return et_g_err; }

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

int et_debug_atomic_probe(float *dst, int64_t rows, int32_t blocks, int32_t iters, void *stream)
{
    if (!dst || rows < 8 || rows > (1LL << 22) - 1 || blocks <= 0 || iters <= 0) return fail("et_debug_atomic_probe: bad arguments");
    hipLaunchKernelGGL(atomic_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dst, (unsigned)rows, iters);
    return check_launch("et_debug_atomic_probe");
}

int et_heatmap_peaks(int64_t num_maps, int32_t H, int32_t W, const float *heatmaps, float radius, float downsample,
                     float threshold, int32_t legacy_floor_division, float *locs, float *scores, void *stream)
{
    if (num_maps <= 0 || num_maps > 0x7fffffffLL || H < 2 || W < 2) return fail("et_heatmap_peaks: bad sizes");
    if (!(radius > 0.f) || (int)(radius + 0.5f) < 1) return fail("et_heatmap_peaks: radius %g", radius);
    if (!heatmaps || !locs || !scores) return fail("et_heatmap_peaks: NULL pointer");
    hipLaunchKernelGGL(heatmap_peaks_kernel, dim3((unsigned)num_maps), dim3(256), 0, (hipStream_t)stream, heatmaps, H, W,
                       radius, downsample, threshold, legacy_floor_division, locs, scores);
    return check_launch("et_heatmap_peaks");
}

}  // extern "C"
