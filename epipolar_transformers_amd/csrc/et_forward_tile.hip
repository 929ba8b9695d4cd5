// libepipolar_amd.so: the MFMA tile formulation of the forward (et_epipolar_forward_tiled).
#include "et_common.h"
#include <cstdlib>

namespace {
#include "kernels_forward_tile.inc"     // tile_order_kernel, epipolar_fwd_tile_kernel / _list_kernel
#include "kernels_forward_tile_ws.inc"  // epipolar_fwd_tile_ws_kernel (warp-specialised, persistent): the default
#include "kernels_source_planes.inc"    // source_planes_kernel (the source maps as split-fp16 planes, once per call)
#include "kernels_forward_tile_ws2.inc" // epipolar_fwd_tile_ws2_kernel (pre-split source planes; ET_VARIANT_WS_V2 only)
}  // namespace
#include "et_tile_host.h"

#ifdef ET_WS_PROFILE
static long long *g_ws_prof = nullptr;   // profiling builds only (python -m epipolar_transformers_amd.build --profile)
extern "C" int et_dev_ws_profile(long long *device_buffer)
{
    g_ws_prof = device_buffer;
    return 0;
}
#endif

extern "C" {

size_t et_epipolar_forward_workspace_bytes(const EtLayerDesc *desc)
{
    if (validate(desc) || !tile_eligible(desc)) return 0;
    const size_t tiles = (size_t)desc->N * (((size_t)desc->H * desc->W + kTilePix - 1) / kTilePix);
    size_t words = tile_workspace_words(tiles, (size_t)desc->N);
    if (tile_ws2_eligible(desc)) words += tile_workspace_plane_words((size_t)desc->N, (size_t)desc->H * desc->W);
    return words * sizeof(int) + 256u;
}

size_t et_epipolar_forward_workspace_error_offset(const EtLayerDesc *desc)
{
    if (validate(desc) || !tile_eligible(desc)) return 0;
    return sizeof(int);   // word 1 of the header: the same place for every shape (a workspace is reused across shapes)
}

size_t et_epipolar_forward_workspace_stats_offset(const EtLayerDesc *desc)
{
    if (validate(desc) || !tile_eligible(desc)) return 0;
    const size_t tiles = (size_t)desc->N * (((size_t)desc->H * desc->W + kTilePix - 1) / kTilePix);
    return (kTileWorkspaceHeaderWords + tiles * kTilePix + tiles) * sizeof(int);
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
    if (total > 0x7fffffffLL / kTilePix) return fail("grid too large");
    p.total_blocks = (int)total;
    tp.hw_words = (HW + 31) / 32;
    tp.rows_cap = tile_rows_cap(desc);
    const TileWorkspace w = carve_tile_workspace(workspace, (size_t)total, (size_t)desc->N, (size_t)HW);
    tp.perm = w.perm;
    tp.stats = w.stats;
    tp.tile_list = w.ovf_list;
    tp.tile_count = w.ovf_count;
    // 1. order every pair's reference pixels by their epipolar line (also clears the overflow counter)
    int n2 = 64;
    while (n2 < HW) n2 <<= 1;
    const size_t lds_sort = (size_t)n2 * sizeof(unsigned long long);
    const int dev = current_device();
    ET_GRANT_LDS(tile_order_kernel, lds_sort, dev);
    // (per-pair scale estimates of the source maps: for the split-fp16 GEMMs of the first-generation persistent kernel and
    //  of the one-block-per-tile kernel; ET_VARIANT_TILE_EXACT keeps the latter in exact fp32)
    float *scales = w.scales;
    // soft-max off: exact fp32 throughout, as the header promises (the first GEMM feeds the `== 0 -> -1e10` mask and the
    // "attention" sim / K is unbounded: no fp16 form of the B rows)
    tp.scales = ((desc->variant & ET_VARIANT_TILE_EXACT) || !desc->softmax_enabled) ? nullptr : w.scales;
    hipLaunchKernelGGL(tile_order_kernel, dim3(desc->N), dim3(1024), lds_sort, st, *desc, xs, ys, cam, n2,
                       tp.tiles_per_pair * kTilePix, w.perm, w.ovf_count, feat_ref, feat_src, scales, w.segs, w.band);
    if (int e = check_launch("et_epipolar_forward_tiled(order)")) return e;
    const int kpl = (desc->K + 63) / 64;
    const int rows = tile_rows(desc);
    const size_t lds = (size_t)(fwd_tile_array_floats(rows) + rows + kTilePix + 48 + kTilePix * 4) * 4 +
                       (size_t)tp.hw_words * 8 + (kpl == 1 ? (size_t)kTilePix * kWave * 8 : 0);
#define ET_SET_LDS(KERNEL, BYTES) ET_GRANT_LDS(KERNEL, BYTES, dev)
    if (tile_ws2_eligible(desc)) {
        // 2a. (ET_VARIANT_WS_V2) the source maps as split-fp16 planes, one pass over the batch (HBM-bound) ...
        const long long nrows = (long long)desc->N * HW;
        const long long pblocks = (nrows + 4 * kPlaneRowsPerWave - 1) / (4 * kPlaneRowsPerWave);
        if (pblocks > 0x7fffffffLL) return fail("grid too large");
        hipLaunchKernelGGL(source_planes_kernel, dim3((unsigned)pblocks), dim3(256), 0, st, feat_src, w.planes, w.rowinv, nrows);
        if (int e = check_launch("et_epipolar_forward_tiled(planes)")) return e;
        // ... 2b. one persistent block per CU, matrix and vector waves specialised (kernels_forward_tile_ws2.inc) ...
        TileWs2Params wp;
        wp.f = p;
        wp.perm = w.perm;
        wp.tiles_per_pair = tp.tiles_per_pair;
        wp.total_tiles = (int)total;
        wp.rows_cap = tp.rows_cap;
        wp.ovf_count = w.ovf_count;
        wp.ovf_list = w.ovf_list;
        wp.stats = w.stats;
        wp.err = w.err;
        wp.segs = w.segs;
        wp.planes = w.planes;
        wp.rowinv = w.rowinv;
        wp.experiment = 0;
#ifdef ET_WS_PROFILE
        wp.prof = g_ws_prof;
        if (const char *e = getenv("ET_WS_EXPERIMENT")) wp.experiment = atoi(e);
#else
        wp.prof = nullptr;
#endif
        const int cus = device_cus(dev);
        const unsigned grid = (unsigned)(total < cus ? total : cus);
        const size_t lds_ws = tile_ws2_lds_bytes(kTileRowsSmall);
        ET_SET_LDS((epipolar_fwd_tile_ws2_kernel<kTileRowsSmall>), lds_ws);
        hipLaunchKernelGGL((epipolar_fwd_tile_ws2_kernel<kTileRowsSmall>), dim3(grid), dim3((kWsMatrixWaves + 8) * kWave), lds_ws, st, wp);
        if (int e = check_launch("et_epipolar_forward_tiled(ws2)")) return e;
        // ... 2c. and the tiles it left over (row sets beyond its arrays; normally none) one block per tile
        const unsigned lgrid = (unsigned)(total < 2LL * cus ? total : 2LL * cus);
        ET_SET_LDS((epipolar_fwd_tile_list_kernel<1, kTileRowsSmall>), lds);
        hipLaunchKernelGGL((epipolar_fwd_tile_list_kernel<1, kTileRowsSmall>), dim3(lgrid), dim3(256), lds, st, tp);
        return check_launch("et_epipolar_forward_tiled(list)");
    }
    if (tile_ws_eligible(desc)) {
        // 2a'. the first-generation persistent kernel (kernels_forward_tile_ws.inc): the default ...
        TileWsParams wp;
        wp.f = p;
        wp.perm = w.perm;
        wp.tiles_per_pair = tp.tiles_per_pair;
        wp.total_tiles = (int)total;
        wp.rows_cap = tp.rows_cap;
        wp.ovf_count = w.ovf_count;
        wp.ovf_list = w.ovf_list;
        wp.stats = w.stats;
        wp.scales = w.scales;
        wp.segs = w.segs;
        wp.band = w.band;
        wp.setprio = (desc->variant & ET_VARIANT_WS_SETPRIO) ? 1 : 0;
#ifdef ET_WS_PROFILE
        wp.prof = g_ws_prof;
        if (const char *e = getenv("ET_WS_EXPERIMENT")) wp.setprio |= atoi(e);
#else
        wp.prof = nullptr;
#endif
        const int cus = device_cus(dev);
        const unsigned grid = (unsigned)(total < cus ? total : cus);
        const size_t lds_ws = tile_ws_lds_bytes(kTileRowsSmall, desc->H, desc->W);
        ET_SET_LDS((epipolar_fwd_tile_ws_kernel<kTileRowsSmall, 8>), lds_ws);
        hipLaunchKernelGGL((epipolar_fwd_tile_ws_kernel<kTileRowsSmall, 8>), dim3(grid), dim3((kWsMatrixWaves + 8) * kWave),
                           lds_ws, st, wp);
        if (int e = check_launch("et_epipolar_forward_tiled(ws)")) return e;
        // ... 2b'. and the tiles it left over one block per tile
        const unsigned lgrid = (unsigned)(total < 2LL * cus ? total : 2LL * cus);
        ET_SET_LDS((epipolar_fwd_tile_list_kernel<1, kTileRowsSmall>), lds);
        hipLaunchKernelGGL((epipolar_fwd_tile_list_kernel<1, kTileRowsSmall>), dim3(lgrid), dim3(256), lds, st, tp);
        return check_launch("et_epipolar_forward_tiled(list)");
    }
    // 2. one block per tile
#define ET_TILE(KK, RR)                                                                                          \
    do {                                                                                                         \
        ET_SET_LDS((epipolar_fwd_tile_kernel<KK, RR>), lds);                                                     \
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
#undef ET_SET_LDS
    return check_launch("et_epipolar_forward_tiled");
}

}  // extern "C"
