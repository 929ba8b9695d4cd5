// libepipolar_amd.so: the MFMA tile formulation of the forward (et_epipolar_forward_tiled).
#include "et_common.h"
#include <cstdlib>

namespace {
#include "kernels_forward_tile.inc"     // tile_order_kernel, epipolar_fwd_tile_kernel / _list_kernel
#include "kernels_forward_tile_ws.inc"  // epipolar_fwd_tile_ws_kernel (warp-specialised, persistent): the default

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
    return tile_workspace_words(tiles, (size_t)desc->N, (size_t)desc->H * desc->W) * sizeof(int) + 256u;
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
    const int dev = current_device();
    // (per-pair scale estimates of the source maps: for the split-fp16 GEMMs of the persistent kernel and
    //  of the one-block-per-tile kernel; ET_VARIANT_TILE_EXACT keeps the latter in exact fp32)
    // soft-max off: exact fp32 throughout, as the header promises (the first GEMM feeds the `== 0 -> -1e10` mask and the
    // "attention" sim / K is unbounded: no fp16 form of the B rows)
    tp.scales = ((desc->variant & ET_VARIANT_TILE_EXACT) || !desc->softmax_enabled) ? nullptr : w.scales;
    if (int e = launch_tile_order(desc, xs, ys, cam, feat_ref, feat_src, w, tp.tiles_per_pair, true, w.scales, true, nullptr, 0, st,
                                  "et_epipolar_forward_tiled(order)"))
        return e;
    const int kpl = (desc->K + 63) / 64;
    const int rows = tile_rows(desc);
    const size_t lds = (size_t)(fwd_tile_array_floats(rows) + rows + kTilePix + 48 + kTilePix * 4) * 4 +
                       (size_t)tp.hw_words * 8 + (kpl == 1 ? (size_t)kTilePix * kWave * 8 : 0);
#define ET_SET_LDS(KERNEL, BYTES) ET_GRANT_LDS(KERNEL, BYTES, dev)
    const bool two_pass = tile_ws_two_pass(desc);
    if (tile_ws_eligible(desc) || two_pass) {
        // 2a. the persistent, warp-specialised kernel (kernels_forward_tile_ws.inc): the default ...
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
        wp.packed_w = nullptr;
        wp.bias = nullptr;
        wp.x = nullptr;
        wp.err = w.err;
        wp.tile_ctr = w.ovf_count + 2;
        wp.setprio = (desc->variant & ET_VARIANT_WS_SETPRIO) ? 1 : 0;
#ifdef ET_WS_PROFILE
        wp.prof = g_ws_prof;
        if (const char *e = getenv("ET_WS_EXPERIMENT")) wp.setprio |= atoi(e);
#else
        wp.prof = nullptr;
#endif
        const int cus = device_cus(dev);
        const unsigned grid = (unsigned)(total < cus ? total : cus);
        if (two_pass) {                 // 64 < K <= 128: two passes of 64 samples per tile, online soft-max (maps up to 128 x 128)
            if (wp.rows_cap > kTileRowsWsLarge) wp.rows_cap = kTileRowsWsLarge;
            const size_t lds_ws = tile_ws_lds_bytes(kTileRowsWsLarge, desc->H, desc->W, true);
            ET_SET_LDS((epipolar_fwd_tile_ws_kernel<kTileRowsWsLarge, 8, false, true, 2>), lds_ws);
            hipLaunchKernelGGL((epipolar_fwd_tile_ws_kernel<kTileRowsWsLarge, 8, false, true, 2>), dim3(grid),
                               dim3((kWsMatrixWaves + 8) * kWave), lds_ws, st, wp);
        } else if (tile_ws_band(desc)) {       // maps above 64 x 64 (up to 96 x 96): 288-row arrays, slot table over the tile's band
            if (wp.rows_cap > kTileRowsWsLarge) wp.rows_cap = kTileRowsWsLarge;
            const size_t lds_ws = tile_ws_lds_bytes(kTileRowsWsLarge, desc->H, desc->W, true);
            ET_SET_LDS((epipolar_fwd_tile_ws_kernel<kTileRowsWsLarge, 8, false, true>), lds_ws);
            hipLaunchKernelGGL((epipolar_fwd_tile_ws_kernel<kTileRowsWsLarge, 8, false, true>), dim3(grid),
                               dim3((kWsMatrixWaves + 8) * kWave), lds_ws, st, wp);
        } else {
            const size_t lds_ws = tile_ws_lds_bytes(kTileRowsSmall, desc->H, desc->W);
            ET_SET_LDS((epipolar_fwd_tile_ws_kernel<kTileRowsSmall, 8>), lds_ws);
            hipLaunchKernelGGL((epipolar_fwd_tile_ws_kernel<kTileRowsSmall, 8>), dim3(grid), dim3((kWsMatrixWaves + 8) * kWave),
                               lds_ws, st, wp);
        }
        if (int e = check_launch("et_epipolar_forward_tiled(ws)")) return e;
        // ... 2b. and the tiles it left over one block per tile
        const unsigned lgrid = (unsigned)(total < 2LL * cus ? total : 2LL * cus);
#define ET_LIST(KK, RR)                                                                                                 \
    do {                                                                                                                \
        ET_SET_LDS((epipolar_fwd_tile_list_kernel<KK, RR>), lds);                                                       \
        hipLaunchKernelGGL((epipolar_fwd_tile_list_kernel<KK, RR>), dim3(lgrid), dim3(256), lds, st, tp);               \
    } while (0)
        if (kpl == 2) {                 // (the two-pass kernel's left-overs: whole tiles, all K samples, one block per tile)
            if (rows == kTileRowsHuge) ET_LIST(2, kTileRowsHuge);
            else if (rows == kTileRowsLarge) ET_LIST(2, kTileRowsLarge);
            else ET_LIST(2, kTileRowsSmall);
        } else if (rows == kTileRowsLarge) {
            ET_LIST(1, kTileRowsLarge);
        } else {
            // (round 6, measured: the left-overs of a map up to 64 x 64 -- all beyond 256 rows -- through the 384-row instance, whole
            //  instead of in pixel groups, take as long: 136 against 131 us for the near-rectified rig's 640 tiles.  A left-over tile
            //  is ~65 us of latency in a block either way and the list is two trips of the resident blocks.)
            ET_LIST(1, kTileRowsSmall);
        }
#undef ET_LIST
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


// The layer's eval-mode forward as ONE data kernel: sampling + attention (as et_epipolar_forward_tiled) with
// x = feat_ref + bias + out . Wf^T -- bn(z(out)) + out + feat with the BN folded into z (epipolar.py:250-253, resnet.py:388;
// what et_residual_gemm computes from `out` in a second pass) -- as a third GEMM of the persistent kernel.
int et_epipolar_forward_fused(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                              const float *cam, const float *feat_ref, const float *feat_src, const void *packed_w,
                              const float *bias, float *x, float *attn, float *corr_pos, float *out_scratch,
                              int32_t want_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int e = validate(desc)) return e;
    if (!xs || !ys || !steps || !cam || !feat_ref || !feat_src || !packed_w || !bias || !x || !out_scratch)
        return fail("et_epipolar_forward_fused: NULL pointer");
    if (reinterpret_cast<uintptr_t>(packed_w) & 15) return fail("et_epipolar_forward_fused: packed weight must be 16-byte aligned");
    if (!tile_eligible(desc) || !tile_ws_eligible(desc))
        return fail("et_epipolar_forward_fused: needs the warp-specialised tile kernel (C == 256, maps up to 96 x 96, K <= 64, "
                    "soft-max on; got C=%d H=%d W=%d K=%d variant=%d): use et_epipolar_forward_tiled + et_residual_gemm",
                    desc->C, desc->H, desc->W, desc->K, desc->variant);
    const size_t need = et_epipolar_forward_workspace_bytes(desc);
    if (!workspace || workspace_bytes < need)
        return fail("et_epipolar_forward_fused: workspace of %zu bytes is smaller than the %zu required",
                    workspace ? workspace_bytes : (size_t)0, need);
    hipStream_t st = (hipStream_t)stream;
    const int HW = desc->H * desc->W;
    TileParams tp;
    FwdParams &p = tp.f;
    p.d = *desc;
    p.xs = xs; p.ys = ys; p.steps = steps; p.cam = cam;
    p.fref = feat_ref; p.fsrc = feat_src;
    p.out = out_scratch; p.attn = attn; p.corr = corr_pos;
    p.res_bias = nullptr; p.res_base = nullptr;
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
    tp.scales = (desc->variant & ET_VARIANT_TILE_EXACT) ? nullptr : w.scales;
    const int dev = current_device();
    if (int e = launch_tile_order(desc, xs, ys, cam, feat_ref, feat_src, w, tp.tiles_per_pair, true, w.scales, true, nullptr, 0, st,
                                  "et_epipolar_forward_fused(order)"))
        return e;
    TileWsParams wp;
    wp.f = p;
    wp.f.out = want_out ? out_scratch : nullptr;      // (the persistent kernel writes `out` on request only)
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
    wp.setprio = 0;
    wp.prof = nullptr;
    wp.packed_w = reinterpret_cast<const unsigned *>(packed_w);
    wp.bias = bias;
    wp.x = x;
    wp.err = w.err;
    wp.tile_ctr = w.ovf_count + 2;
    const int cus = device_cus(dev);
    const unsigned grid = (unsigned)(total < cus ? total : cus);
    if (tile_ws_band(desc)) {
        if (wp.rows_cap > kTileRowsWsLarge) wp.rows_cap = kTileRowsWsLarge;
        const size_t lds_ws = tile_ws_lds_bytes(kTileRowsWsLarge, desc->H, desc->W, true);
        ET_GRANT_LDS((epipolar_fwd_tile_ws_kernel<kTileRowsWsLarge, 8, true, true>), lds_ws, dev);
        hipLaunchKernelGGL((epipolar_fwd_tile_ws_kernel<kTileRowsWsLarge, 8, true, true>), dim3(grid),
                           dim3((kWsMatrixWaves + 8) * kWave), lds_ws, st, wp);
    } else {
        const size_t lds_ws = tile_ws_lds_bytes(kTileRowsSmall, desc->H, desc->W);
        ET_GRANT_LDS((epipolar_fwd_tile_ws_kernel<kTileRowsSmall, 8, true>), lds_ws, dev);
        hipLaunchKernelGGL((epipolar_fwd_tile_ws_kernel<kTileRowsSmall, 8, true>), dim3(grid), dim3((kWsMatrixWaves + 8) * kWave),
                           lds_ws, st, wp);
    }
    if (int e = check_launch("et_epipolar_forward_fused(ws)")) return e;
    // the tiles it left over: `out` rows one block per tile, then their x rows
    const int kpl = 1;
    const int rows = tile_rows(desc);
    const size_t lds = (size_t)(fwd_tile_array_floats(rows) + rows + kTilePix + 48 + kTilePix * 4) * 4 +
                       (size_t)tp.hw_words * 8 + (kpl == 1 ? (size_t)kTilePix * kWave * 8 : 0);
    const unsigned lgrid = (unsigned)(total < 2LL * cus ? total : 2LL * cus);
    if (rows == kTileRowsLarge) {
        ET_GRANT_LDS((epipolar_fwd_tile_list_kernel<1, kTileRowsLarge>), lds, dev);
        hipLaunchKernelGGL((epipolar_fwd_tile_list_kernel<1, kTileRowsLarge>), dim3(lgrid), dim3(256), lds, st, tp);
    } else {
        ET_GRANT_LDS((epipolar_fwd_tile_list_kernel<1, kTileRowsSmall>), lds, dev);
        hipLaunchKernelGGL((epipolar_fwd_tile_list_kernel<1, kTileRowsSmall>), dim3(lgrid), dim3(256), lds, st, tp);
    }
    if (int e = check_launch("et_epipolar_forward_fused(list)")) return e;
    // their x rows: the residual GEMM kernel over the list, two tiles per block and trip (round 6: the plain-fp32 kernel this replaces
    // took 76 us for the near-rectified rig's 640 left-over tiles -- 256 KB of weight fragments per eight pixel rows; now 24 us)
    return et_internal_residual_rows_list(w.perm, w.ovf_list, w.ovf_count, tp.tiles_per_pair, HW, total, out_scratch, feat_ref,
                                          reinterpret_cast<const unsigned *>(packed_w), bias, x, st);
}

}  // extern "C"
