// libepipolar_amd.so: the MFMA tile formulation of the backward (et_epipolar_backward_tiled).
#include <algorithm>
#include "et_common.h"

namespace {
#include "kernels_forward_tile.inc"   // tile_order_kernel and the tile helpers shared with the forward
#include "kernels_backward_tile.inc"  // epipolar_bwd_tile_kernel
}  // namespace
#include "et_tile_host.h"

extern "C" {

size_t et_epipolar_backward_tiled_workspace_bytes(const EtLayerDesc *desc)
{
    if (validate(desc) || !tile_eligible(desc)) return 0;
    return et_epipolar_forward_workspace_bytes(desc);
}

int et_epipolar_backward_tiled(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                               const float *cam, const float *feat_ref, const float *feat_src,
                               const float *grad_out, float *grad_ref, float *grad_src, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    return et_epipolar_backward_tiled_attn(desc, xs, ys, steps, cam, feat_ref, feat_src, nullptr, grad_out, grad_ref,
                                           grad_src, workspace, workspace_bytes, stream);
}

int et_epipolar_backward_tiled_attn(const EtLayerDesc *desc, const float *xs, const float *ys, const float *steps,
                                    const float *cam, const float *feat_ref, const float *feat_src, const float *attn,
                                    const float *grad_out, float *grad_ref, float *grad_src, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    if (int e = validate(desc)) return e;
    if (!xs || !ys || !steps || !cam || !feat_ref || !feat_src || !grad_out || !grad_ref || !grad_src)
        return fail("et_epipolar_backward_tiled: NULL pointer");
    const size_t need = et_epipolar_backward_tiled_workspace_bytes(desc);
    if (need == 0)
        return fail("et_epipolar_backward_tiled: needs C == 256, H*W <= 16384 and 4 min(K, max(W,H)) <= %d "
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
    tp.attn = attn;
    // (with the per-pair scale estimates of the source maps: the merged kernels run their row-type GEMMs as split-fp16
    //  products; the workspace has the forward's layout.  grad_src is cleared by extra blocks of the ordering kernel: the
    //  tile kernel adds into it)
    const TileWorkspace w = carve_tile_workspace(workspace, (size_t)total, (size_t)desc->N, (size_t)HW);
    tp.perm = w.perm;
    tp.scales = w.scales;
    const size_t clear_vec4 = (size_t)desc->N * HW * (desc->C / 4);     // (C == 256)
    // (header = true: the ordering clears the workspace's overflow counter, which the merged launch below counts into)
    if (int e = launch_tile_order(desc, xs, ys, cam, feat_ref, feat_src, w, tp.tiles_per_pair, true, w.scales, false,
                                  reinterpret_cast<float4 *>(grad_src), clear_vec4, st, "et_epipolar_backward_tiled(order)"))
        return e;
    const int dev = current_device();
    const int kpl = (desc->K + 63) / 64;
    // 64 x 64 maps, K <= 64: the merged form (two 192-column arrays, one round of atomics per tile) unless the caller
    // asks for the one-array kernel (ET_VARIANT_TILE_CLASSIC)
    const bool merged = (tile_rows(desc) == kTileRowsSmall || tile_rows(desc) == kTileRowsLarge) && kpl == 1 &&
                        !(desc->variant & ET_VARIANT_TILE_CLASSIC);
    const int rows = !merged ? tile_rows(desc) : tile_rows(desc) == kTileRowsSmall ? kTileRowsMerged : kTileRowsMergedLarge;
    if (merged && tp.rows_cap > rows) tp.rows_cap = rows;
    auto lds_of = [&](int r, bool m) {
        return (size_t)(bwd_tile_array_floats(r) + r + kTilePix + 60 + kTilePix * 4) * 4 + (size_t)tp.hw_words * 8 +
               ((kpl == 1 && !m) ? (size_t)kTilePix * kWave * 8 : 0);
    };
    size_t lds = lds_of(rows, merged);
    // merged launches defer the tiles beyond their columns to a second launch of the one-array kernel (unless the caller tests the
    // splitting: ET_VARIANT_TILE_SPLIT)
    const bool defer = merged && !(desc->variant & (ET_VARIANT_TILE_SPLIT | ET_VARIANT_BWD_SPLIT_IN_PLACE));
    if (defer) {
        tp.ovf_count = w.ovf_count;
        tp.ovf_list = w.ovf_list;
    }
#define ET_BTILE(KK, RR)                                                                                        \
    do {                                                                                                        \
        ET_GRANT_LDS((epipolar_bwd_tile_kernel<KK, RR>), lds, dev);                                             \
        hipLaunchKernelGGL((epipolar_bwd_tile_kernel<KK, RR>), dim3((unsigned)total), dim3(256), lds, st, tp);  \
    } while (0)
    if (rows == kTileRowsMerged || rows == kTileRowsMergedLarge) {
        if (rows == kTileRowsMerged) ET_BTILE(1, kTileRowsMerged);
        else ET_BTILE(1, kTileRowsMergedLarge);
        if (defer) {
            if (int e = check_launch("et_epipolar_backward_tiled(merged)")) return e;
            tp.tile_list = w.ovf_list;
            tp.tile_count = w.ovf_count;
            tp.ovf_list = nullptr;
            tp.ovf_count = nullptr;
            tp.rows_cap = tile_rows_cap(desc);
            lds = lds_of(tile_rows(desc), false);
            // (two blocks per compute unit -- what is resident at once -- walk the list)
            const int cus = device_cus(dev);
            const unsigned lgrid = (unsigned)(total < 2LL * cus ? total : 2LL * cus);
            if (tile_rows(desc) == kTileRowsSmall && ET_BWD_LIST_MERGED) {
                // round 6: the deferred tiles of a 64 x 64 map (193 .. ~280 rows) by the MERGED kernel of 288 columns -- one
                // derivation of the samples' slots for both arrays, no half passes (a tile beyond 288 rows is split there)
                tp.rows_cap = kTileRowsMergedLarge;
                lds = lds_of(kTileRowsMergedLarge, true);
                ET_GRANT_LDS((epipolar_bwd_tile_list_kernel<1, kTileRowsMergedLarge>), lds, dev);
                hipLaunchKernelGGL((epipolar_bwd_tile_list_kernel<1, kTileRowsMergedLarge>), dim3(lgrid), dim3(256), lds, st, tp);
            } else if (tile_rows(desc) == kTileRowsSmall) {
                ET_GRANT_LDS((epipolar_bwd_tile_list_kernel<1, kTileRowsSmall>), lds, dev);
                hipLaunchKernelGGL((epipolar_bwd_tile_list_kernel<1, kTileRowsSmall>), dim3(lgrid), dim3(256), lds, st, tp);
            } else {
                ET_GRANT_LDS((epipolar_bwd_tile_list_kernel<1, kTileRowsLarge>), lds, dev);
                hipLaunchKernelGGL((epipolar_bwd_tile_list_kernel<1, kTileRowsLarge>), dim3(lgrid), dim3(256), lds, st, tp);
            }
        }
    } else if (rows == kTileRowsSmall) {
        if (kpl == 1) ET_BTILE(1, kTileRowsSmall);
        else if (kpl == 2) ET_BTILE(2, kTileRowsSmall);
        else ET_BTILE(4, kTileRowsSmall);
    } else if (rows == kTileRowsLarge) {
        if (kpl == 1) ET_BTILE(1, kTileRowsLarge);
        else if (kpl == 2) ET_BTILE(2, kTileRowsLarge);
        else ET_BTILE(4, kTileRowsLarge);
    } else {
        if (kpl == 2) ET_BTILE(2, kTileRowsHuge);   // (512 rows per pixel need K > 96)
        else ET_BTILE(4, kTileRowsHuge);
    }
#undef ET_BTILE
    return check_launch("et_epipolar_backward_tiled");
}

}  // extern "C"
