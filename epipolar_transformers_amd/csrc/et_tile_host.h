// Host-side helpers shared by the two tile translation units (et_forward_tile.hip, et_backward_tile.hip).
#pragma once
#include <algorithm>
namespace {
// The MFMA tile path applies to the 256-channel head when one reference pixel alone can never
// overflow the tile's row array: a pixel's K samples touch at most 4K source pixels, and a line
// through a W x H map at most 4 per column (or per row, whichever way it runs), i.e. 4 max(W, H).
// one pixel's K samples touch at most 4K source pixels, and a line through a W x H map at most 4 per column (or per
// row, whichever way it runs), i.e. 4 max(W, H)
int tile_rows_per_pixel(const EtLayerDesc *d)
{
    const int longest = d->W > d->H ? d->W : d->H;
    return (d->K < longest) ? 4 * d->K : 4 * longest;
}
// rows per tile the kernel is instantiated with: 256 up to 64 x 64 maps, 384 beyond (longer lines), 512 when a
// single pixel may need more than that
int tile_rows(const EtLayerDesc *d)
{
    if (tile_rows_per_pixel(d) > kTileRowsLarge) return kTileRowsHuge;
    return (d->W > 64 || d->H > 64) ? kTileRowsLarge : kTileRowsSmall;
}
int tile_rows_cap(const EtLayerDesc *d) { return (d->variant & ET_VARIANT_TILE_SPLIT) ? 64 : tile_rows(d); }

bool tile_eligible(const EtLayerDesc *d)
{
    if (d->C != 256 || d->K > 256) return false;
    const long long hw = (long long)d->H * d->W;
    if (hw > 16384) return false;  // bitonic sort of one pair's pixels lives in LDS
    return tile_rows_per_pixel(d) <= tile_rows_cap(d);
}

// the warp-specialised persistent kernels: lanes <-> samples (K <= 64), 256-row arrays.  Soft-max on only: their second
// GEMM holds the B rows (attention x bilinear weights, <= 1 with the soft-max) in fp16 pairs; with
// EPIPOLAR.SOFTMAX_ENABLED False the "attention" is sim / K -- unbounded, -1e10 / K on masked samples -- and the call
// takes the exact-fp32 one-block-per-tile kernel.
// Two instances: 256-row arrays and a whole-map slot table for maps up to 64 x 64; 288-row arrays and a slot table over the
// tile's band for maps up to 96 x 96 (every tile of a 96 x 96 map has at most 280 rows; ET_VARIANT_WS_BAND: also for smaller
// maps, where it must return what the first returns -- the test of the band-table code).
bool tile_ws_band(const EtLayerDesc *d)
{
    const int longest = d->W > d->H ? d->W : d->H;
    if ((d->variant & ET_VARIANT_TILE_CLASSIC) || !d->softmax_enabled || d->K > 64 || d->W < 2 || longest > kWsMaxSideBand)
        return false;
    return tile_rows(d) == kTileRowsLarge || ((d->variant & ET_VARIANT_WS_BAND) && tile_rows(d) == kTileRowsSmall);
}
// 64 < K <= 128 (round 6): the band-table instance in TWO passes of 64 samples per tile with an online soft-max
// (kernels_forward_tile_ws.inc, KH = 2), maps up to 128 x 128 (its column masks hold 128 columns).  Sample + attention only: the
// one-kernel layer keeps K <= 64.
bool tile_ws_two_pass(const EtLayerDesc *d)
{
    const int longest = d->W > d->H ? d->W : d->H;
    return !(d->variant & ET_VARIANT_TILE_CLASSIC) && d->softmax_enabled && d->K > 64 && d->K <= 128 && d->W >= 2 && longest <= kWsMaxSideTwoPass;
}
bool tile_ws_eligible(const EtLayerDesc *d)
{
    return (!(d->variant & ET_VARIANT_TILE_CLASSIC) && d->softmax_enabled && d->K <= 64 && d->W >= 2 &&
            tile_rows(d) == kTileRowsSmall) || tile_ws_band(d);
}
// Workspace of the tile forward (all int32, base aligned up to 256 bytes):
//   header (64 words): [0] overflow count, [1] sticky error word -- at the front, so that their offsets do not depend on
//   the shape of the call (a workspace is reused across shapes) |
//   perm[tiles * 32] | overflow list[tiles] | stats[tiles] | scales[4 * N] (float) |
//   segments[tiles * 32] (float4, 16-byte aligned; in tile order.  Until tile_order_kernel writes them the region of a pair
//   holds the pair's sort keys: 8 bytes per pixel, tile_keys_kernel) |
//   band[tiles] (float4: the tile's base line, warp-specialised kernel) | segments by pixel[N * HW] (float4)
// Header words beyond [0] / [1] are scratch of the kernels of ONE call, cleared by tile_keys_kernel (`header`) at its start:
//   forward:  [2 .. 9]  one tile counter per XCD (the blocks of an XCD draw their tiles from it)
//   backward: [2] four-group tiles met so far, [3] eight-or-more-group tiles met early (kernels_backward_tile.inc: the merged
//             kernel splits the first 128 / 32 of a call in place and defers the rest to its second launch), [4] over-capacity
//             tiles met so far (beyond the first 256 of a call they are deferred without a search)
// The backward REUSES the forward's counter words: safe because every call starts with the ordering kernels on the same stream,
// which zero them -- a change to either user has to keep that.  Which over-capacity tiles a backward call splits in
// place and which it defers depends on the order its blocks reach those counters and on blockIdx relative to gridDim: the
// SET of deferred tiles, and with it the order of the float-atomic additions into grad_src, differs from run to run
// (gradients reproducible to rounding, as documented; ops.backward_deferred_tiles() counts vary by a few tiles).
struct TileWorkspace {
    int *perm, *ovf_count, *err, *ovf_list, *stats;
    float *scales;
    float4 *segs, *band, *segs_pix;
};
constexpr size_t kTileWorkspaceHeaderWords = 64;
size_t tile_workspace_words(size_t tiles, size_t pairs, size_t hw)
{
    return kTileWorkspaceHeaderWords + tiles * kTilePix + 2 * tiles + 4 * pairs + 4 + 4 * tiles * kTilePix + 4 * tiles +
           4 * pairs * hw;
}
TileWorkspace carve_tile_workspace(void *workspace, size_t tiles, size_t pairs, size_t hw)
{
    TileWorkspace w;
    w.ovf_count = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    w.err = w.ovf_count + 1;
    w.perm = w.ovf_count + kTileWorkspaceHeaderWords;
    w.ovf_list = w.perm + tiles * kTilePix;
    w.stats = w.ovf_list + tiles;
    w.scales = reinterpret_cast<float *>(w.stats + tiles);
    w.segs = reinterpret_cast<float4 *>((reinterpret_cast<uintptr_t>(w.scales + 4 * pairs) + 15) & ~(uintptr_t)15);
    w.band = w.segs + tiles * kTilePix;
    w.segs_pix = w.band + tiles;
    return w;
}

// The ordering of a tile call: tile_keys_kernel (segment and sort key of every reference pixel, whole device; clears the
// header words when `header`), then tile_order_kernel (one block per pair: sort -> perm, the segments in tile order and the
// tiles' base lines when `ws_tables`, the scale estimates when `scales`; `clear`: extra blocks that zero a buffer beside the
// sort -- the backward's grad_src).
int launch_tile_order(const EtLayerDesc *desc, const float *xs, const float *ys, const float *cam, const float *feat_ref,
                      const float *feat_src, const TileWorkspace &w, int tiles_per_pair, bool header, float *scales,
                      bool ws_tables, float4 *clear, size_t clear_vec4, hipStream_t st, const char *who)
{
    const int HW = desc->H * desc->W;
    const int perm_stride = tiles_per_pair * kTilePix;
    int n2 = 64;
    while (n2 < HW) n2 <<= 1;
    const size_t lds_sort = order_uses_radix(n2) ? ((tile_order_lds_bytes(n2) + 15) & ~(size_t)15) + order_radix_extra_bytes(n2)
                                                 : tile_order_lds_bytes(n2);
    const int dev = current_device();
    ET_GRANT_LDS(tile_order_kernel, lds_sort, dev);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(w.segs);
    hipLaunchKernelGGL(tile_keys_kernel, dim3((unsigned)((HW + 255) / 256) * desc->N), dim3(256), 0, st, *desc, xs, ys, cam, perm_stride,
                       keys, w.segs_pix, header ? w.ovf_count : (int *)nullptr);
    // (the clearing blocks: >= 8 float4 stores per thread)
    const unsigned clear_blocks = clear ? (unsigned)std::min<size_t>(2048, (clear_vec4 + 8 * 1024 - 1) / (8 * 1024)) : 0u;
    hipLaunchKernelGGL(tile_order_kernel, dim3(desc->N + clear_blocks), dim3(1024), lds_sort, st, *desc, n2, perm_stride, keys,
                       w.segs_pix, w.perm, feat_ref, feat_src, scales, ws_tables ? w.segs : (float4 *)nullptr,
                       ws_tables ? w.band : (float4 *)nullptr, clear, clear_vec4);
    return check_launch(who);
}
}  // namespace
