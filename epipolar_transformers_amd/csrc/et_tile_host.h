// Host-side helpers shared by the two tile translation units (et_forward_tile.hip, et_backward_tile.hip).
#pragma once
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
bool tile_ws_eligible(const EtLayerDesc *d)
{
    return !(d->variant & ET_VARIANT_TILE_CLASSIC) && d->softmax_enabled && d->K <= 64 && d->W >= 2 &&
           tile_rows(d) == kTileRowsSmall;
}
// second generation (pre-split source planes with exact per-row scales, row masks: kernels_forward_tile_ws2.inc), on
// request (ET_VARIANT_WS_V2): measured slower than the first on MI355X (profiles/r03_ws2_*), kept as the variant whose
// arithmetic needs no scale estimate at all
bool tile_ws2_eligible(const EtLayerDesc *d)
{
    return tile_ws_eligible(d) && (d->variant & ET_VARIANT_WS_V2) && d->W <= 64 && d->H <= 64;
}

// Workspace of the tile forward (all int32, base aligned up to 256 bytes):
//   header (64 words): [0] overflow count, [1] sticky error word -- at the front, so that their offsets do not depend on
//   the shape of the call (a workspace is reused across shapes) |
//   perm[tiles * 32] | overflow list[tiles] | stats[tiles] | scales[4 * N] (float) |
//   segments[tiles * 32] (float4, 16-byte aligned) | band[tiles] (float4: the tile's base line, warp-specialised kernel) |
//   -- warp-specialised kernel, second generation only: --
//   rowinv[N * HW] (float) | planes[N * HW * 256] (dwords, 256-byte aligned)
struct TileWorkspace {
    int *perm, *ovf_count, *err, *ovf_list, *stats;
    float *scales;
    float4 *segs, *band;
    float *rowinv;
    unsigned *planes;
};
constexpr size_t kTileWorkspaceHeaderWords = 64;
size_t tile_workspace_words(size_t tiles, size_t pairs)
{
    return kTileWorkspaceHeaderWords + tiles * kTilePix + 2 * tiles + 4 * pairs + 4 + 4 * tiles * kTilePix + 4 * tiles;
}
size_t tile_workspace_plane_words(size_t pairs, size_t hw) { return pairs * hw + 64 + pairs * hw * 256; }
TileWorkspace carve_tile_workspace(void *workspace, size_t tiles, size_t pairs, size_t hw = 0)
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
    w.rowinv = reinterpret_cast<float *>(w.band + tiles);
    w.planes = reinterpret_cast<unsigned *>((reinterpret_cast<uintptr_t>(w.rowinv + pairs * hw) + 255) & ~(uintptr_t)255);
    return w;
}
}  // namespace
