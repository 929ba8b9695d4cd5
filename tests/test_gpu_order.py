"""The ordering in front of every tile kernel (tile_keys_kernel + tile_order_kernel, csrc/kernels_forward_tile.inc), read back
from the workspace the forward leaves (layout: csrc/et_tile_host.h / include/epipolar_amd.h):

  * `perm` of every pair is a permutation of the pair's reference pixels, padded with -1 to whole tiles;
  * the segments in tile order are the per-pixel segments gathered through `perm`, bit for bit (that the segments themselves
    are right is what every parity test of the persistent kernel, which samples from them, checks against the oracle);
  * the pixels are sorted: the direction of the epipolar line, measured from the axis of the pair's fan of lines, never
    decreases along `perm` by more than the key's resolution (a broken sorting network scrambles it);
  * pixels without a segment come last.

Shapes: the bench's 64 x 64 (4096 keys), 10 x 10 (100 pixels: a padded last tile, 128-key bitonic sort), 33 x 20 (non-square, 1024
keys), 96 x 96 (16384-key bitonic sort: the full three-stage rounds); 256 .. 4096 keys take the radix sort (round 6) -- one shape per
width of its counter scan, with and without padding keys.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C = 256


def _regions(ws, n, h, w):
    """perm (n, T*32) int32 | segs in tile order (n, T*32, 4) | base lines (n*T, 4) | segs by pixel (n, h*w, 4) -- views"""
    hw = h * w
    tiles = n * ((hw + 31) // 32)
    base = (-ws.data_ptr()) % 256
    words = ws[base:base + (ws.numel() - base) // 4 * 4].view(torch.int32)
    o = 64
    perm = words[o:o + tiles * 32].view(n, -1)
    o += tiles * 32 + 2 * tiles + 4 * n                       # perm | overflow list | stats | scales
    byte = base + o * 4
    byte += (-(ws.data_ptr() + byte)) % 16                    # float4 arrays are 16-byte aligned
    o = (byte - base) // 4
    fl = words.view(torch.float32)
    segs = fl[o:o + 4 * tiles * 32].view(n, -1, 4)
    o += 4 * tiles * 32
    band = fl[o:o + 4 * tiles].view(tiles, 4)
    o += 4 * tiles
    segs_pix = fl[o:o + 4 * n * hw].view(n, hw, 4)
    return perm, segs, band, segs_pix


@pytest.mark.parametrize("n,h,w,k", [(5, 64, 64, 64), (3, 10, 10, 16), (4, 33, 20, 20), (2, 96, 96, 64),
                                     # (round 6: 256 .. 4096 keys take the radix sort -- every counter-scan width of it)
                                     (3, 16, 16, 16), (3, 15, 15, 16), (3, 20, 20, 16), (2, 40, 40, 32), (2, 48, 64, 48)],
                         ids=["64x64", "10x10-padded-tile", "33x20", "96x96", "16x16-256-keys", "15x15-padded-256", "20x20-512-keys",
                              "40x40-2048-keys", "48x64-padded-4096"])
def test_tile_order_is_a_sorted_permutation_with_gathered_segments(n, h, w, k):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from epipolar_transformers_amd import camera, ops, synthetic as syn

    P1, P2 = syn.make_pairs((n + 3) // 4, 4, 4 * max(h, w), seed=7 + h, jitter=(0.05, 8.0))
    P1, P2 = P1[:n], P2[:n]
    g = torch.Generator().manual_seed(h)
    f1 = torch.randn(n, h, w, C, generator=g).relu_().cuda()
    f2 = torch.randn(n, h, w, C, generator=g).relu_().cuda()
    cam = camera.pair_algebra(P1, P2)
    spec = ops.LayerSpec(H=h, W=w, K=k)
    ws = ops.tile_workspace(spec, n, C, f1.device)
    assert ws.numel() > 0, "the shape was meant to take the tile path"
    ops.forward_nhwc(spec, f1, f2, cam.cuda(), workspace=ws)
    torch.cuda.synchronize()
    perm, segs, band, segs_pix = (t.cpu() for t in _regions(ws, n, h, w))
    hw = h * w
    # permutation + padding
    for i in range(n):
        p = perm[i]
        assert torch.equal(torch.sort(p[:hw]).values, torch.arange(hw, dtype=torch.int32)), "pair %d: not a permutation" % i
        assert (p[hw:] == -1).all()
    # gathered segments, bit for bit; zeros behind the padding
    idx = perm[:, :hw].long()
    want = torch.gather(segs_pix, 1, idx[..., None].expand(-1, -1, 4))
    assert torch.equal(segs[:, :hw].view(torch.int32), want.view(torch.int32))
    assert (segs[:, hw:] == 0).all()
    # sortedness: the line's direction from the fan's axis (modulo pi), pixels without a segment last
    cx = 0.5 * (float(spec.xs[0]) + float(spec.xs[-1]))
    cy = 0.5 * (float(spec.ys[0]) + float(spec.ys[-1]))
    for i in range(n):
        s = want[i].double().numpy()
        valid = (s[:, 2] != 0) | (s[:, 3] != 0)
        nv = int(valid.sum())
        assert valid[:nv].all() and not valid[nv:].any(), "pair %d: pixels without a segment are not at the end" % i
        e2 = cam[i, 24:26].double().numpy()
        th0 = math.atan2(cy - e2[1], cx - e2[0])
        th = np.arctan2(s[:nv, 3], s[:nv, 2])
        tk = np.mod(th - th0 + 0.5 * math.pi, math.pi)
        step = np.diff(tk)
        # (one 14-bit bin of the key + float32 rounding of atan2f; a value that wraps at pi is a step of ~ -pi: none with
        #  the fan's axis at pi / 2)
        assert step.min() >= -2.5 * math.pi / 16384, "pair %d: the order decreases by %g rad" % (i, -step.min())
    # the base line of a tile is that of its first pixel: a finite line wherever the tile has one
    first = perm[:, ::32].reshape(-1)
    has = first >= 0
    assert torch.isfinite(band).all() and (band[has, 1].abs() <= 1.0 + 1e-6).all()

