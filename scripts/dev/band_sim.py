"""CPU simulation of the tile row sets of the warp-specialised forward (no GPU): how the column-mask
construction (one small bit mask per column of the tile's major axis, relative to an analytic base
line) compares with the exact union of taps.  Uses the oracle's sample_locs (test infrastructure).

    python scripts/dev/band_sim.py [H] [K] [frames] [views]
"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import oracle
from epipolar_transformers_amd import synthetic as syn

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2
V = int(sys.argv[4]) if len(sys.argv) > 4 else 4
W = H
P1, P2 = syn.make_pairs(frames, V, H * 4, seed=1000, jitter=(0.05, 8.0))
spec = oracle.LayerSpec(H, W, K)
locs = oracle.sample_locs(spec, P1, P2)          # (K,N,H,W,2) normalised
N = locs.shape[1]
x = (locs[..., 0] + 1.0) * (W / 2.0) - 0.5       # align_corners False
y = (locs[..., 1] + 1.0) * (H / 2.0) - 0.5
x0 = np.clip(np.floor(x), -2, W).astype(np.int64)
y0 = np.clip(np.floor(y), -2, H).astype(np.int64)
TP = 32
WIN = 16
import os
FAN = os.environ.get("FAN", "1") == "1"
E2 = oracle.camera_algebra(P1, P2)[2]
margin = int(sys.argv[5]) if len(sys.argv) > 5 else 2
stats = dict(U=[], need=[], ovf=0, tiles=0, lo=[], hi=[])
for n in range(N):
    xs, ys = x[:, n].reshape(K, -1), y[:, n].reshape(K, -1)
    X0, Y0 = x0[:, n].reshape(K, -1), y0[:, n].reshape(K, -1)
    sx, sy = xs[0], ys[0]
    vx, vy = xs[-1] - xs[0], ys[-1] - ys[0]
    valid = (np.abs(vx) + np.abs(vy)) > 0
    th = np.arctan2(vy, vx)
    th = np.where(th < 0, th + np.pi, th)
    th = np.where(th >= np.pi, th - np.pi, th)
    rho = (sy - H / 2) * np.cos(th) - (sx - W / 2) * np.sin(th)
    if FAN:      # angles measured from the axis of the fan (direction epipole -> image centre), as tile_order_kernel does
        e2 = E2[n]
        # tap-space epipole: image coords -> tap coords is x_tap = (x_img + 0.5 - 2) / 4 * (W / (W - 1)) ... use the affine map of the samples
        th0 = np.arctan2((H * 4 - 1) / 2 - e2[1], (W * 4 - 1) / 2 - e2[0])
        tk = th - th0 + np.pi / 2
        tk = tk - np.pi * np.floor(tk / np.pi)
    else:
        tk = th
    tb = np.clip((tk * (16384 / np.pi)).astype(np.int64), 0, 16383)
    rq = np.clip(((rho / (0.75 * H) * 0.5 + 0.5) * 65535).astype(np.int64), 0, 65535)
    key = np.where(valid, (tb << 16) | rq, 1 << 40)
    order = np.argsort(key, kind="stable")
    for t0 in range(0, H * W, TP):
        pix = order[t0:t0 + TP]
        ax, ay = X0[:, pix], Y0[:, pix]       # (K, 32)
        anyin = (ax >= -1) & (ax < W) & (ay >= -1) & (ay < H) & valid[pix][None, :]
        taps = set()
        for dx in (0, 1):
            for dy in (0, 1):
                xx, yy = ax + dx, ay + dy
                ok = anyin & (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                taps.update((yy[ok] * W + xx[ok]).tolist())
        stats["tiles"] += 1
        stats["U"].append(len(taps))
        if not anyin.any():
            continue
        # base line: the tile's first pixel with a segment
        p0 = pix[np.argmax(valid[pix])]
        xmaj = abs(vx[p0]) >= abs(vy[p0])
        if xmaj:
            a = sy[p0] - sx[p0] * (vy[p0] / vx[p0]); b = vy[p0] / vx[p0]
            u, v = ax, ay
        else:
            a = sx[p0] - sy[p0] * (vx[p0] / vy[p0]); b = vx[p0] / vy[p0]
            u, v = ay, ax
        lo, hi = 99, -99
        for du in (0, 1):
            uu = u + du
            vb = np.floor(np.float32(a) + np.float32(b) * uu.astype(np.float32)).astype(np.int64) - margin
            dv = v - vb                        # rows dv, dv+1
            lo = min(lo, dv[anyin].min()); hi = max(hi, dv[anyin].max() + 1)
        stats["lo"].append(lo); stats["hi"].append(hi)
        if lo < 0 or hi > WIN - 1:
            stats["ovf"] += 1
U = np.array(stats["U"])
print("%dx%d K=%d pairs %d tiles %d: exact U mean %.1f p50 %d p90 %d max %d" % (H, W, K, N, stats["tiles"], U.mean(), np.percentile(U, 50), np.percentile(U, 90), U.max()))
lo, hi = np.array(stats["lo"]), np.array(stats["hi"])
print("dv range (margin %d): lo min %d  hi max %d ; tiles outside a WIN-bit window: %d (%.2f %%)" % (margin, lo.min(), hi.max(), stats["ovf"], 100.0 * stats["ovf"] / stats["tiles"]))
print("hist lo:", np.bincount(np.clip(lo, -5, 10) + 5), " hist hi:", np.bincount(np.clip(hi, 0, 20)))
bad = np.where((lo < 0) | (hi > WIN - 1))[0]
print("outliers (index among non-empty tiles): lo/hi", [(int(lo[i]), int(hi[i])) for i in bad])
# ---- what are the tiles far outside the window?
if len(sys.argv) > 6:
    n_dbg = 0
    for n in range(N):
        xs, ys = x[:, n].reshape(K, -1), y[:, n].reshape(K, -1)
        vx, vy = xs[-1] - xs[0], ys[-1] - ys[0]
        valid = (np.abs(vx) + np.abs(vy)) > 0
        th = np.arctan2(vy, vx); th = np.where(th < 0, th + np.pi, th); th = np.where(th >= np.pi, th - np.pi, th)
        tb = np.clip((th * (16384 / np.pi)).astype(np.int64), 0, 16383)
        rho = (ys[0] - H / 2) * np.cos(th) - (xs[0] - W / 2) * np.sin(th)
        rq = np.clip(((rho / (0.75 * H) * 0.5 + 0.5) * 65535).astype(np.int64), 0, 65535)
        key = np.where(valid, (tb << 16) | rq, 1 << 40)
        order = np.argsort(key, kind="stable")
        print("pair", n, "valid", valid.sum(), "th range %.3f..%.3f" % (th[valid].min(), th[valid].max()))
        ths = th[order][valid[order]]
        jumps = np.where(np.abs(np.diff(ths)) > 0.05)[0]
        print("  angle jumps at sorted positions", jumps.tolist(), ths[jumps].round(3).tolist(), ths[jumps + 1].round(3).tolist())
        # segment lengths (tap space) of the sorted pixels around the suspicious tiles
        ln = np.hypot(vx, vy)[order]
        print("  shortest segments:", np.sort(ln[valid[order]])[:5].round(2).tolist(), " count < 8 px:", int((ln[valid[order]] < 8).sum()))
