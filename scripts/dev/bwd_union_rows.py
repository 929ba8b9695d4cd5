"""CPU simulation (oracle's sample_locs: test infrastructure): source rows per tile of the tiled backward for tiles of 32, 64 and
128 pixels in epipolar-line order -- how many (tile, row) incidences, i.e. KB of float atomics, a larger pixel tile would save.

    python scripts/dev/bwd_union_rows.py H RIG PAIRS"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from epipolar_transformers_amd import synthetic as syn
H = int(sys.argv[1]); K = int(sys.argv[4]) if len(sys.argv) > 4 else 64; W = H
rig = sys.argv[2]; npairs = int(sys.argv[3])
if rig == "ring":
    P1, P2 = syn.make_pairs(npairs // 4, 4, H * 4, seed=1000, jitter=(0.05, 8.0))
else:
    P1, P2 = syn.rig_pairs(rig, npairs // 4, 4 * H, seed=1000, jitter=(0.05, 8.0))
spec = oracle.LayerSpec(H, W, K)
tot = {32: [], 64: [], 128: []}
for n0 in range(0, P1.shape[0], 8):
    with np.errstate(all="ignore"):
        locs = oracle.sample_locs(spec, P1[n0:n0+8], P2[n0:n0+8])
        E2 = oracle.camera_algebra(P1[n0:n0+8], P2[n0:n0+8])[2]
    x = (locs[..., 0] + 1.0) * (W / 2.0) - 0.5
    y = (locs[..., 1] + 1.0) * (H / 2.0) - 0.5
    x0 = np.floor(x).astype(np.int64); y0 = np.floor(y).astype(np.int64)
    for n in range(locs.shape[1]):
        xs, ys = x[:, n].reshape(K, -1), y[:, n].reshape(K, -1)
        X0, Y0 = x0[:, n].reshape(K, -1), y0[:, n].reshape(K, -1)
        sx, sy = xs[0], ys[0]
        vx, vy = xs[-1] - xs[0], ys[-1] - ys[0]
        valid = ((np.abs(vx) + np.abs(vy)) > 0) & (locs[0, n, ..., 0].reshape(-1) > -50)
        th = np.arctan2(vy, vx); th = np.where(th < 0, th + np.pi, th); th = np.where(th >= np.pi, th - np.pi, th)
        rho = (sy - H / 2) * np.cos(th) - (sx - W / 2) * np.sin(th)
        e2 = E2[n]
        th0 = np.arctan2((H * 4 - 1) / 2 - e2[1], (W * 4 - 1) / 2 - e2[0])
        if not abs(th0) <= 4: th0 = 0.0
        tk = th - th0 + np.pi / 2; tk = tk - np.pi * np.floor(tk / np.pi)
        tb = np.clip((tk * (16384 / np.pi)).astype(np.int64), 0, 16383)
        rq = np.clip(((rho / (0.75 * H) * 0.5 + 0.5) * 65535).astype(np.int64), 0, 65535)
        key = np.where(valid, (tb << 16) | rq, 1 << 40)
        order = np.argsort(key, kind="stable")
        for TP in tot:
            for t in range(H * W // TP):
                px = order[t*TP:(t+1)*TP]; px = px[valid[px]]
                if len(px) == 0: continue
                rows = set()
                for dx in (0, 1):
                    for dy in (0, 1):
                        xx = X0[:, px] + dx; yy = Y0[:, px] + dy
                        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                        rows.update((yy[ok] * W + xx[ok]).tolist())
                tot[TP].append(len(rows))
for TP, v in tot.items():
    v = np.array(v)
    print("%s %dx%d K=%d  tile %3d px: tiles %6d  rows/tile mean %.1f p50 %d p95 %d max %d   incidences/pair %.0f (x%.2f of map rows)"
          % (rig, H, W, K, TP, len(v), v.mean(), np.median(v), np.percentile(v, 95), v.max(), v.sum() / P1.shape[0], v.sum() / P1.shape[0] / (H * W)))
