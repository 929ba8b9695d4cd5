"""CPU simulation (no GPU; uses the oracle's sample_locs -- test infrastructure) of the tiled backward's over-capacity tiles:
which 32-pixel tiles (pixels in epipolar-line order, as tile_order_kernel sorts them) touch more source rows than the merged
kernel's 192 (288) columns, and into how many pixel groups they must be split before every group fits.

    python scripts/dev/bwd_hard_tiles.py H RIG PAIRS        e.g.  64 h36m_room 128   |   96 ring 64

Round 5 finding (profiles/r05_bwd_rigs.txt): on the room rig the hard tiles have only 194-232 rows but need 16-32 groups -- their
32 lines nearly coincide (theta spread 0.002 rad) and run diagonally through the whole map, so ONE pixel alone touches ~190
rows and splitting by pixels removes nothing; the 256-column one-array kernel takes them whole.  That is why they are deferred
rather than split in place."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from epipolar_transformers_amd import synthetic as syn
H = int(sys.argv[1]); K = 64; W = H; TP = 32
rig = sys.argv[2]; npairs = int(sys.argv[3])
if rig == "ring":
    P1, P2 = syn.make_pairs(npairs // 4, 4, H * 4, seed=1000, jitter=(0.05, 8.0))
else:
    P1, P2 = syn.rig_pairs(rig, npairs // 4, 4 * H, seed=1000, jitter=(0.05, 8.0))
spec = oracle.LayerSpec(H, W, K)
cap = 192 if H <= 64 else 288
hard = []
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
        ntile = H * W // TP
        for t in range(ntile):
            px = order[t*TP:(t+1)*TP]; px = px[valid[px]]
            if len(px) == 0: continue
            rows = set()
            for dx in (0, 1):
                for dy in (0, 1):
                    xx = X0[:, px] + dx; yy = Y0[:, px] + dy
                    ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                    rows.update((yy[ok] * W + xx[ok]).tolist())
            if len(rows) > cap:
                # groups needed: split pixels until all groups fit
                def nrows(sel):
                    r = set()
                    for dx in (0, 1):
                        for dy in (0, 1):
                            xx = X0[:, sel] + dx; yy = Y0[:, sel] + dy
                            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                            r.update((yy[ok] * W + xx[ok]).tolist())
                    return len(r)
                pxf = order[t*TP:(t+1)*TP]
                g = 2
                while g < 32:
                    gs = 32 // g
                    if all(nrows(pxf[i*gs:(i+1)*gs][valid[pxf[i*gs:(i+1)*gs]]]) <= cap for i in range(g)): break
                    g *= 2
                hard.append((n0 + n, t, len(rows), g, float(tk[px].min()), float(tk[px].max())))
print(rig, H, "pairs", P1.shape[0], "tiles/pair", H*W//TP, "over capacity:", len(hard))
from collections import Counter
print("groups:", sorted(Counter(h[3] for h in hard).items()))
for h in hard:
    if h[3] >= 4: print("  pair %3d tile %3d rows %4d groups %2d  theta %.3f..%.3f" % h)
