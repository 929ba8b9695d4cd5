"""CPU simulation (no GPU) of the persistent forward's per-tile 16-row window on the camera rigs of synthetic.rig_pairs:
how many tiles have a tap outside the window (-> overflow list) with the base line of rounds 2-4 (the tile's first pixel's
line) and with the lower-envelope chord of round 5 (tile_order_kernel).  Uses the oracle's sample_locs (test infrastructure).

    python scripts/dev/band_sim_rigs.py [H] [K]
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from epipolar_transformers_amd import synthetic as syn

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, TP, WIN = H, 32, 16


def line(sx, sy, vx, vy, xm):
    if xm:
        b = vy / vx
        return sy - sx * b, b
    b = vx / vy
    return sx - sy * b, b


for rig in syn.RIGS:
    jitter = None if rig == "epipole_border" else (0.05, 8.0)
    P1, P2 = syn.rig_pairs(rig, 1, 4 * H, seed=700 + H + len(rig), jitter=jitter)
    spec = oracle.LayerSpec(H, W, K)
    with np.errstate(all="ignore"):
        locs = oracle.sample_locs(spec, P1, P2)
        E2 = oracle.camera_algebra(P1, P2)[2]
    N = locs.shape[1]
    x = (locs[..., 0] + 1.0) * (W / 2.0) - 0.5
    y = (locs[..., 1] + 1.0) * (H / 2.0) - 0.5
    x0 = np.clip(np.floor(x), -2, W).astype(np.int64)
    y0 = np.clip(np.floor(y), -2, H).astype(np.int64)
    res = {"first": 0, "envelope": 0}
    tiles = 0
    for n in range(N):
        xs, ys = x[:, n].reshape(K, -1), y[:, n].reshape(K, -1)
        X0, Y0 = x0[:, n].reshape(K, -1), y0[:, n].reshape(K, -1)
        sx, sy = xs[0], ys[0]
        vx, vy = xs[-1] - xs[0], ys[-1] - ys[0]
        valid = ((np.abs(vx) + np.abs(vy)) > 0) & (locs[0, n, ..., 0].reshape(-1) > -50)
        th = np.arctan2(vy, vx)
        th = np.where(th < 0, th + np.pi, th)
        th = np.where(th >= np.pi, th - np.pi, th)
        rho = (sy - H / 2) * np.cos(th) - (sx - W / 2) * np.sin(th)
        e2 = E2[n]
        th0 = np.arctan2((H * 4 - 1) / 2 - e2[1], (W * 4 - 1) / 2 - e2[0])
        if not abs(th0) <= 4:
            th0 = 0.0
        tk = th - th0 + np.pi / 2
        tk = tk - np.pi * np.floor(tk / np.pi)
        tb = np.clip((tk * (16384 / np.pi)).astype(np.int64), 0, 16383)
        rq = np.clip(((rho / (0.75 * H) * 0.5 + 0.5) * 65535).astype(np.int64), 0, 65535)
        key = np.where(valid, (tb << 16) | rq, 1 << 40)
        order = np.argsort(key, kind="stable")
        for t0 in range(0, H * W, TP):
            pix = order[t0:t0 + TP]
            tiles += 1
            if not valid[pix[0]]:
                continue
            ax, ay = X0[:, pix], Y0[:, pix]
            anyin = (ax >= -1) & (ax < W) & (ay >= -1) & (ay < H) & valid[pix][None, :]
            if not anyin.any():
                continue
            p0 = pix[0]
            pl = pix[np.nonzero(valid[pix])[0][-1]]
            xm = abs(vx[p0]) >= abs(vy[p0])
            a1, b1 = line(sx[p0], sy[p0], vx[p0], vy[p0], xm)
            with np.errstate(all="ignore"):
                a2, b2 = line(sx[pl], sy[pl], vx[pl], vy[pl], xm)
            u, v = (ax, ay) if xm else (ay, ax)
            ulo, uhi = -1.0, float(W)
            cands = {"first": (a1, b1)}
            if np.isfinite(a2) and np.isfinite(b2) and abs(b2) <= 4:
                m0 = min(a1 + b1 * ulo, a2 + b2 * ulo)
                m1 = min(a1 + b1 * uhi, a2 + b2 * uhi)
                b = (m1 - m0) / (uhi - ulo)
                cands["envelope"] = (m0 - b * ulo, b)
            else:
                cands["envelope"] = (a1, b1)
            for name, (a, b) in cands.items():
                bad = False
                for du in (0, 1):
                    uu = u + du
                    vb = np.floor(np.float32(a) + np.float32(b) * uu.astype(np.float32)).astype(np.int64) - 1
                    dv = v - vb
                    if dv[anyin].min() < 0 or dv[anyin].max() > 14:
                        bad = True
                res[name] += bad
    print("%-18s %dx%d K=%d  tiles %4d   outside the 16-row window:  first pixel's line %4d   lower-envelope chord %4d" %
          (rig, H, W, K, tiles, res["first"], res["envelope"]))
