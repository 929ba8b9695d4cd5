"""The reference's own OP SEQUENCE for the hot loop, written out in PyTorch (TEST / BASELINE INFRASTRUCTURE ONLY --
never imported by the product package; see oracle/__init__.py).

`oracle/epipolar_oracle.c` restates the algorithm as scalar C; this file restates what the reference actually
EXECUTES on a CPU, op for op (modeling/layers/epipolar.py:188-247 and epipolar_similarity :272-321), so that
`bench.py`'s cpu_baseline can time "the reference CPU path" on the GPU box, where /root/reference does not exist:

    per pair i (the Python loop at :188):
        other1_sampled = F.grid_sample(other1[:, i], sample_locs[:, i])      # (K,C,H,W), stride-0 expanded map   :199
        other2_sampled = F.grid_sample(other2[:, i], sample_locs[:, i])      # sampled AGAIN (`other1 is other2`
                                                                             # is never true)       :210
        sim = (feat1[i] * other1_sampled).sum(1)                             # broadcast mul + sum over C          :294-295
        sim[sim == 0] = -1e10                                                #                                     :298
        sim = softmax(sim * SOFTMAXSCALE, dim 0)   |   sim / K               #                                     :303-311
        idx = sim.argmax(0); corr_pos = de_normalize(gather(sample_locs, idx))                                    :237-242
        out_i = (other2_sampled * sim.view(-1, 1, H, W)).sum(0)              #                                     :243

The sample locations come from the geometry restatement (oracle.sample_locs, bit-equal to the reference's
grid2sample_locs on the golden fixtures); they are <1 % of the reference's time.
tests/test_oracle_golden.py checks this file against the fixtures the real reference produced.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def de_normalize(pos: torch.Tensor, H: int, W: int, correct: bool) -> torch.Tensor:
    """vision/multiview.py:39-57."""
    x, y = pos[..., 0], pos[..., 1]
    if correct:
        return torch.stack([(x + 1) * (W - 1) / 2, (y + 1) * (H - 1) / 2], -1)
    return torch.stack([(x + 1) * W / 2 - 0.5, (y + 1) * H / 2 - 0.5], -1)


def forward(feat1: torch.Tensor, feat2: torch.Tensor, sample_locs: torch.Tensor, softmax_scale: float = 0.125,
            softmax_enabled: bool = True, correct_normalize: bool = True, align_corners: bool = False):
    """feat1, feat2: (N,C,H,W) float32 CPU; sample_locs: (K,N,H,W,2) normalised.  Returns out (N,C,H,W),
    attn (N,K,H,W), corr_pos (N,H,W,2)."""
    N, C, H, W = feat1.shape
    K = sample_locs.shape[0]
    other1 = feat2.view(1, N, C, H, W).expand(K, -1, -1, -1, -1)      # epipolar.py:166-171 (stride-0 expand)
    other2 = feat2.view(1, N, C, H, W).expand(K, -1, -1, -1, -1)
    out, depth, corr = [], [], []
    for i in range(N):                                                # epipolar.py:188
        s1 = F.grid_sample(other1[:, i], sample_locs[:, i], mode="bilinear", padding_mode="zeros",
                           align_corners=align_corners)
        s2 = F.grid_sample(other2[:, i], sample_locs[:, i], mode="bilinear", padding_mode="zeros",
                           align_corners=align_corners)
        sim = (feat1[i] * s1).sum(1)                                   # (K,H,W)
        sim[sim == 0] = -1e10
        if softmax_enabled:
            sim = F.softmax(sim * softmax_scale, 0)
        else:
            sim = sim / K
        idx = sim.argmax(0)
        pos = torch.gather(sample_locs[:, i], 0, idx.view(1, H, W, 1).expand(-1, -1, -1, 2)).squeeze(0)
        corr.append(de_normalize(pos, H, W, correct_normalize))
        out.append((s2 * sim.view(-1, 1, H, W)).sum(0))
        depth.append(sim)
    return torch.stack(out), torch.stack(depth), torch.stack(corr)


def forward_timed(feat1: np.ndarray, feat2: np.ndarray, sample_locs: np.ndarray, threads: int, **kw):
    """Wall time of one `forward` over the given pairs with `threads` intra-op threads (first call of a process is
    ~2.5x slower: warm up before timing).  Returns (seconds, out)."""
    import time

    torch.set_num_threads(threads)
    f1, f2, sl = torch.from_numpy(feat1), torch.from_numpy(feat2), torch.from_numpy(sample_locs)
    with torch.no_grad():
        t0 = time.perf_counter()
        out, _, _ = forward(f1, f2, sl, **kw)
        dt = time.perf_counter() - t0
    return dt, out
