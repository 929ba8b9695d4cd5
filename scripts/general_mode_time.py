"""Development: the parameterised + pooled branch (configs/epipolar/keypoint_h36m_param.yaml's head: theta / phi / g,
BOTTLENECK 2, POOLING, 64 x 64, K = 64, C = 256) through the HIP general kernel and through the chunked torch restatement.
usage: python scripts/general_mode_time.py [pairs]"""
import sys, time, torch
sys.path.insert(0, ".")
from epipolar_transformers_amd import default_cfg, synthetic as syn
from epipolar_transformers_amd.epipolar import Epipolar

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, C, K = 64, 256, 64
cfg = default_cfg()
cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (H, H), "KEYPOINT.NFEATS", C, "EPIPOLAR.SAMPLESIZE", K, "DATASETS.IMAGE_SIZE", (4 * H, 4 * H),
                     "EPIPOLAR.ATTENTION", "avg", "EPIPOLAR.PARAMETERIZED", ("z", "theta", "phi", "g"), "EPIPOLAR.BOTTLENECK", 2,
                     "EPIPOLAR.ZRESIDUAL", False, "EPIPOLAR.POOLING", True])
mod = Epipolar(cfg=cfg).cuda().eval()
P1, P2 = syn.make_pairs(N // 4, 4, 4 * H, seed=3, jitter=(0.05, 8.0))
f1, f2 = syn.make_features(N, C, H, H, seed=5)
f1, f2 = f1.cuda(), f2.cuda()
def timed(fn, reps):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
with torch.no_grad():
    torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
    t_hip = timed(lambda: mod._attend_general_hip(f1, f2, P1, P2), 5)
    m_hip = torch.cuda.max_memory_allocated() - base
    torch.cuda.reset_peak_memory_stats()
    t_torch = timed(lambda: mod._attend_general_chunk(f1[:8], f2[:8], P1[:8], P2[:8]), 2) * (N / 8)
    m_torch = torch.cuda.max_memory_allocated() - base
    a = mod._attend_general_hip(f1[:8], f2[:8], P1[:8], P2[:8]); b = mod._attend_general_chunk(f1[:8], f2[:8], P1[:8], P2[:8])
print("param + POOLING head, %d pairs: HIP general kernel (+ the three 1x1 convolutions) %.2f ms, peak extra memory %.2f GB; "
      "torch restatement %.1f ms (8 pairs at a time, scaled), peak %.2f GB; max |out diff| %.2e, max |attn diff| %.2e"
      % (N, t_hip, m_hip / 2**30, t_torch, m_torch / 2**30, (a[0] - b[0]).abs().max().item(), (a[1] - b[1]).abs().max().item()))
