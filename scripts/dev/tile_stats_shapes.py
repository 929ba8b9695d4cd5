import sys, torch
sys.path.insert(0, ".")
from epipolar_transformers_amd import camera, ops, synthetic as syn
dev = torch.device("cuda:0")
for H, K, frames, V in ((64, 64, 8, 4), (96, 64, 8, 4), (128, 128, 4, 8)):
    P1, P2 = syn.make_pairs(frames, V, H * 4, seed=1000, jitter=(0.05, 8.0))
    N = P1.shape[0]
    g = torch.Generator(device=dev).manual_seed(0)
    ref = torch.randn(N, H, H, 256, device=dev, generator=g).relu_(); src = torch.randn(N, H, H, 256, device=dev, generator=g).relu_()
    cam = camera.pair_algebra(P1, P2).to(dev)
    spec = ops.LayerSpec(H=H, W=H, K=K, variant=65536)
    ws = ops.tile_workspace(spec, N, 256, dev)
    ops.forward_nhwc(spec, ref, src, cam, workspace=ws)
    st = ops.tile_stats(spec, N, 256, ws)
    U = (st & 0xFFFF).float(); grp = (st >> 16)
    q = torch.quantile(U, torch.tensor([0.5, 0.9, 0.99], device=U.device))
    print("%dx%d K=%d: tiles %d  U mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f ; tiles split into groups: %.2f %% (max groups %d); 32-row blocks per tile mean %.2f" %
          (H, H, K, U.numel(), U.mean().item(), q[0].item(), q[1].item(), q[2].item(), U.max().item(), (grp > 1).float().mean().item() * 100, int(grp.max()), ((U + 31) // 32).mean().item()))
