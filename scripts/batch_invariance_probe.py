import sys, torch
sys.path.insert(0, '/root/repo')
from epipolar_transformers_amd import camera, ops, synthetic as syn
dev = 'cuda'
V = 4
P = torch.from_numpy(syn.ring_cameras(V, 64)).float()
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(V, 16, 16, 256, device=dev, generator=g).relu_()
ref_idx = torch.arange(V).repeat(3)
src_idx = torch.cat([torch.arange(V).roll(-s) for s in (1, 2, 3)])
spec = ops.LayerSpec(H=16, W=16, K=16)
cam12 = camera.pair_algebra(P[ref_idx], P[src_idx]).to(dev)
o12, a12, c12 = ops.forward_nhwc(spec, feat[ref_idx].contiguous(), feat[src_idx].contiguous(), cam12)
for s in range(3):
    sl = slice(4 * s, 4 * s + 4)
    cam4 = camera.pair_algebra(P[ref_idx[sl]], P[src_idx[sl]]).to(dev)
    print("shift", s + 1, "cam equal", torch.equal(cam4, cam12[sl]))
    o4, a4, c4 = ops.forward_nhwc(spec, feat[ref_idx[sl]].contiguous(), feat[src_idx[sl]].contiguous(), cam4)
    print("   out max diff %.3e  attn %.3e  corr equal %s" % ((o4 - o12[sl]).abs().max().item(), (a4 - a12[sl]).abs().max().item(), torch.equal(c4, c12[sl])))
    for v in (16384, 65536):
        spv = ops.LayerSpec(H=16, W=16, K=16, variant=v)
        ov, av, cv = ops.forward_nhwc(spv, feat[ref_idx[sl]].contiguous(), feat[src_idx[sl]].contiguous(), cam4)
        print("   variant %d vs ws(4): out %.3e attn %.3e | vs ws(12): out %.3e" % (v, (ov - o4).abs().max().item(), (av - a4).abs().max().item(), (ov - o12[sl]).abs().max().item()))
