import sys, time, torch
sys.path.insert(0, '/root/repo')
from epipolar_transformers_amd import default_cfg, backbones, synthetic as syn
from epipolar_transformers_amd.model import MultiViewPoseModel, ring_sources
dev = torch.device('cuda:0')
cfg = default_cfg()
cfg.merge_from_list(["BACKBONE.BODY", "epipolarposeR-50", "BACKBONE.PRETRAINED", False, "KEYPOINT.HEATMAP_SIZE", (64, 64), "KEYPOINT.NUM_PTS", 17, "KEYPOINT.SIGMA", 8.0,
                     "KEYPOINT.NFEATS", 256, "DATASETS.IMAGE_SIZE", (256, 256), "EPIPOLAR.MERGE", "late", "EPIPOLAR.ATTENTION", "avg", "EPIPOLAR.PARAMETERIZED", ("z",),
                     "EPIPOLAR.ZRESIDUAL", True, "EPIPOLAR.USE_CORRECT_NORMALIZE", True, "EPIPOLAR.SAMPLESIZE", 64])
net = MultiViewPoseModel(cfg).to(dev).eval().to(memory_format=torch.channels_last)
P_ref, P_src = syn.make_pairs(32, 4, 256, seed=1000, jitter=(0.05, 8.0))
img = torch.randn(128, 3, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
idx = ring_sources(32, 4, dev)
def t(fn, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
with torch.no_grad():
    r = net.reference
    print("trunk()            %.2f ms" % t(lambda: r.trunk(img)))
    print("net(img)[0]        %.2f ms" % t(lambda: r(img)[0]))
    feats = r.trunk(img)
    print("feats[idx]         %.2f ms" % t(lambda: feats[idx]))
    other = feats[idx]
    print("fused layer        %.2f ms" % t(lambda: r._fuse(feats, r.epipolar_sampler, other, P_ref, P_ref[idx.cpu()], None, None)))
    x = r._fuse(feats, r.epipolar_sampler, other, P_ref, P_ref[idx.cpu()], None, None)[0]
    print("final_layer        %.2f ms" % t(lambda: r.final_layer(x)))
    heat = r.final_layer(x)
    print("find_peaks (HIP)   %.2f ms" % t(lambda: backbones.find_peaks(heat, 8.0, 4)))
    print("soft_argmax (torch) %.2f ms" % t(lambda: backbones.soft_argmax_peaks(heat, 8.0, 4)))
    print("forward_views      %.2f ms" % t(lambda: net.forward_views(img, P_ref, idx)))
