"""Synthetic 4-view scene for the MPJPE-delta check.

Runs the REAL reference operator + the reference peak finder on the CPU (build
container only) on a planted 17-joint skeleton and freezes its 2-D detections:
tests/golden/mpjpe_scene.npz.  The GPU test pushes the same feature maps through
the MI355X path and triangulates both sets of detections with the same batched
DLT; the mean 3-D difference must stay below 0.1 mm.
    python tests/golden/make_mpjpe_scene.py
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402
from epipolar_transformers_amd import synthetic as syn  # noqa: E402

V, J, C, HS, IMG, K, SIGMA = 4, 17, 256, 16, 64, 64, 2.0


def main():
    warnings.simplefilter("ignore")
    torch.set_num_threads(4)
    g = torch.Generator().manual_seed(2024)
    cams = syn.ring_cameras(V, IMG)                                     # (V,3,4) float64, image coordinates
    joints = torch.tensor([0.0, 0.0, 900.0]) + (torch.rand(J, 3, generator=g) - 0.5) * torch.tensor([900.0, 900.0, 1500.0])
    Xh = torch.cat([joints.double(), torch.ones(J, 1, dtype=torch.float64)], 1)
    proj = torch.from_numpy(cams) @ Xh.T                                # (V,3,J)
    uv = (proj[:, :2] / proj[:, 2:3]).permute(0, 2, 1)                  # (V,J,2) image px
    uv_feat = (uv + 0.5 - 2.0) / 4.0                                    # coord2pix, downsample 4
    ys, xs = torch.meshgrid(torch.arange(HS, dtype=torch.float64), torch.arange(HS, dtype=torch.float64), indexing="ij")
    feat = (torch.randn(V, C, HS, HS, generator=g) * 0.3).relu()
    for v in range(V):
        for j in range(J):
            d2 = (xs - uv_feat[v, j, 0]) ** 2 + (ys - uv_feat[v, j, 1]) ** 2
            feat[v, j] = (5.0 * torch.exp(-d2 / (2 * 1.0 ** 2))).float()
    feat = feat.contiguous()
    P = torch.from_numpy(cams).float()
    P_ref, P_src = P, P.roll(-1, 0)                                     # ring neighbour is the source view
    feat_src = feat.roll(-1, 0).contiguous()

    ov = ["KEYPOINT.HEATMAP_SIZE", "(%d, %d)" % (HS, HS), "KEYPOINT.NFEATS", str(C), "EPIPOLAR.SAMPLESIZE", str(K),
          "DATASETS.IMAGE_SIZE", "(%d,%d)" % (IMG, IMG), "KEYPOINT.NUM_PTS", str(J), "KEYPOINT.SIGMA", str(SIGMA)]
    mod, cfg = rh.reference_epipolar(overrides=ov)
    from modeling.backbones.basic_batch import find_tensor_peak_batch   # reference peak finder

    with torch.no_grad():
        mod.z.weight.normal_(0, 0.02, generator=g)
        mod.z.bias.normal_(0, 0.05, generator=g)
        mod.bn.weight.normal_(1, 0.1, generator=g)
        mod.bn.bias.normal_(0, 0.05, generator=g)
        mod.bn.running_mean.normal_(0, 0.05, generator=g)
        mod.bn.running_var.uniform_(0.8, 1.2, generator=g)
    final_w = torch.randn(J, C, 1, 1, generator=g) * 0.01
    final_w[torch.arange(J), torch.arange(J), 0, 0] = 1.0
    final_b = torch.zeros(J)
    mod.eval()
    with torch.no_grad():
        ret, corr_pos, depth, _ = mod(feat, feat_src, P_ref, P_src)    # reference Epipolar.forward (CPU)
        x = ret + feat                                                  # resnet.py:388
        heat = torch.nn.functional.conv2d(x, final_w, final_b)          # resnet.py:421
        locs, scos = [], []
        for v in range(V):                                              # resnet.py:424-430
            l, s = find_tensor_peak_batch(heat[v], SIGMA, 4)
            locs.append(l); scos.append(s)
        locs, scos = torch.stack(locs), torch.stack(scos)
    a, b, e = orc.camera_algebra(P_ref, P_src)
    cam = np.concatenate([a.reshape(V, 12), b.reshape(V, 12), e.reshape(V, 3)], 1).astype(np.float32)
    flat = heat.view(V, J, -1)
    top2 = flat.topk(2, -1).values
    print("min peak margin", float((top2[..., 0] - top2[..., 1]).min()), " max |loc - gt| px", float((locs.double() - uv).abs().max()))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mpjpe_scene.npz"),
                        feat=feat.numpy(), P=P.numpy(), cam=cam, joints=joints.numpy(),
                        z_weight=mod.z.weight.detach().numpy(), z_bias=mod.z.bias.detach().numpy(), bn_weight=mod.bn.weight.detach().numpy(),
                        bn_bias=mod.bn.bias.detach().numpy(), bn_mean=mod.bn.running_mean.numpy(), bn_var=mod.bn.running_var.numpy(),
                        final_w=final_w.numpy(), final_b=final_b.numpy(), ref_locs=locs.numpy(), ref_scores=scos.numpy(),
                        ref_heat_max=flat.max(-1).values.numpy(), meta=np.array([V, J, C, HS, IMG, K], np.int64),
                        sigma=np.float32(SIGMA))


if __name__ == "__main__":
    main()
