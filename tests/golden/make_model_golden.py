"""Generate tests/golden/model_r18.npz from the REAL reference `Modelbuilder` (row N1 of SURVEY.md section 8f).

Runs only in the build container (needs /root/reference, imported read-only through oracle/ref_harness.py).  What is
executed is the reference's own model code (modeling/model.py:29-58,160-302: `multiview_keypoint`, two backbone passes per
pair; modeling/backbones/resnet.py; modeling/layers/epipolar.py) on the CPU:

  * epipolarposeR-18, 64 x 64 images -> 16 x 16 heat maps, K = 16, 2 frames x 4 views = 8 (reference, source) pairs,
    weights from tests/golden/model_weights.py (rebuilt from the parameter names on both sides);
  * eval: heat maps, detections (batch_locs / scores), corr_pos, depth;
  * one training step: the JointsMSELoss value and the gradients of three parameter tensors.

    python tests/golden/make_model_golden.py
"""
import os
import sys
import tempfile
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from model_weights import deterministic_state_dict  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402
from epipolar_transformers_amd import synthetic as syn  # noqa: E402

FRAMES, V, SIZE, HS, K, J = 2, 4, 64, 16, 16, 17


def main():
    tmp = tempfile.mkdtemp()
    ov = ["FOLDER_NAME", tmp, "OUTPUT_DIR", os.path.join(tmp, "h36m_model_golden"), "BACKBONE.BODY", "epipolarposeR-18",
          "BACKBONE.PRETRAINED", "False", "EPIPOLAR.PRETRAINED", "False", "KEYPOINT.HEATMAP_SIZE", "(%d, %d)" % (HS, HS),
          "KEYPOINT.NUM_PTS", str(J), "KEYPOINT.SIGMA", "2.0", "DATASETS.IMAGE_SIZE", "(%d, %d)" % (SIZE, SIZE),
          "DEVICE", "cpu", "KEYPOINT.NFEATS", "256", "EPIPOLAR.SAMPLESIZE", str(K), "VIS.MULTIVIEW", "False", "TEST.PCK", "False"]
    cfg = rh.load_cfg("configs/epipolar/keypoint_h36m_zresidual_fixed.yaml", ov)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import io
        import contextlib

        with contextlib.redirect_stdout(io.StringIO()):
            from modeling.model import Modelbuilder

            model = Modelbuilder(cfg)
    net = model.reference.module if hasattr(model.reference, "module") else model.reference
    net.load_state_dict(deterministic_state_dict(net.state_dict()))
    assert model.backbone is model.reference                                # SHARE_WEIGHTS

    g = torch.Generator().manual_seed(1234)
    img = torch.randn(FRAMES * V, 3, SIZE, SIZE, generator=g)
    src = torch.arange(FRAMES * V).view(FRAMES, V).roll(-1, 1).reshape(-1)  # ring neighbour inside the frame
    P_ref, P_src = syn.make_pairs(FRAMES, V, SIZE, seed=77, jitter=(0.03, 2.0))
    assert torch.equal(P_src, P_ref[src])
    target = torch.rand(FRAMES * V, J, HS, HS, generator=g)
    vis = (torch.rand(FRAMES * V, J, 1, generator=g) > 0.2).float()
    cam = np.concatenate([a.reshape(FRAMES * V, -1) for a in orc.camera_algebra(P_ref, P_src)], 1).astype(np.float32)

    inputs = lambda: {"img": img.clone(), "other_img": img[src].clone(), "KRT": P_ref.clone(), "other_KRT": P_src.clone(),
                      "heatmap": target.clone(), "visibility": vis.clone()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.eval()
        with torch.no_grad():
            _, metric, out = model(inputs(), is_train=False)          # (eval returns loss_dict, metric_dict, out: model.py:493)
        feat = net(img)[0].detach()                                         # pre-fusion features (resnet.py:406)
        model.train()
        model.zero_grad()
        loss_dict, _ = model(inputs(), is_train=True)
        loss = loss_dict["loss"]                                        # (a single entry is renamed: model.py:482-484)
        loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "model_r18.npz"),
        meta=np.array([FRAMES, V, SIZE, HS, K, J]), img=img.numpy(), src=src.numpy(), KRT=P_ref.numpy(), cam=cam,
        target=target.numpy(), vis=vis.numpy(),
        heat_eval=out["heatmap_pred"].numpy(), locs_eval=np.asarray(out["batch_locs"]), scores_eval=np.asarray(out["score_pred"]),
        corr_pos=out["corr_pos"].numpy(), depth=out["depth"].numpy(), feat_norm=np.array([feat.abs().max().item(), feat.norm().item()]),
        feat_slice=feat[:, :8].numpy(),
        loss=np.array([loss.item()], np.float64),
        grad_conv1=grads["conv1.weight"].numpy(), grad_z_rows=grads["epipolar_sampler.z.weight"][:16].numpy(),
        grad_final=grads["final_layer.weight"].numpy(),
        grad_norms=np.array([grads[k].norm().item() for k in sorted(grads)], np.float64),
        grad_keys=np.array(sorted(grads)))
    print("heat maps: max %.3f; scores mean %.3f; loss %.6f; |feat| max %.2f" % (
        out["heatmap_pred"].abs().max().item(), float(np.mean(np.asarray(out["score_pred"]))), loss.item(), feat.abs().max().item()))
    print("wrote tests/golden/model_r18.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "model_r18.npz")), "bytes")


if __name__ == "__main__":
    main()
