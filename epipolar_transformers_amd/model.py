"""Model level of the path: the `multiview_keypoint` branch of the reference's
`Modelbuilder` (modeling/model.py:29-58 construction, :210-247 forward, :281-302 triangulation) built MI355X-first.

What changes against the reference:
  * N1 -- the trunk runs ONCE per view.  The reference runs the whole pose ResNet twice per (reference, source) pair:
    `self.backbone(other_img)` for the source features (model.py:244) and `self.reference(img, ...)` for the fused
    pass (model.py:246), although in a multi-view batch every view's pre-fusion feature (resnet.py:406,437) is needed
    exactly once.  `forward_views` computes the trunk + deconvolution head of all views in one channels_last pass and
    hands the fused layer `feats` and `feats[source_index]` (exact in eval mode; in training it changes only which
    samples share a BN batch).  MERGE late -- the headline mode; early / both fuse layer1 features with the source's
    DECONVOLUTION features (resnet.py:390-396), which needs the unfused pass first, so they keep two passes.
  * N3 -- test-time lifting stays on the GPU: batched float64 SVD-DLT (`triangulate.py`) instead of the per-joint
    pymvg loop on the CPU (vision/triangulation.py:400-441), no device-to-host copy of the detections.
  * `EPIPOLAR.MULTITEST` (model.py:213-239): every other view in turn as the source, per joint the detection with
    the highest score -- here all (reference, source) combinations go through ONE launch of the fused layer.
  * `BACKBONE.SYNC_BN` converts through `parallel.convert_sync_batchnorm` (model.py:56-58).

The batch-dict keys are the reference's (model.py:166-207).  Training returns the reference's loss dict (`loss`: the
JointsMSELoss of modeling/metrics/metrics2d.py, renamed from `stage_loss0` as model.py:482-484 does for a single entry)
so a training loop written against Modelbuilder keeps working for this task."""
from __future__ import annotations

import torch
from torch import nn

from . import backbones
from .config import get_cfg
from .triangulate import mpjpe, triangulate_dlt


def ring_sources(num_frames: int, num_views: int, device=None) -> torch.Tensor:
    """Index of the source view of every view of a frame-major (frames * views) batch: view v+1 of the same frame
    (the pairing of the synthetic rig; H36M uses `neighbor_cameras`, vision/multiview.py:59-83 -- pass that instead)."""
    return torch.arange(num_frames * num_views, device=device).view(num_frames, num_views).roll(-1, 1).reshape(-1)


class JointsMSELoss(nn.Module):
    """modeling/metrics/metrics2d.py:18-41 JointsMSELoss: per joint the mean squared error of the visibility-weighted
    heat maps (nn.MSELoss(reduction='mean') over batch x pixels), summed over the joints and -- unless
    cfg.KEYPOINT.LOSS_PER_JOINT -- divided by their number.  (No factor 1/2: pinned by tests/golden/model_r18.npz.)"""

    def __init__(self, per_joint: bool = False):
        super().__init__()
        self.per_joint = per_joint

    def forward(self, output, target, target_weight):
        n, j = output.shape[:2]
        pred = output.reshape(n, j, -1)
        gt = target.reshape(n, j, -1)
        w = target_weight.reshape(n, j, 1)
        loss = ((pred * w - gt * w) ** 2).mean(dim=(0, 2)).sum()
        return loss if self.per_joint else loss / j


class MultiViewPoseModel(nn.Module):
    def __init__(self, cfg=None, sharded=None):
        """sharded: a `parallel.ViewShardExchange` -- this process then owns the images of ITS cameras only (one camera
        per GPU at world size = views) and the source features come through the exchange (`forward_views_sharded`);
        what the reference does with nn.DataParallel inside one process (model.py:44,246-247)."""
        super().__init__()
        self.sharded = sharded
        self.cfg = cfg = cfg if cfg is not None else get_cfg()
        assert "epipolarpose" in cfg.BACKBONE.BODY, "MultiViewPoseModel is the multiview_keypoint task of Modelbuilder"
        self.reference = backbones.build_backbone(cfg)                     # model.py:34
        self.backbone = self.reference if cfg.EPIPOLAR.SHARE_WEIGHTS else backbones.build_backbone(cfg)   # :48-57
        if cfg.BACKBONE.SYNC_BN:                                           # :56-58
            from .parallel import convert_sync_batchnorm

            convert_sync_batchnorm(self)
        self.criterion = JointsMSELoss(bool(getattr(cfg.KEYPOINT, "LOSS_PER_JOINT", False)))

    # ------------------------------------------------------------------------------------------ N1
    def forward_views(self, img: torch.Tensor, KRT: torch.Tensor, source_index: torch.Tensor,
                      camera=None, other_camera=None):
        """All views of a batch at once.  img (M,3,Hi,Wi), KRT (M,3,4), source_index (M,) = row of the source view of
        every row.  Returns the backbone's 8-tuple (resnet.py:437) with the trunk run once per view."""
        net = self.reference
        other_KRT = KRT[source_index.to(KRT.device)]            # KRT on the host: pass a host index to avoid a sync
        if self.cfg.EPIPOLAR.MERGE != "late" or self.backbone is not self.reference:
            with torch.set_grad_enabled(torch.is_grad_enabled() and bool(self.cfg.EPIPOLAR.OTHER_GRAD)):
                feats = self.backbone(img)[0]                               # model.py:241-244
            return net(img, [feats[source_index], other_KRT, None, KRT, camera, other_camera, None])
        feature = net.trunk(img)                                            # once per view
        other = feature[source_index]
        if not self.cfg.EPIPOLAR.OTHER_GRAD:
            other = other.detach()
        x, corr_pos, depth, sample_locs = net._fuse(feature, net.epipolar_sampler, other, KRT, other_KRT, camera, other_camera)
        heatmap = net.final_layer(x)
        locs, scos = backbones.find_peaks(heatmap, self.cfg.KEYPOINT.SIGMA, self.cfg.BACKBONE.DOWNSAMPLE)
        return feature, [heatmap], locs, scos, corr_pos, depth, sample_locs, None

    def forward_views_sharded(self, img: torch.Tensor, KRT: torch.Tensor, other_KRT: torch.Tensor, camera=None,
                              other_camera=None, num_chunks: int = 1):
        """The view-sharded partition as a model path, forward AND backward.  img (M,3,Hi,Wi): the images
        of this rank's cameras, camera-major (all frames of camera a, then of camera b, ..: `ViewShardExchange`'s pair
        order); KRT / other_KRT (M,3,4): their projection matrices and those of their source views (the ring neighbour,
        owned by another rank).  The trunk runs on the own images only; the source feature maps arrive through ONE
        all-gather (`parallel.sharded_sources`, channels-last memory as the trunk leaves it; ONE all-to-all with
        EPIPOLAR_AMD.SHARD_P2P: each map only to the rank that samples it), its backward returns
        d(source maps) to their owners with ONE all-to-all; the weight gradients of the shared network are summed by the
        caller (`parallel.allreduce_gradients` or DDP).  Returns the backbone's 8-tuple."""
        from .config import amd_knob
        from .parallel import sharded_sources

        if self.sharded is None:
            raise RuntimeError("MultiViewPoseModel was built without a ViewShardExchange (sharded=...)")
        if self.cfg.EPIPOLAR.MERGE != "late" or self.backbone is not self.reference:
            raise NotImplementedError("the view-sharded path covers MERGE late with SHARE_WEIGHTS (every configs/epipolar/*.yaml)")
        net = self.reference
        feature = net.trunk(img)                                            # own cameras only
        nhwc = feature.permute(0, 2, 3, 1).contiguous()                     # (a view when the trunk ran channels_last)
        other = sharded_sources(nhwc, self.sharded, num_chunks, p2p=bool(amd_knob(self.cfg, "SHARD_P2P", False))).permute(0, 3, 1, 2)
        if not self.cfg.EPIPOLAR.OTHER_GRAD:
            other = other.detach()
        x, corr_pos, depth, sample_locs = net._fuse(feature, net.epipolar_sampler, other, KRT, other_KRT, camera, other_camera)
        heatmap = net.final_layer(x)
        locs, scos = backbones.find_peaks(heatmap, self.cfg.KEYPOINT.SIGMA, self.cfg.BACKBONE.DOWNSAMPLE)
        return feature, [heatmap], locs, scos, corr_pos, depth, sample_locs, None

    def forward_multitest(self, img: torch.Tensor, KRT: torch.Tensor, num_views: int):
        """EPIPOLAR.MULTITEST (model.py:213-239): every other view of the frame as the source in turn; per joint the
        location with the highest score.  img (F*V, ...) frame-major.  One trunk pass per network (two without
        SHARE_WEIGHTS: the sources come from `self.backbone`), ONE fused-layer launch over all F*V*(V-1)
        (reference, source) pairs."""
        net = self.reference
        m = img.shape[0]
        frames = m // num_views
        feature = net.trunk(img)
        # the source features come from `self.backbone` (model.py:219): the same network only with SHARE_WEIGHTS
        source = feature if self.backbone is net else self.backbone(img)[0]
        base = torch.arange(m, device=img.device).view(frames, num_views)
        ref_idx, src_idx = [], []
        for shift in range(1, num_views):
            ref_idx.append(base.reshape(-1))
            src_idx.append(base.roll(-shift, 1).reshape(-1))
        ref_idx, src_idx = torch.cat(ref_idx), torch.cat(src_idx)
        Kc = KRT.to("cpu")
        camera = other_camera = None
        if self.cfg.EPIPOLAR.PRIOR or self.cfg.EPIPOLAR.SIMILARITY == "prior":
            # the learned per-camera-pair tables (epipolar.py:73-80,288-301) are keyed by the camera ids of the data set:
            # view v of a frame-major batch is cfg.DATASETS.CAMERAS[v] (multiview_h36m.py:225-252)
            ids = list(self.cfg.DATASETS.CAMERAS)
            assert len(ids) >= num_views, "EPIPOLAR.PRIOR needs DATASETS.CAMERAS to name every view"
            camera = [ids[int(i) % num_views] for i in ref_idx.tolist()]
            other_camera = [ids[int(i) % num_views] for i in src_idx.tolist()]
        x, _, _, _ = net._fuse(feature[ref_idx], net.epipolar_sampler, source[src_idx], Kc[ref_idx.cpu()], Kc[src_idx.cpu()],
                               camera, other_camera)
        heat = net.final_layer(x)
        locs, scos = backbones.find_peaks(heat, self.cfg.KEYPOINT.SIGMA, self.cfg.BACKBONE.DOWNSAMPLE)
        locs = locs.view(num_views - 1, m, -1, 2)
        scos = scos.view(num_views - 1, m, -1)
        best, which = scos.max(0)                                           # model.py:233-236
        locs = torch.gather(locs, 0, which[None, ..., None].expand(-1, -1, -1, 2)).squeeze(0)
        return locs, best

    # ------------------------------------------------------------------------------------------ N3
    def lift(self, batch_locs: torch.Tensor, batch_scos: torch.Tensor, KRT: torch.Tensor, num_views: int):
        """(F*V, J, 2) detections -> (F, J, 3) world points: the 'naive' / 'pymvg' linear triangulation of
        model.py:281-302, on the device the detections live on."""
        cfg = self.cfg
        m, j, _ = batch_locs.shape
        f = m // num_views
        scale = float(cfg.DATASETS.IMAGE_RESIZE) * float(cfg.DATASETS.PREDICT_RESIZE)
        pts = (batch_locs * scale).view(f, num_views, j, 2)
        return triangulate_dlt(pts, KRT.to(batch_locs.device).view(f, num_views, 3, 4), batch_scos.view(f, num_views, j),
                               conf_thres=float(getattr(cfg.KEYPOINT, "CONF_THRES", 0.05)))

    # ------------------------------------------------------------------------------------------ Modelbuilder.forward
    def forward(self, inputs: dict, is_train: bool = True):
        """The multiview_keypoint branch of Modelbuilder.forward (model.py:160-302).  `inputs` holds the reference's
        keys: img, KRT, and either `other_index` (rows of the source views inside `img`: the de-duplicated path) or
        `other_img` + `other_KRT` (the reference's two-pass form); optional heatmap / visibility (loss) and points-3d
        (MPJPE).  Returns (loss_dict, metric_dict) in training, (loss_dict, metric_dict, out) otherwise, as the reference does
        (model.py:478-493)."""
        cfg = self.cfg
        img = inputs["img"]
        KRT = inputs["KRT"].to(torch.float32)                               # model.py:183-185
        camera, other_camera = inputs.get("camera"), inputs.get("other_camera")
        views = int(inputs.get("num_views", len(cfg.DATASETS.CAMERAS) or 4))
        if cfg.EPIPOLAR.MULTITEST and not is_train:
            with torch.no_grad():
                batch_locs, batch_scos = self.forward_multitest(img, KRT, views)
            heat = corr_pos = depths = None
        else:
            if self.sharded is not None and "other_img" not in inputs and "other_index" not in inputs:
                res = self.forward_views_sharded(img, KRT, inputs["other_KRT"].to(torch.float32), camera, other_camera,
                                                 int(inputs.get("exchange_chunks", 1)))
            elif "other_index" in inputs:
                res = self.forward_views(img, KRT, inputs["other_index"], camera, other_camera)
            else:
                with torch.set_grad_enabled(torch.is_grad_enabled() and bool(cfg.EPIPOLAR.OTHER_GRAD)):
                    other_features = self.backbone(inputs["other_img"])[0]                           # model.py:244
                res = self.reference(img, [other_features, inputs["other_KRT"].to(torch.float32), None, KRT, camera,
                                           other_camera, inputs["other_img"]])                       # model.py:246
            _, heat, batch_locs, batch_scos, corr_pos, depths, _, _ = res
        loss_dict, metric_dict, out = {}, {}, {}
        if is_train and inputs.get("heatmap") is not None:
            vis = inputs.get("visibility")
            vis = torch.ones(heat[0].shape[:2], device=heat[0].device) if vis is None else vis.to(torch.float32)
            loss_dict["stage_loss0"] = self.criterion(heat[0], inputs["heatmap"].to(torch.float32), vis)   # model.py:253
        # the reference's keys (model.py:362-373: heatmap_pred, corr_pos, depth, batch_locs, score_pred) + two aliases
        out.update(heatmap_pred=heat[-1] if heat is not None else None, corr_pos=corr_pos, depth=depths,
                   batch_locs=batch_locs, score_pred=batch_scos, batch_scos=batch_scos,
                   heatmaps=heat[0] if heat is not None else None)
        if not is_train and cfg.VIS.MULTIVIEW:
            pred = self.lift(batch_locs, batch_scos, KRT, views)
            out["points-3d"] = pred
            if inputs.get("points-3d") is not None:
                gt = inputs["points-3d"].to(pred.device).view(-1, views, pred.shape[1], 3)[:, 0]
                metric_dict["MPJPE"] = mpjpe(pred, gt)                      # metrics3d.py:5-46
        # model.py:478-493: several losses are summed into 'loss', a single one is RENAMED to 'loss'; training returns
        # (loss_dict, metric_dict), evaluation (loss_dict, metric_dict, out) with the empty entries of `out` dropped
        if len(loss_dict) > 1:
            loss_dict["loss"] = sum(loss_dict.values())
        elif len(loss_dict) == 1:
            loss_dict["loss"] = loss_dict.popitem()[1]
        if is_train:
            return loss_dict, metric_dict
        return loss_dict, metric_dict, {k: v for k, v in out.items() if v is not None}
