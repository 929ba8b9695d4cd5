"""Multi-process CPU tests (gloo) of the two multi-GPU partitions.  The tensors
are CPU stand-ins; what is under test is the sharding/exchange logic that the
RCCL path runs unchanged on the GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from epipolar_transformers_amd.parallel import ViewShardExchange, frames_partition


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _global_map(frame, view, shape=(2, 3, 4)):
    return torch.full(shape, float(frame * 10 + view))


def _worker(rank, world, port, V, frames, errs):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ex = ViewShardExchange(world, rank, V)
        P_ref, P_src = ex.select_pairs(frames * V, 64, seed=3)
        assert P_ref.shape[0] == len(ex.my_cams) * frames
        fr0 = ex.slice_id * frames                       # global frame ids of this slice
        own = torch.stack([_global_map(fr0 + f, v) for v in ex.my_cams for f in range(frames)])
        src = ex.gather_sources(own)
        want = torch.stack([_global_map(fr0 + f, (v + 1) % V) for v in ex.my_cams for f in range(frames)])
        assert torch.equal(src, want), "rank %d got wrong source maps" % rank
        # point-to-point form (one all-to-all: every block to the one rank that samples it): the same maps, bit for bit
        assert torch.equal(ex.exchange_sources(own), want), "rank %d: all-to-all exchange differs from the all-gather" % rank
        # ... and as the autograd step: forward = exchange_sources, backward = the reverse all-to-all
        from epipolar_transformers_amd.parallel import sharded_sources
        for p2p in (False, True):
            for chunks in (1, 2):
                a = own.clone().requires_grad_(True)
                out = sharded_sources(a, ex, num_chunks=chunks, p2p=p2p)
                assert torch.equal(out, want)
                wgt = torch.stack([_global_map(fr0 + f, v) + 0.25 for v in ex.my_cams for f in range(frames)])
                (out * wgt).sum().backward()
                want_g = torch.stack([_global_map(fr0 + f, (c - 1) % V) + 0.25 for c in ex.my_cams for f in range(frames)])
                assert torch.equal(a.grad, want_g), "rank %d: gradient routing (p2p=%s, chunks=%d)" % (rank, p2p, chunks)
        # chunked / overlappable form: same maps, delivered in frame ranges
        got = torch.empty_like(want)
        seen = 0
        for ranges, maps in ex.gather_sources_chunked(own, 2):
            off = 0
            for a, b in ranges:
                got[a:b] = maps[off:off + (b - a)]
                off += b - a
            seen += off
        assert seen == want.shape[0] and torch.equal(got, want), "rank %d chunked gather mismatch" % rank
        # ... and its point-to-point counterpart (one asynchronous all-to-all per frame range): interchangeable
        for chunks in (1, 2, 3):
            got = torch.full_like(want, float("nan"))
            seen = 0
            for ranges, maps in ex.exchange_sources_chunked(own, chunks):
                off = 0
                for a, b in ranges:
                    got[a:b] = maps[off:off + (b - a)]
                    off += b - a
                seen += off
            assert seen == want.shape[0] and torch.equal(got, want), "rank %d chunked p2p exchange mismatch (%d chunks)" % (rank, chunks)
        # projection matrices follow the same pairing: the source matrix of (frame, v) is the
        # reference matrix of camera (v+1) % V of the same frame
        allref = [torch.empty_like(P_ref) for _ in ex.group_ranks]
        dist.all_gather(allref, P_ref, group=ex._pg())
        for ci, v in enumerate(ex.my_cams):
            owner, idx = ex.source_location(v)
            got = allref[ex.group_ranks.index(owner)][idx * frames:(idx + 1) * frames]
            assert torch.equal(got, P_src[ci * frames:(ci + 1) * frames])
        # backward routing: gradient w.r.t. the gathered sources returns to the owning rank
        g = torch.stack([_global_map(fr0 + f, v) + 0.5 for v in ex.my_cams for f in range(frames)])
        back = ex.scatter_source_grads(g)
        # own map of camera c is the source of reference camera (c-1) % V
        want_b = torch.stack([_global_map(fr0 + f, (c - 1) % V) + 0.5 for c in ex.my_cams for f in range(frames)])
        assert torch.equal(back, want_b), "rank %d got wrong source gradients" % rank
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # pragma: no cover - surfaced in the parent
        errs.put("rank %d: %r" % (rank, exc))
        raise


# (8, 4): two ranks per camera -- two frame slices, each a group of four ranks of its own (dist.new_group), BASELINE
# configs[3] on an 8-GPU node
@pytest.mark.parametrize("world,V", [(2, 4), (4, 4), (2, 8), (8, 4)])
def test_view_sharded_exchange_gloo(world, V):
    ctx = mp.get_context("spawn")
    errs = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, V, 3, errs)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    msgs = []
    while not errs.empty():
        msgs.append(errs.get())
    assert not msgs, msgs
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def _worker_table(rank, world, port, rig, frames, errs):
    """The exchange under a pairing table that is not the ring: the room rig's nearest-neighbour table [2, 3, 0, 1] and the
    uneven arc's [1, 0, 1, 2] (cameras 0 and 2 share a source, camera 3 is nobody's) -- against what ONE process computes
    from the same maps: forward bit for bit, the summed gradient to 1e-6."""
    try:
        from epipolar_transformers_amd import synthetic as syn
        from epipolar_transformers_amd.parallel import sharded_sources

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        V = 4
        table = syn.source_table(rig)
        ex = ViewShardExchange(world, rank, V, source_of=table)
        P_ref, P_src = ex.select_pairs(frames * V, 64, seed=3, rig=rig)
        # the single-process view of the same batch: every map of every (frame, camera), random, the same on every rank
        g = torch.Generator().manual_seed(1234)
        maps = torch.randn(frames, V, 2, 3, 4, generator=g)
        wgt = torch.randn(frames, V, 2, 3, 4, generator=g)
        layer = lambda ref, src: ref * src + 0.5 * src * src            # a differentiable stand-in for the layer
        full = maps.clone().requires_grad_(True)
        src_full = torch.stack([full[:, table[v]] for v in range(V)], 1)             # (frames, V, ...): source map of pair (f, v)
        out_full = layer(full, src_full)
        (out_full * wgt).sum().backward()
        # this rank's share, camera-major
        own = torch.cat([maps[:, v] for v in ex.my_cams])
        want_src = torch.cat([maps[:, table[v]] for v in ex.my_cams])
        assert torch.equal(ex.gather_sources(own), want_src)
        assert torch.equal(ex.exchange_sources(own), want_src)
        for fn in (ex.gather_sources_chunked, ex.exchange_sources_chunked):
            got = torch.full_like(want_src, float("nan"))
            for ranges, m in fn(own, 2):
                off = 0
                for a, b in ranges:
                    got[a:b] = m[off:off + (b - a)]
                    off += b - a
            assert torch.equal(got, want_src), fn.__name__
        # the projection matrices follow the table
        cams = torch.from_numpy(syn.rig_cameras(rig, frames, 64, seed=3, jitter=(0.05, 8.0))).float()
        assert torch.equal(P_ref, torch.cat([cams[:, v] for v in ex.my_cams]))
        assert torch.equal(P_src, torch.cat([cams[:, table[v]] for v in ex.my_cams]))
        if rig == "h36m_room":      # ... and are the pairs the single-process rig generator makes (frame-major there)
            r1, r2 = syn.rig_pairs(rig, frames, 64, seed=3, jitter=(0.05, 8.0))
            assert torch.equal(P_src, torch.cat([r2.view(frames, V, 3, 4)[:, v] for v in ex.my_cams]))
        for p2p in (False, True):
            for chunks in (1, 2):
                a = own.clone().requires_grad_(True)
                src = sharded_sources(a, ex, num_chunks=chunks, p2p=p2p)
                out = layer(a, src)
                want_out = torch.cat([out_full[:, v] for v in ex.my_cams]).detach()
                assert torch.equal(out, want_out), "forward differs from the single process (p2p=%s, chunks=%d)" % (p2p, chunks)
                (out * torch.cat([wgt[:, v] for v in ex.my_cams])).sum().backward()
                want_g = torch.cat([full.grad[:, v] for v in ex.my_cams])
                err = (a.grad - want_g).abs().max().item()
                assert err <= 1e-6 * max(1.0, want_g.abs().max().item()), (err, p2p, chunks)
        # a camera nobody samples gets exactly zero from the exchange's backward
        back = ex.scatter_source_grads(torch.ones_like(own))
        for ci, c in enumerate(ex.my_cams):
            assert torch.equal(back[ci * frames:(ci + 1) * frames], torch.full_like(back[:frames], float(table.count(c))))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # pragma: no cover - surfaced in the parent
        errs.put("rank %d: %r" % (rank, exc))
        raise


@pytest.mark.parametrize("world,rig", [(4, "h36m_room"), (4, "uneven_arc"), (2, "uneven_arc"), (2, "h36m_room")])
def test_view_sharded_exchange_with_the_reference_pairing(world, rig):
    """the view-sharded exchange under the reference's nearest-camera pairing (a table, not the ring)."""
    from epipolar_transformers_amd import synthetic as syn
    assert syn.source_table("h36m_room") == [2, 3, 0, 1] and syn.source_table("uneven_arc") == [1, 0, 1, 2]
    ctx = mp.get_context("spawn")
    errs = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_table, args=(r, world, port, rig, 3, errs)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    msgs = []
    while not errs.empty():
        msgs.append(errs.get())
    assert not msgs, msgs
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def test_view_sharded_layout_world8_views4():
    # two ranks per camera: groups of 4 ranks exchange among themselves
    for rank in range(8):
        ex = ViewShardExchange(8, rank, 4)
        assert ex.my_cams == [rank % 4] and ex.slice_id == rank // 4
        owner, idx = ex.source_location(rank % 4)
        assert owner == (rank // 4) * 4 + (rank % 4 + 1) % 4 and idx == 0


def test_frames_partition_covers_everything():
    for frames in (1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [frames_partition(frames, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == frames
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---------------------------------------------------------------------------------------
# SyncBN of the layer's z-epilogue (training-time coupling across ranks, BACKBONE.SYNC_BN)
# ---------------------------------------------------------------------------------------
def _syncbn_worker(rank, world, port, errs):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from epipolar_transformers_amd import default_cfg
        from epipolar_transformers_amd.epipolar import Epipolar
        from epipolar_transformers_amd.parallel import SyncBatchNorm2d, convert_sync_batchnorm

        C, H = 8, 6
        cfg = default_cfg()
        cfg.merge_from_list(["KEYPOINT.HEATMAP_SIZE", (H, H), "KEYPOINT.NFEATS", C, "EPIPOLAR.PARAMETERIZED", ("z",),
                             "EPIPOLAR.ZRESIDUAL", True])
        torch.manual_seed(0)                               # same weights and the same GLOBAL batch on every rank
        single = Epipolar(cfg=cfg)
        with torch.no_grad():
            single.bn.weight.normal_(1, 0.1)
            single.bn.bias.normal_(0, 0.1)
        out_all = torch.randn(2 * world, C, H, H)
        gout_all = torch.randn(2 * world, C, H, H)
        keys_before = sorted(single.state_dict())
        import copy
        multi = convert_sync_batchnorm(copy.deepcopy(single))
        assert isinstance(multi.bn, SyncBatchNorm2d) and sorted(multi.state_dict()) == keys_before
        single.train()
        multi.train()
        # reference: ONE process, the whole batch, plain batch norm (what the reference gets from its
        # SynchronizedBatchNorm2d across DataParallel replicas, sync_batchnorm/batchnorm.py:114-122)
        a = out_all.clone().requires_grad_(True)
        fin_all, _ = single._epilogue_torch(a)
        (fin_all * gout_all).sum().backward()
        # this rank: its shard only, statistics synchronised
        sl = slice(2 * rank, 2 * rank + 2)
        b = out_all[sl].clone().requires_grad_(True)
        fin, _ = multi._epilogue_torch(b)
        (fin * gout_all[sl]).sum().backward()
        assert torch.allclose(fin, fin_all[sl], atol=1e-5), "rank %d: synced BN output differs" % rank
        assert torch.allclose(b.grad, a.grad[sl], atol=1e-5), "rank %d: synced BN input gradient differs" % rank
        assert torch.allclose(multi.bn.running_mean, single.bn.running_mean, atol=1e-6)
        assert torch.allclose(multi.bn.running_var, single.bn.running_var, atol=1e-5)
        # parameter gradients are per-rank partial sums (DDP / an all-reduce adds them up)
        gw = multi.bn.weight.grad.clone()
        dist.all_reduce(gw)
        assert torch.allclose(gw, single.bn.weight.grad, atol=1e-4)
        gz = multi.z.weight.grad.clone()
        dist.all_reduce(gz)
        assert torch.allclose(gz, single.z.weight.grad, atol=1e-4)
        # eval mode: no coupling, identical to the plain layer
        multi.eval(); single.eval()
        with torch.no_grad():
            assert torch.allclose(multi._epilogue_torch(out_all[sl])[0], single._epilogue_torch(out_all[sl])[0], atol=1e-6)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # pragma: no cover
        errs.put("rank %d: %r" % (rank, exc))
        raise


def test_sync_batchnorm_epilogue_matches_single_process_gloo():
    ctx = mp.get_context("spawn")
    errs = ctx.SimpleQueue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_syncbn_worker, args=(r, world, port, errs)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    msgs = []
    while not errs.empty():
        msgs.append(errs.get())
    assert not msgs, msgs
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


# ---------------------------------------------------------------------------------------
# The view-sharded partition as a TRAINING path: MultiViewPoseModel(sharded=...) forward + backward over the exchange
# ---------------------------------------------------------------------------------------
def _cpu_attend(self, feat1, feat2, P1, P2):
    """CPU stand-in for the fused HIP operator in this test (the product path has none): the reference's own op
    sequence (oracle/torch_ref_path.py, autograd included) over the oracle's sample locations."""
    from oracle import oracle as orc, torch_ref_path as trp

    spec = orc.LayerSpec(self.feat_h, self.feat_w, self.sample_size)
    locs = torch.from_numpy(orc.sample_locs(spec, P1, P2))
    return trp.forward(feat1, feat2, locs)


def _sharded_train_worker(rank, world, port, V, frames, chunks, errs):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        import copy

        from epipolar_transformers_amd import default_cfg, synthetic as syn
        from epipolar_transformers_amd.epipolar import Epipolar
        from epipolar_transformers_amd.model import MultiViewPoseModel
        from epipolar_transformers_amd.parallel import allreduce_gradients, convert_sync_batchnorm

        Epipolar.attend = _cpu_attend
        size, hs, J = 64, 16, 17
        cfg = default_cfg()
        cfg.merge_from_list(["BACKBONE.BODY", "epipolarposeR-18", "BACKBONE.PRETRAINED", False, "DATASETS.TASK", "multiview_keypoint",
                             "KEYPOINT.HEATMAP_SIZE", (hs, hs), "KEYPOINT.NUM_PTS", J, "KEYPOINT.SIGMA", 2.0,
                             "DATASETS.IMAGE_SIZE", (size, size), "EPIPOLAR.MERGE", "late", "EPIPOLAR.ATTENTION", "avg",
                             "EPIPOLAR.PARAMETERIZED", ("z",), "EPIPOLAR.ZRESIDUAL", True, "EPIPOLAR.USE_CORRECT_NORMALIZE", True,
                             "EPIPOLAR.SAMPLESIZE", 8, "EPIPOLAR.SHARE_WEIGHTS", True,
                             "EPIPOLAR_AMD.SHARD_P2P", chunks == 2])    # (the chunked case also takes the all-to-all forward)
        ex = ViewShardExchange(world, rank, V)
        torch.manual_seed(5)                                   # the same network and the same GLOBAL batch on every rank
        single = MultiViewPoseModel(cfg)
        with torch.no_grad():
            single.reference.epipolar_sampler.bn.weight.normal_(1, 0.1)
            single.reference.epipolar_sampler.bn.bias.normal_(0, 0.1)
        sharded = MultiViewPoseModel(cfg, sharded=ex)
        sharded.load_state_dict(copy.deepcopy(single.state_dict()))
        convert_sync_batchnorm(sharded)                        # BACKBONE.SYNC_BN: batch statistics span the ranks
        g = torch.Generator().manual_seed(9)
        img_all = torch.randn(frames, V, 3, size, size, generator=g)
        tgt_all = torch.rand(frames, V, J, hs, hs, generator=g)
        P_ref_all, P_src_all = syn.make_pairs(frames, V, size, seed=3, jitter=(0.05, 8.0))        # frame-major
        # ---- one process, the whole batch (frame-major), plain batch norm: what the reference's DataParallel model computes
        single.train()
        src_index = torch.arange(frames * V).view(frames, V).roll(-1, 1).reshape(-1)
        def keep(store):
            def hook_fn(module, inputs, output):               # (returns None: the output itself is passed on)
                output.retain_grad()
                store["f"] = output
            return hook_fn

        feats = {}
        hook = single.reference.deconv_layers.register_forward_hook(keep(feats))
        loss_full, _ = single({"img": img_all.reshape(-1, 3, size, size), "KRT": P_ref_all, "other_index": src_index,
                               "heatmap": tgt_all.reshape(-1, J, hs, hs), "num_views": V}, is_train=True)
        loss_full["loss"].backward()
        hook.remove()
        dfeat_full = feats["f"].grad.view(frames, V, 256, hs, hs)
        # ---- this rank: the images of its cameras only (camera-major), sources through the exchange
        P_ref, P_src = ex.select_pairs(frames * V, size, seed=3)
        own = lambda t: torch.cat([t[:, v] for v in ex.my_cams])
        assert torch.equal(P_ref, own(P_ref_all.view(frames, V, 3, 4))) and torch.equal(P_src, own(P_src_all.view(frames, V, 3, 4)))
        sharded.train()
        feats_s = {}
        hook = sharded.reference.deconv_layers.register_forward_hook(keep(feats_s))
        loss_own, _ = sharded({"img": own(img_all), "KRT": P_ref, "other_KRT": P_src, "heatmap": own(tgt_all), "num_views": V,
                               "exchange_chunks": chunks}, is_train=True)
        (loss_own["loss"] / world).backward()                  # global loss = mean of the ranks' (equally sized) means
        hook.remove()
        total = loss_own["loss"].detach().clone()
        dist.all_reduce(total)
        assert abs(total.item() / world - loss_full["loss"].item()) <= 1e-5 * abs(loss_full["loss"].item()), \
            "rank %d: sharded loss %.8f vs single-process %.8f" % (rank, total.item() / world, loss_full["loss"].item())
        # d loss / d (pre-fusion feature map) of every own view: the reference role AND the source role (returned by the
        # all-to-all of the exchange's backward) -- equal to the single-process gradient of that view
        want = own(dfeat_full)
        got = feats_s["f"].grad
        scale = want.abs().max().item()
        assert (got - want).abs().max().item() <= 2e-4 * scale, "rank %d: d feat differs by %g (scale %g)" % (
            rank, (got - want).abs().max().item(), scale)
        # weight gradients of the shared network: summed over the ranks they equal the single-process ones
        allreduce_gradients(sharded)
        ps, pf = dict(sharded.named_parameters()), dict(single.named_parameters())
        assert sorted(ps) == sorted(pf)
        for k in ("reference.conv1.weight", "reference.epipolar_sampler.z.weight", "reference.final_layer.weight",
                  "reference.layer3.0.conv1.weight", "reference.epipolar_sampler.bn.weight"):
            a, b = ps[k].grad, pf[k].grad
            assert (a - b).abs().max().item() <= 5e-4 * max(b.abs().max().item(), 1e-8), "rank %d: grad of %s differs" % (rank, k)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # pragma: no cover
        errs.put("rank %d: %r" % (rank, exc))
        raise


@pytest.mark.parametrize("chunks", [1, 2])
def test_view_sharded_training_step_matches_single_process_gloo(chunks):
    """north_star: "per-view forward/backward is sharded one-camera-per-GPU ... with RCCL all-gather of source feature
    maps".  World 2 (two cameras per rank), 2 frames x 4 views of epipolarposeR-18: a sharded training step --
    trunk on the own images, `parallel.sharded_sources` (all-gather forward -- one all-to-all with EPIPOLAR_AMD.SHARD_P2P, the
    chunked case --, all-to-all backward), SyncBN, the summed
    weight gradients -- reproduces the single-process loss, d feat of every view and the parameter gradients.
    (No multi-GPU curve has been measured; this covers correctness of the path the GPUs run unchanged over RCCL.)"""
    ctx = mp.get_context("spawn")
    errs = ctx.SimpleQueue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_sharded_train_worker, args=(r, world, port, 4, 2, chunks, errs)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    msgs = []
    while not errs.empty():
        msgs.append(errs.get())
    assert not msgs, msgs
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
