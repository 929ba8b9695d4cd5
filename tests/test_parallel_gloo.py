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


@pytest.mark.parametrize("world,V", [(2, 4), (4, 4), (2, 8)])
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
