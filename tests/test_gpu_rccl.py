"""The view-sharded exchange on RCCL, on ONE MI355X.

`parallel.ViewShardExchange` / `sharded_sources` have run on gloo with CPU tensors only (tests/test_parallel_gloo.py);
no multi-GPU box is available to this build, so no scaling curve exists.  What one GPU CAN prove is the device side of
the plumbing the gloo tests cannot reach: `all_gather_into_tensor` / `all_to_all_single` of backend "nccl" (= RCCL) on
device NHWC tensors, asynchronous chunks on RCCL's stream handed to the custom HIP kernels on torch's current stream,
and the autograd function that returns d(source maps) through the all-to-all.  A 1-rank process group owns all four
cameras (world <= V), so every collective really executes in RCCL and the result must equal the no-exchange path.

Runs in a child process: the RCCL communicator and the process group stay out of the pytest process.
Replaces the reference's nn.DataParallel coupling (modeling/model.py:44,246-247).
"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["ET_ROOT"])
import torch
import torch.distributed as dist
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn
from epipolar_transformers_amd.parallel import ViewShardExchange, sharded_sources

_lib.load()
ops.POISON_OUTPUTS = True
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
V, frames, H, C, K = 4, 4, 16, 256, 16
ex = ViewShardExchange(1, 0, V)
P_ref, P_src = ex.select_pairs(frames * V, H * 4, seed=11)             # camera-major, as the exchange orders them
n = P_ref.shape[0]
g = torch.Generator(device=dev).manual_seed(3)
own = torch.randn(n, H, H, C, device=dev, generator=g).relu_()          # the maps this rank's trunk produced (NHWC)
cam = camera.pair_algebra(P_ref, P_src).to(dev)
spec = ops.LayerSpec(H=H, W=H, K=K)

# --- no-exchange path: the source map of (camera v, frame f) is the map of camera (v + 1) % V of the same frame
idx = torch.cat([torch.arange(frames) + ((v + 1) % V) * frames for v in range(V)]).to(dev)
src_local = own[idx].contiguous()
out0, attn0, corr0 = ops.forward_nhwc(spec, own, src_local, cam)

# --- 1. one all-gather on RCCL
src_rccl = ex.gather_sources(own)
assert src_rccl.is_cuda and torch.equal(src_rccl, src_local), "all_gather_into_tensor on RCCL returned other maps"

# --- 1b. the point-to-point form: one all_to_all_single on RCCL, the same maps
src_p2p = ex.exchange_sources(own)
assert src_p2p.is_cuda and torch.equal(src_p2p, src_local), "all_to_all_single (exchange_sources) on RCCL returned other maps"

# --- 2. asynchronous chunks (RCCL's stream) feeding the fused kernel (torch's current stream), range by range
out1 = torch.full_like(out0, float("nan")); attn1 = torch.full_like(attn0, float("nan")); corr1 = torch.full_like(corr0, float("nan"))
seen = 0
for ranges, maps in ex.gather_sources_chunked(own, 2):
    ref_c = torch.cat([own[a:b] for a, b in ranges]).contiguous()
    cam_c = torch.cat([cam[a:b] for a, b in ranges]).contiguous()
    o, a_, c_ = ops.forward_nhwc(spec, ref_c, maps.contiguous(), cam_c)
    off = 0
    for a, b in ranges:
        out1[a:b], attn1[a:b], corr1[a:b] = o[off:off + b - a], a_[off:off + b - a], c_[off:off + b - a]
        off += b - a
    seen += off
torch.cuda.synchronize()
assert seen == n
# (a pair's result does not depend on which other pairs share its launch: bit for bit)
assert torch.equal(out1, out0) and torch.equal(attn1, attn0) and torch.equal(corr1, corr0), "chunked exchange + kernel differs"

# --- 3. the exchange as an autograd step: all-gather forward, all-to-all backward, through the fused operator
w = torch.randn(n, C, H, H, device=dev, generator=g)

def loss_through(src_of):
    a = own.detach().clone().requires_grad_(True)
    src = src_of(a)                                                   # (n, H, W, C)
    out, _, _ = ops.EpipolarAttend.apply(a.permute(0, 3, 1, 2), src.permute(0, 3, 1, 2), cam, spec)
    ((out * w).sum() + (src * src).sum()).backward()
    return a.grad

for chunks, p2p in ((1, False), (2, False), (2, True)):
    g_rccl = loss_through(lambda a: sharded_sources(a, ex, num_chunks=chunks, p2p=p2p))
    g_loc = loss_through(lambda a: a[idx])
    torch.cuda.synchronize()
    assert torch.isfinite(g_rccl).all()
    # the tile backward sums d(feat_src) with float atomics: reproducible to rounding, not bit for bit
    err = (g_rccl - g_loc).abs().max().item()
    assert err <= 1e-4 * g_loc.abs().max().item(), (chunks, p2p, err)
# the routing itself (no kernel in between) is a permutation: exact
a = own.detach().clone().requires_grad_(True)
(sharded_sources(a, ex, num_chunks=2) * src_local).sum().backward()
want = torch.zeros_like(own); want[idx] = src_local
assert torch.equal(a.grad, want), "all_to_all_single on RCCL routed the gradients elsewhere"
ops.check_tile_errors()
dist.barrier()
dist.destroy_process_group()
print("rccl-1rank ok")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_view_shard_exchange_runs_on_rccl_with_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", ET_ROOT=ROOT)
    proc = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=600)
    assert proc.returncode == 0 and "rccl-1rank ok" in proc.stdout, proc.stdout[-4000:]


@pytest.mark.parametrize("world,p2p,rig", [(4, False, "ring"), (4, True, "ring"), (2, True, "ring"), (8, False, "ring"),
                                           (4, True, "h36m_room"), (4, True, "uneven_arc"), (2, False, "uneven_arc")],
                         ids=["4ranks-allgather", "4ranks-p2p", "2ranks-p2p", "8ranks-allgather", "4ranks-p2p-room-rig",
                              "4ranks-p2p-shared-source", "2ranks-allgather-shared-source"])
def test_bench_view_partition_with_several_ranks_on_one_gpu(tmp_path, world, p2p, rig):
    """`bench.py --gpus N --partition views` with N processes on cuda:0 (BENCH_SINGLE_DEVICE=1, gloo instead of RCCL: a 1-GPU
    box has no second device): `layer_step_view_sharded`, `select_pairs`, the chunked exchange (all-gather and point-to-point)
    and its range bookkeeping with world > 1 on DEVICE tensors.  Every rank dumps its maps, matrices and the x of one step;
    the test recomputes each rank's x in this process from the maps of the rank that owns its source camera."""
    import json

    import torch

    hw, frames, V = 32, 2, 4
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1", BENCH_DIST_BACKEND="gloo", BENCH_DUMP_DIR=str(tmp_path),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--partition", "views",
           "--steps", "2", "--warmup", "1", "--frames", str(frames), "--hw", str(hw), "--samples", "16", "--exchange-chunks", "2",
           "--no-cpu-baseline", "--no-end-to-end", "--no-other-configs", "--rig", rig] + (["--p2p"] if p2p else [])
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, cwd=ROOT)
    assert proc.returncode == 0, proc.stdout[-4000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == world and res["config"]["partition"] == "views" and res["value"] > 0
    assert ("all_to_all" in res["config"]["exchange"]) == p2p

    from epipolar_transformers_amd import camera, ops, synthetic as syn
    from epipolar_transformers_amd.parallel import ViewShardExchange

    # (the pairing table: the ring neighbour, or the reference's nearest-camera rule -- room rig [2, 3, 0, 1], uneven arc
    #  [1, 0, 1, 2]: two views share a source, one camera is nobody's)
    table = None if rig == "ring" else syn.source_table(rig)
    dumps = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    spec = ops.LayerSpec(H=hw, W=hw, K=16)
    n = frames * V
    for r, d in enumerate(dumps):
        ex = ViewShardExchange(world, r, V, source_of=table)
        assert d["my_cams"] == ex.my_cams and d["x"].shape[0] == n
        f = n // len(ex.my_cams)
        src = torch.empty_like(d["feat"])
        for ci, cam_id in enumerate(ex.my_cams):
            owner, idx = ex.source_location(cam_id)
            src[ci * f:(ci + 1) * f] = dumps[owner]["feat"][idx * f:(idx + 1) * f]
            # the matrices follow the same pairing: the source matrix of my camera is the reference matrix of its owner
            assert torch.equal(d["P_src"][ci * f:(ci + 1) * f], dumps[owner]["P_ref"][idx * f:(idx + 1) * f])
        cam = camera.pair_algebra(d["P_ref"], d["P_src"]).cuda()
        packed = ops.residual_gemm_pack(d["w_fold_t"].t().contiguous().cuda())
        x, _, _ = ops.forward_fused_nhwc(spec, d["feat"].cuda(), src.cuda(), cam, packed, d["b_fold"].cuda())
        # (a pair's result does not depend on which pairs share its launch: the chunked step must reproduce it bit for bit)
        assert torch.equal(x.cpu(), d["x"]), "rank %d of %d: x of the sharded step differs (max %g)" % (
            r, world, (x.cpu() - d["x"]).abs().max().item())
