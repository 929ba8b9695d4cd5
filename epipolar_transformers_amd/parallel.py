"""Multi-GPU partitions of the path.  One process per GPU,
`torch.distributed` ("nccl" is RCCL on ROCm; tests use "gloo" on CPU).

Pairs are independent units, so the natural partition is collective-free:
  * frames-DP  -- a rank owns whole frames (all V views local): no exchange.
The north-star partition shards by camera and has ONE real exchange step:
  * view-sharded -- rank r owns camera(s) v == r (mod G) for its frames; the
    source map of pair (frame, v) is the map of camera source_of[v] of the same
    frame -- the ring neighbour (v+1) mod V by default, the reference's
    NEAREST camera in general (vision/multiview.py:59-83 neighbor_cameras +
    data/datasets/multiview_h36m.py:231-238; `synthetic.source_table`) --,
    produced on another rank, so the per-rank feature maps are all-gathered
    (RCCL over xGMI) before the fused kernel consumes them.  The nearest-camera
    rule is not a permutation: two views may share a source and a view may be
    nobody's, so the point-to-point form sends a block to EVERY rank that
    samples it and the backward SUMS what comes back.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import synthetic as syn


def frames_partition(num_frames: int, world: int, rank: int):
    """Contiguous frame range [lo, hi) owned by `rank` (frames-DP)."""
    base, rem = divmod(num_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ViewShardExchange:
    """Book-keeping of the view-sharded partition for V cameras on `world` ranks.

    world <= V : rank r owns cameras {v : v % world == r} for every frame.
    world >  V : world must be a multiple of V; rank r owns camera r % V for the
                 frame slice r // V (groups of V ranks exchange among themselves).
    A rank's local pair list is ordered (camera-major, then frame)."""

    def __init__(self, world: int, rank: int, num_views: int, group=None, source_of=None):
        """source_of: the pairing table, source_of[v] = the camera reference camera v samples (default: the ring neighbour;
        `synthetic.source_table(rig)` / the reference's neighbor_cameras in general -- any map of 0..V-1 into itself)."""
        self.source_of = [(v + 1) % num_views for v in range(num_views)] if source_of is None else [int(v) for v in source_of]
        if len(self.source_of) != num_views or any(not 0 <= v < num_views for v in self.source_of):
            raise ValueError("source_of must name one camera of 0..%d per view, got %r" % (num_views - 1, source_of))
        if world <= num_views:
            if num_views % world:
                raise ValueError("views (%d) must be a multiple of world size (%d)" % (num_views, world))
            self.group_ranks = list(range(world))
            self.cams_of = {r: [v for v in range(num_views) if v % world == r] for r in range(world)}
            self.slice_id, self.num_slices = 0, 1
        else:
            if world % num_views:
                raise ValueError("world size (%d) must be a multiple of views (%d)" % (world, num_views))
            self.slice_id, self.num_slices = rank // num_views, world // num_views
            first = self.slice_id * num_views
            self.group_ranks = list(range(first, first + num_views))
            self.cams_of = {first + v: [v] for v in range(num_views)}
        self.world, self.rank, self.V = world, rank, num_views
        self.my_cams = self.cams_of[rank]
        self.group = group
        self._own_group = None

    # ------------------------------------------------------------------ pairs
    def select_pairs(self, total_pairs: int, image: int, seed: int, rig: str = "ring"):
        """Projection matrices of this rank's pairs.  total_pairs = frames * V of
        ONE frame slice; every rank regenerates the same rig from `seed`.  rig: "ring" (any V) or a four-camera rig of
        synthetic.rig_cameras, paired by THIS exchange's source_of table."""
        frames = total_pairs // self.V
        if rig == "ring" and self.source_of == [(v + 1) % self.V for v in range(self.V)]:
            P_ref, P_src = syn.make_pairs(frames, self.V, image, seed=seed + self.slice_id, jitter=(0.05, 8.0))
            P_ref = P_ref.view(frames, self.V, 3, 4)
            P_src = P_src.view(frames, self.V, 3, 4)
        else:
            cams = torch.from_numpy(syn.rig_cameras(rig, frames, image, seed=seed + self.slice_id, jitter=(0.05, 8.0))).float()
            if cams.shape[1] != self.V:
                raise ValueError("rig %r has %d cameras, the exchange %d" % (rig, cams.shape[1], self.V))
            P_ref = cams
            P_src = torch.stack([cams[:, self.source_of[v]] for v in range(self.V)], 1)
        self.frames = frames
        ref = torch.cat([P_ref[:, v] for v in self.my_cams])            # camera-major
        src = torch.cat([P_src[:, v] for v in self.my_cams])
        return ref.contiguous(), src.contiguous()

    def source_location(self, cam: int):
        """(owner rank, index of that camera in the owner's camera list) of the
        source view of reference camera `cam` (source_of: multiview_h36m.py:231-238)."""
        s = self.source_of[cam]
        for r in self.group_ranks:
            if s in self.cams_of[r]:
                return r, self.cams_of[r].index(s)
        raise AssertionError

    # --------------------------------------------------------------- exchange
    def _pg(self):
        if self.group is not None or self.num_slices == 1:
            return self.group
        if self._own_group is None:
            # every rank must create every group, in the same order
            groups = [dist.new_group(list(range(s * self.V, (s + 1) * self.V))) for s in range(self.num_slices)]
            self._own_group = groups[self.slice_id]
        return self._own_group

    def _gather_flat(self, send: torch.Tensor, pg, async_op=False):
        """ONE all_gather_into_tensor of `send` over the view group: (G, *send.shape), no per-rank list copies."""
        g = len(self.group_ranks)
        flat = torch.empty((g * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        work = dist.all_gather_into_tensor(flat, send.contiguous(), group=pg, async_op=async_op)   # concatenation along dim 0
        return flat.view((g,) + tuple(send.shape)), work

    def gather_sources(self, own_maps: torch.Tensor) -> torch.Tensor:
        """own_maps: this rank's (len(my_cams)*frames, H, W, C) maps, camera-major.
        Returns the source maps of its pairs, same order, after ONE all-gather of
        the per-rank maps over the view group (a VIEW of the receive buffer when the rank owns one camera)."""
        pg = self._pg()
        recv, _ = self._gather_flat(own_maps, pg)
        f = own_maps.shape[0] // len(self.my_cams)
        chunks = []
        for cam in self.my_cams:
            owner, idx = self.source_location(cam)
            chunks.append(recv[self.group_ranks.index(owner), idx * f:(idx + 1) * f])
        return chunks[0] if len(chunks) == 1 else torch.cat(chunks)

    def gather_sources_chunked(self, own_maps: torch.Tensor, num_chunks: int):
        """Overlappable form of gather_sources: the frames of every camera are split in `num_chunks` ranges and
        each range is all-gathered as its own asynchronous collective (RCCL runs them on its stream).  Yields
        `(ranges, source_maps)` per chunk after waiting for THAT chunk only -- `ranges` = one (start, stop) slice of
        the rank's pair list per owned camera (contiguous frame ranges: the caller slices its reference maps and
        camera algebra with them, no index gather) -- so the caller's fused kernel on chunk i overlaps the transfer
        of chunks i+1.. (xGMI: a 128 MiB shard takes ~0.9 ms per link -- the same order as the kernel)."""
        pg = self._pg()
        ncam = len(self.my_cams)
        f = own_maps.shape[0] // ncam
        num_chunks = max(1, min(num_chunks, f))
        bounds = [(i * f) // num_chunks for i in range(num_chunks + 1)]
        per_cam = own_maps.view(ncam, f, *own_maps.shape[1:])
        inflight = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            send = per_cam[:, lo:hi]                                   # (ncam, hi-lo, ...): contiguous when ncam == 1
            recv, work = self._gather_flat(send, pg, async_op=True)
            inflight.append((lo, hi, recv, work))
        for lo, hi, recv, work in inflight:
            work.wait()
            chunks, ranges = [], []
            for ci, cam in enumerate(self.my_cams):
                owner, idx = self.source_location(cam)
                chunks.append(recv[self.group_ranks.index(owner), idx])
                ranges.append((ci * f + lo, ci * f + hi))
            yield ranges, (chunks[0] if len(chunks) == 1 else torch.cat(chunks))

    def _routes(self):
        """The pairing as an all-to-all: (send_order, send_split, recv_split, recv_slots) for moving every camera block
        from the rank that OWNS it to every rank one of whose reference cameras samples it (a block that two views share
        appears twice in send_order; a block nobody samples not at all).
          send_order : indices into my camera list, grouped by destination rank (group order), inside a destination in the
                       order of ITS reference cameras;   send_split / recv_split: blocks per rank;
          recv_slots : for the blocks as they arrive (source rank major), the index of MY reference camera they belong to."""
        send_order, send_split = [], []
        for q in self.group_ranks:
            mine = [self.source_location(cam)[1] for cam in self.cams_of[q] if self.source_location(cam)[0] == self.rank]
            send_order += mine
            send_split.append(len(mine))
        recv_split, recv_slots = [], []
        for o in self.group_ranks:
            slots = [ci for ci, cam in enumerate(self.my_cams) if self.source_location(cam)[0] == o]
            recv_split.append(len(slots))
            recv_slots += slots
        return send_order, send_split, recv_split, recv_slots

    def exchange_sources(self, own_maps: torch.Tensor) -> torch.Tensor:
        """Point-to-point form of gather_sources: ONE all_to_all_single in which every camera block goes only to the rank
        that samples it -- 1 x the bytes a rank consumes, where the all-gather stages G x (at BASELINE config 5, 1 GiB of
        maps per rank on 8 ranks, 8 GiB to use one).  Same result as gather_sources, bit for bit (copies only)."""
        pg = self._pg()
        ncam = len(self.my_cams)
        f = own_maps.shape[0] // ncam
        per = own_maps.reshape(ncam, f, -1)
        send_order, send_split, recv_split, recv_slots = self._routes()
        send = torch.cat([per[i] for i in send_order]) if send_order else per.new_zeros((0, per.shape[2]))
        recv = torch.empty((sum(recv_split) * f, per.shape[2]), dtype=own_maps.dtype, device=own_maps.device)
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=[n * f for n in recv_split],
                               input_split_sizes=[n * f for n in send_split], group=pg)
        if recv_slots == list(range(ncam)):                    # (arrival order = my camera order: the buffer is the result)
            return recv.view(own_maps.shape)
        out = torch.empty_like(per)
        for k, ci in enumerate(recv_slots):
            out[ci] = recv[k * f:(k + 1) * f]
        return out.view(own_maps.shape)

    def exchange_sources_chunked(self, own_maps: torch.Tensor, num_chunks: int):
        """Overlappable form of exchange_sources, the point-to-point counterpart of gather_sources_chunked: the frames of
        every camera are split in `num_chunks` ranges and each range travels as its own asynchronous all_to_all_single in
        which a camera block goes only to the rank that samples it (1 x the bytes, where the all-gather stages G x).  Yields
        `(ranges, source_maps)` per chunk exactly as gather_sources_chunked does -- the two are interchangeable."""
        pg = self._pg()
        ncam = len(self.my_cams)
        f = own_maps.shape[0] // ncam
        num_chunks = max(1, min(num_chunks, f))
        bounds = [(i * f) // num_chunks for i in range(num_chunks + 1)]
        rest = tuple(own_maps.shape[1:])
        per_cam = own_maps.view((ncam, f) + rest)
        send_order, send_split, recv_split, recv_slots = self._routes()
        inflight = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            n = hi - lo
            if len(send_order) == 1:
                send = per_cam[send_order[0], lo:hi]                       # contiguous: no staging copy
            else:
                send = torch.cat([per_cam[i, lo:hi] for i in send_order]) if send_order else own_maps.new_zeros((0,) + rest)
            recv = torch.empty((sum(recv_split) * n,) + rest, dtype=own_maps.dtype, device=own_maps.device)
            work = dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=[c * n for c in recv_split],
                                          input_split_sizes=[c * n for c in send_split], group=pg, async_op=True)
            inflight.append((lo, hi, recv, work))
        for lo, hi, recv, work in inflight:
            work.wait()
            n = hi - lo
            ranges = [(ci * f + lo, ci * f + hi) for ci in range(ncam)]
            if recv_slots == list(range(ncam)):                            # arrival order = my camera order
                yield ranges, recv
            else:
                blocks = [None] * ncam
                for k, ci in enumerate(recv_slots):
                    blocks[ci] = recv[k * n:(k + 1) * n]
                yield ranges, torch.cat(blocks)

    def scatter_source_grads(self, grad_src: torch.Tensor) -> torch.Tensor:
        """Backward of gather_sources: route d(source maps) back to the ranks that own those maps: one
        all_to_all_single in which a rank sends each camera's gradient block to its owner only -- 1x the data, where
        an all-gather (or a reduce-scatter over zero-padded slots) would move G x.  With the ring pairing every map is
        the source of exactly one reference camera and this is a permutation; with the nearest-camera rule a map may be
        sampled by several views (their blocks are SUMMED, in arrival order = group-rank order, then camera order: fixed,
        so the sum is reproducible) or by none (zeros)."""
        pg = self._pg()
        ncam = len(self.my_cams)
        f = grad_src.shape[0] // ncam
        per = grad_src.reshape(ncam, f, -1)
        width = per.shape[2]
        # send side: my blocks grouped by destination rank (group order), camera order inside
        send_blocks, send_split = [], []
        for r in self.group_ranks:
            mine = [ci for ci, cam in enumerate(self.my_cams) if self.source_location(cam)[0] == r]
            send_blocks += [per[ci] for ci in mine]
            send_split.append(len(mine) * f)
        # receive side: from rank q, the blocks of q's cameras whose source I own, in q's camera order
        recv_split, recv_slots = [], []
        for q in self.group_ranks:
            slots = [self.source_location(cam)[1] for cam in self.cams_of[q] if self.source_location(cam)[0] == self.rank]
            recv_split.append(len(slots) * f)
            recv_slots += slots
        send = torch.cat(send_blocks) if send_blocks else per.new_zeros((0, width))
        recv = torch.empty((sum(recv_split), width), dtype=grad_src.dtype, device=grad_src.device)
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_split, input_split_sizes=send_split, group=pg)
        out = torch.zeros_like(per)
        for k, idx in enumerate(recv_slots):
            out[idx] += recv[k * f:(k + 1) * f]
        return out.view_as(grad_src)


# ---------------------------------------------------------------------------------------------------------------
# The exchange as a differentiable step: what replaces the reference's nn.DataParallel scatter / gather of
# `other_features` (modeling/model.py:44,246-247) when the views are sharded one camera per GPU
# ---------------------------------------------------------------------------------------------------------------
class _ShardedSources(torch.autograd.Function):
    """own maps (this rank's cameras, camera-major) -> the source maps of this rank's pairs.  Forward: ONE all-gather
    over the view group (`gather_sources`); backward: ONE all-to-all that returns d(source maps) to the ranks that own
    those maps (`scatter_source_grads`: summed where several views sample one map)."""

    @staticmethod
    def forward(ctx, own_maps, exchange, p2p=False):
        ctx.exchange = exchange
        if p2p:
            return exchange.exchange_sources(own_maps.contiguous())
        out = exchange.gather_sources(own_maps.contiguous())
        # With one camera per rank the result is a slice of the G x receive buffer of the all-gather, and the fused op saves
        # it for its backward: as a view it would keep the WHOLE buffer alive through forward and backward (8 GiB held to use
        # 1 GiB at BASELINE config 5 on 8 ranks).  Copy the slice out and let the buffer go; with more cameras per rank
        # gather_sources has already concatenated into a tensor of its own.  (p2p stages 1 x and needs no copy.)
        base = out._base
        if base is not None and base.numel() > out.numel():
            out = out.clone()
        return out

    @staticmethod
    def backward(ctx, grad):
        return ctx.exchange.scatter_source_grads(grad.contiguous()), None, None


def sharded_sources(own_maps: torch.Tensor, exchange: "ViewShardExchange", num_chunks: int = 1, p2p: bool = False) -> torch.Tensor:
    """Differentiable `gather_sources`: (len(my_cams) * frames, ...) own maps -> same shape, the source map of every
    pair of this rank.  num_chunks > 1 splits the frames of every camera into ranges and exchanges range by range
    (each range its own collective in the forward and in the backward), so that on the GPUs the fused kernel of
    range i can run while range i + 1 is on the links.  p2p: the forward as one all-to-all (`exchange_sources`: every
    block to the one rank that samples it) instead of the all-gather BASELINE.json's north star names -- same result,
    1 / G of the bytes staged; the backward is that all-to-all reversed either way."""
    ncam = len(exchange.my_cams)
    f = own_maps.shape[0] // ncam
    num_chunks = max(1, min(int(num_chunks), f))
    if num_chunks == 1:
        return _ShardedSources.apply(own_maps, exchange, p2p)
    per_cam = own_maps.reshape((ncam, f) + tuple(own_maps.shape[1:]))
    bounds = [(i * f) // num_chunks for i in range(num_chunks + 1)]
    parts = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        chunk = per_cam[:, lo:hi].reshape((ncam * (hi - lo),) + tuple(own_maps.shape[1:]))
        parts.append(_ShardedSources.apply(chunk, exchange, p2p).reshape((ncam, hi - lo) + tuple(own_maps.shape[1:])))
    return torch.cat(parts, 1).reshape(own_maps.shape)


def allreduce_gradients(module: torch.nn.Module, group=None, bucket_bytes: int = 64 << 20, average: bool = False):
    """Sum (or average) the parameter gradients of `module` over the process group in flat buckets -- the weight-gradient
    all-reduce of the shared network (EPIPOLAR.SHARE_WEIGHTS) that closes a sharded training step.  (torch's
    DistributedDataParallel does the same, overlapped with the backward; this is the explicit form for loops that
    scale the loss themselves.)  Parameters without a gradient on this rank contribute zeros."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    params = [p for p in module.parameters() if p.requires_grad]
    world = dist.get_world_size(group)
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, group=group)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
        bucket, size = [], 0

    for p in params:
        bucket.append(p)
        size += p.numel() * p.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


# ---------------------------------------------------------------------------------------------------------------
# Cross-rank coupling of the layer in TRAINING: the batch statistics of the z-epilogue's BN
# ---------------------------------------------------------------------------------------------------------------
class _SyncBNFunction(torch.autograd.Function):
    """Batch norm over (N,H,W) of ALL ranks: per-channel sum / sum of squares / count go through one all-reduce in the
    forward, sum(dy) / sum(dy * xhat) through one in the backward (what the reference's SynchronizedBatchNorm2d does
    between DataParallel replicas, modeling/sync_batchnorm/batchnorm.py:114-122, here between processes)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        c = x.shape[1]
        xf = x.float()
        stats = torch.cat([xf.sum((0, 2, 3)), (xf * xf).sum((0, 2, 3)), xf.new_tensor([x.numel() / c])])
        dist.all_reduce(stats, group=group)
        count = stats[-1]
        mean = stats[:c] / count
        var = (stats[c:2 * c] / count - mean * mean).clamp_min_(0)
        invstd = torch.rsqrt(var + eps)
        xhat = (xf - mean.view(1, c, 1, 1)) * invstd.view(1, c, 1, 1)
        ctx.save_for_backward(xhat, invstd, weight)
        ctx.group, ctx.count = group, count
        y = xhat * weight.view(1, c, 1, 1) + bias.view(1, c, 1, 1)
        ctx.mark_non_differentiable(mean, var)
        return y.to(x.dtype), mean, var, count

    @staticmethod
    def backward(ctx, dy, _dm, _dv, _dc):
        xhat, invstd, weight = ctx.saved_tensors
        c = dy.shape[1]
        dyf = dy.float()
        local = torch.cat([dyf.sum((0, 2, 3)), (dyf * xhat).sum((0, 2, 3))])
        dweight, dbias = local[c:].clone(), local[:c].clone()             # parameter grads stay local (DDP sums them)
        dist.all_reduce(local, group=ctx.group)
        mean_dy = (local[:c] / ctx.count).view(1, c, 1, 1)
        mean_dy_xhat = (local[c:] / ctx.count).view(1, c, 1, 1)
        dx = (dyf - mean_dy - xhat * mean_dy_xhat) * (invstd * weight).view(1, c, 1, 1)
        return dx.to(dy.dtype), dweight, dbias, None, None


class SyncBatchNorm2d(torch.nn.BatchNorm2d):
    """Drop-in for the layer's `bn` (same parameters / buffers / state_dict keys) whose TRAINING statistics span the
    process group -- cfg.BACKBONE.SYNC_BN (configs/epipolar/keypoint_h36m_resnet152_384_pretrained_8gpu.yaml:21;
    the reference converts with convert_model at modeling/model.py:56-58).  Works on any torch.distributed backend
    (RCCL on the GPUs; gloo in the CPU tests), falls back to plain batch norm outside a process group or in eval."""

    process_group = None

    def forward(self, x):
        if not (self.training and dist.is_available() and dist.is_initialized() and
                dist.get_world_size(self.process_group) > 1):
            return super().forward(x)
        y, mean, var, count = _SyncBNFunction.apply(x, self.weight, self.bias, self.eps, self.process_group)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                unbiased = var * (count / (count - 1).clamp_min(1))
                self.running_mean.mul_(1 - m).add_(mean.to(self.running_mean.dtype), alpha=m)
                self.running_var.mul_(1 - m).add_(unbiased.to(self.running_var.dtype), alpha=m)
        return y


def convert_sync_batchnorm(module: torch.nn.Module, process_group=None) -> torch.nn.Module:
    """Replace every BatchNorm2d (incl. the layer's zeroinitBN) below `module` by SyncBatchNorm2d, in place of the
    reference's `convert_model` (sync_batchnorm/batchnorm.py:379-386).  Parameters and buffers are shared, so
    checkpoints keep loading."""
    for name, child in list(module.named_children()):
        if isinstance(child, torch.nn.BatchNorm2d) and not isinstance(child, SyncBatchNorm2d):
            new = SyncBatchNorm2d(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
            new.process_group = process_group
            if child.affine:
                new.weight, new.bias = child.weight, child.bias
            if child.track_running_stats:
                new.running_mean, new.running_var = child.running_mean, child.running_var
                new.num_batches_tracked = child.num_batches_tracked
            new.train(child.training)
            setattr(module, name, new)
        else:
            convert_sync_batchnorm(child, process_group)
    return module
