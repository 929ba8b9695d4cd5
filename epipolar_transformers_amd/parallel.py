"""Multi-GPU partitions of the path.  One process per GPU,
`torch.distributed` ("nccl" is RCCL on ROCm; tests use "gloo" on CPU).

Pairs are independent units, so the natural partition is collective-free:
  * frames-DP  -- a rank owns whole frames (all V views local): no exchange.
The north-star partition shards by camera and has ONE real exchange step:
  * view-sharded -- rank r owns camera(s) v == r (mod G) for its frames; the
    source map of pair (frame, v) is the map of camera (v+1) mod V of the same
    frame, produced on another rank, so the per-rank feature maps are
    all-gathered (RCCL over xGMI) before the fused kernel consumes them.
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

    def __init__(self, world: int, rank: int, num_views: int, group=None):
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
    def select_pairs(self, total_pairs: int, image: int, seed: int):
        """Projection matrices of this rank's pairs.  total_pairs = frames * V of
        ONE frame slice; every rank regenerates the same rig from `seed`."""
        frames = total_pairs // self.V
        P_ref, P_src = syn.make_pairs(frames, self.V, image, seed=seed + self.slice_id, jitter=(0.05, 8.0))
        P_ref = P_ref.view(frames, self.V, 3, 4)
        P_src = P_src.view(frames, self.V, 3, 4)
        self.frames = frames
        ref = torch.cat([P_ref[:, v] for v in self.my_cams])            # camera-major
        src = torch.cat([P_src[:, v] for v in self.my_cams])
        return ref.contiguous(), src.contiguous()

    def source_location(self, cam: int):
        """(owner rank, index of that camera in the owner's camera list) of the
        source view of reference camera `cam` (ring neighbour, multiview_h36m.py:231-238)."""
        s = (cam + 1) % self.V
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

    def gather_sources(self, own_maps: torch.Tensor) -> torch.Tensor:
        """own_maps: this rank's (len(my_cams)*frames, H, W, C) maps, camera-major.
        Returns the source maps of its pairs, same order, after ONE all-gather of
        the per-rank maps over the view group."""
        pg = self._pg()
        parts = [torch.empty_like(own_maps) for _ in self.group_ranks]
        dist.all_gather(parts, own_maps.contiguous(), group=pg)
        f = own_maps.shape[0] // len(self.my_cams)
        chunks = []
        for cam in self.my_cams:
            owner, idx = self.source_location(cam)
            chunks.append(parts[self.group_ranks.index(owner)][idx * f:(idx + 1) * f])
        return chunks[0] if len(chunks) == 1 else torch.cat(chunks)

    def gather_sources_chunked(self, own_maps: torch.Tensor, num_chunks: int):
        """Overlappable form of gather_sources: the frames of every camera are split in `num_chunks` ranges and
        each range is all-gathered as its own asynchronous collective (RCCL runs them on its stream).  Yields
        `(pair_index_tensor, source_maps)` per chunk after waiting for THAT chunk only, so the caller's fused
        kernel on chunk i overlaps the transfer of chunks i+1.. (xGMI: a 128 MiB shard takes ~0.9 ms per
        link -- the same order as the kernel)."""
        pg = self._pg()
        ncam = len(self.my_cams)
        f = own_maps.shape[0] // ncam
        num_chunks = max(1, min(num_chunks, f))
        bounds = [(i * f) // num_chunks for i in range(num_chunks + 1)]
        per_cam = own_maps.view(ncam, f, *own_maps.shape[1:])
        inflight = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            send = per_cam[:, lo:hi].contiguous()                       # (ncam, hi-lo, ...)
            parts = [torch.empty_like(send) for _ in self.group_ranks]
            work = dist.all_gather(parts, send, group=pg, async_op=True)
            inflight.append((lo, hi, parts, work))
        for lo, hi, parts, work in inflight:
            work.wait()
            chunks, index = [], []
            for ci, cam in enumerate(self.my_cams):
                owner, idx = self.source_location(cam)
                chunks.append(parts[self.group_ranks.index(owner)][idx])
                index.append(torch.arange(ci * f + lo, ci * f + hi))
            yield torch.cat(index), (chunks[0] if len(chunks) == 1 else torch.cat(chunks))

    def scatter_source_grads(self, grad_src: torch.Tensor) -> torch.Tensor:
        """Backward of gather_sources: route d(source maps) back to the ranks that
        own those maps and sum (reduce-scatter semantics, done as all-gather + local
        sum so it also runs on gloo)."""
        pg = self._pg()
        parts = [torch.empty_like(grad_src) for _ in self.group_ranks]
        dist.all_gather(parts, grad_src.contiguous(), group=pg)
        f = grad_src.shape[0] // len(self.my_cams)
        out = torch.zeros_like(grad_src)
        for gi, r in enumerate(self.group_ranks):
            for ci, cam in enumerate(self.cams_of[r]):
                owner, idx = self.source_location(cam)
                if owner == self.rank:
                    out[idx * f:(idx + 1) * f] += parts[gi][ci * f:(ci + 1) * f]
        return out
