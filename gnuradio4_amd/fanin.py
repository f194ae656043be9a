"""Multi-GPU host logic: independent SDR-channel sub-graphs (one per rank) and the fan-in combiner edge.

The flowgraph shards only across independent branches: channel c -> rank c mod world.  The one exchange step is the
combiner `gr::blocks::math::Add<float>` with n_inputs = channels (blocks/math/.../Math.hpp:73-108) that sums frame-aligned |X|^2
vectors; across GPUs every rank ends up with 1/world of the frames of the sum (never a reduce-to-root): an RCCL reduce_scatter, or an
all_to_all of the shards (one xGMI link per peer) folded in rank order -- see fan_in_sum.  Backend "nccl" is RCCL on ROCm; "gloo" (CPU tests,
several ranks on one GPU) is the functional stand-in, staged through the host.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def channel_plan(n_channels: int, world: int) -> List[List[int]]:
    """channels owned by each rank (1/2/4/8 GPUs -> 8/4/2/1 channels per device for the 8-channel graph)."""
    if n_channels < 1 or world < 1:
        raise ValueError("n_channels and world must be >= 1")
    return [[c for c in range(n_channels) if c % world == r] for r in range(world)]


def shard_frames(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """[lo, hi) frame range of the reduced spectra that `rank` owns after the reduce_scatter."""
    if n_frames % world:
        raise ValueError(f"n_frames={n_frames} must be a multiple of world={world} (launch sizes are chosen accordingly)")
    per = n_frames // world
    return rank * per, (rank + 1) * per


def local_sum(channels: List[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """left fold over the channels a rank owns -- the order of MathOpMultiPortImpl::processBulk (Math.hpp:100-107).
    Device tensors take the library's one-pass n-ary kernel (gr4hip_math_nary); host tensors (CPU tests) the same fold in torch."""
    if channels[0].is_cuda:
        from .blocks import math_nary
        return math_nary("Add", channels, out=out)
    acc = channels[0].clone() if out is None else out.copy_(channels[0])
    for c in channels[1:]:
        acc += c
    return acc


class Communicator:
    """RCCL communicator owned by the kernel library (gr4hip_fanin_*, include/gr4hip.h): the collectives of the fan-in edge are queued through the C-ABI on the
    caller's HIP stream -- the entry points the C++ engine (gr4/hip.hpp: FanInRun) uses -- and torch.distributed only ships the 128-byte id from rank 0 to
    the others when the communicator is made (any backend).  One rank per GPU: RCCL refuses two ranks on one device."""

    def __init__(self, group=None):
        import ctypes as C
        from . import capi
        self._capi, self._C = capi, C
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        L = capi.lib()
        box = [None]
        if self.rank == 0:
            ident = (C.c_char * 128)()
            capi.check(L.gr4hip_fanin_unique_id(ident), "fanin_unique_id")
            box[0] = bytes(ident.raw)
        dist.broadcast_object_list(box, src=0, group=group)
        self._h = C.c_void_p()
        capi.check(L.gr4hip_fanin_create(C.byref(self._h), box[0], self.rank, self.world), "fanin_create")  # collective over all ranks

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._capi.lib().gr4hip_fanin_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fan_in_sum(self, local: torch.Tensor, out: torch.Tensor, algo: str = "reduce_scatter", recv: Optional[torch.Tensor] = None) -> torch.Tensor:
        L, st = self._capi.lib(), torch.cuda.current_stream().cuda_stream
        if algo == "reduce_scatter":
            self._capi.check(L.gr4hip_fanin_reduce_scatter_sum_f32(self._h, local.data_ptr(), out.data_ptr(), out.numel(), st), "fanin_reduce_scatter")
        elif algo == "all_to_all":
            if recv is None:
                recv = torch.empty_like(local)
            self._capi.check(L.gr4hip_fanin_all_to_all_sum_f32(self._h, local.data_ptr(), recv.data_ptr(), out.data_ptr(), out.numel(), st), "fanin_all_to_all")
        else:
            raise ValueError(f"unknown fan-in algorithm '{algo}'")
        return out


def fan_in_sum(local: torch.Tensor, out: Optional[torch.Tensor] = None, group=None, async_op: bool = False, algo: str = "reduce_scatter",
               recv: Optional[torch.Tensor] = None, comm: Optional[Communicator] = None):
    """Sum `local` ([frames, fft_size] mag2 of this rank's channels) over all ranks; every rank keeps its shard of the frames.
    Returns (shard, work-or-None).  Two ways to move the same (N-1)/N of every rank's partial sum:
      "reduce_scatter"  one RCCL reduce_scatter(sum)
      "all_to_all"      every rank sends shard j of its partial sum straight to rank j (all_to_all_single: one xGMI link per peer, all of them busy at once --
                        xGMI is point-to-point, not a switch) and folds the `world` shards it received in RANK ORDER with the library's n-ary Add: the same
                        left fold on every rank, so the result does not depend on the collective's internal reduction order.  `recv`: scratch of local's shape.
    Which one is faster is a property of the node's RCCL; bench.py measures both before the timed region and keeps the faster (--fanin-algo auto).
    comm: a Communicator -- the collectives go through the library's C entry points (gr4hip_fanin_*) instead of torch.distributed's."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_frames(local.shape[0], world, rank)
    if out is None:
        out = torch.empty((hi - lo,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if comm is not None and local.is_cuda:  # the library's own RCCL communicator (no torch in the data path), asynchronous on the current stream
        if not (local.is_contiguous() and out.is_contiguous()):
            raise ValueError("fan_in_sum: contiguous tensors expected")
        return comm.fan_in_sum(local, out, algo=algo, recv=recv), None
    gloo = dist.get_backend(group) == "gloo"  # functional path (CPU tests, several ranks on one GPU): same result, staged through the host
    if algo == "all_to_all":
        if recv is None:
            recv = torch.empty_like(local)
        if gloo:
            src, dst = local.detach().to("cpu", copy=True).contiguous(), torch.empty(local.shape, dtype=local.dtype)
            dist.all_to_all_single(dst.reshape(-1), src.reshape(-1), group=group)
            recv.copy_(dst)
        else:
            dist.all_to_all_single(recv.reshape(-1), local.reshape(-1), group=group)
        per = hi - lo
        local_sum([recv[j * per:(j + 1) * per] for j in range(world)], out=out)
        return out, None
    if algo != "reduce_scatter":
        raise ValueError(f"unknown fan-in algorithm '{algo}'")
    if gloo:
        tmp = local.detach().to("cpu", copy=True)
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group, async_op=False)
        out.copy_(tmp[lo:hi])
        return out, None
    work = dist.reduce_scatter_tensor(out.reshape(-1), local.reshape(-1), op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return out, (work if async_op else None)
