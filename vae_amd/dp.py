"""Data-parallel gradient exchange: bucketed all-reduce of the flat gradient buffer over
RCCL (torch.distributed backend "nccl" on ROCm), overlapped with the backward pass.

The reference is single-process (no torch.distributed call anywhere); this is the batch
data-parallel step the MI355X build adds.  Parameters and optimizer state are replicated (same
start, same averaged gradients).  BatchNorm batch statistics stay per rank (the single-process
semantics at the local batch), so the running statistics -- broadcast once at start -- drift
apart from rank to rank afterwards; `sync_bn_stats` averages them (call it before an eval pass
or a checkpoint if one set of statistics for all ranks is wanted).  The backward pass finishes layers in three stages (decoder + latent
heads, conv4+conv3, conv2+conv1); each stage's slice of the flat gradient buffer is contiguous,
so its buckets are plain views (no copy) and are all-reduced on the process group's own stream
while the next stage computes."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

# RCCL shares device buffers between the ranks of a node through dmabuf handles; with the legacy IPC mode the first
# collective fails with "hipIpcGetMemHandle: invalid argument" on hosts whose driver only supports dmabuf.  It is read
# when the HIP runtime starts, i.e. before the first GPU call of the process: import this module (or set the variable
# in the launcher's environment) before touching the GPU.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist


def init_process_group(backend: str = "nccl", local_rank: int | None = None, **kw):
    """torch.distributed.init_process_group for one process per GPU: picks the GPU from LOCAL_RANK (torchrun's
    variable), binds the RCCL communicator to it and returns (rank, world_size, device).  Rendezvous address and
    port come from MASTER_ADDR / MASTER_PORT as usual (use 127.0.0.1 on a single node)."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=dev, **kw)
    else:
        dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1)) if torch.cuda.is_available() else torch.device("cpu")
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size(), dev


def stage_ranges(offsets: Sequence[Tuple[int, int]], total: int) -> List[Tuple[int, int]]:
    """[begin, end) of the flat buffer finished by backward stage 0, 1, 2.
    offsets: (offset, numel) of the 52 parameters in state_dict order:
    0-15 conv1+conv2, 16-31 conv3+conv4, 32-51 latent heads + decoder."""
    s1, s0 = offsets[16][0], offsets[32][0]
    return [(s0, total), (s1, s0), (0, s1)]


def split_buckets(begin: int, end: int, bucket_elems: int) -> List[Tuple[int, int]]:
    out = []
    a = begin
    while a < end:
        b = min(end, a + bucket_elems)
        out.append((a, b))
        a = b
    return out


class BucketedReducer:
    """all-reduce (mean) of `flat_grads` in per-stage buckets."""

    def __init__(self, flat_grads: torch.Tensor, ranges: Sequence[Tuple[int, int]], bucket_bytes: int = 16 << 20,
                 group=None, side_stream: bool = False, merge_first: bool = False):
        self.flat = flat_grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        ranges = [tuple(r) for r in ranges]
        # merge_first: the first stage's range is exchanged together with the second's when the two are adjacent in
        # the flat buffer (one collective fewer per step; it then runs under the last stage instead of the second).
        # On one rank it measures the same as three collectives, so it is off unless asked for.
        if merge_first and len(ranges) >= 3 and ranges[1][1] == ranges[0][0]:
            ranges = [(0, 0), (ranges[1][0], ranges[0][1])] + ranges[2:]
        self.buckets = [split_buckets(a, b, max(1, bucket_bytes // flat_grads.element_size())) for a, b in ranges]
        self.cuda = flat_grads.is_cuda
        # torch's NCCL process group runs a collective on its own stream, ordered after the stream it was issued from;
        # side_stream=False issues from the compute stream itself (one hand-over fewer per stage)
        self.comm_stream = torch.cuda.Stream() if self.cuda and side_stream else None
        self.handles = []
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self.avg_op = backend == "nccl"          # RCCL has ncclAvg; gloo needs SUM then a divide

    def stage_done(self, stage: int):
        """call right after the kernels of backward stage `stage` were enqueued."""
        if self.world == 1 or not self.buckets[stage]:
            return
        op = dist.ReduceOp.AVG if self.avg_op else dist.ReduceOp.SUM
        if self.cuda and self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                for a, b in self.buckets[stage]:
                    self.handles.append((dist.all_reduce(self.flat[a:b], op=op, group=self.group, async_op=True), a, b))
        else:
            for a, b in self.buckets[stage]:
                self.handles.append((dist.all_reduce(self.flat[a:b], op=op, group=self.group, async_op=True), a, b))

    def wait(self):
        """make the current stream wait for every outstanding bucket (and finish the mean)."""
        for h, a, b in self.handles:
            h.wait()
            if not self.avg_op:
                self.flat[a:b].div_(self.world)
        self.handles = []


def sync_bn_stats(model, group=None):
    """running_mean / running_var of every BatchNorm := mean over ranks (num_batches_tracked is already equal:
    every rank counts the same steps)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(model._bn_stats, group=group)
    model._bn_stats.div_(dist.get_world_size(group))


def broadcast_state(model, src: int = 0, group=None):
    """same initial parameters / running stats on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.broadcast(model.flat_params, src, group=group)
    dist.broadcast(model._bn_stats, src, group=group)
    dist.broadcast(model._nbt, src, group=group)
    model.repack_weights()
