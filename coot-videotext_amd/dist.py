"""
Data-parallel training of the retrieval path: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference only has single-process nn.DataParallel (nntrainer/trainer_base.py:126-129: encoders scattered,
outputs gathered to GPU 0, loss on the full batch).  Same semantics here, restated for multi-process DP
(SURVEY 8e):
  * videos are sharded by rank; all encoder work and the cycle-consistency loss are per video;
  * ONE packed all-gather of the six embedding sets per step; every rank evaluates the contrastive loss on the
    FULL gathered batch (mean over the global N^2, identical numerics to one GPU) and keeps the gradient of
    its own rows — no second collective for the embedding gradients;
  * global max clips/sentences per video (all-reduce MAX) so every rank pads to the same Cmax — required for
    reference-exact avg_special pooling;
  * parameter gradients: all-reduce(SUM) of the four flat gradient arenas (one collective per network,
    issued on a side stream as soon as backward has produced them).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class DirectRccl:
    """The step's collectives as direct RCCL calls on the CALLER'S HIP stream (ncclAllGather / ncclAllReduce of the librccl.so the
    process already has loaded through torch), on a communicator of the same ranks as the torch group.  torch.distributed's wrapper
    around the same two calls costs ~7 us of device time per collective even with one rank (work-tracking events around it, each a
    barrier packet on the stream: 1.229 -> 1.208 ms per step with the step's four collectives stubbed out, round 6), and a step has
    four.  The torch group stays what it is for everything off the step's device path (setup, shape exchange, the agreement below).
    Creation is collective and its outcome is AGREED over the torch group: either every rank gets a communicator or none does
    (DataParallelContext then keeps the torch.distributed calls and says so once)."""

    FLOAT32, SUM = 7, 0  # ncclFloat32, ncclSum (rccl.h)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_ubyte * 128)]  # NCCL_UNIQUE_ID_BYTES (c_ubyte: a c_char array reads back cut at the first NUL)

    def __init__(self, group=None, lib_path: Optional[str] = None):
        self.comm = None
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # (the agreement tensors live where the torch group's backend wants them: a gloo group in the CPU tests exercises this
        # constructor's control flow up to the communicator that cannot exist without a GPU)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        err = None
        try:
            lib = C.CDLL(lib_path or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
            lib.ncclGetUniqueId.argtypes = [C.POINTER(self.UniqueId)]
            lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, self.UniqueId, C.c_int]
            lib.ncclCommDestroy.argtypes = [C.c_void_p]
            lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
            lib.ncclGetErrorString.argtypes, lib.ncclGetErrorString.restype = [C.c_int], C.c_char_p
            self.lib = lib
        except (OSError, AttributeError) as e:
            err, self.lib = e, None
        if not self._agree(self.lib is not None, dev):  # a rank without the library must not leave the others inside the init
            self._report(f"librccl.so not loadable on every rank ({err})")
            return
        # rank 0's id to everyone through the torch group (bytes in a one-element object list)
        uid = self.UniqueId()
        if self.rank == 0:
            self._check(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [bytes(uid.internal) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        assert isinstance(box[0], bytes) and len(box[0]) == 128
        C.memmove(C.byref(uid), box[0], 128)
        comm = C.c_void_p()
        rc = lib.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank)  # collective
        if not self._agree(rc == 0, dev):
            if rc == 0:
                lib.ncclCommDestroy(comm)
            self._report(f"ncclCommInitRank failed on a rank (here: {self._err(rc)})")
            return
        self.comm = comm

    def _agree(self, ok: bool, dev) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item()) == 1

    def _report(self, why: str) -> None:
        if self.rank == 0:
            print(f"coot dist: no direct RCCL communicator ({why}); the step's collectives go through torch.distributed", flush=True)

    def _err(self, rc: int) -> str:
        return "ok" if rc == 0 else (self.lib.ncclGetErrorString(rc) or b"?").decode()

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what}: {self._err(rc)}")

    def all_gather(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        assert send.dtype == torch.float32 and recv.dtype == torch.float32 and recv.numel() == self.world * send.numel()
        assert send.is_contiguous() and recv.is_contiguous() and send.is_cuda and recv.is_cuda
        self._check(self.lib.ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), self.FLOAT32, self.comm,
                                           torch.cuda.current_stream().cuda_stream), "ncclAllGather")

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        self._check(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), self.FLOAT32, self.SUM, self.comm,
                                           torch.cuda.current_stream().cuda_stream), "ncclAllReduce")

    def close(self) -> None:
        if self.comm is not None:
            torch.cuda.synchronize()
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


class _GatherRows(torch.autograd.Function):
    """all-gather row blocks of possibly different heights; backward = own slice of the incoming gradient
    (every rank computes the same full loss, so no reduction is needed)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, counts: List[int], rank: int, group):
        world = len(counts)
        maxc = max(counts)
        pad = x.new_zeros((maxc,) + tuple(x.shape[1:]))
        pad[: x.shape[0]] = x
        out = x.new_empty((world * maxc,) + tuple(x.shape[1:]))
        _all_gather_into(out, pad.contiguous(), group)
        ctx.counts, ctx.rank, ctx.maxc = counts, rank, maxc
        if all(c == maxc for c in counts):
            return out
        idx = torch.cat([torch.arange(r * maxc, r * maxc + c, device=x.device) for r, c in enumerate(counts)])
        return out.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        start = sum(ctx.counts[: ctx.rank])
        return g[start:start + ctx.counts[ctx.rank]].contiguous(), None, None, None


def gather_rows(x: torch.Tensor, counts: List[int], rank: int, group=None) -> torch.Tensor:
    return _GatherRows.apply(x, counts, rank, group)


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """Backend "gloo" with device tensors (several ranks sharing ONE GPU in tests/test_gpu_dp_procs.py — RCCL refuses two ranks on
    one device): the collective runs on a host copy.  With "nccl" (RCCL) tensors never leave the device."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather_into(out: torch.Tensor, x: torch.Tensor, group=None) -> None:
    if _host_staged(x, group):
        xc = x.cpu()  # (synchronises the current stream: the rows are final)
        oc = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(oc, xc, group=group)
        out.copy_(oc)
    else:
        dist.all_gather_into_tensor(out, x, group=group)


def _all_reduce(t: torch.Tensor, op, group=None) -> None:
    if _host_staged(t, group):
        c = t.cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op, group=group)


def gather_rows_nograd(x: torch.Tensor, counts: List[int], group=None) -> torch.Tensor:
    """The collective of gather_rows without autograd (native step: gradients are sliced by the caller)."""
    world, maxc = len(counts), max(counts)
    if x.shape[0] != maxc:
        pad = x.new_zeros((maxc,) + tuple(x.shape[1:]))
        pad[: x.shape[0]] = x
        x = pad
    out = x.new_empty((world * maxc,) + tuple(x.shape[1:]))
    _all_gather_into(out, x.contiguous(), group)
    if all(c == maxc for c in counts):
        return out
    idx = torch.cat([torch.arange(r * maxc, r * maxc + c, device=x.device) for r, c in enumerate(counts)])
    return out.index_select(0, idx)


class DataParallelContext:
    """Holds rank/world and implements the collectives of one training step."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def host_group(self):
        """A gloo group over the same ranks for the few integers a step exchanges about its batch: a collective on HOST tensors
        does not touch the device queue, so the host keeps running ahead of the GPU (the same integers through RCCL cost a
        device round trip + a stream synchronisation per step: the pipeline drains).  Created collectively on first use."""
        g = getattr(self, "_host_group", None)
        if g is None:
            if dist.get_backend(self.group) == "gloo":
                g = self.group if self.group is not None else dist.group.WORLD
            else:
                err = None
                try:
                    ranks = dist.get_process_group_ranks(self.group if self.group is not None else dist.group.WORLD)
                    g = dist.new_group(ranks=ranks, backend="gloo")
                except Exception as e:  # (e.g. no resolvable host name for gloo's TCP transport)
                    err, g = e, False
                # The outcome is agreed COLLECTIVELY over the main group: a rank that fell back on its own would enter the device
                # all-gather of exchange_shapes while the others enter the gloo one — a deadlock instead of an error.  The side group
                # is used only if it exists on every rank.
                ok = torch.tensor([0 if g is False else 1], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
                if int(ok.item()) == 0:
                    if g is not False:
                        dist.destroy_process_group(g)
                    print(f"coot dist: no gloo side group on every rank ({err}); batch shapes go through the device collectives "
                          "(one stream sync per step)")
                    g = False
            self._host_group = g
        return g

    def exchange_shapes(self, values: List[int]) -> List[List[int]]:
        """All-gather of a short list of host integers (this rank's batch shape: videos, clips, max clips / sentences per
        video): result[r] = rank r's list.  ONE host collective per batch instead of three device ones."""
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        out = torch.empty(self.world * t.numel(), dtype=torch.int64)
        g = self.host_group()
        if g is False:  # device collective of the main group
            dev = torch.device("cuda", torch.cuda.current_device())
            od = out.to(dev)
            dist.all_gather_into_tensor(od, t.to(dev), group=self.group)
            out = od.cpu()
        else:
            dist.all_gather_into_tensor(out, t, group=g)
        return out.view(self.world, t.numel()).tolist()

    def global_max(self, value: int, device) -> int:
        t = torch.tensor([int(value)], dtype=torch.int32, device=device)
        _all_reduce(t, dist.ReduceOp.MAX, self.group)
        return int(t.item())

    def global_max_pair(self, a: int, b: int, device) -> Tuple[int, int]:
        t = torch.tensor([int(a), int(b)], dtype=torch.int32, device=device)
        _all_reduce(t, dist.ReduceOp.MAX, self.group)
        v = t.tolist()
        return int(v[0]), int(v[1])

    def global_counts(self, n: int, device) -> List[int]:
        t = torch.tensor([int(n)], dtype=torch.int64, device=device)
        out = torch.empty(self.world, dtype=torch.int64, device=device)
        _all_gather_into(out, t, self.group)
        return [int(v) for v in out.tolist()]

    def gather_rows_nograd(self, x: torch.Tensor, counts: List[int]) -> torch.Tensor:
        """All-gather of row blocks (counts[r] rows from rank r), no autograd: the native step slices gradients itself."""
        return gather_rows_nograd(x, counts, self.group)

    def gather_block(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        """ONE all-gather of equally sized blocks: recv [world * n] <- every rank's send [n] in rank order (the native step lays
        its six embedding sets out in one block and the loss reads the gathered blocks in place)."""
        r = self._direct(send)
        if r is not None:
            r.all_gather(recv, send)
        else:
            _all_gather_into(recv, send, self.group)

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        """In-place sum over ranks on the current stream."""
        r = self._direct(t)
        if r is not None:
            r.all_reduce_sum(t)
        else:
            _all_reduce(t, dist.ReduceOp.SUM, self.group)

    def _direct(self, t: torch.Tensor) -> Optional[DirectRccl]:
        """The direct RCCL communicator for the step's device collectives (backend "nccl", fp32 device tensors), created collectively
        at the first one — every rank's first is the step's embedding gather.  COOT_DP_COLLECTIVES=torch keeps torch.distributed's."""
        d = getattr(self, "_rccl", None)
        if d is None:
            use = (t.is_cuda and t.dtype == torch.float32 and dist.get_backend(self.group) == "nccl"
                   and os.environ.get("COOT_DP_COLLECTIVES", "direct") != "torch")
            d = self._rccl = DirectRccl(self.group) if use else False
            if d is not False and d.comm is None:
                d = self._rccl = False
        return d if d is not False else None

    def prepare_device_collectives(self) -> str:
        """Create the direct communicator NOW (collective) instead of at the first step's gather — e.g. while the caller has stdout
        redirected: RCCL prints its version banner there when a process creates its first communicator.  Returns the route."""
        if dist.get_backend(self.group) == "nccl":
            self._direct(torch.empty(1, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device())))
        return self.collectives_route()

    def collectives_route(self) -> str:
        """"direct" once the step's collectives run as direct RCCL calls, "torch" if they go through torch.distributed."""
        d = getattr(self, "_rccl", None)
        return "direct" if (d is not None and d is not False) else "torch"

    def close(self) -> None:
        d = getattr(self, "_rccl", None)
        if d is not None and d is not False:
            d.close()
        self._rccl = None

    def gather_embeddings(self, vis, txt, clip_counts: Optional[List[int]] = None, vid_counts: Optional[List[int]] = None):
        """Returns the six full-batch embedding sets (vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx).
        Row counts per rank may be passed when known on the host (fixed-shape batches) to avoid two tiny
        all-gathers + host syncs."""
        dev = vis.vid_emb.device
        if vid_counts is None:
            vid_counts = self.global_counts(vis.vid_emb.shape[0], dev)
        if clip_counts is None:
            clip_counts = self.global_counts(vis.clip_emb.shape[0], dev)
        # pack the high-level (per video) sets into one buffer and the low-level (per clip) sets into another:
        # two collectives instead of six
        high = torch.cat([vis.vid_emb, txt.par_emb, vis.vid_context, txt.par_context], dim=1)
        low = torch.cat([vis.clip_emb, txt.sent_emb], dim=1)
        high_all = gather_rows(high, vid_counts, self.rank, self.group)
        low_all = gather_rows(low, clip_counts, self.rank, self.group)
        dg, dl = vis.vid_emb.shape[1], vis.clip_emb.shape[1]
        vid_emb, par_emb = high_all[:, :dg], high_all[:, dg:2 * dg]
        vid_ctx, par_ctx = high_all[:, 2 * dg:2 * dg + dl], high_all[:, 2 * dg + dl:]
        clip_emb, sent_emb = low_all[:, :dl], low_all[:, dl:]
        return vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx, sum(vid_counts)

    def allreduce_grads(self, flat_grads: List[torch.Tensor], side_stream=None) -> None:
        """Sum the flat gradient arenas over ranks.  With a side stream the collectives are enqueued there
        (ordered after the current stream) and the current stream waits for them afterwards."""
        if side_stream is None:
            for g in flat_grads:
                _all_reduce(g, dist.ReduceOp.SUM, self.group)
            return
        side_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side_stream):
            for g in flat_grads:
                _all_reduce(g, dist.ReduceOp.SUM, self.group)
        torch.cuda.current_stream().wait_stream(side_stream)
