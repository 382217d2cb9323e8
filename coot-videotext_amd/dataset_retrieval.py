"""Input side of the path: batch collation and host-to-device staging.

Mirrors the data types and ``collate_fn`` of coot/dataset_retrieval.py (``RetrievalDataPointTuple`` :25-61,
``RetrievalDataBatchTuple`` :64-102, ``RetrievalDataset.collate_fn`` :335-463) — same fields, same padded layout, same masks
and lengths — built for one GPU with a lot of HBM behind a PCIe link:

* the whole batch lives in ONE arena: four padded feature blocks, six int64 length vectors and four bool masks at 256-byte
  aligned offsets of a single (pinned) host allocation.  The reference allocates 16 tensors per batch, pins them one by one in
  the DataLoader and issues 16 ``.cuda()`` copies (nntrainer/typext.py:248-260);
* the padded blocks are written by ``coot_collate_level`` (libcoot_hip.so, host code: memcpy / zero-fill per sequence on a few
  threads instead of Python slice assignments), optionally as bf16 (round-to-nearest-even — the first thing the networks do
  with a feature is LayerNorm statistics in fp32 and a bf16 MFMA operand, so bf16 staging halves the PCIe bytes; fp32 is the
  default and is bit-identical to the reference's batch);
* ``DeviceLoader`` moves each arena with ONE asynchronous copy on a copy stream, ``depth`` batches ahead of the consumer,
  into rotating device arenas; the consumer's stream only waits on the copy's event.  Host and device arenas are reused —
  nothing is allocated per batch after warm-up.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib
from .model_retrieval import RetrievalDataBatchTuple

_ALIGN = 256


@dataclass
class RetrievalDataPointTuple:
    """One video with its clips, paragraph and sentences (coot/dataset_retrieval.py:25-61, same field names).  Features are
    fp32 ``torch.Tensor`` or ``numpy.ndarray`` of shape [length, dim]."""
    key: str
    data_key: str
    sentences: List[str]
    vid_feat: Any
    vid_feat_len: int
    par_feat: Any
    par_feat_len: int
    clip_num: int
    clip_feat_list: List[Any]
    clip_feat_len_list: List[int]
    sent_num: int
    sent_feat_list: List[Any]
    sent_feat_len_list: List[int]


def _host_ptr(x) -> Tuple[int, Any]:
    """(address, keep-alive object) of a contiguous fp32 [rows, dim] feature."""
    if isinstance(x, torch.Tensor):
        if x.dtype != torch.float32 or not x.is_contiguous() or x.is_cuda:
            x = x.detach().to("cpu", torch.float32).contiguous()
        return x.data_ptr(), x
    a = np.ascontiguousarray(x, dtype=np.float32)
    return a.ctypes.data, a


class BatchArena:
    """One host allocation (pinned when a GPU is present) that holds a whole collated batch; grows, never shrinks."""

    def __init__(self, pin: Optional[bool] = None):
        self.pin = torch.cuda.is_available() if pin is None else pin
        self.host = torch.empty(0, dtype=torch.uint8)
        self.nbytes = 0

    def reserve(self, nbytes: int) -> None:
        if self.host.numel() < nbytes:
            cap = max(nbytes, int(self.host.numel() * 1.5))
            self.host = torch.empty(cap, dtype=torch.uint8, pin_memory=self.pin)
        self.nbytes = nbytes

    def view(self, off: int, shape: Sequence[int], dtype: torch.dtype) -> torch.Tensor:
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        return self.host[off:off + n].view(dtype).view(*shape)


def _plan(segments: List[Tuple[str, Tuple[int, ...], torch.dtype]]):
    off, table = 0, {}
    for name, shape, dtype in segments:
        table[name] = (off, shape, dtype)
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        off = (off + n + _ALIGN - 1) // _ALIGN * _ALIGN
    return table, off


def collate_fn(data_batch: List[RetrievalDataPointTuple], arena: Optional[BatchArena] = None, bf16: bool = False,
               threads: int = 4) -> RetrievalDataBatchTuple:
    """coot/dataset_retrieval.py:335-463 with the batch written into ``arena`` (a fresh pageable one if None).  The returned
    tuple's tensors are views of the arena (``batch.arena`` / ``batch.arena_table`` describe it for DeviceLoader); padding
    lengths are the batch maxima exactly as in the reference (vid/par: longest video / paragraph, clip/sent: longest clip /
    sentence of the batch), sentences are cut out of the paragraph features by the running pointer (:438-452)."""
    lib = _lib.load()
    B = len(data_batch)
    assert B > 0, "empty batch"
    arena = arena if arena is not None else BatchArena(pin=False)
    fdt = torch.bfloat16 if bf16 else torch.float32
    keep: List[Any] = []

    def ptrs(items):
        out = []
        for it in items:
            p, k = _host_ptr(it)
            keep.append(k)
            out.append(p)
        return out

    vid_ptr = ptrs(d.vid_feat for d in data_batch)
    par_ptr = ptrs(d.par_feat for d in data_batch)
    vid_dim = int(data_batch[0].vid_feat.shape[-1])
    par_dim = int(data_batch[0].par_feat.shape[-1])
    vid_len = [int(d.vid_feat_len) for d in data_batch]
    par_len = [int(d.par_feat_len) for d in data_batch]
    clip_num = [int(d.clip_num) for d in data_batch]
    sent_num = [int(d.sent_num) for d in data_batch]
    clip_ptr = ptrs(c for d in data_batch for c in d.clip_feat_list)
    clip_len = [int(c.shape[0]) for d in data_batch for c in d.clip_feat_list]
    sent_len = [int(n) for d in data_batch for n in d.sent_feat_len_list]
    sent_ptr = []
    for b, d in enumerate(data_batch):  # sentences are pieces of the paragraph features
        pointer = 0
        for n in d.sent_feat_len_list:
            assert pointer + int(n) <= par_len[b], f"sentence lengths of {d.key} exceed its paragraph features"
            sent_ptr.append(par_ptr[b] + pointer * par_dim * 4)
            pointer += int(n)
    assert len(clip_len) == sum(clip_num) and len(sent_len) == sum(sent_num)
    Lv, Lp, Lc, Ls = max(vid_len), max(par_len), max(clip_len), max(sent_len)
    Nc, Ns = len(clip_len), len(sent_len)
    table, total = _plan([
        ("vid_feat", (B, Lv, vid_dim), fdt), ("clip_feat", (Nc, Lc, vid_dim), fdt),
        ("par_feat", (B, Lp, par_dim), fdt), ("sent_feat", (Ns, Ls, par_dim), fdt),
        ("vid_feat_len", (B,), torch.int64), ("par_feat_len", (B,), torch.int64), ("clip_num", (B,), torch.int64),
        ("clip_feat_len", (Nc,), torch.int64), ("sent_num", (B,), torch.int64), ("sent_feat_len", (Ns,), torch.int64),
        ("vid_feat_mask", (B, Lv), torch.bool), ("clip_feat_mask", (Nc, Lc), torch.bool),
        ("par_feat_mask", (B, Lp), torch.bool), ("sent_feat_mask", (Ns, Ls), torch.bool)])
    arena.reserve(total)
    t = {name: arena.view(off, shape, dtype) for name, (off, shape, dtype) in table.items()}
    for name, vals in (("vid_feat_len", vid_len), ("par_feat_len", par_len), ("clip_num", clip_num), ("clip_feat_len", clip_len),
                       ("sent_num", sent_num), ("sent_feat_len", sent_len)):
        t[name].copy_(torch.tensor(vals, dtype=torch.int64))
    for name, plist, lens, dim, L in (("vid_feat", vid_ptr, vid_len, vid_dim, Lv), ("clip_feat", clip_ptr, clip_len, vid_dim, Lc),
                                      ("par_feat", par_ptr, par_len, par_dim, Lp), ("sent_feat", sent_ptr, sent_len, par_dim, Ls)):
        n = len(plist)
        seq = (C.c_void_p * n)(*plist)
        rows = (C.c_int64 * n)(*lens)
        _lib.check(lib.coot_collate_level(seq, rows, n, dim, L, int(bf16), t[name].data_ptr(), t[name + "_mask"].data_ptr(), threads),
                   "coot_collate_level")
    batch = RetrievalDataBatchTuple(
        [d.key for d in data_batch], [d.data_key for d in data_batch], [d.sentences for d in data_batch],
        t["vid_feat"], t["vid_feat_mask"], t["vid_feat_len"], t["par_feat"], t["par_feat_mask"], t["par_feat_len"],
        t["clip_num"], t["clip_feat"], t["clip_feat_mask"], t["clip_feat_len"],
        t["sent_num"], t["sent_feat"], t["sent_feat_mask"], t["sent_feat_len"],
        max_clip_num=max(clip_num), max_sent_num=max(sent_num))
    batch.arena, batch.arena_table = arena, table
    return batch


TENSOR_FIELDS = ("vid_feat", "vid_feat_mask", "vid_feat_len", "par_feat", "par_feat_mask", "par_feat_len", "clip_num", "clip_feat",
                 "clip_feat_mask", "clip_feat_len", "sent_num", "sent_feat", "sent_feat_mask", "sent_feat_len")


class DeviceLoader:
    """Iterates ``source`` (lists of RetrievalDataPointTuple — collated here — or batches that collate_fn already wrote into
    their own BatchArena) and yields device-resident batches, ``depth`` arenas ahead of the consumer: one async H2D copy per
    batch on a copy stream, the consumer's stream waits on its event only.  A device arena is rewritten only after the work
    the consumer enqueued on it (everything up to its next ``next()``) has finished; a host arena only after its copy has."""

    def __init__(self, source: Iterable, depth: int = 2, device: str = "cuda", bf16: bool = False, threads: int = 8):
        assert depth >= 1
        self.source, self.depth, self.device, self.bf16, self.threads = source, depth, torch.device(device), bf16, threads
        self.copy_stream = torch.cuda.Stream(device=self.device)
        n = depth + 1
        self.host = [BatchArena(pin=True) for _ in range(n)]
        self.dev = [torch.empty(0, dtype=torch.uint8, device=self.device) for _ in range(n)]
        self.copied = [None] * n    # event: H2D copy of slot k finished (host arena k reusable, device arena k readable)
        self.consumed = [None] * n  # event: the consumer's work on device arena k is enqueued up to here

    def __len__(self) -> int:
        return len(self.source)  # type: ignore[arg-type]

    def _stage(self, item, k: int) -> RetrievalDataBatchTuple:
        if self.copied[k] is not None:
            self.copied[k].synchronize()  # the host arena is about to be rewritten
        hb = collate_fn(item, self.host[k], self.bf16, self.threads) if isinstance(item, list) else item
        nbytes = hb.arena.nbytes
        if self.dev[k].numel() < nbytes:
            # (Re)allocate ON the copy stream: the caching allocator may hand out a block the consumer's stream freed while kernels
            # reading it are still queued there; a block allocated under the copy stream is only re-used in that stream's order,
            # and record_stream tells the allocator that the consumer's stream reads it too.  The copy stream first waits for
            # everything the consumer has enqueued so far (the old arena of this slot may still be in use until then).
            consumer = torch.cuda.current_stream(self.device)
            self.copy_stream.wait_stream(consumer)
            with torch.cuda.stream(self.copy_stream):
                self.dev[k] = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=self.device)
            self.dev[k].record_stream(consumer)
        if self.consumed[k] is not None:
            self.copy_stream.wait_event(self.consumed[k])
        with torch.cuda.stream(self.copy_stream):
            self.dev[k][:nbytes].copy_(hb.arena.host[:nbytes], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.copied[k] = ev
        fields = {}
        for name in TENSOR_FIELDS:
            off, shape, dtype = hb.arena_table[name]
            nb = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
            fields[name] = self.dev[k][off:off + nb].view(dtype).view(*shape)
        db = RetrievalDataBatchTuple(hb.key, hb.data_key, hb.sentences, **fields, max_clip_num=hb.max_clip_num,
                                     max_sent_num=hb.max_sent_num)
        db.ready = ev
        return db

    def __iter__(self) -> Iterator[RetrievalDataBatchTuple]:
        n = self.depth + 1
        it = iter(self.source)
        queue: List[Tuple[int, RetrievalDataBatchTuple]] = []
        slot = 0
        exhausted = False

        def fill():
            nonlocal slot, exhausted
            while not exhausted and len(queue) < self.depth:
                try:
                    item = next(it)
                except StopIteration:
                    exhausted = True
                    return
                queue.append((slot, self._stage(item, slot)))
                slot = (slot + 1) % n

        fill()
        while queue:
            k, db = queue.pop(0)
            torch.cuda.current_stream(self.device).wait_event(db.ready)
            if self.bf16:  # the kernels take fp32 features: widen on the device (HBM pass, not a PCIe one)
                for name in ("vid_feat", "clip_feat", "par_feat", "sent_feat"):
                    setattr(db, name, getattr(db, name).float())
            yield db
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.consumed[k] = ev
            fill()
