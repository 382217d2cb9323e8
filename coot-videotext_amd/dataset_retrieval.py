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
* ``collate_fn(..., packed=True)`` writes the batch PACKED AT THE SOURCE instead (``coot_collate_packed``): the frames of the B videos
  and of the Nc clips back to back in one matrix, the words of the paragraphs and sentences in another, int32 row starts
  (cu_seqlens) next to them — no padding row is written, crosses PCIe or is read from HBM (real ActivityNet clips average far
  fewer than the 80-frame maximum).  ``unpack_batch`` rebuilds the reference's padded batch bit-exactly;
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
from .model_retrieval import RetrievalDataBatchTuple, RetrievalPackedBatchTuple

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
               threads: int = 4, packed: bool = False):
    """coot/dataset_retrieval.py:335-463 with the batch written into ``arena`` (a fresh pageable one if None).  The returned
    tuple's tensors are views of the arena (``batch.arena`` / ``batch.arena_table`` describe it for DeviceLoader); padding
    lengths are the batch maxima exactly as in the reference (vid/par: longest video / paragraph, clip/sent: longest clip /
    sentence of the batch), sentences are cut out of the paragraph features by the running pointer (:438-452).
    packed=True: a RetrievalPackedBatchTuple (no padding rows, cu_seqlens) instead of the padded RetrievalDataBatchTuple."""
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
    if packed:
        assert Nc == Ns, "one sentence per clip (coot/dataset_retrieval.py:404-452)"
        tok_vis, tok_txt = sum(vid_len) + sum(clip_len), sum(par_len) + sum(sent_len)
        table, total = _plan([
            ("vis_tokens", (tok_vis, vid_dim), fdt), ("txt_tokens", (tok_txt, par_dim), fdt),
            ("cu_vis", (B + Nc + 1,), torch.int32), ("cu_txt", (B + Ns + 1,), torch.int32),
            ("vid_feat_len", (B,), torch.int64), ("par_feat_len", (B,), torch.int64), ("clip_num", (B,), torch.int64),
            ("clip_feat_len", (Nc,), torch.int64), ("sent_num", (B,), torch.int64), ("sent_feat_len", (Ns,), torch.int64)])
        arena.reserve(total)
        t = {name: arena.view(off, shape, dtype) for name, (off, shape, dtype) in table.items()}
        for name, vals in (("vid_feat_len", vid_len), ("par_feat_len", par_len), ("clip_num", clip_num), ("clip_feat_len", clip_len),
                           ("sent_num", sent_num), ("sent_feat_len", sent_len)):
            t[name].copy_(torch.tensor(vals, dtype=torch.int64))
        for name, cu_name, plist, lens, dim in (("vis_tokens", "cu_vis", vid_ptr + clip_ptr, vid_len + clip_len, vid_dim),
                                                ("txt_tokens", "cu_txt", par_ptr + sent_ptr, par_len + sent_len, par_dim)):
            n = len(plist)
            seq = (C.c_void_p * n)(*plist)
            rows = (C.c_int64 * n)(*lens)
            _lib.check(lib.coot_collate_packed(seq, rows, n, dim, int(bf16), t[name].data_ptr(), t[cu_name].data_ptr(), threads),
                       "coot_collate_packed")
        pb = RetrievalPackedBatchTuple(
            [d.key for d in data_batch], [d.data_key for d in data_batch], [d.sentences for d in data_batch],
            t["vis_tokens"], t["txt_tokens"], t["cu_vis"], t["cu_txt"], t["vid_feat_len"], t["par_feat_len"], t["clip_num"],
            t["clip_feat_len"], t["sent_num"], t["sent_feat_len"], max_lens=(Lv, Lc, Lp, Ls), max_clip_num=max(clip_num),
            max_sent_num=max(sent_num), tok_vis=tok_vis, tok_txt=tok_txt)
        pb.arena, pb.arena_table = arena, table
        return pb
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


PACKED_TENSOR_FIELDS = ("vis_tokens", "txt_tokens", "cu_vis", "cu_txt", "vid_feat_len", "par_feat_len", "clip_num", "clip_feat_len", "sent_num",
                        "sent_feat_len")


def unpack_batch(pb: RetrievalPackedBatchTuple) -> RetrievalDataBatchTuple:
    """The reference's padded batch (coot/dataset_retrieval.py:335-463: zero-padded feature blocks, bool masks with True = padding)
    rebuilt from a packed one, on the packed batch's device.  fp32 packed batches give the reference's tensors bit for bit."""
    dev = pb.vis_tokens.device
    Lv, Lc, Lp, Ls = pb.max_lens
    B, Nc = pb.vid_feat_len.numel(), pb.clip_feat_len.numel()

    def level(tokens, cu, lens, first, n, L):
        out = torch.zeros(n, L, tokens.shape[1], dtype=torch.float32, device=dev)
        mask = torch.ones(n, L, dtype=torch.bool, device=dev)
        cu_h, lens_h = cu.cpu().tolist(), lens.cpu().tolist()
        for i in range(n):
            r0, r = cu_h[first + i], lens_h[i]
            out[i, :r] = tokens[r0:r0 + r].float()
            mask[i, :r] = False
        return out, mask

    vid, vid_m = level(pb.vis_tokens, pb.cu_vis, pb.vid_feat_len, 0, B, Lv)
    clip, clip_m = level(pb.vis_tokens, pb.cu_vis, pb.clip_feat_len, B, Nc, Lc)
    par, par_m = level(pb.txt_tokens, pb.cu_txt, pb.par_feat_len, 0, B, Lp)
    sent, sent_m = level(pb.txt_tokens, pb.cu_txt, pb.sent_feat_len, B, Nc, Ls)
    return RetrievalDataBatchTuple(pb.key, pb.data_key, pb.sentences, vid, vid_m, pb.vid_feat_len, par, par_m, pb.par_feat_len, pb.clip_num, clip,
                                   clip_m, pb.clip_feat_len, pb.sent_num, sent, sent_m, pb.sent_feat_len, max_clip_num=pb.max_clip_num,
                                   max_sent_num=pb.max_sent_num)


TENSOR_FIELDS = ("vid_feat", "vid_feat_mask", "vid_feat_len", "par_feat", "par_feat_mask", "par_feat_len", "clip_num", "clip_feat",
                 "clip_feat_mask", "clip_feat_len", "sent_num", "sent_feat", "sent_feat_mask", "sent_feat_len")


class DeviceLoader:
    """Iterates ``source`` (lists of RetrievalDataPointTuple — collated here — or batches that collate_fn already wrote into
    their own BatchArena) and yields device-resident batches, ``depth`` arenas ahead of the consumer: one async H2D copy per
    batch on a copy stream, the consumer's stream waits on its event only.  A device arena is rewritten only after the work
    the consumer enqueued on it (everything up to its next ``next()``) has finished; a host arena only after its copy has.
    ``lookahead``: the consumer asks for batch t + 1 BEFORE it works on batch t (RetrievalTrainer.train_model: the step on batch t also
    normalises batch t + 1, train_step_native(next_batch=)); a batch's arena is then released ``lookahead`` requests later."""

    def __init__(self, source: Iterable, depth: int = 2, device: str = "cuda", bf16: bool = False, threads: int = 8, packed: bool = False,
                 background: bool = False, lookahead: int = 0):
        assert depth >= 1 and lookahead >= 0
        self.lookahead = lookahead
        self.source, self.depth, self.device, self.bf16, self.threads = source, depth, torch.device(device), bf16, threads
        self.packed = packed  # collate packed at the source: no padding rows in the arena; bf16 rows are consumed as they are
        # background: collation + the H2D enqueue run on a loader thread (the C collation releases the GIL), so the consumer's thread
        # only issues its training step: per batch the host then costs max(collation, step issue) instead of their sum
        self.background = background
        self._consumer_stream = None
        self.copy_stream = torch.cuda.Stream(device=self.device)
        n = depth + 1 + lookahead
        self.host = [BatchArena(pin=True) for _ in range(n)]
        self.dev = [torch.empty(0, dtype=torch.uint8, device=self.device) for _ in range(n)]
        self.copied = [None] * n    # event: H2D copy of slot k finished (host arena k reusable, device arena k readable)
        self.consumed = [None] * n  # event: the consumer's work on device arena k is enqueued up to here

    def __len__(self) -> int:
        return len(self.source)  # type: ignore[arg-type]

    def _stage(self, item, k: int) -> RetrievalDataBatchTuple:
        if self.copied[k] is not None:
            self.copied[k].synchronize()  # the host arena is about to be rewritten
        hb = collate_fn(item, self.host[k], self.bf16, self.threads, packed=self.packed) if isinstance(item, list) else item
        nbytes = hb.arena.nbytes
        if self.dev[k].numel() < nbytes:
            # (Re)allocate ON the copy stream: the caching allocator may hand out a block the consumer's stream freed while kernels
            # reading it are still queued there; a block allocated under the copy stream is only re-used in that stream's order,
            # and record_stream tells the allocator that the consumer's stream reads it too.  The copy stream first waits for
            # everything the consumer has enqueued so far (the old arena of this slot may still be in use until then).
            consumer = self._consumer_stream if self._consumer_stream is not None else torch.cuda.current_stream(self.device)
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
        is_packed = isinstance(hb, RetrievalPackedBatchTuple)
        for name in (PACKED_TENSOR_FIELDS if is_packed else TENSOR_FIELDS):
            off, shape, dtype = hb.arena_table[name]
            nb = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
            fields[name] = self.dev[k][off:off + nb].view(dtype).view(*shape)
        if is_packed:
            db = RetrievalPackedBatchTuple(hb.key, hb.data_key, hb.sentences, **fields, max_lens=hb.max_lens, max_clip_num=hb.max_clip_num,
                                           max_sent_num=hb.max_sent_num, tok_vis=hb.tok_vis, tok_txt=hb.tok_txt)
            db.ready = ev
            return db
        db = RetrievalDataBatchTuple(hb.key, hb.data_key, hb.sentences, **fields, max_clip_num=hb.max_clip_num,
                                     max_sent_num=hb.max_sent_num)
        db.ready = ev
        return db

    def _finish(self, db):
        torch.cuda.current_stream(self.device).wait_event(db.ready)
        if self.bf16 and not isinstance(db, RetrievalPackedBatchTuple):  # padded batches: the kernels take fp32 features, widen on the device (HBM pass, not a PCIe one); packed bf16 rows are read as they are
            for name in ("vid_feat", "clip_feat", "par_feat", "sent_feat"):
                setattr(db, name, getattr(db, name).float())
        return db

    def _iter_background(self) -> Iterator[RetrievalDataBatchTuple]:
        """Producer thread: takes a free slot, collates into its host arena, enqueues the copy; consumer: waits for the copy's event,
        yields, records how far its stream has got on that slot and hands the slot back."""
        import queue
        import threading
        n = self.depth + 1 + self.lookahead
        held: List[int] = []  # slots the consumer may still be working on (lookahead)
        free: "queue.Queue[int]" = queue.Queue()
        ready: "queue.Queue" = queue.Queue()
        for k in range(n):
            free.put(k)
        self._consumer_stream = torch.cuda.current_stream(self.device)
        stop = threading.Event()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

        def produce():
            try:
                torch.cuda.set_device(dev_index)
                for item in self.source:
                    k = free.get()
                    if stop.is_set():
                        return
                    ready.put((k, self._stage(item, k)))
                ready.put(None)
            except BaseException as e:  # surfaces in the consumer
                ready.put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        try:
            while True:
                got = ready.get()
                if got is None:
                    break
                if isinstance(got, BaseException):
                    raise got
                k, db = got
                yield self._finish(db)
                held.append(k)
                while len(held) > self.lookahead:  # everything enqueued so far covers the work on the batch yielded `lookahead` requests ago
                    kk = held.pop(0)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    self.consumed[kk] = ev
                    free.put(kk)
        finally:
            stop.set()
            free.put(0)  # wakes a producer that waits for a slot
            th.join(timeout=10)
            self._consumer_stream = None

    def __iter__(self) -> Iterator[RetrievalDataBatchTuple]:
        if self.background:
            yield from self._iter_background()
            return
        n = self.depth + 1 + self.lookahead
        it = iter(self.source)
        queue: List[Tuple[int, RetrievalDataBatchTuple]] = []
        held: List[int] = []
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
            yield self._finish(db)
            held.append(k)
            while len(held) > self.lookahead:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self.consumed[held.pop(0)] = ev
            fill()
