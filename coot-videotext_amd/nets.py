"""
TransformerHip — drop-in for nntrainer.models.TransformerLegacy
(nntrainer/models/transformer_legacy.py:115-288) backed by libcoot_hip.so.

* Parameters stay ``nn.Parameter``s with the reference's state-dict names and shapes (SURVEY 8a
  row a2) so checkpoints (``model_N.pth`` = {net_name: state_dict}), ``torch.optim`` and the
  reference's init (nntrainer/initialization.py:51-111) keep working.  Physically they are views
  into ONE flat fp32 arena per network, whose layout the C library defines
  (coot_net_param_info); gradients use the same flat layout.
* ``forward(features, mask, lengths, hidden_state) -> (pooled, per_token)`` has the reference
  signature.  All math runs in HIP kernels through the C ABI; PyTorch only owns memory, streams
  and the autograd graph edges.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import lib as _lib
from .config import TransformerConfig


def sincos_pe(max_len: int, dim: int) -> torch.Tensor:
    """PositionalEncodingSinCos buffer: verbatim the six lines of nntrainer/models/encoder.py:84-90 (a parity-defining
    formula: the exponent uses the dim index itself, not index // 2, and the buffer is part of the state dict)."""
    pe = torch.zeros(max_len, dim).float()
    position = torch.arange(0, max_len).unsqueeze(1).float()
    dimension = torch.arange(0, dim).float()
    div_term = 10000 ** (2 * dimension / dim)
    pe[:, 0::2] = torch.sin(position / div_term[0::2])
    pe[:, 1::2] = torch.cos(position / div_term[1::2])
    return pe


def fill_truncnorm_(t: torch.Tensor, std: float, limit: float = 2.0, generator=None) -> None:
    """Truncated normal, verbatim the draw of nntrainer/utils_torch.py:87-92 (8 candidates per element, the first one inside
    +-limit sigma): the RNG consumption order defines which weights a seed gives, so the lines are kept as they are."""
    tmp = torch.empty(tuple(t.shape) + (8,)).normal_(generator=generator)
    valid = (tmp < limit) & (tmp > -limit)
    _, ind = valid.max(-1, keepdim=True)
    t.copy_(tmp.gather(-1, ind).squeeze(-1).mul_(std))


class TransformerHip(nn.Module):
    def __init__(self, cfg: TransformerConfig, feature_dim: Optional[int] = None):
        super().__init__()
        if feature_dim is not None:
            assert feature_dim == cfg.input_dim, (feature_dim, cfg.input_dim)
        self.cfg = cfg
        self.c_cfg = cfg.to_c()
        self.numel, self.table = _lib.param_table(self.c_cfg)
        self.output_dim = cfg.hidden_dim * (2 if cfg.use_context else 1)
        flat = torch.zeros(self.numel, dtype=torch.float32)
        self._flat: torch.Tensor = flat
        self._params: List[nn.Parameter] = []
        for name, off, shape in self.table:
            p = nn.Parameter(flat[off:off + math.prod(shape)].view(shape), requires_grad=True)
            self._register(name, p)
            self._params.append(p)
        # non-trainable state-dict entries of the reference
        self._submodule("embedding").register_buffer("pe", sincos_pe(1000, cfg.hidden_dim))
        if cfg.pooler == "atn":
            self._register("pooler.pools.0.genpool_one", nn.Parameter(torch.ones(1), requires_grad=False))
        self._wpack: Optional[torch.Tensor] = None
        self._dirty = True
        self._grad_flat: Optional[torch.Tensor] = None
        self.accumulate_into_flat = False  # set by RetrievalTrainer.train_step (see _NetFn.backward)
        self.seed_dev: Optional[torch.Tensor] = None  # device int64 base seed (advanced once per step by the trainer)
        self.call_counter = 0
        self.init_network(cfg.weight_init_type, cfg.weight_init_std)

    # ---- module tree that reproduces the reference state-dict keys ---------------------------------
    def _submodule(self, path: str) -> nn.Module:
        mod: nn.Module = self
        for part in path.split("."):
            if part not in mod._modules:
                mod.add_module(part, nn.Module())
            mod = mod._modules[part]
        return mod

    def _register(self, name: str, p: nn.Parameter) -> None:
        path, leaf = name.rsplit(".", 1) if "." in name else ("", name)
        (self._submodule(path) if path else self).register_parameter(leaf, p)

    def init_network(self, init_type: str, init_std: float) -> None:
        """nntrainer/initialization.py:51-111: truncnorm on every weight AND bias, LN gain 1 / bias 0."""
        if init_type == "none":
            return
        assert init_type == "truncnorm", init_type
        with torch.no_grad():
            for (name, _, _), p in zip(self.table, self._params):
                if "layer_normalization" in name or "norm_input." in name:
                    p.fill_(1.0 if name.endswith("gain") else 0.0)
                else:
                    fill_truncnorm_(p, init_std)
        self._dirty = True

    # ---- flat arena maintenance -----------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        super()._apply(fn)
        self._reflatten()
        return self

    def _reflatten(self) -> None:
        dev = self._params[0].device
        flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for (name, off, shape), p in zip(self.table, self._params):
                flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + p.numel()].view(shape)
        self._flat = flat
        self._wpack = None
        self._grad_flat = None
        self._dirty = True

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._dirty = True
        return out

    def set_compute_dtype(self, dtype: str) -> None:
        """"bf16" (default): the fast path.  "f32": the fp32 reference mode of the library (coot_net_config.dtype = COOT_DTYPE_F32: the
        reference's op sequence and its derivative in fp32, eval mode) — outputs agree with the reference to ~1e-6 of their scale,
        parameter gradients to ~1e-5, so comparing the two modes measures the bf16 rounding of the fast path."""
        assert dtype in ("bf16", "f16", "f32"), dtype  # ("f16" / "bf16": whichever build is loaded — the other one is refused by the library)
        self.cfg.dtype = dtype
        self.c_cfg = self.cfg.to_c()

    def mark_dirty(self) -> None:
        """Call after an optimizer step: the bf16 weight pack is rebuilt before the next forward."""
        self._dirty = True

    def pack_is_fresh(self) -> bool:
        """True when the bf16 weight pack matches the fp32 parameters (nothing touched them since the last packing)."""
        return (self._wpack is not None and not self._dirty and self._param_version() == getattr(self, "_packed_version", None)
                and self._wpack.data_ptr() == getattr(self, "_packed_ptr", None))

    def mark_packed(self) -> None:
        """The library updated the parameters through the flat arena AND rebuilt the pack (coot_train_step with
        COOT_STEP_REPACK): raw-pointer writes do not bump the tensors' version counters, so record the pack as current."""
        self._dirty = False
        self._packed_version = self._param_version()
        self._packed_ptr = self._wpack.data_ptr() if self._wpack is not None else None

    def bind_flat_grads(self) -> torch.Tensor:
        """Pre-assign every param.grad as a view of one flat fp32 gradient arena (zeroed), so the
        autograd accumulation lands in place and a fused optimizer / one all-reduce can use it."""
        if self._grad_flat is None or self._grad_flat.device != self._flat.device:
            self.rebind_flat_grads(torch.zeros_like(self._flat))
        return self._grad_flat

    def rebind_flat_grads(self, storage: torch.Tensor) -> None:
        """Use `storage` (fp32, numel == self.numel, e.g. a slice of one arena shared by the four networks so that the
        data-parallel step zeroes and all-reduces ONE tensor) as the gradient arena; the current gradients are copied over."""
        assert storage.dtype == torch.float32 and storage.numel() == self.numel and storage.device == self._flat.device
        if self._grad_flat is not None and self._grad_flat.device == storage.device and self._grad_flat.data_ptr() != storage.data_ptr():
            storage.copy_(self._grad_flat)
        self._grad_flat = storage
        for (name, off, shape), p in zip(self.table, self._params):
            p.grad = self._grad_flat[off:off + p.numel()].view(shape)

    def _param_version(self) -> int:
        return sum(p._version for p in self._params)

    def ensure_packed(self) -> torch.Tensor:
        """(Re)build the bf16 weight pack when the fp32 parameters changed (in-place optimizer updates bump the
        tensors' version counters) — once per optimizer step, two kernel launches."""
        lib = _lib.load()
        if self._wpack is None:
            nbytes = lib.coot_net_wpack_bytes(C.byref(self.c_cfg))
            self._wpack = torch.empty(nbytes, dtype=torch.uint8, device=self._flat.device)
            self._dirty = True
        ver = self._param_version()
        # (a pack that was COPIED — copy.deepcopy of the module — lives at a new address: the library tracks which layouts of a pack are
        # current by its address, so such a pack is rebuilt rather than trusted)
        if self._dirty or ver != getattr(self, "_packed_version", None) or self._wpack.data_ptr() != getattr(self, "_packed_ptr", None):
            _lib.check(lib.coot_net_pack_weights(C.byref(self.c_cfg), _lib.ptr(self._flat), _lib.ptr(self._wpack),
                                                 _lib.stream_ptr()), "coot_net_pack_weights")
            self._dirty = False
            self._packed_version = ver
            self._packed_ptr = self._wpack.data_ptr()
        return self._wpack

    # ---- forward ----------------------------------------------------------------------------------------
    def forward(self, features: torch.Tensor, mask: Optional[torch.Tensor], lengths: torch.Tensor,
                hidden_state: Optional[torch.Tensor], want_tokens: bool = True, seed: Optional[int] = None):
        """Reference signature (transformer_legacy.py:200-215).  `mask` (True = padding) must be
        consistent with `lengths`; the kernels use `lengths`."""
        pooled, tokens = self._run(features, lengths, None, None, hidden_state, want_tokens, seed)
        return pooled, tokens

    def forward_pair(self, features: torch.Tensor, lengths: torch.Tensor, features2: torch.Tensor, lengths2: torch.Tensor,
                     seed: Optional[int] = None, packed: Optional[Tuple[torch.Tensor, int]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Two sets of sequences through the same weights in ONE call (video frames + clip frames of a batch,
        coot/model_retrieval.py:104,:120).  Returns (pooled_1 [N1, D], pooled_2 [N2, D]).
        packed = (cu_seqlens, total): int32 device tensor [N1 + N2 + 1] of packed row starts (first set first) and its last entry as
        a host int — the network then processes only the valid tokens (coot_packed_seqs, include/coot_hip.h); same outputs."""
        pooled, _ = self._run(features, lengths, features2, lengths2, None, False, seed, packed)
        n1 = features.shape[0]
        return pooled[:n1], pooled[n1:]

    def _run(self, features, lengths, features2, lengths2, hidden_state, want_tokens, seed, packed=None):
        if not features.is_cuda:
            raise RuntimeError("TransformerHip runs on an MI355X only (features must be a cuda tensor); "
                               "there is no CPU fallback")
        if self.cfg.use_context and hidden_state is None:
            raise AssertionError("hidden_state required for use_context (transformer_legacy.py:252)")
        self.ensure_packed()
        if seed is None:
            self.call_counter += 1
            seed = (torch.initial_seed() * 1000003 + self.call_counter) & 0xFFFFFFFFFFFFFFFF
        return _NetFn.apply(self, features, lengths, features2, lengths2, hidden_state, bool(want_tokens), int(seed), packed,
                            *self._params)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: TransformerHip, feats, lengths, feats2, lengths2, hidden, want_tokens, seed, packed, *params):
        lib = _lib.load()
        cfg = net.c_cfg
        feats = feats.contiguous().float()
        lengths = lengths.contiguous().long()
        N, L, Din = feats.shape
        assert Din == net.cfg.input_dim, (Din, net.cfg.input_dim)
        N2 = L2 = 0
        if feats2 is not None:
            feats2 = feats2.contiguous().float()
            lengths2 = lengths2.contiguous().long()
            N2, L2, Din2 = feats2.shape
            assert Din2 == Din
        dev = feats.device
        hid = hidden.contiguous().float() if hidden is not None else None
        pooled = torch.empty(N + N2, net.output_dim, dtype=torch.float32, device=dev)
        tokens = torch.empty(N, L, net.cfg.hidden_dim, dtype=torch.float32, device=dev) if want_tokens else None
        saved = torch.empty(lib.coot_net_saved_bytes(C.byref(cfg), N, L, N2, L2), dtype=torch.uint8, device=dev)
        train = 1 if net.training else 0
        pe = net.embedding.pe
        pk = None
        if packed is not None:
            cu, total = packed
            assert cu.dtype == torch.int32 and cu.is_cuda and cu.numel() == N + N2 + 1, "packed = (int32 device cu_seqlens [N + N2 + 1], total)"
            pk = _lib.PackedSeqs(cu.data_ptr(), int(total))
        _lib.check(lib.coot_net_fwd(C.byref(cfg), _lib.ptr(net._flat), _lib.ptr(net._wpack), _lib.ptr(pe), _lib.ptr(feats),
                                    _lib.ptr(lengths), N, L, _lib.ptr(feats2), _lib.ptr(lengths2), N2, L2, _lib.ptr(hid),
                                    _lib.ptr(pooled), _lib.ptr(tokens), _lib.ptr(saved), saved.numel(), None, 0, train, seed,
                                    _lib.ptr(net.seed_dev), _lib.stream_ptr(), C.byref(pk) if pk is not None else None), "coot_net_fwd")
        ctx.net, ctx.saved, ctx.seed, ctx.train, ctx.packed = net, saved, seed, train, packed
        ctx.feats, ctx.lengths, ctx.hid, ctx.feats2, ctx.lengths2 = feats, lengths, hid, feats2, lengths2
        ctx.need_dfeats = bool(ctx.needs_input_grad[1])
        ctx.need_dhid = hidden is not None
        ctx.mark_non_differentiable(*([tokens] if tokens is not None else []))
        return pooled, tokens

    @staticmethod
    def backward(ctx, dpooled, _dtokens):
        lib = _lib.load()
        net: TransformerHip = ctx.net
        cfg = net.c_cfg
        feats, lengths, hid, feats2, lengths2 = ctx.feats, ctx.lengths, ctx.hid, ctx.feats2, ctx.lengths2
        N, L, Din = feats.shape
        N2, L2 = (feats2.shape[0], feats2.shape[1]) if feats2 is not None else (0, 0)
        dev = feats.device
        dpooled = dpooled.contiguous().float()
        # gradient sink: the network's persistent flat arena (train_step; param.grad are views of it) or a
        # fresh zeroed arena whose views are returned to autograd (generic drop-in use)
        direct = net.accumulate_into_flat and net._grad_flat is not None
        gflat = net._grad_flat if direct else torch.zeros(net.numel, dtype=torch.float32, device=dev)
        dhid = torch.empty(N, net.cfg.hidden_dim, dtype=torch.float32, device=dev) if ctx.need_dhid else None
        dfeats = None
        if ctx.need_dfeats:
            if net.cfg.use_input_fc:
                raise RuntimeError("gradient wrt input features is only available for networks without input_fc")
            dfeats = torch.empty_like(feats)
        scratch = torch.empty(lib.coot_net_scratch_bytes(C.byref(cfg), N, L, N2, L2), dtype=torch.uint8, device=dev)
        _lib.check(lib.coot_net_bwd(C.byref(cfg), _lib.ptr(net._flat), _lib.ptr(net._wpack), _lib.ptr(net.embedding.pe),
                                    _lib.ptr(feats), _lib.ptr(lengths), N, L, _lib.ptr(feats2), _lib.ptr(lengths2), N2, L2,
                                    _lib.ptr(hid), _lib.ptr(dpooled), _lib.ptr(gflat), _lib.ptr(dhid), _lib.ptr(dfeats),
                                    _lib.ptr(ctx.saved), ctx.saved.numel(), _lib.ptr(scratch), scratch.numel(), ctx.train,
                                    ctx.seed, _lib.ptr(net.seed_dev), _lib.stream_ptr(),
                                    C.byref(_lib.PackedSeqs(ctx.packed[0].data_ptr(), int(ctx.packed[1]))) if ctx.packed is not None else None),
                   "coot_net_bwd")
        if direct:
            grads = (None,) * len(net.table)
        else:
            grads = tuple(gflat[off:off + math.prod(shape)].view(shape) for (_, off, shape) in net.table)
        return (None, dfeats, None, None, None, dhid, None, None, None) + grads


class _PackFn(torch.autograd.Function):
    """The pack loop of encode_visual/encode_text (coot/model_retrieval.py:121-136)."""

    @staticmethod
    def forward(ctx, emb, counts, cmax: int):
        lib = _lib.load()
        emb = emb.contiguous().float()
        counts = counts.contiguous().long()
        B, D = counts.shape[0], emb.shape[1]
        dev = emb.device
        out = torch.empty(B, cmax, D, dtype=torch.float32, device=dev)
        mask = torch.empty(B, cmax, dtype=torch.bool, device=dev)
        lens = torch.empty(B, dtype=torch.long, device=dev)
        _lib.check(lib.coot_pack_fwd(_lib.ptr(emb), _lib.ptr(counts), B, cmax, D, _lib.ptr(out), _lib.ptr(mask), _lib.ptr(lens),
                                     _lib.stream_ptr()), "coot_pack_fwd")
        ctx.counts, ctx.shape = counts, (emb.shape[0], D, B, cmax)
        ctx.mark_non_differentiable(mask, lens)
        return out, mask, lens

    @staticmethod
    def backward(ctx, dout, _dm, _dl):
        lib = _lib.load()
        Nc, D, B, cmax = ctx.shape
        dout = dout.contiguous().float()
        demb = torch.zeros(Nc, D, dtype=torch.float32, device=dout.device)
        _lib.check(lib.coot_pack_bwd(_lib.ptr(dout), _lib.ptr(ctx.counts), B, cmax, D, _lib.ptr(demb), _lib.stream_ptr()),
                   "coot_pack_bwd")
        return demb, None, None


def pack_by_count(emb: torch.Tensor, counts: torch.Tensor, cmax: Optional[int] = None):
    if cmax is None:
        cmax = int(counts.max())  # host sync, as th.max(batch.clip_num) in the reference (:122)
    return _PackFn.apply(emb, counts, int(cmax))
