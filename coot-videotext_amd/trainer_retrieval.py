"""
Training/validation step of the retrieval path with the reference's trainer hooks
(coot/trainer_retrieval.py: compute_align_loss :122-133, compute_cluster_loss :135-146,
compute_total_constrastive_loss :148-182, compute_cyclecons_loss :216-233, step body :253-291,
validate_epoch :312-477).  Experiment scaffolding (checkpoint dirs, tensorboard, LR schedule) is the
reference's BaseTrainer and out of scope; in the reference these methods are
the overrides a ``HipRetrievalTrainer(RetrievalTrainer)`` subclass carries.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import math
import os
import numpy as np
import torch

import ctypes as C

from . import lib as _lib
from . import loss_fn
from .config import RetrievalConfig, RetrievalNetworksConst
from .model_retrieval import (RetrievalDataBatchTuple, RetrievalModelManager, RetrievalPackedBatchTuple, RetrievalTextEmbTuple,
                              RetrievalVisualEmbTuple)
from .retrieval import compute_retrieval, compute_retrieval_device  # noqa: F401


class RAdam(torch.optim.Optimizer):
    """Rectified Adam as the reference's in-file class behaves (nntrainer/optimization.py:79-181; Liu et al. 2019): moments
    from the raw gradient, decoupled weight decay ``p -= wd * lr * p``, the variance-rectified adaptive step once the SMA length
    N_sma >= 5, before that a momentum-SGD step (``degenerated_to_sgd``) or no parameter update at all.  Written on the
    torch._foreach primitives; the native step uses the same rule in the library (coot_radam_step / coot_train_step)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, degenerated_to_sgd=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.degenerated_to_sgd = degenerated_to_sgd

    @staticmethod
    def scalars(step: int, beta1: float, beta2: float, degenerated_to_sgd: bool):
        beta2_t = beta2 ** step
        n_max = 2.0 / (1.0 - beta2) - 1.0
        n_sma = n_max - 2.0 * step * beta2_t / (1.0 - beta2_t)
        if n_sma >= 5:
            rect = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2))
            return "rect", rect / (1 - beta1 ** step)
        if degenerated_to_sgd:
            return "sgd", 1.0 / (1 - beta1 ** step)
        return "none", -1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            beta1, beta2 = group["betas"]
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            steps = {self.state[p]["step"] for p in ps}
            assert len(steps) == 1, "parameters of one group step together"
            t = steps.pop() + 1
            gs = [p.grad for p in ps]
            ms = [self.state[p]["exp_avg"] for p in ps]
            vs = [self.state[p]["exp_avg_sq"] for p in ps]
            torch._foreach_mul_(vs, beta2); torch._foreach_addcmul_(vs, gs, gs, value=1 - beta2)
            torch._foreach_mul_(ms, beta1); torch._foreach_add_(ms, gs, alpha=1 - beta1)
            for p in ps:
                self.state[p]["step"] = t
            mode, step_size = self.scalars(t, beta1, beta2, self.degenerated_to_sgd)
            if mode == "none":
                continue
            if group["weight_decay"] != 0:
                torch._foreach_mul_(ps, 1.0 - group["weight_decay"] * group["lr"])
            if mode == "rect":
                den = torch._foreach_sqrt(vs)
                torch._foreach_add_(den, group["eps"])
                torch._foreach_addcdiv_(ps, ms, den, value=-step_size * group["lr"])
            else:
                torch._foreach_add_(ps, ms, alpha=-step_size * group["lr"])
        return loss


def make_optimizer(cfg_opt, params, capturable: bool = False) -> torch.optim.Optimizer:
    """nntrainer/optimization.py:45-74: Adam(lr, betas=(momentum, adam_beta2), eps, amsgrad) or RAdam(..., degenerated_to_sgd =
    radam_degentosgd), weight_decay * decay_mult per parameter.  Parameters are grouped by decay_mult (same update rule,
    fewer groups)."""
    name = getattr(cfg_opt, "name", "adam")
    lr, wd = float(cfg_opt.lr), float(cfg_opt.weight_decay)
    groups: Dict[float, list] = {}
    for p in params:
        groups.setdefault(p["decay_mult"] * wd, []).append(p["params"])
    param_groups = [{"params": ps, "weight_decay": w, "lr": lr} for w, ps in groups.items()]
    betas = (float(cfg_opt.momentum), float(cfg_opt.adam_beta2))
    if name == "adam":
        return torch.optim.Adam(param_groups, lr=lr, betas=betas, eps=float(cfg_opt.adam_eps),
                                amsgrad=bool(getattr(cfg_opt, "adam_amsgrad", False)), foreach=True, capturable=capturable)
    if name == "radam":
        return RAdam(param_groups, lr=lr, betas=betas, eps=float(cfg_opt.adam_eps),
                     degenerated_to_sgd=bool(getattr(cfg_opt, "radam_degentosgd", False)))
    raise NotImplementedError(f"Unknown optimizer {name}")  # as nntrainer/optimization.py:62


def _with_next(iterable):
    """(item, the item after it or None) for every item: a lookahead of one."""
    it = iter(iterable)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


class RetrievalTrainer:
    # data-parallel native step: Adam + weight pack of the global networks on the communication stream behind their gradient bucket
    # (COOT_UPDATE_GLOBAL_ONLY), as coot_train_step does on one GPU.  Built and parity-tested in round 5; with ONE rank it measured 0.5 %
    # slower than updating all four networks at the tail (1.258 against 1.252 ms, profiles/r05_ab_dp_early_update.txt: two more launches
    # on a third stream with nothing to hide behind them), and no multi-GPU box was available to measure it where the bucket's
    # all-reduce takes ~100 us: off unless COOT_DP_EARLY=1.
    dp_early_global_update = os.environ.get("COOT_DP_EARLY", "0") == "1"
    lookahead_min_stage_bytes = 32 << 20  # train_step_native(next_batch=): only batches whose normalised features reach this size
    _stage_owner = None  # id() of the trainer whose input stages the library currently holds (thread-local there; one training thread here)

    def __init__(self, cfg: RetrievalConfig, model_mgr: RetrievalModelManager, is_test: bool = False,
                 world_size: int = 1):
        self.cfg = cfg
        self.model_mgr = model_mgr
        self.loss_cfg = loss_fn.ContrastiveLossConfig.from_section(cfg.train.contrastive_loss_config)
        self.loss_contr = loss_fn.ContrastiveLoss(self.loss_cfg.margin)
        self.world_size = world_size
        self.optimizer = None
        if not is_test:
            params, _names, _flat = model_mgr.get_all_params()
            on_gpu = any(p.is_cuda for p in _flat)
            self.optimizer = make_optimizer(cfg.optimizer, params, capturable=on_gpu)  # capturable: HIP-graph safe
        self.cc_generator: Optional[torch.Generator] = None
        self.total_step = 0
        # epoch-loop state (nntrainer/trainer_configs.py:16-40, the fields train_model needs; kept in memory — checkpoint files,
        # metric meters and tensorboard are the reference's control plane and stay there)
        self.current_epoch = 0
        self.det_best_field_current: float = 0.0
        self.det_best_field_best: Optional[float] = None
        self.infos_val_epochs: list = []
        self.infos_val_is_good: list = []
        self.lr_scheduler = None

    # ---- ownership of what the library retains (include/coot_hip.h: "Retained pointers") ------------------------------------------
    def close(self) -> None:
        """Withdraws every pointer of THIS trainer that the library keeps between calls — the input stages and a pending next-batch
        announcement (thread-local), the deterministic table (process-global) — and drops the captured native steps, whose nodes
        carry this trainer's buffers.  Called by __del__; a caller that frees the trainer's tensors earlier calls it first.
        Idempotent; a trainer that never ran a native step has nothing registered."""
        st = self.__dict__.get("_native")
        if st is None and not self.__dict__.get("deterministic", False) and not self.__dict__.get("_side_cs"):
            return
        try:
            lib = _lib.load()
            if torch.cuda.is_available():
                torch.cuda.synchronize()  # (the library's internal stream may still be writing a stage)
            if st is not None:
                if st.__dict__.get("graphs"):
                    st.graphs.clear()
                if getattr(st, "stages", None) is not None:
                    # the registration is thread-local and one per thread: withdraw it only if it is still this trainer's (another
                    # trainer of the thread may have registered its own stages since)
                    if RetrievalTrainer._stage_owner == id(self):
                        lib.coot_step_set_next_batch(None, None)
                        lib.coot_step_set_input_stages(None, None, 0)
                        RetrievalTrainer._stage_owner = None
                    st.stages = None
                for name in ("text_cs", "comm_cs"):
                    cs = st.__dict__.get(name)
                    if cs is not None:
                        cs.close()
                        setattr(st, name, None)
                st.comm = None
            for cs in self.__dict__.get("_side_cs", ()):
                cs.close()
            self._side_cs, self._side_streams = (), None
            self._next_desc = None
            if self.__dict__.get("deterministic", False):
                self.set_deterministic(False)
        except Exception:  # (interpreter shutdown: the library or torch may already be gone)
            pass

    def __del__(self):
        self.close()

    # ---- loss hooks ------------------------------------------------------------------------------------
    def compute_align_loss(self, visual_emb: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
        return self.loss_contr(visual_emb, text_emb)

    def compute_cluster_loss(self, visual_emb: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
        return (self.loss_contr(visual_emb, visual_emb) + self.loss_contr(text_emb, text_emb)) / 2

    def compute_total_constrastive_loss(self, visual_data: RetrievalVisualEmbTuple,
                                        text_data: RetrievalTextEmbTuple) -> torch.Tensor:
        # networks in the fp32 reference mode (TransformerHip.set_compute_dtype("f32")) take the fp32 mode of the loss with them: the
        # checker then runs forward, loss and backward in the library in fp32 end to end (tests/test_gpu_f32_mode.py)
        dts = {getattr(net.cfg, "dtype", "bf16") for net in self.model_mgr.model_dict.values()}
        return loss_fn.total_contrastive_loss(self.loss_cfg, visual_data.vid_emb, text_data.par_emb, visual_data.clip_emb,
                                              text_data.sent_emb, visual_data.vid_context, text_data.par_context,
                                              dtype="f32" if dts == {"f32"} else "bf16")

    def compute_cyclecons_loss(self, visual_data: RetrievalVisualEmbTuple, text_data: RetrievalTextEmbTuple,
                               idx_clip: Optional[torch.Tensor] = None, idx_sent: Optional[torch.Tensor] = None):
        w = float(self.cfg.train.loss_cycle_cons)
        if w == 0:
            return 0
        if idx_clip is None:
            idx_clip = loss_fn.sample_cycle_indices(visual_data.clip_emb_lens, self.cc_generator)
        if idx_sent is None:
            idx_sent = loss_fn.sample_cycle_indices(text_data.sent_emb_lens, self.cc_generator)
        return loss_fn.cycle_consistency_loss(visual_data.clip_emb_reshape, visual_data.clip_emb_lens,
                                              text_data.sent_emb_reshape, text_data.sent_emb_lens, w, idx_clip, idx_sent)

    # ---- video side and text side are independent until the loss: run them on two HIP streams -------------------
    overlap_sides = True

    def encode_both(self, batch: RetrievalDataBatchTuple):
        """encode_visual and encode_text (coot/trainer_retrieval.py:265-266) on two side streams so the small
        text-side kernels fill the CUs the video side leaves idle.  Autograd replays each backward node on the
        stream its forward ran on, so the backward passes overlap the same way."""
        if not (self.overlap_sides and batch.vid_feat.is_cuda):
            return self.model_mgr.encode_visual(batch), self.model_mgr.encode_text(batch)
        if getattr(self, "_side_streams", None) is None:  # two streams verified to overlap each other (_lib.ConcurrentStream)
            a = _lib.ConcurrentStream([])
            self._side_cs = (a, _lib.ConcurrentStream([a]))
            self._side_streams = tuple(c.torch for c in self._side_cs)
        main = torch.cuda.current_stream()
        sv, st = self._side_streams
        sv.wait_stream(main)
        st.wait_stream(main)
        with torch.cuda.stream(sv):
            visual_data = self.model_mgr.encode_visual(batch)
        with torch.cuda.stream(st):
            text_data = self.model_mgr.encode_text(batch)
        main.wait_stream(sv)
        main.wait_stream(st)
        return visual_data, text_data

    def _join_side_streams(self) -> None:
        if getattr(self, "_side_streams", None) is not None:
            main = torch.cuda.current_stream()
            for s in self._side_streams:
                main.wait_stream(s)

    # ---- one optimisation step (coot/trainer_retrieval.py:253-291) ---------------------------------------
    def train_step(self, batch: RetrievalDataBatchTuple, vid_counts=None, clip_counts=None
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """forward + losses + backward (+ gradient all-reduce) + optimizer step; returns (loss, contr_loss,
        cc_loss) as 0-dim device tensors (no host sync; the reference's per-step .item() logging is the
        caller's choice).  With ``self.dp`` set (dist.DataParallelContext) the batch is this rank's shard.
        This is the AUTOGRAD route (torch autograd + torch.optim around the library's per-network calls): always eager.  A captured
        (hipGraph) step exists on the native route only — train_step_native(use_graph=True), one C call inside the capture; the
        autograd route's whole-step capture (rounds 1-5) was removed in round 6: its replay crashed inside hipGraphLaunch under one
        test ordering and was never root-caused (docs/NOTEBOOK_r1-r5.md 12), and nothing defaulted to it."""
        return self._step_impl(batch, vid_counts, clip_counts)

    def _step_prepare_seed(self, batch, nets) -> None:
        self._seed_dev = torch.zeros(1, dtype=torch.int64, device=batch.vid_feat.device)
        for net in nets:
            net.seed_dev = self._seed_dev

    def _sync_global_max(self, batch) -> None:
        """Data parallel: every rank pads its packed clip / sentence embeddings to the GLOBAL max count per video (one
        all-reduce(MAX) of two ints per batch) — avg_special pooling sums the padded rows (nntrainer/models/poolers.py:237), so the
        global embeddings depend on Cmax and must see the full batch's value.  Fixed-shape callers set
        batch.global_max_synced themselves and skip the collective."""
        dp = getattr(self, "dp", None)
        if dp is None or getattr(batch, "global_max_synced", False):
            return
        if batch.max_clip_num is None or batch.max_sent_num is None:
            batch.max_clip_num, batch.max_sent_num = int(batch.clip_num.max()), int(batch.sent_num.max())
        batch.max_clip_num, batch.max_sent_num = dp.global_max_pair(batch.max_clip_num, batch.max_sent_num, batch.clip_num.device)
        batch.global_max_synced = True

    def _dp_batch_shapes(self, batch, vid_counts=None, clip_counts=None):
        """Data parallel: what every rank must know about the GLOBAL batch before it can size its step — videos and clips per rank
        (the gathered embedding blocks) and the global max clips / sentences per video (_sync_global_max) — in ONE collective on
        host integers (dist.DataParallelContext.exchange_shapes): nothing here waits for the device.  Callers with fixed shapes
        pass the counts and set batch.global_max_synced: no collective at all."""
        dp = self.dp
        synced = getattr(batch, "global_max_synced", False)
        if vid_counts is not None and clip_counts is not None and synced:
            return list(vid_counts), list(clip_counts)
        if not hasattr(dp, "exchange_shapes"):  # (a context that only implements the device collectives)
            self._sync_global_max(batch)
            dev = batch.clip_num.device
            return (vid_counts if vid_counts is not None else dp.global_counts(int(batch.clip_num.shape[0]), dev),
                    clip_counts if clip_counts is not None else dp.global_counts(int(batch.clip_feat_len.shape[0]), dev))
        if batch.max_clip_num is None or batch.max_sent_num is None:  # (collate_fn sets them on the host; a hand-made batch pays a device sync)
            batch.max_clip_num, batch.max_sent_num = int(batch.clip_num.max()), int(batch.sent_num.max())
        rows = dp.exchange_shapes([int(batch.clip_num.shape[0]), int(batch.clip_feat_len.shape[0]), batch.max_clip_num, batch.max_sent_num])
        if not synced:
            batch.max_clip_num, batch.max_sent_num = max(r[2] for r in rows), max(r[3] for r in rows)
            batch.global_max_synced = True
        return (list(vid_counts) if vid_counts is not None else [r[0] for r in rows],
                list(clip_counts) if clip_counts is not None else [r[1] for r in rows])

    def _step_impl(self, batch, vid_counts=None, clip_counts=None):
        self.join_streams()  # (a native step with defer_join may still be updating the text networks on its side stream)
        if getattr(self, "dp", None) is not None:
            vid_counts, clip_counts = self._dp_batch_shapes(batch, vid_counts, clip_counts)
        nets = list(self.model_mgr.model_dict.values())
        if getattr(self, "_seed_dev", None) is None:
            self._step_prepare_seed(batch, nets)
        self._seed_dev += 1  # device-side dropout seed: advances on every step, also under graph replay
        flat_grads = [net.bind_flat_grads() for net in nets]
        det_extra = self._det_extra(flat_grads[0].device)
        self._det_sync(flat_grads + det_extra)
        for net, g in zip(nets, flat_grads):
            g.zero_()
            net.accumulate_into_flat = True  # backward kernels accumulate straight into the flat arenas
        visual_data, text_data = self.encode_both(batch)
        dp = getattr(self, "dp", None)
        if dp is None:
            contr_loss = self.compute_total_constrastive_loss(visual_data, text_data)
            cc_loss = self.compute_cyclecons_loss(visual_data, text_data)
        else:
            *sets, global_b = dp.gather_embeddings(visual_data, text_data, clip_counts, vid_counts)
            contr_loss = loss_fn.total_contrastive_loss(self.loss_cfg, *sets)
            w = float(self.cfg.train.loss_cycle_cons)
            cc_loss = 0
            if w != 0:
                cc_loss = loss_fn.cycle_consistency_loss(
                    visual_data.clip_emb_reshape, visual_data.clip_emb_lens, text_data.sent_emb_reshape,
                    text_data.sent_emb_lens, w, loss_fn.sample_cycle_indices(visual_data.clip_emb_lens, self.cc_generator),
                    loss_fn.sample_cycle_indices(text_data.sent_emb_lens, self.cc_generator), global_batch=global_b)
        loss = contr_loss + cc_loss
        loss.backward()
        self._join_side_streams()
        self._det_flush(flat_grads)
        for net in nets:
            net.accumulate_into_flat = False
        if dp is not None:
            dp.allreduce_grads(flat_grads, getattr(self, "comm_stream", None))
        self.optimizer.step()
        self.model_mgr.mark_weights_dirty()
        if not torch.cuda.is_current_stream_capturing():
            self.total_step += 1
        return loss.detach(), contr_loss.detach(), (cc_loss.detach() if torch.is_tensor(cc_loss) else torch.zeros_like(loss))

    # ---- native step: the whole optimisation step sequenced inside libcoot_hip.so ---------------------------------
    def _native_setup(self, batch: RetrievalDataBatchTuple):
        lib = _lib.load()
        nets = [self.model_mgr.model_dict[k] for k in RetrievalNetworksConst.values()]
        dev = batch.vis_tokens.device if isinstance(batch, RetrievalPackedBatchTuple) else batch.vid_feat.device
        st = getattr(self, "_native", None)
        if st is None:
            st = type("NativeState", (), {})()
            st.nets = nets
            st.cfg = _lib.StepConfig()
            for i, net in enumerate(nets):
                st.cfg.net[i] = net.c_cfg
                net.bind_flat_grads()
                net.ensure_packed()
            st.cfg.contr = self.loss_cfg.to_c()
            st.cfg.cc_weight = float(self.cfg.train.loss_cycle_cons)
            o = self.cfg.optimizer
            st.cfg.lr, st.cfg.beta1, st.cfg.beta2 = float(o.lr), float(o.momentum), float(o.adam_beta2)
            st.cfg.eps, st.cfg.weight_decay = float(o.adam_eps), float(o.weight_decay)
            oname = getattr(o, "name", "adam")
            if oname not in ("adam", "radam"):
                raise NotImplementedError(f"Unknown optimizer {oname}")
            st.cfg.optimizer = 1 if oname == "radam" else 0
            st.cfg.radam_degentosgd = int(bool(getattr(o, "radam_degentosgd", False)))
            wd_bias = bool(getattr(o, "weight_decay_for_bias", False))
            st.m = [torch.zeros_like(n._flat) for n in nets]
            st.v = [torch.zeros_like(n._flat) for n in nets]
            st.decay = []
            for n in nets:  # decay_mult = 0 for biases when weight_decay_for_bias (model_manager_base.py:152-154, sic)
                mask = torch.ones_like(n._flat)
                for (name, off, shape) in n.table:
                    if wd_bias and "bias" in name:
                        mask[off:off + int(np.prod(shape))] = 0.0
                st.decay.append(mask)
            # one byte per 1 024 elements: "the mask is 1.0 on this whole block" — the update launch then skips the mask there
            # (coot_step_buffers.decay_block_all; the mask is 0 only on bias vectors)
            st.decay_all = []
            for mask in st.decay:
                nb = (mask.numel() + 1023) // 1024
                pad = torch.ones(nb * 1024, dtype=mask.dtype, device=mask.device)
                pad[:mask.numel()] = mask
                st.decay_all.append((pad.view(nb, 1024).min(dim=1).values == 1.0).to(torch.uint8).contiguous())
            st.bufs = _lib.StepBuffers()
            st.losses = torch.zeros(3, dtype=torch.float32, device=dev)
            # text side: a stream that is VERIFIED to run
            # beside the caller's stream (which carries the video side): _lib.ConcurrentStream.  A stream taken on trust may share a
            # hardware queue with it (HIP maps streams to 4 queues in creation order) and the two sides then run one after the other.
            st.text_cs = st.comm_cs = None
            self._bind_text_stream(st, torch.cuda.current_stream())
            st.step = 0
            st.dims_key = None
            self._native = st
            pending = self.__dict__.pop("_native_pending", None)
            if pending is not None:  # optimizer state loaded before the first native step (load_optimizer_state_dict)
                st.step = int(pending["step"])
                for dst, src in zip(st.m, pending["m"]):
                    dst.copy_(src)
                for dst, src in zip(st.v, pending["v"]):
                    dst.copy_(src)
            elif self.__dict__.pop("_native_seed_from_torch", False):
                self._seed_native_from_torch_optimizer(st)
        for i, n in enumerate(nets):  # arenas move when a module is re-flattened (.cuda()/load): refresh every call
            st.bufs.params[i], st.bufs.grads[i], st.bufs.wpack[i] = n._flat.data_ptr(), n._grad_flat.data_ptr(), n._wpack.data_ptr()
            st.bufs.adam_m[i], st.bufs.adam_v[i], st.bufs.decay_mask[i] = st.m[i].data_ptr(), st.v[i].data_ptr(), st.decay[i].data_ptr()
            st.bufs.decay_block_all[i] = st.decay_all[i].data_ptr()
            st.bufs.pe[i] = n.embedding.pe.data_ptr()
        kept = getattr(self, "_next_desc", None)
        # this very object was announced as `next_batch` by the previous native step (COOT_STEP_STAGE_ANNOUNCED: only then may the step
        # use the x^ that step prepared — equal pointers and shapes do not identify a batch, arena slots are refilled in place)
        st.batch_announced = kept is not None and kept[0] is batch
        key, dims_args, x, held = self._native_describe(batch)
        # `x` is raw device pointers.  For a packed-source batch outside the packed-row kernels' domain they point into the padded copy
        # _native_describe made (its 4th return value): that copy — or the caller's batch — stays referenced from here until the NEXT step's
        # setup, i.e. past this step's launches on every stream (the main stream of the next step is ordered behind this step's forward on
        # both sides), so the caching allocator cannot hand its blocks to an allocation that runs ahead of the launch.
        st.batch_ref = held
        if key != st.dims_key:
            st.dims = _lib.StepDims(*dims_args)
            need = lib.coot_step_workspace_bytes(C.byref(st.cfg), C.byref(st.dims))
            if getattr(st, "ws", None) is None or st.ws.numel() < need:  # ragged batches change shape every step: grow only
                if getattr(st, "ws", None) is not None:
                    torch.cuda.synchronize()  # (the text stream may still work on the old workspace; the allocator only tracks the current stream)
                st.ws = torch.empty(int(need * 1.1), dtype=torch.uint8, device=dev)
            st.dims_key = key
        return st, x

    def _native_describe(self, batch):
        """(shape key, coot_step_dims arguments, coot_step_batch, the batch object whose tensors the pointers refer to) of a batch.
        The description of the batch announced as `next_batch` of the previous native step is kept (same object -> same pointers:
        what coot_train_step compares to know that the batch's input LayerNorm already ran)."""
        kept = getattr(self, "_next_desc", None)
        if kept is not None and kept[0] is batch:
            return kept[1]
        src_packed = isinstance(batch, RetrievalPackedBatchTuple)  # packed at the source (dataset_retrieval.collate_fn(packed=True))
        if src_packed and (max(batch.max_lens) > 128 or min(batch.tok_vis, batch.tok_txt) < 1024):
            # the packed-row kernels cover sequences of <= 128 rows and launches of >= 1 024 tokens (csrc/api.hip: packed_ok); a
            # packed matrix has no padded tensor to fall back to, so such a batch is unpacked on the device first
            from .dataset_retrieval import unpack_batch
            batch = unpack_batch(batch)
            src_packed = False
        if batch.max_clip_num is None or batch.max_sent_num is None:
            batch.max_clip_num, batch.max_sent_num = int(batch.clip_num.max()), int(batch.sent_num.max())
        packed = getattr(batch, "cu_vis", None) is not None and getattr(batch, "cu_txt", None) is not None
        tok_vis, tok_txt = (int(batch.tok_vis), int(batch.tok_txt)) if packed else (0, 0)
        if src_packed:
            B, Nc = batch.vid_feat_len.numel(), batch.clip_feat_len.numel()
            Lv, Lc, Lp, Ls = (int(v) for v in batch.max_lens)
            source = _lib.SOURCE_PACKED_BF16 if batch.vis_tokens.dtype == torch.bfloat16 else _lib.SOURCE_PACKED_F32
            assert batch.txt_tokens.dtype == batch.vis_tokens.dtype and batch.sent_feat_len.numel() == Nc and batch.par_feat_len.numel() == B
            key = ("packed", B, Nc, Lv, Lc, Lp, Ls, batch.max_clip_num, batch.max_sent_num, tok_vis, tok_txt, source)
        else:
            B, Lv, _ = batch.vid_feat.shape
            Nc, Lc, _ = batch.clip_feat.shape
            Lp, Ls, source = batch.par_feat.shape[1], batch.sent_feat.shape[1], _lib.SOURCE_PADDED
            assert batch.sent_feat.shape[0] == Nc and batch.par_feat.shape[0] == B
            key = (batch.vid_feat.shape, batch.clip_feat.shape, batch.par_feat.shape, batch.sent_feat.shape, batch.max_clip_num, batch.max_sent_num,
                   tok_vis, tok_txt)
        dims_args = (B, Nc, Lv, Lc, Lp, Ls, batch.max_clip_num, batch.max_sent_num, tok_vis, tok_txt, source)
        x = _lib.StepBatch()
        if src_packed:
            assert batch.vis_tokens.is_contiguous() and batch.txt_tokens.is_contiguous() and batch.vis_tokens.shape[0] == tok_vis
            x.vid_feat, x.par_feat = batch.vis_tokens.data_ptr(), batch.txt_tokens.data_ptr()
        else:
            for f in ("vid_feat", "clip_feat", "par_feat", "sent_feat"):
                t_ = getattr(batch, f)
                assert t_.dtype == torch.float32 and t_.is_contiguous(), f
                setattr(x, f, t_.data_ptr())
        for f, src in (("vid_len", "vid_feat_len"), ("clip_len", "clip_feat_len"), ("par_len", "par_feat_len"),
                       ("sent_len", "sent_feat_len"), ("clip_num", "clip_num"), ("sent_num", "sent_num")):
            t_ = getattr(batch, src)
            assert t_.dtype == torch.int64 and t_.is_contiguous(), src
            setattr(x, f, t_.data_ptr())
        if packed:  # valid tokens only through the local networks (coot_packed_seqs)
            assert batch.cu_vis.dtype == torch.int32 and batch.cu_vis.numel() == B + Nc + 1 and batch.cu_vis.is_cuda
            x.cu_vis, x.cu_txt = batch.cu_vis.data_ptr(), batch.cu_txt.data_ptr()
        return key, dims_args, x, batch

    def _announce_next_batch(self, lib, st, batch, next_batch) -> bool:
        """Input stages of the native step (include/coot_hip.h: COOT_STEP_INPUT_STAGES): two device buffers for the normalised input
        features; this step normalises `next_batch` (the data loader's lookahead) into the one it does not use, behind its local forward
        passes.  Returns whether the step runs with the stages (False: nothing to announce and no stages yet)."""
        if next_batch is None and getattr(st, "stages", None) is None:
            return False
        need = lib.coot_step_input_stage_bytes(C.byref(st.cfg), C.byref(st.dims))
        if getattr(st, "stages", None) is None and need < self.lookahead_min_stage_bytes:
            # small inputs: the LayerNorm is a few microseconds and the third stream costs more than it saves (measured: the 16-video
            # YouCook2 batch 0.763 against 0.737 ms per step with the lookahead; ActivityNet shapes 1.217 against 1.238)
            self._next_desc = None
            return False
        nd = None
        if next_batch is not None:
            desc = self._native_describe(next_batch)
            # keeps the object, its description and a possibly unpacked copy alive until its own step (the CURRENT batch's copy, if it came
            # out of the previous announcement, is held by st.batch_ref: replacing _next_desc here does not free it)
            self._next_desc = (next_batch, desc)
            nd = _lib.StepDims(*desc[1])
            need = max(need, lib.coot_step_input_stage_bytes(C.byref(st.cfg), C.byref(nd)))
        else:
            self._next_desc = None
        if getattr(st, "stages", None) is None or st.stages[0].numel() < need:  # (a change of buffers resets what the stages hold)
            if getattr(st, "stages", None) is not None:
                # the library's prefetch stream may still be writing the old stages, and the caching allocator does not know that stream:
                # drain the device before the old buffers go back to it (rare: the stages only grow, with 10 % slack)
                torch.cuda.synchronize()
            dev = st.ws.device
            st.stages = (torch.empty(int(need * 1.1), dtype=torch.uint8, device=dev), torch.empty(int(need * 1.1), dtype=torch.uint8, device=dev))
        _lib.check(lib.coot_step_set_input_stages(st.stages[0].data_ptr(), st.stages[1].data_ptr(), st.stages[0].numel()), "coot_step_set_input_stages")
        RetrievalTrainer._stage_owner = id(self)  # (close() withdraws the registration only while it is still this trainer's)
        if next_batch is not None:
            _lib.check(lib.coot_step_set_next_batch(C.byref(desc[2]), C.byref(nd)), "coot_step_set_next_batch")
        else:
            _lib.check(lib.coot_step_set_next_batch(None, None), "coot_step_set_next_batch")
        return True

    # ---- deterministic mode (include/coot_hip.h: coot_det_configure; tests_nntrainer/integration_deter.py:18-66) -----------------
    def set_deterministic(self, on: bool = True) -> None:
        """Run-to-run determinism of the training step: the few fp32 atomic accumulations of the library (bias / LayerNorm parameter
        gradients, the cycle-consistency loss word) go through order-independent fixed-point accumulators.  Two runs of the same steps
        from the same state are then bit-identical (parameters, losses); the step computes the same numbers as without it to fp32
        round-off.  Process-wide (the library's mode is); costs two extra launches and ~60 MB of accumulators per step."""
        if bool(on) != getattr(self, "deterministic", False):
            self._drop_graphs()  # captured steps carry the mode they were captured in (flush nodes, the shadow's address)
        self.deterministic = bool(on)
        if not on:
            loss_fn.set_det_loss_word(None)  # (module-global in loss_fn: must not outlive the mode it belongs to)
        if not on and getattr(self, "_det_key", None) is not None:
            torch.cuda.synchronize()  # (no launch of any stream may still add into — or flush — the shadow that is freed below)
            _lib.check(_lib.load().coot_det_configure(0, None, None, None, 0, torch.cuda.current_stream().cuda_stream), "coot_det_configure")
            self._det_key, self._det_shadow, self._det_ranges = None, None, []

    def det_bypass_count(self) -> int:
        """Deterministic mode: how many gradient / loss addends since the mode was configured were NaN, Inf or >= 2^22 and therefore took the
        plain float atomic instead of the fixed-point shadow (csrc/det.h) — a run with a non-zero count is not bit-reproducible.
        Synchronises the device; -1 while the mode is off."""
        v = C.c_int(0)
        _lib.check(_lib.load().coot_get_option(b"det_bypasses", C.byref(v)), "coot_get_option")
        return int(v.value)

    def _drop_graphs(self) -> None:
        """Forgets every captured native step.  The library consults its process-wide deterministic table
        at RUN time and a captured step holds the flush launches — or their absence — and the shadow's address of the mode it was
        captured in: a replay under another configuration would drop gradient addends (no flush node) or read a freed shadow."""
        st = getattr(self, "_native", None)
        if st is not None and st.__dict__.get("graphs"):
            torch.cuda.synchronize()
            st.graphs.clear()

    def _det_extra(self, device) -> list:
        """The fp32 words outside the gradient arenas that several workgroups add into on the autograd route: the cycle-consistency
        loss word (loss_fn._CycleConsFn) — a persistent word, registered with the arenas."""
        if not getattr(self, "deterministic", False):
            loss_fn.set_det_loss_word(None)
            return []
        w = getattr(self, "_det_loss_word", None)
        if w is None or w.device != device:
            w = self._det_loss_word = torch.zeros(1, dtype=torch.float32, device=device)
        loss_fn.set_det_loss_word(w)
        return [w]

    def _det_state(self):
        """What a captured step depends on besides shapes: the deterministic mode and the configuration it was captured under."""
        return (bool(getattr(self, "deterministic", False)), getattr(self, "_det_epoch", 0))

    def _det_sync(self, tensors) -> None:
        """(Re)registers the fp32 tensors the step accumulates into — the gradient arenas and the loss words — when they changed."""
        if not getattr(self, "deterministic", False):
            return
        ranges = sorted({(t.data_ptr(), t.numel() * 4) for t in tensors})
        if ranges == getattr(self, "_det_key", None):
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("deterministic mode must be configured before a step is captured (the registered ranges changed inside a capture)")
        self._drop_graphs()
        self._det_epoch = getattr(self, "_det_epoch", 0) + 1
        lib = _lib.load()
        n = len(ranges)
        bases = (C.c_void_p * n)(*[r[0] for r in ranges])
        sizes = (C.c_size_t * n)(*[r[1] for r in ranges])
        nbytes = int(lib.coot_det_shadow_bytes(n, sizes))
        torch.cuda.synchronize()  # (the old shadow may still be in use on a side stream)
        self._det_shadow = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        _lib.check(lib.coot_det_configure(n, bases, sizes, self._det_shadow.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream),
                   "coot_det_configure")
        self._det_key, self._det_ranges = ranges, list(tensors)

    def _det_flush(self, tensors) -> None:
        if getattr(self, "deterministic", False):
            lib, sp = _lib.load(), torch.cuda.current_stream().cuda_stream
            for t in tensors:
                _lib.check(lib.coot_det_flush(t.data_ptr(), t.numel() * 4, sp), "coot_det_flush")

    # ---- optimizer state of either path, for optimizer_<epoch>.pth (nntrainer/trainer_base.py:685-707) ------------------------
    def get_opt_state(self) -> Dict[str, Any]:
        """The reference's name for it (nntrainer/trainer_base.py:251-261)."""
        return self.optimizer_state_dict()

    def set_opt_state(self, opt_state: Dict[str, Any]) -> None:
        """nntrainer/trainer_base.py:263-271; also takes a reference-written optimizer_<epoch>.pth ({"optimizer", "lr_scheduler"})."""
        self.load_optimizer_state_dict(opt_state)

    def optimizer_state_dict(self) -> Dict[str, Any]:
        """What a checkpoint must hold to resume training: ``self.optimizer.state_dict()`` (the autograd path's torch optimizer)
        and, when native steps have run, the library's optimizer state — the flat first / second moment arenas of the four
        networks and the step count (coot_train_step keeps them outside torch.optim, so the torch state dict alone would
        resume with zero moments and bias-correction step 1)."""
        self.join_streams()
        out: Dict[str, Any] = {"optimizer": self.optimizer.state_dict() if self.optimizer is not None else None, "total_step": self.total_step}
        if self.lr_scheduler is not None:  # the reference's second key (nntrainer/trainer_base.py:251-261: get_opt_state)
            out["lr_scheduler"] = self.lr_scheduler.state_dict()
        st = getattr(self, "_native", None)
        if st is not None:
            out["native"] = {"step": int(st.step), "m": [t.detach().cpu().clone() for t in st.m], "v": [t.detach().cpu().clone() for t in st.v]}
        return out

    def load_optimizer_state_dict(self, state: Dict[str, Any], batch_for_setup: Optional[RetrievalDataBatchTuple] = None) -> None:
        """Inverse of optimizer_state_dict.  The native state is created lazily by the first native step; when it does not exist
        yet the loaded moments are kept and installed by _native_setup."""
        if state.get("optimizer") is not None and self.optimizer is not None:
            self.optimizer.load_state_dict(state["optimizer"])
        if state.get("lr_scheduler") is not None and self.lr_scheduler is not None:  # (a plain reference opt_state has exactly these two keys)
            self.lr_scheduler.load_state_dict(state["lr_scheduler"])
        self.total_step = int(state.get("total_step", self.total_step))
        nat = state.get("native")
        if nat is None:
            # a state the REFERENCE wrote (optimizer_<epoch>.pth = {"optimizer", "lr_scheduler"}, nntrainer/trainer_base.py:251-261) or one
            # of this trainer's autograd route: the native step keeps its moments outside torch.optim, so they are seeded from the torch
            # optimizer's exp_avg / exp_avg_sq / step through the flat parameter layout — a native run resumed from such a state
            # continues Adam's moments and bias correction instead of silently restarting them next to a continuing LR schedule
            if state.get("optimizer") is not None and self.optimizer is not None:
                st = getattr(self, "_native", None)
                if st is None:
                    self._native_seed_from_torch = True
                else:
                    self._seed_native_from_torch_optimizer(st)
            return
        st = getattr(self, "_native", None)
        if st is None:
            self._native_pending = nat
            return
        st.step = int(nat["step"])
        for dst, src in zip(st.m, nat["m"]):
            dst.copy_(src)
        for dst, src in zip(st.v, nat["v"]):
            dst.copy_(src)

    def _seed_native_from_torch_optimizer(self, st) -> int:
        """Copies exp_avg / exp_avg_sq / step of self.optimizer's per-parameter state into the native step's flat moment arenas (the
        parameters are views of each network's flat arena at the offsets of net.table).  Returns the number of parameters seeded;
        parameters the torch optimizer holds no state for keep zero moments."""
        if self.optimizer is None:
            return 0
        seeded, steps = 0, []
        for i, net in enumerate(st.nets):
            for (_name, off, shape), p in zip(net.table, net._params):
                ps = self.optimizer.state.get(p)
                if not ps or "exp_avg" not in ps or "exp_avg_sq" not in ps:
                    continue
                n = int(np.prod(shape))
                st.m[i][off:off + n].copy_(ps["exp_avg"].reshape(-1))
                st.v[i][off:off + n].copy_(ps["exp_avg_sq"].reshape(-1))
                steps.append(int(ps["step"]) if "step" in ps else 0)
                seeded += 1
        if seeded:
            if len(set(steps)) > 1:
                raise ValueError(f"optimizer state: parameters at different step counts {sorted(set(steps))}; the native step keeps one")
            st.step = steps[0]
        return seeded

    # ---- the native step as a replayed hipGraph ----------------------------------------------------------------------
    _GRAPH_FEATS = ("vid_feat", "clip_feat", "par_feat", "sent_feat")
    _GRAPH_LENS = ("vid_feat_len", "clip_feat_len", "par_feat_len", "sent_feat_len", "clip_num", "sent_num")

    def _graph_state_bytes(self, st, lr: float) -> bytes:
        import struct
        seed = (torch.initial_seed() * 1000003 + 7919 * self.total_step) & 0xFFFFFFFFFFFFFFFF  # the kernel adds 7919 first
        return struct.pack("<QQfi", seed, int(st.step), float(lr), 0)

    def _train_step_native_graph(self, batch: RetrievalDataBatchTuple):
        """coot_train_step captured ONCE per batch shape and replayed (torch.cuda.CUDAGraph around the C call): a dependent
        launch costs 1.7 us in a replay against 3.1 us launched one by one (tools/micro/launchgap.hip), and the step is a
        chain of ~150 of them.  Per-step scalars (dropout seed, optimizer step count and its scalars, learning rate) live in a
        device state block the step's first node advances (coot_step_set_device_state), the batch is copied into static
        buffers.  Same masks, same update as the eager native step (tests/test_gpu_path.py).  Falls back to the eager call
        for the first step of a shape (lazy state is created outside a capture)."""
        lib = _lib.load()
        if getattr(batch, "cu_vis", None) is not None or isinstance(batch, RetrievalPackedBatchTuple):
            # packed (cu_seqlens) batches carry their token totals as host scalars of the launches and change them every batch:
            # the captured step would replay the first batch's totals.  They run eagerly (the caller falls back on None).
            return None
        st, x = self._native_setup(batch)
        lr = float(self.optimizer.param_groups[0]["lr"]) if self.optimizer is not None else float(st.cfg.lr)
        ptrs = tuple(int(st.bufs.params[i]) for i in range(4)) + tuple(int(st.bufs.wpack[i]) for i in range(4))
        # deterministic mode: configured here, BEFORE any capture (a reconfiguration drops every cached graph, _det_sync); the key carries
        # the mode and its configuration epoch, so a graph captured under another one is never replayed
        self._det_sync([n._grad_flat for n in st.nets] + [st.losses])
        key = (st.dims_key, ptrs, st.ws.data_ptr(), self._det_state())
        graphs = st.__dict__.setdefault("graphs", {})
        g = graphs.get(key)
        if g is None:
            if st.step == 0 or not all(n.pack_is_fresh() for n in st.nets):
                return None  # caller runs the eager step (creates lazy state, leaves fresh packs); captured on the next call
            g = type("NativeGraph", (), {})()
            dev = batch.vid_feat.device
            nbytes = int(lib.coot_step_device_state_bytes())
            assert nbytes >= 24
            g.state = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            g.static = {f: getattr(batch, f).clone() for f in self._GRAPH_FEATS + self._GRAPH_LENS}
            g.x = _lib.StepBatch()
            for f in self._GRAPH_FEATS:
                setattr(g.x, f, g.static[f].data_ptr())
            for f, src in (("vid_len", "vid_feat_len"), ("clip_len", "clip_feat_len"), ("par_len", "par_feat_len"),
                           ("sent_len", "sent_feat_len"), ("clip_num", "clip_num"), ("sent_num", "sent_num")):
                setattr(g.x, f, g.static[src].data_ptr())
            g.dims = _lib.StepDims(*[getattr(st.dims, f[0]) for f in st.dims._fields_])
            g.cfg = _lib.StepConfig.from_buffer_copy(st.cfg)
            g.cfg.lr = 0.0  # unused: the learning rate is read from the state block
            g.mirror = None
            g.last_batch, g.last_versions = None, None
            flags = _lib.STEP_OPTIMIZER | _lib.STEP_REPACK | _lib.STEP_PACKS_FRESH
            torch.cuda.synchronize()
            g.flags = flags

            lib.coot_step_set_device_state(g.state.data_ptr())
            try:
                g.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g.graph):
                    cur = torch.cuda.current_stream()
                    _lib.check(lib.coot_train_step(C.byref(g.cfg), C.byref(st.bufs), C.byref(g.x), C.byref(g.dims), st.losses.data_ptr(),
                                                   st.ws.data_ptr(), st.ws.numel(), 1, 0, 1, flags, cur.cuda_stream, cur.cuda_stream,
                                                   st.streams[1].cuda_stream), "coot_train_step (capture)")
            finally:
                lib.coot_step_set_device_state(None)
            graphs[key] = g
        if not all(n.pack_is_fresh() for n in st.nets):
            return None
        # inputs -> static buffers (skipped when the caller passes the very same, unmodified batch object again)
        versions = tuple(getattr(batch, f)._version for f in self._GRAPH_FEATS + self._GRAPH_LENS)
        if batch is not g.last_batch or versions != g.last_versions:
            for f in self._GRAPH_FEATS + self._GRAPH_LENS:
                g.static[f].copy_(getattr(batch, f), non_blocking=True)
            g.last_batch, g.last_versions = batch, versions
        # device counters: rewritten only when the host moved on without the graph (eager steps in between) or the LR changed
        want = (self.total_step, st.step, lr)
        if g.mirror != want:
            hdr = torch.frombuffer(bytearray(self._graph_state_bytes(st, lr)), dtype=torch.uint8)
            g.state[:24].copy_(hdr)
        g.graph.replay()
        st.step += 1
        self.total_step += 1
        g.mirror = (self.total_step, st.step, lr)
        for n in st.nets:
            n.mark_packed()
        return st.losses[0], st.losses[1], st.losses[2]

    @staticmethod
    def _bind_text_stream(st, main) -> None:
        """st.streams[1] = a stream whose kernels overlap those of `main` (re-made when the caller comes with another current stream)."""
        if st.text_cs is not None and st.text_main == main.cuda_stream:
            return
        if st.text_cs is not None:
            torch.cuda.synchronize()
            st.text_cs.close()
            if st.comm_cs is not None:
                st.comm_cs.close()
                st.comm_cs = st.comm = None
        # Default priority, as every stream of the step always had in effect (rounds 2-5 asked torch for priority 1, which torch clamps to
        # its lowest = HIP's default).  A REAL low-priority queue (COOT_TEXT_STREAM_PRIORITY=1) is time-sliced against the default ones:
        # 8-us kernels take 50 us on both sides and the step 1.9-2.3 ms, depending on where the queues land
        # (profiles/r06_stream_queues.txt).
        st.text_cs = _lib.ConcurrentStream([main], priority=int(os.environ.get("COOT_TEXT_STREAM_PRIORITY", "0")))
        st.text_main = main.cuda_stream
        st.streams = (None, st.text_cs.torch)

    def join_streams(self) -> None:
        """Orders the current stream after the native step's text stream (train_step_native(defer_join=True) leaves the text side's
        update running there).  Call before reading the text networks' parameters or weight packs on the current stream (validation,
        checkpoints, the autograd route); a torch.cuda.synchronize() does the same for the host.  The step's three loss words are written
        on the caller's stream (they ride on the video side's update launch) and need no join."""
        st = getattr(self, "_native", None)
        if st is not None and getattr(st, "join_pending", False):
            torch.cuda.current_stream().wait_stream(st.streams[1])
            st.join_pending = False

    def train_step_native(self, batch: RetrievalDataBatchTuple, do_optimizer: bool = True, seed: Optional[int] = None,
                          vid_counts=None, clip_counts=None, use_graph=False, cc_indices: Optional[torch.Tensor] = None,
                          defer_join: bool = False, next_batch=None):
        """One optimisation step as ONE call into libcoot_hip.so (coot_train_step): forward of both sides on two
        HIP streams, losses, backward, fused Adam — no Python between the kernel launches.  Returns views of the
        device loss vector (total, contrastive, cycle-consistency).  With ``self.dp`` set the step runs as native phases
        with the RCCL collectives between them (_train_step_native_dp).  Adam state is the library's
        (flat moment arenas), the learning rate is read from self.optimizer's first param group on every call so the
        reference's LR schedulers keep working.
        next_batch: the batch of the FOLLOWING step, if the data loader already has it on the device (lookahead of one): its
        parameter-free input LayerNorm runs inside this step, next to the global networks, instead of at the head of the next one
        (COOT_STEP_INPUT_STAGES; same results).  Pass the same object to the next call."""
        lib = _lib.load()
        if getattr(self, "dp", None) is not None:
            return self._train_step_native_dp(batch, do_optimizer, seed, vid_counts, clip_counts, cc_indices, next_batch, defer_join)
        if use_graph and cc_indices is not None:
            raise ValueError("train_step_native: injected cycle-consistency positions are not available under graph replay")
        if use_graph and do_optimizer and seed is None and self.model_mgr.is_train:
            out = self._train_step_native_graph(batch)
            if out is not None:
                return out
        st, x = self._native_setup(batch)
        if self.optimizer is not None:
            st.cfg.lr = float(self.optimizer.param_groups[0]["lr"])
        if do_optimizer:
            st.step += 1
        if seed is None:
            seed = (torch.initial_seed() * 1000003 + 7919 * (self.total_step + 1)) & 0xFFFFFFFFFFFFFFFF
        train = 1 if self.model_mgr.is_train else 0
        main = torch.cuda.current_stream()
        self._bind_text_stream(st, main)
        # the video side runs on the caller's stream itself (no cross-stream hop on the critical path), the text side on a side stream
        flags = 0
        if do_optimizer:
            flags |= _lib.STEP_OPTIMIZER | _lib.STEP_REPACK
        if all(n.pack_is_fresh() for n in st.nets):
            flags |= _lib.STEP_PACKS_FRESH
        if self._announce_next_batch(lib, st, batch, next_batch):
            flags |= _lib.STEP_INPUT_STAGES | (_lib.STEP_STAGE_ANNOUNCED if st.batch_announced else 0)
        if defer_join and do_optimizer:
            flags |= _lib.STEP_DEFER_TEXT_JOIN
            st.join_pending = True
        else:
            st.join_pending = False  # this call ends with the text stream joined into the current one
        if cc_indices is not None:  # a given draw of the cycle-consistency positions ([2B] int64: clips, then sentences)
            assert cc_indices.dtype == torch.int64 and cc_indices.is_cuda and cc_indices.numel() == 2 * st.dims.B
            lib.coot_step_set_cycle_indices(cc_indices.data_ptr())
        self._det_sync([n._grad_flat for n in st.nets] + [st.losses])  # (deterministic mode: the step flushes these ranges itself)
        try:
            _lib.check(lib.coot_train_step(C.byref(st.cfg), C.byref(st.bufs), C.byref(x), C.byref(st.dims), st.losses.data_ptr(),
                                           st.ws.data_ptr(), st.ws.numel(), train, int(seed), max(st.step, 1), flags,
                                           main.cuda_stream, main.cuda_stream, st.streams[1].cuda_stream), "coot_train_step")
        finally:
            if cc_indices is not None:
                lib.coot_step_set_cycle_indices(None)
        if do_optimizer:  # the library rebuilt the bf16 packs right after its Adam update
            for n in st.nets:
                n.mark_packed()
        self.total_step += 1
        return st.losses[0], st.losses[1], st.losses[2]

    def _train_step_native_dp(self, batch, do_optimizer=True, seed=None, vid_counts=None, clip_counts=None, cc_indices=None, next_batch=None,
                              defer_join=False):
        """Data-parallel native step: this rank's videos through coot_step_forward (C, two streams), ONE
        packed all-gather per embedding level over RCCL, the contrastive loss on the full gathered batch (every rank
        keeps the gradient rows of its own videos — no collective for embedding gradients), the per-video
        cycle-consistency loss scaled by 1 / global batch, coot_step_backward (C), all-reduce(SUM) of the flat gradient
        arenas, fused Adam.  Same kernels and the same C sequencing as the single-GPU native step."""
        lib = _lib.load()
        dp = self.dp  # every collective of the step goes through this object (dist.DataParallelContext)
        dev = batch.vis_tokens.device if isinstance(batch, RetrievalPackedBatchTuple) else batch.vid_feat.device
        vid_counts, clip_counts = self._dp_batch_shapes(batch, vid_counts, clip_counts)
        st, x = self._native_setup(batch)
        if self.optimizer is not None:
            st.cfg.lr = float(self.optimizer.param_groups[0]["lr"])
        if do_optimizer:
            st.step += 1
        if seed is None:
            seed = (torch.initial_seed() * 1000003 + 7919 * (self.total_step + 1) + 104729 * dp.rank) & 0xFFFFFFFFFFFFFFFF
        train = 1 if self.model_mgr.is_train else 0
        d = st.dims
        B, Nc, D = d.B, d.Nc, st.cfg.net[0].hidden_dim
        assert vid_counts[dp.rank] == B and clip_counts[dp.rank] == Nc, (vid_counts, clip_counts, dp.rank, B, Nc)
        key = (st.dims_key, tuple(vid_counts), tuple(clip_counts))
        gb, gn = sum(vid_counts), sum(clip_counts)
        if getattr(st, "dp_key", None) != key:
            f32 = dict(dtype=torch.float32, device=dev)
            # The four embedding outputs of the forward live in ONE block [glob_v | glob_t | local_v | local_t] laid out at the MAX
            # row counts over the ranks, so that a single all-gather of equally sized blocks is the whole exchange and the loss
            # reads the gathered blocks in place (coot_contrastive_fwd_bwd_dp_blocks: per-rank base offsets of the six sets).
            # More ranks than the loss addresses (lib.DP_MAX_RANKS), or a context without gather_block: two packed row gathers.
            W = len(vid_counts)
            mB, mN = max(vid_counts), max(clip_counts)
            off_gv, off_gt, off_lv = 0, mB * 2 * D, 2 * mB * 2 * D
            off_lt = off_lv + (mB + mN) * D
            blk = (off_lt + (mB + mN) * D + 63) // 64 * 64
            st.blocks_on = hasattr(dp, "gather_block") and W <= _lib.DP_MAX_RANKS
            st.sendbuf = torch.zeros(blk, **f32)
            st.recvbuf = torch.empty(W * blk, **f32) if st.blocks_on else None
            cut = lambda o, r, c: st.sendbuf[o:o + r * c].view(r, c)
            st.emb = [cut(off_lv, B + Nc, D), cut(off_lt, B + Nc, D), cut(off_gv, B, 2 * D), cut(off_gt, B, 2 * D),
                      torch.empty(B, d.Cmax_clip, D, **f32), torch.empty(B, d.Cmax_sent, D, **f32)]
            base = []  # [6][W]: vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx (rank r's clips sit behind ITS context rows)
            for s_ in range(6):
                for r in range(W):
                    o = (off_gv, off_gt, off_lv + vid_counts[r] * D, off_lt + vid_counts[r] * D, off_lv, off_lt)[s_]
                    base.append(r * blk + o)
            st.blk_args = ((C.c_int64 * W)(*vid_counts), (C.c_int64 * W)(*clip_counts), (C.c_int64 * (6 * W))(*base),
                           (C.c_int64 * 6)(2 * D, 2 * D, D, D, D, D))
            # every buffer the step zeroes, in ONE allocation (one fill per step instead of six): the gradients wrt this rank's
            # embeddings (same shapes as st.emb)
            sizes = [t.numel() for t in st.emb]
            st.zbuf = torch.zeros(sum(sizes), **f32)
            views, off = [], 0
            for t_, n_ in zip(st.emb, sizes):
                views.append(st.zbuf[off:off + n_].view(t_.shape))
                off += n_
            st.demb = views
            d_local_v, d_local_t, d_glob_v, d_glob_t = views[:4]
            st.down = (C.c_void_p * 6)(d_glob_v.data_ptr(), d_glob_t.data_ptr(), d_local_v[B:].data_ptr(), d_local_t[B:].data_ptr(),
                                       d_local_v.data_ptr(), d_local_t.data_ptr())
            st.emb_ptrs = [t.data_ptr() for t in st.emb]
            st.loss_scratch = torch.empty(lib.coot_contrastive_scratch_bytes(gb, gn, 2 * D, D), dtype=torch.uint8, device=dev)
            st.cyc_idx = torch.empty(2 * B, dtype=torch.int64, device=dev)
            st.dp_key = key
        if getattr(st, "gall", None) is None or st.gall.device != dev:
            # ONE gradient arena for the four networks + the loss words: one fill per step, reduced in THREE buckets in the order the
            # backward finishes them.  Layout: [video global | text global | text local | video local | total, contrastive,
            # cycle-consistency, pad] — the global networks' gradients are final after the global backward (two thirds of the pass
            # before its end) and the text side's local backward ends before the video side's (a third of its tokens): both are
            # reduced on a communication stream under the backward that is still running; only the last bucket — the video local
            # network with the loss words, per-rank partial sums of global means — is exposed behind the pass on the main stream
            st.gall = torch.zeros(sum(n.numel for n in st.nets) + 4, **dict(dtype=torch.float32, device=dev))
            off, cuts = 0, {}
            for i in (1, 3, 2, 0):
                n = st.nets[i]
                n.rebind_flat_grads(st.gall[off:off + n.numel])
                st.bufs.grads[i] = n._grad_flat.data_ptr()
                cuts[i] = off
                off += n.numel
            st.g_glob = st.gall[:cuts[2]]
            st.g_loc_t = st.gall[cuts[2]:cuts[0]]
            st.losses = st.gall[off:off + 3]       # (total, contrastive, cycle-consistency): the layout coot_step_update completes
            st.cl_word = st.gall[off + 1:off + 2]  # this rank's share of the contrastive loss (its rows against the gathered batch)
            st.cc_word = st.gall[off + 2:off + 3]
            st.g_loc_v = st.gall[cuts[0]:off + 3]  # video local network + the loss words
            st.comm = None
            # stream ordering through the library's fence-free events (include/coot_hip.h: coot_event_* / coot_stream_hop): a default event
            # — torch.cuda.Event, wait_stream — performs a system-scope L2 writeback / invalidation at every record
            st.EV_GLOB_V, st.EV_GLOB_T, st.EV_TEXT, st.EV_GLOB_RED = 0, 1, 2, 3
        local_v, local_t, glob_v, glob_t, resh_v, resh_t = st.emb
        d_local_v, d_local_t, d_glob_v, d_glob_t, d_resh_v, d_resh_t = st.demb
        main = torch.cuda.current_stream()
        self._bind_text_stream(st, main)
        if getattr(st, "comm", None) is None:  # the communication stream: beside both sides (their backward runs under the buckets)
            st.comm_cs = _lib.ConcurrentStream([main, st.streams[1]])
            st.comm = st.comm_cs.torch
        sv, stt = main, st.streams[1]  # video side on the caller's stream (no hop), text on a side stream
        sp = main.cuda_stream
        # the backward WRITES the weight-matrix gradients (coot_net_grads_overwrite): only the vectors that are still accumulated (biases,
        # LayerNorm parameters), the cycle-consistency word and the embedding-gradient block are zeroed — one launch instead of two fills
        if getattr(st, "zero_args", None) is None or st.zero_key != (st.gall.data_ptr(), st.zbuf.data_ptr()):
            cfgs = (C.POINTER(_lib.NetConfig) * 4)(*[C.pointer(st.cfg.net[i]) for i in range(4)])
            grads = (C.c_void_p * 4)(*[st.bufs.grads[i] for i in range(4)])
            extra = (C.c_void_p * 2)(st.losses.data_ptr(), st.zbuf.data_ptr())
            extra_n = (C.c_int64 * 2)(4, st.zbuf.numel())   # the arena's 4 tail words (loss words + padding)
            st.zero_args, st.zero_key = (cfgs, grads, extra, extra_n), (st.gall.data_ptr(), st.zbuf.data_ptr())
        self._det_sync([st.gall])  # (deterministic mode: coot_step_backward flushes the arenas, _dp_finish the loss words)
        za = st.zero_args
        # ... on the TEXT stream, behind whatever the previous step left running there (with defer_join: the text networks' update, which
        # reads these buffers) and behind the previous step's work on the main stream (the video networks' update, the gradient buckets'
        # reduction).  The video side's forward on the main stream touches none of it and starts at once; everything that accumulates into
        # the zeroed buffers (losses, backward) is ordered after both sides' forward, i.e. after this launch.
        hop = lambda a, b: _lib.check(lib.coot_stream_hop(a.cuda_stream, b.cuda_stream), "coot_stream_hop")
        hop(main, stt)
        _lib.check(lib.coot_nets_zero_grads_ex(4, za[0], za[1], 1, za[2], za[3], 2, stt.cuda_stream), "coot_nets_zero_grads_ex")
        ws, wsn = st.ws.data_ptr(), st.ws.numel()
        fresh = _lib.FWD_PACKS_FRESH if all(n.pack_is_fresh() for n in st.nets) else 0
        if next_batch is not None and not getattr(next_batch, "global_max_synced", False):
            next_batch = None  # (its padded shapes are not final before ITS batch-shape exchange: no lookahead for it)
        if self._announce_next_batch(lib, st, batch, next_batch):  # the next batch's input LayerNorm runs under the exchange and the loss
            fresh |= _lib.FWD_INPUT_STAGES | (_lib.FWD_STAGE_ANNOUNCED if st.batch_announced else 0)
        _lib.check(lib.coot_step_forward(C.byref(st.cfg), C.byref(st.bufs), C.byref(x), C.byref(d), *st.emb_ptrs, ws, wsn,
                                         train, int(seed), fresh, main.cuda_stream, sv.cuda_stream, stt.cuda_stream), "coot_step_forward")
        # cycle-consistency (per video, no exchange) on the text stream, next to the gathers and the contrastive loss on the main stream
        use_cc = st.cfg.cc_weight != 0.0
        if use_cc:
            hop(main, stt)  # main is ordered after both sides' forward and the zero fills
            with torch.cuda.stream(stt):
                if cc_indices is not None:  # a given draw ([2B] int64: this rank's clip positions, then its sentence positions)
                    st.cyc_idx.copy_(cc_indices)
                else:
                    _lib.check(lib.coot_sample_cycle_indices(batch.clip_num.data_ptr(), batch.sent_num.data_ptr(), B, int(seed),
                                                             st.cyc_idx.data_ptr(), stt.cuda_stream), "coot_sample_cycle_indices")
                _lib.check(lib.coot_cyclecons_fwd_bwd(resh_v.data_ptr(), resh_t.data_ptr(), batch.clip_num.data_ptr(), batch.sent_num.data_ptr(),
                                                      st.cyc_idx.data_ptr(), st.cyc_idx[B:].data_ptr(), B, d.Cmax_clip, d.Cmax_sent, D,
                                                      float(st.cfg.cc_weight), 1.0 / float(gb), st.cc_word.data_ptr(), None, None,
                                                      d_resh_v.data_ptr(), d_resh_t.data_ptr(), stt.cuda_stream), "coot_cyclecons_fwd_bwd")
        # ---- exchange + contrastive loss on the gathered batch (this rank's rows against every column) ----
        if st.blocks_on:  # ONE all-gather of the embedding block; the loss reads the gathered blocks in place
            dp.gather_block(st.sendbuf, st.recvbuf)
            ba = st.blk_args
            _lib.check(lib.coot_contrastive_fwd_bwd_dp_blocks(C.byref(st.cfg.contr), len(vid_counts), dp.rank, ba[0], ba[1], 2 * D, D,
                                                              st.recvbuf.data_ptr(), ba[2], C.byref(ba[3]), st.cl_word.data_ptr(), C.byref(st.down),
                                                              st.loss_scratch.data_ptr(), st.loss_scratch.numel(), sp),
                       "coot_contrastive_fwd_bwd_dp_blocks")
        else:  # two packed row gathers (per-video sets, per-clip sets)
            high = torch.cat([glob_v, glob_t, local_v[:B], local_t[:B]], dim=1)      # [B, 2D | 2D | D | D]
            low = torch.cat([local_v[B:], local_t[B:]], dim=1)                        # [Nc, D | D]
            high_all = dp.gather_rows_nograd(high, vid_counts)
            low_all = dp.gather_rows_nograd(low, clip_counts)
            wh, wl, e4 = 6 * D, 2 * D, 4
            hp, lp = high_all.data_ptr(), low_all.data_ptr()
            sets = (C.c_void_p * 6)(hp, hp + 2 * D * e4, lp, lp + D * e4, hp + 4 * D * e4, hp + 5 * D * e4)  # vid, par, clip, sent, vid_ctx, par_ctx
            lds = (C.c_int64 * 6)(wh, wh, wl, wl, wh, wh)
            v0, c0 = sum(vid_counts[:dp.rank]), sum(clip_counts[:dp.rank])
            _lib.check(lib.coot_contrastive_fwd_bwd_dp(C.byref(st.cfg.contr), gb, gn, 2 * D, D, C.byref(sets), C.byref(lds), st.cl_word.data_ptr(),
                                                       C.byref(st.down), v0, B, c0, Nc, st.loss_scratch.data_ptr(), st.loss_scratch.numel(), sp),
                       "coot_contrastive_fwd_bwd_dp")
        if use_cc:
            hop(stt, main)  # the backward reads d_resh / the cycle-consistency word
        lib.coot_step_set_global_done_events(lib.coot_event_handle(st.EV_GLOB_V), lib.coot_event_handle(st.EV_GLOB_T))
        lib.coot_net_grads_overwrite(1)
        try:
            self._dp_backward(lib, st, x, d, local_v, local_t, resh_v, resh_t, d_local_v, d_local_t, d_glob_v, d_glob_t, d_resh_v, d_resh_t, use_cc,
                              ws, wsn, train, seed, main, sv, stt)
        finally:
            lib.coot_net_grads_overwrite(0)
        self._dp_finish(lib, dp, st, main, sv, stt, do_optimizer, defer_join)
        return st.losses[0], st.losses[1], st.losses[2]

    @staticmethod
    def _dp_backward(lib, st, x, d, local_v, local_t, resh_v, resh_t, d_local_v, d_local_t, d_glob_v, d_glob_t, d_resh_v, d_resh_t, use_cc,
                     ws, wsn, train, seed, main, sv, stt):
        _lib.check(lib.coot_step_backward(C.byref(st.cfg), C.byref(st.bufs), C.byref(x), C.byref(d), local_v.data_ptr(), local_t.data_ptr(),
                                          resh_v.data_ptr(), resh_t.data_ptr(), d_local_v.data_ptr(), d_local_t.data_ptr(), d_glob_v.data_ptr(),
                                          d_glob_t.data_ptr(), d_resh_v.data_ptr() if use_cc else None, d_resh_t.data_ptr() if use_cc else None,
                                          ws, wsn, train, int(seed), main.cuda_stream, sv.cuda_stream, stt.cuda_stream), "coot_step_backward")
        _lib.check(lib.coot_event_record(st.EV_TEXT, stt.cuda_stream), "coot_event_record")  # the text side's backward is the last thing on its stream

    def _dp_finish(self, lib, dp, st, main, sv, stt, do_optimizer, defer_join=False):
        # gradient all-reduce in the order the backward produces its results (every rank issues the three collectives in this order):
        #   1. the global networks' bucket on the communication stream as soon as both global backward passes are done (events recorded
        #      inside coot_step_backward): it runs under the local backward passes;
        #   2. the text local network's bucket, also on the communication stream, when the text stream's backward ends (the event the
        #      text stream recorded behind coot_step_backward): the video side's local backward, three times its tokens, still runs;
        #   3. the video local network + the loss words behind the pass on the main stream — the only exposed bucket.
        lib.coot_step_set_global_done_events(None, None)  # the events belong to this trainer: no other step may record them
        self._det_flush([st.losses])  # (deterministic mode: the cycle-consistency word's fixed-point sum, behind the backward on main)
        for slot in (st.EV_GLOB_V, st.EV_GLOB_T):
            _lib.check(lib.coot_event_wait(slot, st.comm.cuda_stream), "coot_event_wait")
        # The global networks are updated EARLY, as in coot_train_step: their reduced gradients exist a whole local backward before the
        # step's end, so Adam + weight pack of those two networks run on the communication stream right behind their bucket and the
        # tail of the step updates the local networks only.  Same rule as the single call: only when the local backward is long enough
        # to hide the two extra launches (kEarlyMinTokens rows on the video side), and not in deterministic mode.
        d = st.dims
        early = (do_optimizer and self.dp_early_global_update and not getattr(self, "deterministic", False)
                 and d.B * d.Lv + d.Nc * d.Lc >= 8192)
        with torch.cuda.stream(st.comm):
            dp.all_reduce_sum(st.g_glob)
            if early:
                cs = st.comm.cuda_stream
                _lib.check(lib.coot_step_update(C.byref(st.cfg), C.byref(st.bufs), max(st.step, 1), _lib.UPDATE_REPACK | _lib.UPDATE_GLOBAL_ONLY,
                                                None, cs, cs, cs), "coot_step_update (global networks)")
            _lib.check(lib.coot_event_record(st.EV_GLOB_RED, st.comm.cuda_stream), "coot_event_record")
            _lib.check(lib.coot_event_wait(st.EV_TEXT, st.comm.cuda_stream), "coot_event_wait")
            dp.all_reduce_sum(st.g_loc_t)
        dp.all_reduce_sum(st.g_loc_v)
        if do_optimizer and defer_join:
            # Each side's update waits for ITS buckets only: the video side (this stream) for the global networks' bucket and its own
            # local one, the text side for the communication stream's end.  The text side's backward ends up to 90 us after the video
            # side's (it gets the CUs the video side leaves); with one join of the communication stream into this stream the video
            # side's update — and the next step's video forward behind it — waited for the text bucket it never reads.
            _lib.check(lib.coot_event_wait(st.EV_GLOB_RED, main.cuda_stream), "coot_event_wait")
            _lib.check(lib.coot_stream_hop(st.comm.cuda_stream, stt.cuda_stream), "coot_stream_hop")
        else:  # the caller reads every gradient behind this call on its stream
            _lib.check(lib.coot_stream_hop(st.comm.cuda_stream, main.cuda_stream), "coot_stream_hop")
        if do_optimizer:  # (the video side's update launch also writes total = contrastive + cycle-consistency)
            flags = _lib.UPDATE_REPACK | (_lib.UPDATE_DEFER_TEXT_JOIN if defer_join else 0) | (_lib.UPDATE_SKIP_GLOBAL if early else 0)
            _lib.check(lib.coot_step_update(C.byref(st.cfg), C.byref(st.bufs), max(st.step, 1), flags, st.losses.data_ptr(), main.cuda_stream,
                                            sv.cuda_stream, stt.cuda_stream), "coot_step_update")
            st.join_pending = bool(defer_join)
            for n in st.nets:
                n.mark_packed()
        else:
            torch.add(st.cl_word, st.cc_word, out=st.losses[0:1])
        self.total_step += 1

    # ---- epoch loop (coot/trainer_retrieval.py:235-310 and the nntrainer/trainer_base.py hooks it calls) --------------
    def check_is_val_epoch(self) -> bool:
        """nntrainer/trainer_base.py:312-325."""
        v = self.cfg.val
        do_val = self.current_epoch % v.val_freq == 0 and v.val_freq > -1 and self.current_epoch >= v.val_start
        return do_val or self.current_epoch == self.cfg.train.num_epochs

    def check_is_new_best(self, result: float) -> bool:
        """nntrainer/trainer_base.py:327-353, :632-669: better than the old best by det_best_threshold_value (relative or
        absolute), smaller or bigger depending on det_best_compare_mode; the first result is always a new best."""
        v = self.cfg.val
        best, eps = self.det_best_field_best, float(v.det_best_threshold_value)
        if v.det_best_compare_mode not in ("min", "max"):
            raise ValueError(f"Compare mode for determining best field not understood: {v.det_best_compare_mode}")
        if v.det_best_threshold_mode not in ("rel", "abs"):
            raise ValueError(f"Threshold mode for metric comparison not understood: {v.det_best_threshold_mode}")
        if best is None:
            is_best = True
        elif v.det_best_compare_mode == "min":
            is_best = result < (best * (1 - eps) if v.det_best_threshold_mode == "rel" else best - eps)
        else:
            is_best = result > (best * (1 + eps) if v.det_best_threshold_mode == "rel" else best + eps)
        self.det_best_field_current = result
        if is_best:
            self.det_best_field_best = result
        return is_best

    def find_best_epoch(self) -> int:
        """The last validated epoch that was a new best, -1 before any validation (nntrainer/experiment_organization.py:79-102,
        read from the in-memory flags instead of the trainer-state file of the last checkpoint)."""
        good = [e for e, g in zip(self.infos_val_epochs, self.infos_val_is_good) if g]
        return good[-1] if good else -1

    def check_early_stop(self) -> bool:
        """nntrainer/trainer_base.py:285-310: stop after det_best_terminate_after epochs without a new best (-1: never)."""
        current = self.current_epoch - 1
        best = self.find_best_epoch()
        if best == -1:
            best = current
        after = self.cfg.val.det_best_terminate_after
        return after > -1 and current - best >= after  # the reference's '>= after' with after = -1 is guarded by its config assert

    def train_model(self, train_loader, val_loader, native: Optional[bool] = None, on_epoch_end=None) -> Dict[str, list]:
        """Train epochs until done or early-stopped (coot/trainer_retrieval.py:235-310).  ``train_loader`` /
        ``val_loader``: sized iterables of RetrievalDataBatchTuple already resident on the device.  Every step is
        train_step_native (one C call) on a GPU; the schedule (lr_scheduler.py) runs on the host between steps and hands the
        library one scalar.  Returns the per-epoch history {"epoch", "lr", "train_loss", "val"}."""
        from . import lr_scheduler as lrs
        assert self.optimizer is not None, "trainer was built with is_test=True"
        if native is None:
            native = any(p.is_cuda for g in self.optimizer.param_groups for p in g["params"])
        steps_per_epoch = len(train_loader)
        if self.lr_scheduler is None:
            self.lr_scheduler = lrs.make_lr_scheduler(self.optimizer, lrs.SchedulerConfig(self.cfg.raw["lr_scheduler"]),
                                                      float(self.cfg.optimizer.lr), self.cfg.train.num_epochs, steps_per_epoch)
        hist: Dict[str, list] = {"epoch": [], "lr": [], "train_loss": [], "val": []}
        for _epoch in range(self.current_epoch, self.cfg.train.num_epochs):
            if self.check_early_stop():
                break
            self.model_mgr.set_all_models_train()
            loss_sum = None
            # native single-GPU steps: a lookahead of one batch (the loader has it on the device while the step runs: DeviceLoader) lets
            # the step run the next batch's input LayerNorm off its critical path (train_step_native(next_batch=))
            # (only with a loader that keeps a batch's tensors alive while the NEXT one is requested: a list of device batches, or
            # DeviceLoader(lookahead=1))
            lookahead = (native and getattr(self, "dp", None) is None
                         and (isinstance(train_loader, (list, tuple)) or int(getattr(train_loader, "lookahead", 0)) >= 1))
            for batch, nxt in (_with_next(train_loader) if lookahead else ((b, None) for b in train_loader)):
                out = self.train_step_native(batch, next_batch=nxt) if native else self.train_step(batch)
                loss = out[0].detach()
                loss_sum = loss.clone() if loss_sum is None else loss_sum + loss  # device-side: no sync per step
                self.lr_scheduler.step()
            do_val, is_best, val = self.check_is_val_epoch(), False, None
            if do_val:
                v = self.cfg.val
                val_clips = bool(v.val_clips) and v.val_clips_freq > 0 and self.current_epoch % v.val_clips_freq == 0
                val = self.validate_epoch(val_loader, val_clips=val_clips)
                field = v.det_best_field
                if field not in ("val_score_at_1", "val_loss", "val_clip_sent_score_at_1"):
                    raise NotImplementedError(f"best field {field} not known")
                is_best = self.check_is_new_best(val["loss"] if field == "val_loss" else val[field])
                self.infos_val_epochs.append(self.current_epoch)
                self.infos_val_is_good.append(is_best)
            self.lr_scheduler.step_epoch(do_val, is_best)
            hist["epoch"].append(self.current_epoch); hist["lr"].append(self.lr_scheduler.current_lr)
            hist["train_loss"].append(float(loss_sum) / max(steps_per_epoch, 1) if loss_sum is not None else float("nan"))
            hist["val"].append(val)
            if on_epoch_end is not None:
                on_epoch_end(self, do_val, is_best, val)
            self.current_epoch += 1
        return hist

    @torch.no_grad()
    def validate_epoch(self, data_loader, val_clips: bool = True, save_embs: bool = False, save_path: Optional[str] = None):
        """coot/trainer_retrieval.py:312-477 (metric part): eval forward of every batch, both losses, retrieval metrics of
        the collected embeddings; with ``save_embs`` also the embedding export of :404-415 — the dictionary under
        ``out["embeddings"]`` carries exactly the datasets of the reference's ``embeddings_<epoch>.h5`` (``clip_num``,
        ``sent_num`` — written from clip_num there too, :360 —, ``key``, and for each of vid_emb / par_emb / clip_emb /
        sent_emb / vid_context / par_context the L2-normalised rows plus ``<name>_before_norm``); ``save_path`` writes it
        (``.h5`` through h5py when that is importable — the consumers' format, mart/recursive_caption_dataset.py:159-201 —
        otherwise ``.npz`` with the same keys)."""
        self.join_streams()
        self.model_mgr.set_all_models_eval()
        keys = ["vid_emb", "par_emb", "clip_emb", "sent_emb"] + (["vid_context", "par_context"] if save_embs else [])
        coll: Dict[str, list] = {k: [] for k in keys}
        losses, save_clip_num, save_key = [], [], []
        for batch in data_loader:
            visual_data = self.model_mgr.encode_visual(batch)
            text_data = self.model_mgr.encode_text(batch)
            contr = self.compute_total_constrastive_loss(visual_data, text_data)
            cc = self.compute_cyclecons_loss(visual_data, text_data)
            losses.append(contr + cc)
            both = {**visual_data.__dict__, **text_data.__dict__}
            for k in keys:
                coll[k].append(both[k])
            if save_embs:
                save_clip_num.append(batch.clip_num)
                save_key.extend(batch.key)
        data = {k: torch.cat(v, 0).float() for k, v in coll.items()}
        # The reference moves every batch to the host, normalises there (manual L2 without eps, :397-402) and ranks with one
        # numpy argsort per row (nntrainer/retrieval.py:68-98).  Here the embeddings never leave the GPU: normalisation,
        # similarities, ranks and the metric dictionaries are libcoot_hip.so kernels (coot_retrieval_ranks).
        v2p, p2v, vp_sum = compute_retrieval_device(data["vid_emb"], data["par_emb"], normalize=True)
        out = {"v2p": v2p, "p2v": p2v, "val_score_at_1": vp_sum}
        if val_clips:
            c2s, s2c, cs_sum = compute_retrieval_device(data["clip_emb"], data["sent_emb"], normalize=True)
            out.update({"c2s": c2s, "s2c": s2c, "val_clip_sent_score_at_1": cs_sum})
        out["loss"] = float(torch.stack(losses).mean())
        if save_embs:
            clip_num = torch.cat(save_clip_num).cpu().numpy()
            emb: Dict[str, Any] = {"clip_num": clip_num, "sent_num": clip_num.copy(), "key": list(save_key)}
            for k in keys:  # one D2H copy per tensor, after the last batch (the reference copies every batch of every key)
                x = data[k]
                emb[k] = (x / (x * x).sum(dim=-1).sqrt().unsqueeze(-1)).cpu().numpy()
                emb[f"{k}_before_norm"] = x.cpu().numpy()
            out["embeddings"] = emb
            if save_path is not None:
                out["embeddings_file"] = save_embeddings(emb, save_path)
        return out


def save_embeddings(emb: Dict[str, Any], path: str) -> str:
    """Write the export dictionary of validate_epoch(save_embs=True): HDF5 with the reference's dataset names when h5py is
    importable (coot/trainer_retrieval.py:404-415), else numpy ``.npz`` with the same keys (``key`` as a unicode array)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    try:
        import h5py  # noqa: F401
    except ImportError:
        h5py = None
    base = path[:-3] if path.endswith(".h5") else (path[:-4] if path.endswith(".npz") else path)
    if h5py is not None:
        fn = base + ".h5"
        with h5py.File(fn, mode="w") as h5:
            for k, v in emb.items():
                h5[k] = v
        return fn
    fn = base + ".npz"
    np.savez(fn, **{k: (np.array(v) if k == "key" else v) for k, v in emb.items()})
    return fn
