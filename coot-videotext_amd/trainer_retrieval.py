"""
Training/validation step of the retrieval path with the reference's trainer hooks
(coot/trainer_retrieval.py: compute_align_loss :122-133, compute_cluster_loss :135-146,
compute_total_constrastive_loss :148-182, compute_cyclecons_loss :216-233, step body :253-291,
validate_epoch :312-477).  Experiment scaffolding (checkpoint dirs, tensorboard, LR schedule) is the
reference's BaseTrainer and out of scope; in the reference these methods are
the overrides a ``HipRetrievalTrainer(RetrievalTrainer)`` subclass carries.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import loss_fn
from .config import RetrievalConfig
from .model_retrieval import (RetrievalDataBatchTuple, RetrievalModelManager, RetrievalTextEmbTuple,
                              RetrievalVisualEmbTuple)
from .retrieval import compute_retrieval


def make_optimizer(cfg_opt, params) -> torch.optim.Optimizer:
    """nntrainer/optimization.py:45-74 for name == 'adam': Adam(lr, betas=(momentum, adam_beta2), eps,
    weight_decay * decay_mult per parameter).  Parameters are grouped by decay_mult (same update rule,
    fewer groups)."""
    name = getattr(cfg_opt, "name", "adam")
    if name != "adam":
        raise NotImplementedError(f"optimizer {name}: only adam is wired up (RAdam is SURVEY 8f row 3, next)")
    lr, wd = float(cfg_opt.lr), float(cfg_opt.weight_decay)
    groups: Dict[float, list] = {}
    for p in params:
        groups.setdefault(p["decay_mult"] * wd, []).append(p["params"])
    param_groups = [{"params": ps, "weight_decay": w, "lr": lr} for w, ps in groups.items()]
    return torch.optim.Adam(param_groups, lr=lr, betas=(float(cfg_opt.momentum), float(cfg_opt.adam_beta2)),
                            eps=float(cfg_opt.adam_eps), amsgrad=bool(getattr(cfg_opt, "adam_amsgrad", False)),
                            foreach=True)


class RetrievalTrainer:
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
            self.optimizer = make_optimizer(cfg.optimizer, params)
        self.cc_generator: Optional[torch.Generator] = None
        self.total_step = 0

    # ---- loss hooks ------------------------------------------------------------------------------------
    def compute_align_loss(self, visual_emb: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
        return self.loss_contr(visual_emb, text_emb)

    def compute_cluster_loss(self, visual_emb: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
        return (self.loss_contr(visual_emb, visual_emb) + self.loss_contr(text_emb, text_emb)) / 2

    def compute_total_constrastive_loss(self, visual_data: RetrievalVisualEmbTuple,
                                        text_data: RetrievalTextEmbTuple) -> torch.Tensor:
        return loss_fn.total_contrastive_loss(self.loss_cfg, visual_data.vid_emb, text_data.par_emb, visual_data.clip_emb,
                                              text_data.sent_emb, visual_data.vid_context, text_data.par_context)

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
        if getattr(self, "_side_streams", None) is None:
            self._side_streams = (torch.cuda.Stream(), torch.cuda.Stream())
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
        caller's choice).  With ``self.dp`` set (dist.DataParallelContext) the batch is this rank's shard."""
        nets = list(self.model_mgr.model_dict.values())
        flat_grads = [net.bind_flat_grads() for net in nets]
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
        for net in nets:
            net.accumulate_into_flat = False
        if dp is not None:
            dp.allreduce_grads(flat_grads, getattr(self, "comm_stream", None))
        self.optimizer.step()
        self.model_mgr.mark_weights_dirty()
        self.total_step += 1
        return loss.detach(), contr_loss.detach(), (cc_loss.detach() if torch.is_tensor(cc_loss) else torch.zeros_like(loss))

    # ---- validation (coot/trainer_retrieval.py:312-477, metrics part) ---------------------------------------
    @torch.no_grad()
    def validate_epoch(self, data_loader, val_clips: bool = True):
        self.model_mgr.set_all_models_eval()
        coll: Dict[str, list] = {k: [] for k in ("vid_emb", "par_emb", "clip_emb", "sent_emb")}
        losses = []
        for batch in data_loader:
            visual_data = self.model_mgr.encode_visual(batch)
            text_data = self.model_mgr.encode_text(batch)
            contr = self.compute_total_constrastive_loss(visual_data, text_data)
            cc = self.compute_cyclecons_loss(visual_data, text_data)
            losses.append(contr + cc)
            coll["vid_emb"].append(visual_data.vid_emb); coll["par_emb"].append(text_data.par_emb)
            coll["clip_emb"].append(visual_data.clip_emb); coll["sent_emb"].append(text_data.sent_emb)
        data = {k: torch.cat(v, 0) for k, v in coll.items()}
        # manual L2 normalisation without eps (:397-402)
        data = {k: (v / (v * v).sum(-1).sqrt().unsqueeze(-1)).float().cpu().numpy() for k, v in data.items()}
        v2p, p2v, vp_sum = compute_retrieval(data["vid_emb"], data["par_emb"])
        out = {"v2p": v2p, "p2v": p2v, "val_score_at_1": vp_sum}
        if val_clips:
            c2s, s2c, cs_sum = compute_retrieval(data["clip_emb"], data["sent_emb"])
            out.update({"c2s": c2s, "s2c": s2c, "val_clip_sent_score_at_1": cs_sum})
        out["loss"] = float(torch.stack(losses).mean())
        return out
