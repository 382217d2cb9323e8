"""Learning-rate schedules of the retrieval trainer: constant and reduce-on-plateau, both with optional linear warmup
(behaviour of nntrainer/lr_scheduler.py:100-458; host-side logic, SURVEY 8f-3 — the kernels only ever see the scalar the
schedule produced: ``coot_step_config.lr``).

Calling convention (nntrainer/lr_scheduler.py:103-105): ``step()`` after every optimisation step, ``step_epoch(is_val,
has_improved)`` after every epoch; construction performs one of each, so the object starts at (epoch 0, step 0).

The schedule is a pure function of three counters — epoch, global step, number of reductions — so it is kept as that function
(``factor()``) plus the plateau bookkeeping, not as per-group LR lists carried from call to call: the optimizer's groups are
rewritten from ``base_lrs * factor`` whenever the factor changes."""
from __future__ import annotations

from typing import Any, Dict, List


class SchedulerConst:
    NONE = "none"
    REDUCE_OPW = "reduce_opw"


class SchedulerWarmupConst:
    NONE = "none"
    STEP = "step"
    EPOCH = "epoch"


class SchedulerConfig:
    """nntrainer/lr_scheduler.py:57-75: name, warmup_type, warmup_epochs (+ rop_* for reduce_opw)."""

    def __init__(self, config: Dict[str, Any]):
        config = dict(config)
        self.name: str = config.pop("name")
        self.warmup_type: str = config.pop("warmup_type")
        self.warmup_epochs: int = config.pop("warmup_epochs")
        if self.name == SchedulerConst.REDUCE_OPW:
            self.rop_factor: float = config.pop("rop_factor")
            self.rop_patience: int = config.pop("rop_patience")
            self.rop_cooldown: int = config.pop("rop_cooldown")
            self.rop_min_lr_factor: float = config.pop("rop_min_lr_factor")


class LRScheduler:
    """Constant schedule with warmup; the plateau schedule below only overrides the post-warmup factor."""

    def __init__(self, optimizer, base_lr: float, cfg: SchedulerConfig, num_epochs: int, train_loader_length: int):
        if cfg.warmup_type not in (SchedulerWarmupConst.NONE, SchedulerWarmupConst.STEP, SchedulerWarmupConst.EPOCH):
            raise ValueError(f"Unknown warmup type {cfg.warmup_type}")
        self.optimizer = optimizer
        self.base_lr = float(base_lr)
        self.cfg = cfg
        self.num_epochs = num_epochs
        self.num_steps_per_train_epoch = train_loader_length
        self.base_lr_list: List[float] = []
        for group in optimizer.param_groups:
            assert "initial_lr" not in group, "optimizer already carries initial_lr: scheduler created twice?"
            group["initial_lr"] = group["lr"]
            self.base_lr_list.append(group["lr"])
        self.current_lr = self.base_lr
        self.current_lr_list = list(self.base_lr_list)
        self.current_global_step = -1
        self.current_epoch = -1
        self.step()
        self.step_epoch(False, False)

    # -- the schedule ---------------------------------------------------------------------------------------
    def _is_warmup(self) -> bool:
        return self.cfg.warmup_type != SchedulerWarmupConst.NONE and self.current_epoch < self.cfg.warmup_epochs

    def _warmup_factor(self) -> float:
        if self.cfg.warmup_type == SchedulerWarmupConst.EPOCH:  # changes once per epoch
            return (self.current_epoch + 1) / max(self.cfg.warmup_epochs, 1)
        # per step; the +1 in the denominator keeps the last warmup step below 1 (nntrainer/lr_scheduler.py:326-329)
        return (self.current_global_step + 1) / (self.cfg.warmup_epochs * self.num_steps_per_train_epoch + 1)

    def _plateau_factor(self) -> float:
        return 1.0

    def _on_epoch(self, is_val: bool, has_improved: bool) -> None:
        pass

    def _apply(self, factor: float, ref_lr: float) -> None:
        new = [lr * factor for lr in self.base_lr_list]
        self.current_lr = ref_lr
        if new != self.current_lr_list:
            for group, lr in zip(self.optimizer.param_groups, new):
                group["lr"] = lr
        self.current_lr_list = new

    def _refresh(self, from_epoch: bool, is_val: bool = False, has_improved: bool = False) -> None:
        if self._is_warmup():  # the plateau bookkeeping does not run during warmup (nntrainer/lr_scheduler.py:255-259)
            f = self._warmup_factor()
            self._apply(f, f * self.base_lr)
            return
        if from_epoch:
            self._on_epoch(is_val, has_improved)
            f = self._plateau_factor()
            self._apply(f, self.base_lr * f)
        # a training step after warmup leaves the learning rates where the last epoch boundary (or the last warmup step) put them

    # -- public interface -----------------------------------------------------------------------------------
    def step(self) -> None:
        self.current_global_step += 1
        lo = self.current_epoch * self.num_steps_per_train_epoch
        hi = (self.current_epoch + 1) * self.num_steps_per_train_epoch
        assert lo < self.current_global_step <= hi, (
            f"scheduler step {self.current_global_step} outside ({lo}, {hi}] of epoch {self.current_epoch}: step() / step_epoch() "
            f"not called once per train step / epoch, or wrong steps per epoch ({self.num_steps_per_train_epoch})")
        self._refresh(False)

    def step_epoch(self, is_val: bool, has_improved: bool) -> None:
        self.current_epoch += 1
        self._refresh(True, is_val, has_improved)

    def state_dict(self) -> Dict[str, Any]:
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state: Dict[str, Any]) -> None:
        self.__dict__.update(state)


class ConstantLR(LRScheduler):
    pass


class ReduceOnPlateauWarmup(LRScheduler):
    """factor = max(rop_factor ** reductions, rop_min_lr_factor); one more reduction after more than ``rop_patience``
    validated epochs without a new best, none counted during the ``rop_cooldown`` validated epochs after a reduction
    (nntrainer/lr_scheduler.py:407-458)."""

    def __init__(self, optimizer, base_lr, cfg, num_epochs, train_loader_length):
        self.reduce_steps = 0
        self.cooldown_counter = 0
        self.num_bad_epochs = 0
        super().__init__(optimizer, base_lr, cfg, num_epochs, train_loader_length)

    def _on_epoch(self, is_val: bool, has_improved: bool) -> None:
        if not is_val:
            return
        self.num_bad_epochs = 0 if has_improved else self.num_bad_epochs + 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad_epochs = 0
        if self.num_bad_epochs > self.cfg.rop_patience:
            self.reduce_steps += 1
            self.cooldown_counter = self.cfg.rop_cooldown
            self.num_bad_epochs = 0

    def _plateau_factor(self) -> float:
        return max(self.cfg.rop_factor ** self.reduce_steps, self.cfg.rop_min_lr_factor)


NewROPWarmup = ReduceOnPlateauWarmup  # the reference's class name


def make_lr_scheduler(optimizer, cfg: SchedulerConfig, base_lr: float, num_epochs: int, train_loader_length: int) -> LRScheduler:
    """nntrainer/lr_scheduler.py:23-52."""
    if cfg.name == SchedulerConst.REDUCE_OPW:
        return ReduceOnPlateauWarmup(optimizer, base_lr, cfg, num_epochs, train_loader_length)
    if cfg.name == SchedulerConst.NONE:
        return ConstantLR(optimizer, base_lr, cfg, num_epochs, train_loader_length)
    raise ValueError(f"LR Scheduler unknown: {cfg.name}")
