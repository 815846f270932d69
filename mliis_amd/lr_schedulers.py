"""Inner-loop learning-rate schedules (host, float64) -- same names and call surface as
models/lr_schedulers.py:20-52 of the reference (`cur_lr(cur_step)`), values pinned by tests/golden/host_logic.json."""
import math
from typing import Optional


class LRScheduler:
    def __init__(self, initial_lr: float, total_steps: Optional[int]):
        self.initial_lr = initial_lr
        self.total_steps = total_steps

    def anneal_lr(self, cur_step: int) -> float:
        raise NotImplementedError

    def cur_lr(self, cur_step: int) -> float:
        return self.anneal_lr(cur_step)


class CosineLRScheduler(LRScheduler):
    """lr0/2 * (1 + cos(pi * t / T)), floored at `min_to_decay_to`."""

    def anneal_lr(self, cur_step: int, min_to_decay_to: float = 0.0) -> float:
        return max(0.5 * self.initial_lr * (1.0 + math.cos(math.pi * cur_step / self.total_steps)), min_to_decay_to)


class StepDecay(LRScheduler):
    """lr0 * rate^(t // n), floored at min_lr."""

    def __init__(self, initial_lr: float, total_steps: Optional[int] = None, decay_rate: float = 0.5,
                 decay_after_n_steps: int = 5, min_lr: float = 1e-7):
        super().__init__(initial_lr, total_steps)
        if decay_rate is None or decay_after_n_steps is None:
            raise ValueError("decay_rate and decay_after_n_steps are required")
        self.decay_rate, self.decay_after_n_steps, self.min_lr = decay_rate, decay_after_n_steps, min_lr

    def anneal_lr(self, cur_step: int, decay_rate: Optional[float] = None, decay_after_n_steps: Optional[int] = None) -> float:
        n = self.decay_after_n_steps if decay_after_n_steps is None else decay_after_n_steps
        r = self.decay_rate if decay_rate is None else decay_rate
        return max(self.initial_lr * r ** (cur_step // n), self.min_lr)


supported_learning_rate_schedulers = {"cosine_anneal": CosineLRScheduler, "fixed": None, "constant": None,
                                      "step": StepDecay, "step_decay": StepDecay}
