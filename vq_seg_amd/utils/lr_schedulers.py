"""Per-iteration learning-rate schedules (reference: utils/lr_schedulers.py:86-112)."""
import math


class WarmUpPolyLR:
    def __init__(self, start_lr, lr_power, total_iters, warmup_steps):
        self.start_lr, self.lr_power = start_lr, lr_power
        self.total_iters, self.warmup_steps = float(total_iters), warmup_steps

    def get_lr(self, cur_iter):
        if cur_iter < self.warmup_steps:
            return self.start_lr * (cur_iter / self.warmup_steps)
        return self.start_lr * ((1 - float(cur_iter) / self.total_iters) ** self.lr_power)


class CosineAnnealingLR:
    """lr(i) = min + (start - min)/2 * (1 + cos(pi * i / (total - warmup)))   (:110-112; the
    warm-up only shortens the period)."""

    def __init__(self, start_lr, min_lr, total_iters, warmup_steps):
        self.start_lr, self.min_lr = start_lr, min_lr
        self.total_iters, self.warmup_steps = float(total_iters), warmup_steps

    def get_lr(self, cur_iter):
        phase = math.pi * cur_iter / (self.total_iters - self.warmup_steps)
        return self.min_lr + 0.5 * (self.start_lr - self.min_lr) * (1 + math.cos(phase))
