"""Drop-in for the reference's ``Scheduler.py``: ``GradualWarmupScheduler(optimizer, multiplier, warm_epoch, after_scheduler)``.

Every base learning rate ramps linearly from ``base`` (epoch 0) to ``base * multiplier`` (epoch ``warm_epoch``); from then on
``after_scheduler`` -- the reference pairs it with CosineAnnealingLR (TrainCondition.py:41-44) -- takes over, with its own
base rates re-based to ``base * multiplier`` at the hand-over and its epoch counter starting there.  The resulting
learning-rate sequence is pinned against the reference's (tests/golden/lr_schedule.json, equal to 1e-12).

Public attributes keep the reference's names: ``multiplier``, ``total_epoch``, ``after_scheduler``, ``finished``.
"""
from torch.optim.lr_scheduler import LRScheduler


class GradualWarmupScheduler(LRScheduler):
    def __init__(self, optimizer, multiplier, warm_epoch, after_scheduler=None):
        self.multiplier = multiplier
        self.total_epoch = warm_epoch
        self.after_scheduler = after_scheduler
        self.finished = False          # True once the follow-up scheduler has been re-based and is in charge
        super().__init__(optimizer)

    def _ramp(self) -> float:
        """Factor on the base rate during warm-up: 1 at epoch 0, ``multiplier`` at ``total_epoch``."""
        return 1.0 + (self.multiplier - 1.0) * self.last_epoch / self.total_epoch

    def _peak(self):
        return [base * self.multiplier for base in self.base_lrs]

    def get_lr(self):
        warming_up = self.last_epoch <= self.total_epoch
        if warming_up:
            factor = self._ramp()
            return [base * factor for base in self.base_lrs]
        follow_up = self.after_scheduler
        if follow_up is None:
            return self._peak()
        if not self.finished:
            follow_up.base_lrs = self._peak()
            self.finished = True
        return follow_up.get_lr()

    def step(self, epoch=None, metrics=None):
        handed_over = self.finished and self.after_scheduler is not None
        if not handed_over:
            return super().step(epoch)
        self.after_scheduler.step(None if epoch is None else epoch - self.total_epoch)
