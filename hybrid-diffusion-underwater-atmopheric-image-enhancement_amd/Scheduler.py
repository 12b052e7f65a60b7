"""Drop-in for the reference's ``Scheduler.py``: ``GradualWarmupScheduler(optimizer, multiplier, warm_epoch, after_scheduler)``.

Linear warm-up of every base learning rate from ``base`` to ``base * multiplier`` over ``warm_epoch`` epochs, then hand-over
to ``after_scheduler`` (the reference pairs it with CosineAnnealingLR, TrainCondition.py:41-44) whose base rates are
re-based to ``base * multiplier`` at the hand-over.  The learning-rate sequence is pinned against the reference's
(tests/golden/lr_schedule.json).
"""
from torch.optim.lr_scheduler import LRScheduler


class GradualWarmupScheduler(LRScheduler):
    def __init__(self, optimizer, multiplier, warm_epoch, after_scheduler=None):
        self.multiplier = multiplier
        self.total_epoch = warm_epoch
        self.after_scheduler = after_scheduler
        self.finished = False
        super().__init__(optimizer)

    def _warm(self, base_lr):
        return base_lr * ((self.multiplier - 1.0) * self.last_epoch / self.total_epoch + 1.0)

    def get_lr(self):
        if self.last_epoch <= self.total_epoch:
            return [self._warm(b) for b in self.base_lrs]
        if self.after_scheduler is None:
            return [b * self.multiplier for b in self.base_lrs]
        if not self.finished:
            self.after_scheduler.base_lrs = [b * self.multiplier for b in self.base_lrs]
            self.finished = True
        return self.after_scheduler.get_lr()

    def step(self, epoch=None, metrics=None):
        if self.finished and self.after_scheduler is not None:
            self.after_scheduler.step(None if epoch is None else epoch - self.total_epoch)
        else:
            super().step(epoch)
