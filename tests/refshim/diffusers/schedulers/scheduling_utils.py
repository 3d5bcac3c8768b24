"""Plumbing symbols imported by the reference's vendored scheduler text (vsr/diffusion/scheduling_ddim.py:33):
an enum of scheduler names and an empty mixin.  No arithmetic."""
from enum import Enum


class KarrasDiffusionSchedulers(Enum):
    DDIMScheduler = 1
    DDPMScheduler = 2
    EulerDiscreteScheduler = 3


class SchedulerMixin:
    pass
