"""Stand-in for ``torchrl.objectives.LossModule`` (tests only): an nn.Module base class that, like the real one, owns bookkeeping that must
exist before a subclass assigns attributes, and refuses a second ``__init__``.  It lets ``class TRPLLoss(LossModule)`` be constructed and
called with the torchrl branch of geometry_rl_amd.trpl active."""
import torch.nn as nn


class LossModule(nn.Module):
    def __init__(self):
        if getattr(self, "_loss_module_ready", False):
            raise RuntimeError("LossModule.__init__ called twice")
        super().__init__()
        self._loss_module_ready = True
        self._tensor_keys = None

    def __setattr__(self, name, value):
        if not name.startswith("_") and not self.__dict__.get("_loss_module_ready", False):
            raise AttributeError(f"attribute {name!r} assigned before LossModule.__init__")
        super().__setattr__(name, value)
