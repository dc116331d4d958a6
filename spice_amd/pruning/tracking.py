"""Experiment tracker with the reference tracker's method names
(pruning/wandb_tracking/wandb_tracker.py:19-204).  The remote service client is not part of this
package: every call is accepted and dropped, which is what the reference does when its tracker
is disabled (wandb_tracker.py:48-50)."""


class WandBTracker:
    def __init__(self, config=None, enabled: bool = False):
        self.config = config
        self.enabled = False

    def __getattr__(self, name):
        if name.startswith(("log_", "save_", "finish", "watch", "init")):
            return lambda *a, **k: None
        raise AttributeError(name)
