"""Experiment tracker with the reference tracker's constructor and method names
(pruning/wandb_tracking/wandb_tracker.py:19-204).  The remote service client is not part of this
package: every call is accepted and dropped, which is what the reference does when its tracker
is disabled (wandb_tracker.py:48-50)."""


class WandBTracker:
    def __init__(self, config=None, project=None, entity=None, name=None, tags=None, enabled: bool = False):
        self.config, self.project, self.entity, self.name, self.tags = config, project, entity, name, tags
        self.enabled = False  # nothing is ever sent anywhere

    def __bool__(self) -> bool:
        return True

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return lambda *a, **k: None
