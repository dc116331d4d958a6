"""`baseline`: the reference's package name (`from baseline.eval.metrics import update_confusion_matrix`,
baseline/eval/metrics.py:8; `python -m baseline.eval.run_eval`), served by spice_amd.baseline."""
from spice_amd._alias import install as _install

_install(__name__, "spice_amd.baseline")
