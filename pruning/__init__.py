"""`pruning`: the reference's package name (pruning/__init__.py:18-31, `python -m pruning.cli`,
pruning/cli.py:23), served by spice_amd.pruning.  Every `pruning.X` is the module `spice_amd.pruning.X`."""
from spice_amd._alias import install as _install

_install(__name__, "spice_amd.pruning")

from spice_amd.pruning import *  # noqa: E402,F401,F403
from spice_amd.pruning import __all__  # noqa: E402,F401
