"""Serve a package under a second name.

The reference's callers say `from pruning import ...`, `import pruning.losses`,
`python -m pruning.cli`, `from baseline.eval.metrics import ...` (pruning/__init__.py:18-31,
pruning/cli.py:23, baseline/eval/metrics.py:8).  install("pruning", "spice_amd.pruning") makes every
`pruning.X` resolve to the module object of `spice_amd.pruning.X` (one module, two names: no second
copy is executed), including under `python -m`.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.util
import sys


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, alias: str, target: str):
        self.alias, self.target = alias, target

    def _real(self, name: str) -> str:
        return self.target + name[len(self.alias):]

    def find_spec(self, name, path=None, target=None):
        if not name.startswith(self.alias + "."):
            return None
        real_spec = importlib.util.find_spec(self._real(name))
        if real_spec is None:
            return None
        spec = importlib.util.spec_from_loader(name, self, origin=real_spec.origin,
                                               is_package=real_spec.submodule_search_locations is not None)
        spec.has_location = real_spec.has_location
        return spec

    def create_module(self, spec):
        return importlib.import_module(self._real(spec.name))

    def exec_module(self, module):
        pass  # already executed under its real name

    # `python -m alias.mod` asks the loader for the code to run as __main__: a trampoline that runs the
    # real module as __main__ inside its real package (its relative imports reach above the alias)
    def get_code(self, fullname):
        real = self._real(fullname)
        src = f"import runpy\nrunpy.run_module({real!r}, run_name='__main__', alter_sys=True)\n"
        return compile(src, f"<alias {fullname} -> {real}>", "exec")

    def get_source(self, fullname):
        real = self._real(fullname)
        return importlib.util.find_spec(real).loader.get_source(real)

    def is_package(self, fullname):
        return importlib.util.find_spec(self._real(fullname)).submodule_search_locations is not None


def install(alias: str, target: str) -> None:
    for f in sys.meta_path:
        if isinstance(f, _AliasFinder) and f.alias == alias:
            return
    sys.meta_path.insert(0, _AliasFinder(alias, target))
