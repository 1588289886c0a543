"""Import alias: `puresound.*` -> `puresound_amd.*` (the HIP mirror of the reference's module tree).

The reference's recipes import `puresound.nnet...`, `puresound.streaming...`.  Put this directory and the repository
root on PYTHONPATH *instead of* the reference checkout and those imports resolve to the MI355X modules unchanged:

    PYTHONPATH=/path/to/repo/compat:/path/to/repo python egs/tse/main.py ...

It lives under compat/ (not at the repository root) so that it never shadows a real `puresound` checkout by accident:
the golden-vector generator (tests/golden/make_golden.py) imports the reference under that name.
"""
import importlib
import pkgutil
import sys

import puresound_amd

_SUBPACKAGES = ("nnet", "streaming")


def _alias(name: str) -> None:
    mod = importlib.import_module("puresound_amd." + name)
    sys.modules["puresound." + name] = mod
    parent, _, leaf = name.rpartition(".")
    setattr(sys.modules["puresound." + parent] if parent else sys.modules[__name__], leaf, mod)
    if hasattr(mod, "__path__"):
        for info in pkgutil.iter_modules(mod.__path__):
            if not info.name.startswith("_"):
                _alias(name + "." + info.name)


for _p in _SUBPACKAGES:
    _alias(_p)
__version__ = getattr(puresound_amd, "__version__", "0")
