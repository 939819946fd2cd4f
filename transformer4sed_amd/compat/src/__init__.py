"""Adapter for `import src....` as the reference recipes do.  Extends the package path with any other `src/` found later
on sys.path (the reference checkout), so only the hot-path modules defined here are overridden."""
import os
import sys

__path__ = [os.path.dirname(os.path.abspath(__file__))]
for _p in sys.path:
    _cand = os.path.join(_p, "src")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != __path__[0] and os.path.exists(os.path.join(_cand, "models")):
        __path__.append(os.path.abspath(_cand))
