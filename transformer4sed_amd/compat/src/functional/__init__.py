import os
import sys

# hot-path overrides first, then the reference's own sub-package of the same name
__path__ = [os.path.dirname(os.path.abspath(__file__))]
_rel = __name__.replace(".", os.sep)
for _p in sys.path:
    _cand = os.path.join(_p, _rel)
    if os.path.isdir(_cand) and os.path.abspath(_cand) != __path__[0]:
        __path__.append(os.path.abspath(_cand))
