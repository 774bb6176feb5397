"""Import alias for the package that lives in ``long-context-attention_amd/`` (a directory name
Python cannot import directly).  ``import yunchang_amd`` == that package: this module's
``__path__`` points at the real directory, so ``yunchang_amd.kernels``, ``yunchang_amd.ring`` ...
resolve to ``long-context-attention_amd/kernels`` etc., mirroring ``yunchang.kernels`` ...
"""
import os as _os

_impl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "long-context-attention_amd")
__path__ = [_impl]
with open(_os.path.join(_impl, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_impl, "__init__.py"), "exec"))
del _f
