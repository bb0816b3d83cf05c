"""Import alias for the product package.

The build contract fixes the package directory name to ``bayes-od-rc_amd/``, which is not a
valid Python identifier; this shim makes it importable as ``bayes_od_rc_amd``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "bayes-od-rc_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
