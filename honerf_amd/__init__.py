"""Importable alias of the ``ho-nerf_amd/`` package.

The product package directory carries the upstream project's hyphenated name,
which is not a Python identifier; this shim makes ``import honerf_amd`` (and
``honerf_amd.<submodule>``) resolve to the files under ``ho-nerf_amd/``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'ho-nerf_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _f, _real
