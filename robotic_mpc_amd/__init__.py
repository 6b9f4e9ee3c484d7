"""Importable alias of the ``robotic-mpc_amd/`` package directory (a hyphen is not a
valid Python identifier, so ``import robotic_mpc_amd`` resolves to that directory)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "robotic-mpc_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
