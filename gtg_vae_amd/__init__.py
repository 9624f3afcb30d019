"""Import alias: the package sources live in ``gtg-vae_amd/`` (a directory name
Python cannot import directly); this shim makes them importable as
``gtg_vae_amd`` by pointing the package path at that directory."""
import os as _os

_SRC = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gtg-vae_amd")
__path__.insert(0, _SRC)
with open(_os.path.join(_SRC, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_SRC, "__init__.py"), "exec"))
del _f
