"""Import alias: the product package lives in ``proto-clip_amd/`` (the name the build contract
fixes), which is not a valid Python identifier.  This stub makes ``import proto_clip_amd`` (and
``proto_clip_amd.utils`` etc.) resolve to that directory.  No code lives here."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "proto-clip_amd")
__path__ = [_real]
_init = _os.path.join(_real, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
del _f, _init
