"""Import shim: the product package lives in the directory `cvpr23-e3dge_amd/` (not a valid Python
identifier), this module makes it importable as `e3dge_amd` (`import e3dge_amd.op`, ...)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "cvpr23-e3dge_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
