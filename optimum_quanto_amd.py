"""Import shim: the package sources live in ``optimum-quanto_amd/`` (a directory name Python cannot import).

``import optimum_quanto_amd`` from the repository root (or with the root on ``sys.path``) resolves to this
module, which turns itself into a package whose search path is that directory and then executes its
``__init__.py``.  An installed copy maps the directory with
``package_dir={"optimum_quanto_amd": "optimum-quanto_amd"}`` (setup.py) and does not need this file.
"""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "optimum-quanto_amd")
__path__ = [_pkg_dir]
__package__ = __name__
__file__ = _os.path.join(_pkg_dir, "__init__.py")
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f, _os
