"""Import shim: the product package lives in the directory `pyannote-audio_amd/` (the name the
build contract prescribes), which is not a valid Python identifier.  This module turns itself into
that package (sub-modules included) so it is importable as `pyannote_audio_amd`, without copying or
symlinking anything."""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pyannote-audio_amd")
__path__ = [_pkg_dir]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(_pkg_dir, "__init__.py")
with open(__file__, "r") as _fp:
    exec(compile(_fp.read(), __file__, "exec"))
