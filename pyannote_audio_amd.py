"""Import shim: the product package lives in the directory `pyannote-audio_amd/` (the name the
build contract prescribes), which is not a valid Python identifier.  This module makes it
importable as `pyannote_audio_amd` (sub-modules included) without copying or symlinking."""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pyannote-audio_amd")
__path__ = [_pkg_dir]
__file__ = _os.path.join(_pkg_dir, "__init__.py")
with open(__file__, "r") as _fp:
    exec(compile(_fp.read(), __file__, "exec"))
