"""Loader of the torch C++ extension (amq_amd/csrc/amq_torch_ext.cpp -> amq_amd/_amq_ext.so): the host-side fast path of the
drop-in modules.  It only shortens the HOST side of a call (one C++ call instead of Python checks + torch.empty + a ctypes call);
what runs on the device is libamq_hip.so either way, so the modules use the ctypes binding when the extension has not been built
(``get()`` returns None) -- and fail loudly, as always, when the library itself is missing."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_mod = None
_tried = False


def get():
    """the initialised extension module, or None when amq_amd/_amq_ext.so does not exist (not built)"""
    global _mod, _tried
    if _tried:
        return _mod
    _tried = True
    if not os.path.exists(os.path.join(_HERE, "_amq_ext.so")):
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)
    from . import _lib
    _lib.load()                      # raises if libamq_hip.so is missing
    from . import _amq_ext           # an ImportError here is a real build problem: let it propagate
    _amq_ext.init(_lib.LIB_PATH)
    _mod = _amq_ext
    return _mod
