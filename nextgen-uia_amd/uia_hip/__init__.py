"""uia_hip — Python face of libuia_hip.so: ctypes loader (_lib) and tensor-level op wrappers (ops)."""
from ._lib import BF16, F32, LIB_PATH, UiaError, lib  # noqa: F401
