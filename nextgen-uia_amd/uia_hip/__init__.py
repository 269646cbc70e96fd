"""uia_hip — Python face of libuia_hip.so: ctypes loader (_lib) and tensor-level op wrappers (ops)."""
import os as _os

# dmabuf IPC (the only kind the MI355X hosts of this pool support; RCCL between the ranks of a node needs it).  The HSA runtime reads it at the first HIP call of the
# process: importing this package before the GPU is touched is early enough, and an exported value wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from ._lib import BF16, F32, LIB_PATH, UiaError, lib  # noqa: F401
