"""imgan: MI355X-native SLIC / RAG / MPEG-7 hot path (ctypes over libimgan_hip.so)."""
from . import _lib          # noqa: F401  (raises if the HIP library is missing)
from . import ops           # noqa: F401
from . import shard         # noqa: F401
