"""MI355X-native GP posterior core behind the osurdml/GPtest call surface.

Host code is plain Python + NumPy; all arithmetic on the hot path (kernel-matrix
build, blocked Cholesky, triangular solves) runs in hand-written HIP kernels for
gfx950 reached through the ctypes C-ABI of ``libgpk.so`` (``include/gpk.h``).
There is no CPU fallback: importing is cheap, but the first compute call raises
``GpkError`` if the library or an MI355X is missing.
"""
from ._lib import GpkError, lib, init, shutdown, set_option, get_option  # noqa: F401

__all__ = ["GpkError", "lib", "init", "shutdown", "set_option", "get_option"]
