"""Kernel durations of one chain step, alone: the three launches (mode 0) against potf2_chain_kernel without a flag (mode 1).
Run under rocprofv3 --kernel-trace --stats."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gptest_amd import _lib
from gpk_hooks import hooks
_lib.init()
P = _lib.c_double_p
rng = np.random.RandomState(5)
Q = rng.randn(256, 300)
B = np.ascontiguousarray(Q @ Q.T / 300 + 0.5 * np.eye(256))
L, D, A11 = np.zeros((256, 128)), np.zeros((128, 128)), np.zeros((128, 128))
for mode in (0, 1):
    for _ in range(200):
        assert hooks().gpk_test_chain_step(B.ctypes.data_as(P), L.ctypes.data_as(P), D.ctypes.data_as(P), A11.ctypes.data_as(P), mode) == 0
_lib.shutdown()
