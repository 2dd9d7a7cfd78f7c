"""TEST INFRASTRUCTURE ONLY (tests/hostsim): one rank of `bench.py --gpus N` on the CPU.

tests/test_hostsim.py starts N of these with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as
torch.distributed.run would set them.  The launcher points the ctypes binding at build/libgpk_hostsim.so -- libgpk's own
host code compiled against the host-memory HIP stand-in and the inter-process RCCL stand-in -- and then runs bench.py's
main() unchanged: argument parsing, rank discovery, the TCP bootstrap of the unique id, gpk_comm_init with N > 1,
gpk_bench_posterior's multi-rank leg, the closing barriers / max-reduction, the JSON line on rank 0.
The swap lives HERE, in the test's launcher; the product has no switch that could select this library."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from gptest_amd import _lib  # noqa: E402

_lib._LIB_PATH = os.path.join(HERE, "build", "libgpk_hostsim.so")
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
