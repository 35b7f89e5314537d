"""Interpreter start-up hook of tests/test_bench_launch.py ONLY: with tests/standin on PYTHONPATH and REID_TEST_STANDIN=1 every
Python process of the job (bench.py, the torch.distributed.run launcher it starts, the ranks) gets the stand-in for
libreid_hip.so (tests/standin_lib.py: host memory + gloo collectives) installed as the library reid_amd._ffi.lib() returns, so
that bench.py's launch / exit-code logic runs on a CPU-only box.  Nothing outside that test sets the variable."""
import os
import sys

if os.environ.get("REID_TEST_STANDIN") == "1":
    _tests = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _root = os.path.dirname(_tests)
    for _p in (_tests, _root):
        if _p not in sys.path:
            sys.path.insert(0, _p)
    if os.path.basename(sys.argv[0] if sys.argv else "") != "run.py" and "torch.distributed.run" not in " ".join(sys.argv):
        from standin_lib import FakeLib
        from reid_amd import _ffi, parallel

        _ffi._lib = FakeLib()
        parallel.RcclComm.unique_id = staticmethod(lambda: bytes(128))      # ncclGetUniqueId needs librccl + a GPU
