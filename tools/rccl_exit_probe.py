"""Which part of bench.py's embed workload makes a process with a 1-rank RCCL communicator abort at exit on the /opt/rocm stack?
python tools/rccl_exit_probe.py <flags joined by _>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
v = sys.argv[1].split("_")
if "torch" in v:
    import torch  # noqa: F401
import numpy as np
from reid_amd import _ffi, parallel, synth, weights
from reid_amd.engine import Engine, get_engine
eng = get_engine(0) if "getengine" in v else Engine(0)
comm = parallel.RcclComm(eng, 0, 1, parallel.RcclComm.unique_id() if "nocomm" not in v else None)
print("allreduce:", comm.all_reduce([1.0, 2.0], "max"), flush=True)
n = 256
if "load" in v:
    eng.set_chunk(1024)
    eng.load_seres18(*weights.pack_seres18(synth.seres18_state_dict(0, gem_p=3.0))[:2])
if "embed" in v:
    crops = parallel.DevArray.from_numpy(eng, synth.crops_u8(n, seed=1))
    emb_all = parallel.DevArray(eng, (n, 512)); emb_local = parallel.DevArray(eng, (n, 512)); dm = parallel.DevArray(eng, (n, n))
    eng.set_precision(2)
    parallel.embed_sharded_dev(eng, comm, crops.ptr, n, n, emb_all, emb_local.ptr)
    parallel.distmat_row_block(eng, emb_all, 0, n, _ffi.METRIC_L2, dm)
    eng.sync()
if "profile" in v:
    eng.profile_reset(); eng.profile(True)
    parallel.embed_sharded_dev(eng, comm, crops.ptr, n, n, emb_all, emb_local.ptr)
    eng.sync()
    print(eng.profile_get(0)); eng.profile(False)
if "timer" in v:
    eng.timer_start(); parallel.embed_sharded_dev(eng, comm, crops.ptr, n, n, emb_all, emb_local.ptr); print(eng.timer_stop())
if "devsync" in v:
    eng.device_sync()
if "dup" in v:
    fd = os.dup(1); os.dup2(2, 1)
comm.close()
print("python done", flush=True)
