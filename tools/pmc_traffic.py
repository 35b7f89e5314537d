"""HBM-side traffic per launch of the convolution kernel class from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).  On the GPU box:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --single
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --single
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch/p_results.db gpurun_out/pmc_write/p_results.db [f32|f16] > profiles/rNN_traffic_conv_<mode>.json

gfx950 correction (same guide, HBM section): FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  Both counters are in KiB."""
import json
import re
import sqlite3
import sys

MODE = sys.argv[3] if len(sys.argv) > 3 else "f16"
if MODE == "f32":   # convolution class of the exact-fp32 path: conv_f32.hip kernels + the 7x7 stem (gemm_f32_kernel<A_STEM_*>)
    CONV = re.compile(r"conv_f32_dma_kernel|conv_f32_kernel|stem_f32_kernel|gemm_f32_kernel<[123],|gemm_f32_kernelILi[123]E")
    LABEL = "convolution kernels of the fp32 path (conv_f32_dma, conv_f32, 7x7 stem)"
elif MODE == "f16x3":      # convolution class of the fp32-class path: SPLIT builds on the f16 pipe + the fp32 stem (+ small fp32 leftovers)
    CONV = re.compile(r"conv3x3_x3|conv_x3s_kernel|conv3x3_f16_kernel|gemm_f16_kernel|stem_split_kernel|stem_f32_kernel|conv_f32_dma_kernel")
    LABEL = "convolution kernels of the fp32-class path (conv3x3_x3 two-blocks-per-CU halo kernel, conv_x3s strided / 1x1, conv3x3_f16 / gemm_f16 SPLIT builds, stem_split)"
elif MODE == "swin_f32":   # Swin contractions in exact fp32: dense LDS-DMA GEMM + general conv kernel
    CONV = re.compile(r"gemm_f32_dma_kernel|conv_f32_dma_kernel|gemm_f32_kernel")
    LABEL = "Swin Linear / conv contractions of the fp32 path (gemm_f32_dma, conv_f32_dma general variant)"
elif MODE == "market":     # distance matrix of the Market-size retrieval (bench.py --workload market)
    CONV = re.compile(r"gemm_f32_dma_kernel")
    LABEL = "distance-matrix GEMM of the Market-size search (gemm_f32_dma_kernel<E_DIST>)"
elif MODE == "select":     # fused distance + selection of the Market-size search (no matrix): sample-bound pass + sweep
    CONV = re.compile(r"dist_select_kernelILi\d+ELi\d+E")   # <metric, mode>: mode 0 = sweep with lists, 1 = sample bound, 2 = arg-min sweep
    LABEL = "fused distance + top-k selection of the Market-size search (dist_select_kernel: bound pass + sweep)"
elif MODE == "swin_f16":
    CONV = re.compile(r"gemm_f16_kernel")
    LABEL = "Swin Linear / conv contractions of the fp16-storage path (gemm_f16 linear builds)"
elif MODE == "swin_f16x3":
    CONV = re.compile(r"gemm_f16_kernel|lin_x3_kernel|two_linear_f16x3_kernel|ln_linear_f16x3_kernel")
    LABEL = ("Swin Linear / conv contractions of the fp32-class path (lin_x3_kernel / gemm_f16 linear builds over hi/lo-split operands, the fused "
             "pairs of linears and LayerNorm + to_qkv of stages 1-2: two_linear_f16.hip)")
else:
    CONV = re.compile(r"conv3x3_f16_kernel|conv3x3_c64_f16_kernel|stem_pool_f16_kernel|gemm_f16_kernel")
    LABEL = "convolution kernels of the fp16 path (conv3x3_f16, conv3x3_c64_f16, stem_pool_f16, gemm_f16)"


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda pre: [x for x in tabs if x.startswith(pre)][0]
    q = (f"select s.kernel_name, p.value from {t('rocpd_pmc_event')} p "
         f"join {t('rocpd_info_pmc')} i on p.pmc_id = i.id "
         f"join {t('rocpd_kernel_dispatch')} d on p.event_id = d.event_id "
         f"join {t('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id where i.name = ?")
    out = {}
    for name, val in c.execute(q, (counter,)):
        m = CONV.search(name)
        if not m:
            continue
        k = m.group(0)
        a = out.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(val)
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
launches = sum(v[0] for v in fetch.values())
fetch_kb = sum(v[1] for v in fetch.values())
write_kb = sum(v[1] for v in write.values())
import os
res = {
    "kernel_class": LABEL,
    "commit": os.environ.get("REID_COMMIT", "unknown"),    # the tree the profiled library was built from (passed by the caller: the GPU box has no .git)
    # the argv that was actually profiled, recorded by the script that ran it (round 5's file claimed --crops 512 for a run of 1024)
    "command": ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- " + os.environ["REID_PROFILED_COMMAND"])
               if os.environ.get("REID_PROFILED_COMMAND") else "not recorded (the caller did not export REID_PROFILED_COMMAND)",
    "launches": launches,
    "fetch_size_kb_raw_per_launch": fetch_kb / max(1, launches),
    "write_size_kb_per_launch": write_kb / max(1, sum(v[0] for v in write.values())),
    "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
    "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0 / max(1, launches),
    "per_kernel": {k: {"launches": fetch[k][0], "hbm_bytes_per_launch": (2.0 * fetch[k][1] + write.get(k, [0, 0.0])[1]) * 1024.0 / fetch[k][0]}
                   for k in fetch},
}
print(json.dumps(res, indent=1))
