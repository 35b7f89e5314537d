#!/usr/bin/env python3
"""Benchmark of the re-ID embed + match hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = BASELINE config 2 on every rank: embed 4096 synthetic uint8 crops (128x256, already resident in HBM)
with ResNet18-IBN-SE, [N>1: one RCCL all-gather of the 512-d embeddings], then the L2 distance matrix of this
rank's 4096 embeddings against all gathered ones.  Weak scaling: per-GPU work is fixed, value = crops of ALL
ranks / max-over-ranks time.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (spec, no sparsity)
PEAK_HBM_GBS = 8000.0
FLOP_PER_CROP = 3.980e9        # 1 990 145 536 MAC
FUSED_BYTES_PER_CROP = 13.78e6 # layer-fused fp32 activation traffic model, SURVEY.md section 8(d)


def host_cores():
    """Cores this process may actually use: min(cpu_count, affinity, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def traffic_from_profile(f16):
    """HBM-side bytes per launch of the conv-GEMM class from the committed rocprofv3 PMC passes of this same command
    (tools/pmc_traffic.py: FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as is); PMC counters cannot be read
    from inside the process.  The newest profiles/rNN_traffic_conv_<mode>.json wins."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_conv_%s.json" % ("f16" if f16 else "f32"))))
    if not cands:
        return None
    try:
        return round(json.load(open(cands[-1]))["hbm_bytes_per_launch"], 1)
    except (KeyError, ValueError):
        return None


def cpu_baseline(sd, budget_s=12.0):
    """The oracle (CPU restatement of the reference path) on a bounded sample of the same workload:
    batch 64 (reference default --bs 64), all host cores."""
    import torch
    from oracle import matching, seres18
    from reid_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    crops = synth.crops_u8(64, seed=1)
    seres18.embed_u8(sd, crops)                      # warm-up batch
    n, t0 = 0, time.perf_counter()
    embs = []
    while True:
        embs.append(seres18.embed_u8(sd, crops))
        n += 64
        el = time.perf_counter() - t0
        if el > budget_s or n >= 8192:
            break
    emb = np.concatenate(embs, 0)
    x = np.tile(emb, (max(1, 1024 // emb.shape[0]) + 1, 1))[:1024]
    t1 = time.perf_counter()
    matching.euclidean_dist(x, x)
    dist_ms = (time.perf_counter() - t1) * 1e3
    return {"value": round(n / el, 2), "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": "%d crops (batches of 64, %.1f s) through oracle/seres18.py (torch-CPU restatement, %d threads); "
                      "1024x1024x512 L2 distmat via numpy in %.1f ms" % (n, el, torch.get_num_threads(), dist_ms),
            "distmat_1024_ms": round(dist_ms, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--crops", type=int, default=4096, help="crops per GPU per step (BASELINE config 2: 4096)")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("REID_CHUNK", "1024")))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--single", action="store_true", help="measure only --precision (skip the other arithmetic)")
    ap.add_argument("--precision", choices=["f32", "f16"], default=os.environ.get("REID_PRECISION", "f32"),
                    help="arithmetic of the headline: f32 = the reference's (exact fp32 MFMA, default); f16 = fp16 storage / fp32 "
                         "accumulate (inside north_star's 1e-3 cosine tolerance, reported as the labelled side run by default)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from reid_amd import _ffi, synth, weights
    from reid_amd.engine import get_engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI

    eng = get_engine(local_rank)
    # one explicit (non-null) HIP stream shared by torch (events, RCCL ordering) and the C ABI launches
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    eng.set_chunk(args.chunk)
    sd = synth.seres18_state_dict(0, gem_p=3.0)
    blob, manifest, _ = weights.pack_seres18(sd)
    eng.load_seres18(blob, manifest)

    n, d = args.crops, 512
    # synthetic crops, resident in HBM before the timed region: n DISTINCT crops per rank (BASELINE config 2, seed 1 + rank)
    crops = torch.from_numpy(synth.crops_u8(n, seed=1 + rank)).cuda()
    emb = torch.empty((n, d), dtype=torch.float32, device="cuda")
    gathered = torch.empty((n * world, d), dtype=torch.float32, device="cuda") if world > 1 else emb
    distmat = torch.empty((n, n * world), dtype=torch.float32, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step(timed=None):
        if timed:
            ev[0].record()
        eng.embed_u8_dev(crops.data_ptr(), n, emb.data_ptr())
        if timed:
            ev[1].record()
        if world > 1:
            dist.all_gather_into_tensor(gathered, emb)
        eng.distmat_dev(emb.data_ptr(), n, gathered.data_ptr(), n * world, d, _ffi.METRIC_L2, distmat.data_ptr())
        if timed:
            ev[2].record()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(precision, steps, warmup):
        """Timed region per the bench contract + a profiled repeat of the same steps for the roofline object."""
        eng.set_precision(1 if precision == "f16" else 0)
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # split of one step (embed / all-gather+distmat), HIP events on the launch stream
        step(timed=True)
        torch.cuda.synchronize()
        embed_ms, match_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        # roofline of the dominant kernel class (implicit-GEMM convolutions): the same steps again with every launch
        # of the class bracketed by HIP events on its stream
        eng.profile_reset()
        eng.profile(True)
        tp = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        prof_ms = (time.perf_counter() - tp) * 1e3 / steps
        conv, dgm, elt = (eng.profile_get(k) for k in (_ffi.K_CONV_GEMM, _ffi.K_DIST_GEMM, _ffi.K_ELEMENTWISE))
        eng.profile(False)
        f16 = precision == "f16"
        peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
        conv_tflops = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        return {
            "value": round(n * world * steps / elapsed, 1), "ms_per_step": round(elapsed * 1e3 / steps, 3),
            "embed_ms": round(embed_ms, 3), "embed_crops_per_s_per_gpu": round(n / (embed_ms * 1e-3), 1),
            "distmat_ms": round(match_ms, 3),
            "whole_net_fraction_of_mfma_peak": round(FLOP_PER_CROP * n / (embed_ms * 1e-3) / 1e12 / peak, 4),
            "whole_net_fraction_of_hbm_roofline": round(FUSED_BYTES_PER_CROP * (0.5 if f16 else 1.0) * n / (embed_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "roofline": {
                "kernel": ("convolution kernels of the fp16 path, v_mfma_f32_32x32x16_f16: conv3x3_f16 (LDS halo, layers 2-4), conv3x3_c64_f16 (layer 1, weights in registers), stem_pool_f16 (7x7 + BN + maxpool), gemm_f16 (strided / 1x1)" if f16 else
                           "convolution kernels of the fp32 path, v_mfma_f32_32x32x2_f32 (exact fp32): conv_f32_dma_kernel (implicit GEMM, LDS-DMA staging, all 3x3 / 1x1 convs) + the 7x7 stem"),
                "bound": "mfma", "achieved": round(conv_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(conv_tflops / peak, 4), "traffic": traffic_from_profile(f16),
                "launches": conv["launches"], "avg_launch_us": round(conv["ms"] * 1e3 / max(1, conv["launches"]), 2),
                "algorithmic_gflop_per_launch": round(conv["flops"] / max(1, conv["launches"]) / 1e9, 3),
                "algorithmic_bytes_per_launch": round(conv["bytes"] / max(1, conv["launches"]), 1),
                "profiled_ms_per_step": round(prof_ms, 3),
            },
            "other_kernels": {
                "distmat_gemm": {"ms_per_launch": round(dgm["ms"] / max(1, dgm["launches"]), 3),
                                 "tflops": round(dgm["flops"] / max(dgm["ms"], 1e-9) / 1e9, 2),
                                 "gbs": round(dgm["bytes"] / max(dgm["ms"], 1e-9) / 1e6, 1)},
                "elementwise": {"ms_per_step": round(elt["ms"] / steps, 3),
                                "gbs": round(elt["bytes"] / max(elt["ms"], 1e-9) / 1e6, 1)},
            },
        }

    main_res = run(args.precision, args.steps, args.warmup)
    other = "f32" if args.precision == "f16" else "f16"
    other_res = run(other, max(1, min(3, args.steps)), 1) if not args.single else None

    # parity inside the bench: the two precisions agree on the embeddings of this rank (cosine) and on row arg-mins
    eng.set_precision(1)
    eng.embed_u8_dev(crops.data_ptr(), 256, emb.data_ptr())
    e16 = emb[:256].clone()
    eng.set_precision(0)
    eng.embed_u8_dev(crops.data_ptr(), 256, emb.data_ptr())
    torch.cuda.synchronize()
    e32 = emb[:256]
    cos_err = float((1 - torch.nn.functional.cosine_similarity(e16, e32, dim=1)).max().item())

    if rank == 0:
        f16 = args.precision == "f16"
        out = {
            "metric": "crops/sec embedded + NxM distmat ms, ResNet18-SE 128x256",
            "value": main_res["value"],
            "unit": "crops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if f16 else "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ResNet18-SE embed %d uint8 crops (128x256) per GPU + %dx%d L2 distmat"
                                   % (n, n, n * world),
                       "crops_per_gpu": n, "embed_dim": d, "chunk": args.chunk,
                       "arithmetic": ("fp16 storage, fp32 accumulate (north_star tolerance 1e-3 cosine; measured below)" if f16
                                      else "exact fp32 (v_mfma_f32_32x32x2_f32)"),
                       "sharding": "crops sharded by rank, one RCCL all-gather of [N,512] embeddings" if world > 1 else "single GPU"},
            "f16_vs_f32_max_cosine_err": cos_err,
        }
        out.update({k: v for k, v in main_res.items() if k not in ("value", "ms_per_step")})
        if other_res is not None:
            out[other + "_path"] = other_res
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(sd)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
