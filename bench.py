#!/usr/bin/env python3
"""Benchmark of the similarity-navigated aggregation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one fused forward of the aggregation (F.normalize + per-edge cosine +
top-k/threshold selection + similarity-weighted mean: SURVEY.md 8a rows a3-a8) over
the whole synthetic graph, inputs resident in HBM.  Workload = BASELINE.json
configs[3]: ogbn-arxiv-sized graph, SNGNN_Plus top_k=16, thr=0.0, self-loops
removed, C=40 (the config BASELINE.md quotes the roofline target on).

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling - every
rank owns an arxiv-sized node range of an N-times larger graph whose sources are
global, and a step adds the RCCL all-gather of the [n_local, C] feature shards.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def algorithmic_bytes(e_prime: int, n: int, c: int) -> int:
    """per edge one source row (4C) + column index (4) + source
    inverse norm (4); per node row pointer (4) + own row (4C) + own inverse norm (4)
    + output row (4C)."""
    return e_prime * (4 * c + 8) + n * (8 * c + 8)


def make_rank_inputs(name, rank, world, seed, device, channels=None, scale=1.0):
    from sngnn_amd import synth
    n, e, f, classes, max_deg, kind, dens = synth.SHAPES[name]
    if scale != 1.0:
        n, e = int(n * scale), int(e * scale)
    rng = np.random.default_rng(seed + 7919 * rank)
    ei = synth.make_edges(rng, n, e, max_deg, n_src=n * world, dst_offset=rank * n)
    x = synth.make_features(rng, n, f, kind, dens)
    torch.manual_seed(seed)                     # same lin on every rank
    classes = channels or classes               # conv output width (default: #classes, 1 layer)
    lin = torch.nn.Linear(f, classes)
    ei = torch.from_numpy(ei).to(device)
    x = torch.from_numpy(x).to(device)
    with torch.no_grad():
        h = lin.to(device)(x).contiguous()
    return n, classes, ei, x, h, lin


def cpu_baseline(h_cpu, ei_cpu, top_k, thr, reps):
    """The oracle's core-torch restatement of the reference op sequence
    (models.py:233-263) timed on the host cores - a reported baseline only."""
    from oracle import sngnn_oracle as O
    # the GPU box gives one GPU's job a 16-core share; more threads only oversubscribe
    cores = int(os.environ.get("SNGNN_CPU_THREADS", min(16, len(os.sched_getaffinity(0)))))
    torch.set_num_threads(cores)
    O.aggregate_reference(h_cpu, ei_cpu, add_loops=True, remove_loops=True, top_k=top_k, thr=thr)
    t0 = time.perf_counter()
    for _ in range(reps):
        res = O.aggregate_reference(h_cpu, ei_cpu, add_loops=True, remove_loops=True,
                                    top_k=top_k, thr=thr)
    dt = (time.perf_counter() - t0) / reps
    return res, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="arxiv")
    ap.add_argument("--top_k", type=int, default=16)
    ap.add_argument("--thr", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (exploration only)")
    ap.add_argument("--channels", type=int, default=None,
                    help="conv output width C (default: the dataset's class count, 40 for arxiv)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-epoch", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    # SNGNN_BENCH_REHEARSAL=1: the multi-rank code path on however many GPUs there are
    # (ranks share devices, collectives over gloo) - a functional rehearsal for boxes with
    # fewer GPUs than ranks; the numbers it prints mean nothing.
    rehearsal = os.environ.get("SNGNN_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from sngnn_amd import _lib, ops
    from sngnn_amd import dist as sn_dist
    from sngnn_amd.graph import Graph
    lib = _lib.load()

    part = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    n, c, ei, x, h_local, lin = make_rank_inputs(args.workload, rank, world, args.seed, device,
                                                 args.channels, args.scale)
    n_total = n * world
    if world > 1:
        part = sn_dist.Partition(rank, world, n)
    graph = Graph(ei, n_total, True, True, row_range=(rank * n, (rank + 1) * n))
    e_prime = graph.num_edges

    h_full = torch.empty((n_total, c), dtype=torch.float32, device=device) if world > 1 else None

    def step():
        if world > 1:
            dist.all_gather_into_tensor(h_full, h_local)
            src = h_full
        else:
            src = h_local
        return ops.aggregate_forward(graph, src, args.top_k, args.thr)[0]

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    sync_all()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(e_prime)], dtype=torch.float64, device=device)
    if world > 1:
        tmax = tt[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        esum = tt[1:].clone()
        dist.all_reduce(esum)
        dt, e_all = float(tmax.item()), int(esum.item())
    else:
        e_all = e_prime
    ms_per_step = dt / args.steps * 1e3

    result = None
    if rank == 0:
        # --- roofline leg: device time of the dominant kernel, HIP events on its stream
        lib.sngnn_profile_enable(1)
        norms, mains, fins, empties = [], [], [], []
        src = h_full if world > 1 else h_local
        z, m, f, e0 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        for _ in range(min(args.steps, 100)):
            ops.aggregate_forward(graph, src, args.top_k, args.thr)
            _lib.check(lib.sngnn_profile_last_forward(C.byref(z), C.byref(m), C.byref(f), C.byref(e0)),
                       "profile")
            norms.append(z.value)
            mains.append(m.value)
            fins.append(f.value)
            empties.append(e0.value)
        lib.sngnn_profile_enable(0)
        # an event pair itself adds a few microseconds to an interval: the empty interval the
        # library records behind the last launch measures it, and it comes off every figure
        pair_ms = float(np.mean(empties))
        norm_ms, main_ms, fin_ms = (max(float(np.mean(v)) - pair_ms, 0.0) for v in (norms, mains, fins))
        b_alg = algorithmic_bytes(e_prime, n, c)
        achieved = b_alg / (main_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.workload}_k{args.top_k}")
            except Exception:
                traffic = None
        result = {
            "metric": "similarity-aggregation edges/sec",
            "value": e_all / (dt / args.steps),
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ogbn-{args.workload}-sized synthetic graph, SNGNN_Plus "
                                   f"aggregation forward, top_k={args.top_k}, thr={args.thr}, "
                                   "self-loops removed",
                       "nodes_per_gpu": n, "edges_per_gpu": e_prime, "channels": c,
                       "parallelism": "node-range partition + RCCL all-gather of h"
                                      if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_agg_fwd", "kernel_ms": main_ms,
                         "normalize_kernel_ms": norm_ms, "finalize_kernel_ms": fin_ms,
                         "timer": "HIP events on the launch stream, recorded inside the library around "
                                  "each launch, minus the duration of an empty event interval "
                                  f"({pair_ms * 1e3:.1f} us) recorded behind the last launch",
                         "algorithmic_bytes": b_alg},
        }

    # --- extras on one GPU: training-mode forward+backward and a full epoch
    if world == 1 and rank == 0:
        hg = h_local.clone().requires_grad_(True)
        gout = torch.randn_like(h_local)
        for _ in range(5):
            ops.aggregate(hg, graph, args.top_k, args.thr).backward(gout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            hg.grad = None
            ops.aggregate(hg, graph, args.top_k, args.thr).backward(gout)
        torch.cuda.synchronize()
        result["fwd_bwd_ms"] = (time.perf_counter() - t0) / reps * 1e3
        if not args.no_epoch:
            from sngnn_amd.train import epoch_time_ms
            # reference-style eager loop (train.py:73-143) and the same epoch replayed
            # from a HIP graph (sngnn_amd/train.py:GraphedEpoch)
            result["epoch_ms_eager"] = epoch_time_ms(args.workload, x, ei, n, c, args.top_k,
                                                     args.thr, seed=args.seed)
            result["epoch_ms"] = epoch_time_ms(args.workload, x, ei, n, c, args.top_k, args.thr,
                                               seed=args.seed, graphed=True)
        if not args.no_cpu_baseline:
            h_cpu, ei_cpu = h_local.cpu(), ei.cpu()
            res, cpu_dt = cpu_baseline(h_cpu, ei_cpu, args.top_k, args.thr, reps=3)
            result["cpu_baseline"] = {
                "value": e_prime / cpu_dt, "unit": "edges/s",
                "cores": torch.get_num_threads(), "kind": "port",
                "sample": "3 full forward passes of the same graph through the oracle's "
                          "core-torch restatement of the reference op sequence "
                          f"({cpu_dt * 1e3:.0f} ms each)"}
            # the benchmarked output is the checked output
            err = (out.cpu() - res["out"]).abs().max().item()
            result["max_abs_err_vs_oracle"] = err
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
