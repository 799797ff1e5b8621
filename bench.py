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
global, and a step adds the RCCL exchange of the feature rows: the halo exchange
(all-to-all-v of the rows the rank's in-edges reference; default) or the full all-gather
of the [n_local, C] shards (--exchange allgather).  --locality p draws a fraction p of
every row's sources from the rank's own range (0 = uniform over all ranks, the worst case
for a node-range partition); the line reports the bytes each exchange form moves.

--workload products: BASELINE config 5's graph (ogbn-products size) through one
SNGNN_Plus_Plus layer - adjacency branch (Linear(num_nodes, C) on the sparse adjacency) +
aggregation + blend - at C = 48 (47 classes padded to 16-byte rows).

The line's keys (beyond the driver's contract):
  roofline        frac = SURVEY.md 8d's algorithmic bytes of one forward over ms_per_step - the whole
                  step (normalise + main + finalize launches and the gaps between them), the gate
                  BASELINE.md defines.  Beside it the dominant kernel k_agg_fwd alone: kernel_frac =
                  the same bytes over kernel_ms = the larger of (a) HIP events on the launch stream
                  around BATCHES of 20 back-to-back launches recorded inside the library
                  (sngnn_profile_enable(20)), nothing subtracted, and (b) the committed rocprofv3
                  average of the same command (kernel_ms_source names the one used; both and the
                  single-launch event interval are in the line).  traffic = counter bytes per launch of
                  the main kernel in the committed PMC passes (profiles/traffic.json), not re-measured.
  roofline_step   the headline figure under its round-3 name.
  variants        BASELINE.md's other config-4 cases (C = 32; thr = 0.9) timed the same way; their fractions
                  are EFFECTIVE (B_fwd of the unpruned byte model over the time - a pruning threshold moves
                  fewer bytes: `traffic`).
  roofline.gather_floor_ms / kernel_over_floor   sngnn_gather_floor timed in this run: the forward's row reads
                  and stores on this graph and table with no arithmetic - what the memory system needs for the
                  access pattern - and the main kernel's time over it.
  epoch_ms        train + validation + test step replayed from one HIP graph, the reference's
                  THREE forwards (train.py:134-143); epoch_ms_shared_eval = validation and test
                  read one eval-mode forward (bit-identical metrics); epoch_ms_eager = the
                  reference-style loop.
  --workload products also prints the ++ layer's forward+backward, its backward roofline and the
  Adam step over the dense 460 MB w (train.py:376).

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


def algorithmic_bytes(e_prime: int, n: int, c: int, plus_plus: bool = False) -> int:
    """per edge one source row (4C) + column index (4) + source
    inverse norm (4); per node row pointer (4) + own row (4C) + own inverse norm (4)
    + output row (4C).  SNGNN++ adds the adjacency branch: per edge one W^T row (4C) +
    its index (4), per node one more output row (4C)."""
    b = e_prime * (4 * c + 8) + n * (8 * c + 8)
    if plus_plus:
        b += e_prime * (4 * c + 4) + n * 4 * c
    return b


def backward_bytes(e_prime: int, n_sel: int, n: int, c: int) -> int:
    """SURVEY.md 8(d), secondary accounting model of the backward (selected-edge formulation):
    per edge of the static CSC weight + id (8); per kept edge the rows h_src, g'_dst, n_dst
    (12C) + 8; per node G, h, n, grad_h rows (16C)."""
    return e_prime * 8 + n_sel * (12 * c + 8) + n * 16 * c


def make_rank_inputs(name, rank, world, seed, device, channels=None, scale=1.0, locality=0.0):
    from sngnn_amd import synth
    n, e, f, classes, max_deg, kind, dens = synth.SHAPES[name]
    if scale != 1.0:
        n, e = int(n * scale), int(e * scale)
    rng = np.random.default_rng(seed + 7919 * rank)
    ei = synth.make_edges(rng, n, e, max_deg, n_src=n * world, dst_offset=rank * n)
    if locality > 0.0 and world > 1:
        # a fraction of every row's sources comes from the rank's own node range (graphs that
        # are partitioned by a locality-preserving order look like this); re-coalesced
        local = rng.random(ei.shape[1]) < locality
        src = np.where(local, rng.integers(rank * n, (rank + 1) * n, size=ei.shape[1]), ei[0])
        key = np.unique(src * (n * world + 1) + ei[1])
        ei = np.stack([key // (n * world + 1), key % (n * world + 1)])
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


def cpu_reference_literal(h_cpu, ei_cpu, top_k, thr, reps):
    """where torch_scatter imports on this box, the SAME op sequence with the
    reference's literal third-party calls - ``torch_scatter.scatter_max`` in the top_k rounds
    (models.py:252) and ``torch_scatter.scatter(reduce='mean')`` as PyG's aggregate (models.py:217) -
    timed beside the port.  Returns (seconds per forward, max |out - port|) or a string saying which
    package is absent."""
    try:
        import torch_scatter as ts
    except Exception as ex:      # noqa: BLE001
        return f"torch_scatter does not import here ({type(ex).__name__}): only the port is timed"
    from oracle import sngnn_oracle as O
    kw = dict(add_loops=True, remove_loops=True, top_k=top_k, thr=thr,
              smax=lambda src, index: ts.scatter_max(src, index, dim=0),
              smean=lambda msg, index, n: ts.scatter(msg, index, dim=-2, dim_size=n, reduce="mean"))
    res = O.aggregate_reference(h_cpu, ei_cpu, **kw)
    t0 = time.perf_counter()
    for _ in range(reps):
        res = O.aggregate_reference(h_cpu, ei_cpu, **kw)
    return (time.perf_counter() - t0) / reps, res


def profile_forward(lib, _lib, ops, graph, table, top_k, thr, calls, reps):
    """Device time of the forward's three launches from HIP events recorded inside the library on
    the launch stream: every launch issued `reps` times back to back between its two events
    (sngnn_profile_enable(reps); reps = 1: the single-launch interval).  Means over `calls` calls:
    (normalise, main, finalize, empty event interval) in ms."""
    lib.sngnn_profile_enable(int(reps))
    z, m, f, e0 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    acc = []
    for _ in range(calls):
        ops.aggregate_forward(graph, table, top_k, thr)
        _lib.check(lib.sngnn_profile_last_forward(C.byref(z), C.byref(m), C.byref(f), C.byref(e0)), "profile")
        acc.append((z.value, m.value, f.value, e0.value))
    lib.sngnn_profile_enable(0)
    return tuple(float(v) for v in np.mean(np.array(acc[1:] if len(acc) > 1 else acc), axis=0))


def gather_floor(lib, _lib, graph, table, mode, batches=6, reps=20):
    """``sngnn_gather_floor``: the memory work of the forward's main kernel with none of its arithmetic, on
    THIS graph and THIS table (mode 0: one row read per entry of the graph's column list; mode 1: + own row
    read and output row written per node = every byte of B_fwd).  Average launch duration in ms: torch events
    on the launch stream (the current one) around batches of ``reps`` back-to-back launches, median of the
    batches after the first."""
    c = table.size(1)
    ws = torch.empty(int(lib.sngnn_gather_floor_workspace_bytes()), dtype=torch.uint8, device=table.device)
    out = torch.empty((graph.num_nodes, c), dtype=torch.float32, device=table.device) if mode == 1 else None
    st = torch.cuda.current_stream(table.device).cuda_stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for _ in range(batches):
        ev[0].record()
        for _ in range(reps):
            _lib.check(lib.sngnn_gather_floor(graph.handle, table.data_ptr(), c, mode, _lib.ptr(out), ws.data_ptr(), st),
                       "sngnn_gather_floor")
        ev[1].record()
        ev[1].synchronize()
        ms.append(ev[0].elapsed_time(ev[1]) / reps)
    return float(np.median(ms[1:]))


def copy_floor(table, batches=6, reps=20):
    """The normalisation pass's floor on this box: a plain device copy of the table (4NC bytes read + 4NC written =
    the pass's bytes but for the 4N of norms, no arithmetic), average duration in ms, timed like gather_floor."""
    dst = torch.empty_like(table)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for _ in range(batches):
        ev[0].record()
        for _ in range(reps):
            dst.copy_(table)
        ev[1].record()
        ev[1].synchronize()
        ms.append(ev[0].elapsed_time(ev[1]) / reps)
    return float(np.median(ms[1:]))


def time_loop(fn, warmup, steps, sync):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    sync()
    return time.perf_counter() - t0, out


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
    ap.add_argument("--exchange", choices=["halo", "allgather"], default="halo",
                    help="N > 1: how the feature rows reach the ranks that reference them")
    ap.add_argument("--locality", type=float, default=0.0,
                    help="N > 1: fraction of a row's sources drawn from the rank's own node range "
                         "(default 0: sources uniform over ALL ranks' nodes, SURVEY.md 8d's generator and "
                         "the adversarial case for a node-range partition; the edge cut of a "
                         "locality-preserving 8-way partition of a real graph is 15-30 %%, i.e. 0.7-0.85 - "
                         "the line's `locality_0.8` sub-result is that case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-epoch", action="store_true")
    ap.add_argument("--preheat-ms", type=float, default=60.0,
                    help="untimed run of the same step in front of the W warmup steps, until the device has been busy this "
                         "long: after an idle spell the first ~15 ms of work run ~7 %% slower (tools/micro/ramp.py: 67.6 us "
                         "per call in the first 200-call window, 63.2 in every later one) - the clocks ramp; the figure a "
                         "training run sees is the steady one.  0 = off; the cold figure is reported beside it either way")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-finalize-ab", action="store_true",
                    help="skip the leg that times the same step with the split rows' finalize as a launch of its own "
                         "(profiling runs: one form of the main kernel per trace)")
    args = ap.parse_args()

    # (before anything initialises the HIP runtime, which reads it once: the host driver only
    # supports dmabuf IPC, and RCCL fails with hipIpcGetMemHandle otherwise)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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

    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    plus_plus = args.workload == "products"          # config 5: one SNGNN_Plus_Plus layer
    if plus_plus and args.channels is None:
        args.channels = 48                           # 47 classes, rows padded to 16 bytes
    strong = plus_plus and world > 1                 # config 5 partitions ONE products graph

    def build_rank(locality, exchange):
        """Inputs, partition, local graph(s) and the step function of this rank.  N > 1 runs the
        SAME exchange + aggregation code the conv layers run (sngnn_amd/dist.py:halo_aggregate -
        the halo rows received straight into a fresh [own | halo] table, interior rows aggregated
        while they are in flight - or all_gather_rows + ops.aggregate)."""
        n, c, ei, x, h_local, lin = make_rank_inputs(args.workload, rank, world, args.seed, device, args.channels,
                                                     args.scale / world if strong else args.scale, locality)
        R = dict(n=n, c=c, ei=ei, x=x, h_local=h_local, plan=None, plan_f=None)
        if world > 1:
            part = sn_dist.Partition(rank, world, n, exchange=exchange)
            R["part"] = part
            if exchange == "halo":
                plan = R["plan"] = sn_dist.HaloPlan(ei, part)
                graph = Graph(plan.edge_index, plan.table_rows, True, True, row_range=(0, n))
            else:
                graph = Graph(ei, n * world, True, True, row_range=(rank * n, (rank + 1) * n))
        else:
            graph = Graph(ei, n, True, True)
        R["graph"] = graph
        if plus_plus:
            # the adjacency branch: W^T rows [n, C] of Linear(num_nodes, C), sharded by node range
            gen = torch.Generator(device="cpu").manual_seed(args.seed + 1)
            R["wt_local"] = wt_local = (torch.randn(n, c, generator=gen) * 0.01).to(device)
            R["w_bias"] = w_bias = torch.zeros(c, device=device)
            R["beta"] = beta = torch.full((1,), 0.3, device=device)
            if world > 1:
                ei_f = ei.flip(0).contiguous()
                if exchange == "halo":
                    plan_f = R["plan_f"] = sn_dist.HaloPlan(ei_f, part)
                    graph_f = Graph(plan_f.edge_index, plan_f.table_rows, True, True, row_range=(0, n))
                else:
                    graph_f = Graph(ei_f, n * world, True, True, row_range=(rank * n, (rank + 1) * n))

        @torch.no_grad()
        def step():
            if world == 1:
                out1 = ops.aggregate_forward(graph, h_local, args.top_k, args.thr)[0]
            elif exchange == "halo":
                out1 = sn_dist.halo_aggregate(h_local, R["plan"], graph, args.top_k, args.thr)
            else:
                out1 = ops.aggregate(sn_dist.all_gather_rows(h_local, part), graph, args.top_k, args.thr)
            if not plus_plus:
                return out1
            if world == 1:
                out0 = ops.adj_linear_forward(graph, wt_local, w_bias)
            elif exchange == "halo":
                out0 = ops.gather_sum(sn_dist.halo_exchange(wt_local, R["plan_f"]), w_bias, graph_f)
            else:
                out0 = ops.gather_sum(sn_dist.all_gather_rows(wt_local, part), w_bias, graph_f)
            return ops.blend(out0, out1, beta)
        R["step"] = step
        return R

    def measure(R, steps, warmup):
        dt, out = time_loop(R["step"], warmup, steps, sync_all)
        e_prime = R["graph"].num_edges
        halo_rows = (R["plan"].n_halo if R["plan"] is not None else R["n"] * (world - 1)) if world > 1 else 0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            es = torch.tensor([float(e_prime), float(halo_rows)], dtype=torch.float64, device=device)
            dist.all_reduce(es)
            return float(tt.item()), int(es[0].item()), int(es[1].item()), out
        return dt, e_prime, 0, out

    R = build_rank(args.locality, args.exchange)
    n, c, ei, x, h_local, graph, plan = R["n"], R["c"], R["ei"], R["x"], R["h_local"], R["graph"], R["plan"]
    e_prime = graph.num_edges
    # device preheat (--preheat-ms): the same step, untimed, until the device has been busy that long
    preheat_steps = 0
    if args.preheat_ms > 0:
        t_probe, _ = time_loop(R["step"], 1, 3, sync_all)
        preheat_steps = int(min(20000, max(1, np.ceil(args.preheat_ms * 1e-3 / max(t_probe / 3, 1e-6)))))
        if world > 1:       # every rank runs the same number of steps (collectives inside)
            cnt = torch.tensor([preheat_steps], dtype=torch.int64, device=device)
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
            preheat_steps = int(cnt.item())
        time_loop(R["step"], 0, preheat_steps, sync_all)
    dt, e_all, halo_all, out = measure(R, args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3

    # N > 1: the two exchange forms must agree on this rank's rows (RCCL all_to_all_single vs
    # all_gather_into_tensor: a self-check of the collectives on the real node)
    exchange_check = None
    if world > 1 and not plus_plus:
        other = build_rank(args.locality, "allgather" if args.exchange == "halo" else "halo")
        diff = (other["step"]() - R["step"]()).abs().max()
        dist.all_reduce(diff, op=dist.ReduceOp.MAX)
        exchange_check = float(diff.item())
        if not exchange_check <= 1e-6:
            raise SystemExit(f"halo and all-gather forms disagree: max |diff| = {exchange_check}")
        del other

    result = None
    table = h_local
    if world > 1:       # the local table of this rank, as the timed step sees it (a collective: every rank)
        with torch.no_grad():
            table = (sn_dist.halo_exchange(h_local, plan) if plan is not None
                     else sn_dist.all_gather_rows(h_local, R["part"]))
    if rank == 0:
        # --- roofline leg: the dominant kernel's average launch duration, HIP events on its stream
        # around batches of 20 back-to-back launches (nothing subtracted); beside it the interval
        # of a single launch between two events (which carries the event pair's own ~4.5 us)
        norm_ms, main_ms, fin_ms, _ = profile_forward(lib, _lib, ops, graph, table, args.top_k, args.thr, 6, 20)
        _, main_single_ms, _, empty_ms = profile_forward(lib, _lib, ops, graph, table, args.top_k, args.thr, 30, 1)
        # where the split rows were finalized (sngnn_tuning_set knob 9): inside the main launch - then kernel_ms
        # holds that work too and finalize_kernel_ms is an empty interval - or in a launch of its own
        fin_wgs = int(lib.sngnn_last_forward_finalize_workgroups())
        fin_ab = None
        if fin_wgs > 0 and world == 1 and not plus_plus and not args.no_finalize_ab:
            # the same step with the finalize as a launch of its own (round 4's form), same box, same graph
            lib.sngnn_tuning_set(9, 0)
            dt0, _, _, _ = measure(R, args.steps, args.warmup)
            n0, m0, f0, _ = profile_forward(lib, _lib, ops, graph, table, args.top_k, args.thr, 6, 20)
            lib.sngnn_tuning_set(9, 1)
            fin_ab = {"ms_per_step": dt0 / args.steps * 1e3, "normalize_kernel_ms": n0, "kernel_ms_batched_events": m0,
                      "finalize_kernel_ms": f0}
        b_alg = algorithmic_bytes(e_prime, n, c)
        traffic = rocprof_us = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                prof = json.load(open(tpath))
                traffic = prof.get(f"{args.workload}_k{args.top_k}")
                # the same kernel's average duration in the committed rocprofv3 run of this command
                rocprof_us = prof.get(f"{args.workload}_k{args.top_k}_kernel_us_rocprofv3")
                traffic_src = prof.get(f"source_{args.workload}", prof.get("source") if args.workload == "arxiv" else None)
            except Exception:
                traffic = rocprof_us = None
        # the floor under the kernel: the same graph's column list gathered over the same table with no
        # arithmetic at all (sngnn_gather_floor), timed here, on this box
        floor_ms = floor_edges_ms = None
        if c % 4 == 0 and c <= 256:
            floor_edges_ms = gather_floor(lib, _lib, graph, table, 0)
            floor_ms = gather_floor(lib, _lib, graph, table, 1)
        # the floor under the normalisation pass: a device copy of the same table
        copy_ms = copy_floor(table[:n] if table.size(0) != n else table)
        # the kernel figure: the live batched-event average, but never less than the committed
        # rocprofv3 average of the same command (dispatch to completion of one kernel: back-to-back
        # launches of one kernel can overlap a ramp with a drain and read lower); kernel_ms_source
        # says which of the two it is
        main_batched_ms = main_ms
        kernel_ms_source = "batched_events"
        if rocprof_us is not None and world == 1 and not plus_plus and args.scale == 1.0 and rocprof_us * 1e-3 > main_ms:
            main_ms = rocprof_us * 1e-3
            kernel_ms_source = "rocprofv3"
        kernel_achieved = b_alg / (main_ms * 1e-3) / 1e9
        # the headline: B_fwd over the WHOLE forward as the driver times it (every launch of a step -
        # normalise, main kernel, finalize - and the gaps between them): BASELINE.md's gate B_fwd / t_fwd
        achieved = b_alg / (ms_per_step * 1e-3) / 1e9
        layer = ("one SNGNN_Plus_Plus layer forward (adjacency branch + aggregation + blend)" if plus_plus
                 else "SNGNN_Plus aggregation forward")
        if world == 1:
            parallelism = "single GPU"
        else:
            parallelism = ("node-range partition + RCCL " +
                           ("halo exchange (all-to-all-v of the referenced rows, received into the [own | halo] "
                            "table; interior rows aggregated while it is in flight)" if plan is not None
                            else "all-gather of h") + (", w sharded by node range" if plus_plus else ""))
        result = {
            "metric": "similarity-aggregation edges/sec",
            "value": e_all / (dt / args.steps),
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "preheat": {"ms": args.preheat_ms, "steps": preheat_steps,
                        "what": "untimed steps in front of the warmup steps (the same step): a device coming out of idle runs "
                                "its first ~15 ms ~7 % slower; cold_start = the same W + K measurement from an idle device"},
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ogbn-{args.workload}-sized synthetic graph, {layer}, "
                                   f"top_k={args.top_k}, thr={args.thr}, self-loops removed"
                                   + (f", {args.locality:.0%} of a row's sources inside its rank's node range"
                                      if world > 1 and not strong else ""),
                       "nodes_per_gpu": n, "edges_per_gpu": e_prime, "channels": c,
                       "parallelism": parallelism},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "what": "SURVEY.md 8d's algorithmic bytes of one forward over ms_per_step: the whole step "
                                 "(normalise + main + finalize launches and the gaps), not one kernel of it; "
                                 "the dominant kernel alone is priced under kernel_frac",
                         "algorithmic_bytes": b_alg, "ms": ms_per_step,
                         "kernel": "k_agg_fwd", "kernel_ms": main_ms, "kernel_ms_source": kernel_ms_source,
                         "kernel_achieved": kernel_achieved, "kernel_frac": kernel_achieved / HBM_PEAK_GBS,
                         "gather_floor_ms": floor_ms, "gather_floor_edges_only_ms": floor_edges_ms,
                         "kernel_over_floor": None if not floor_ms else main_batched_ms / floor_ms,
                         "gather_floor_what": "sngnn_gather_floor on this graph and table, this box, batched events like "
                                              "kernel_ms_batched_events: every row read and row store of B_fwd (one 4C-byte row "
                                              "per entry of the column list in CSR order, own row in, output row out) with no "
                                              "arithmetic, no selection, no second fetch of kept rows; edges_only = the column "
                                              "list's gather alone.  kernel_over_floor = kernel_ms_batched_events / gather_floor_ms",
                         "normalize_floor_ms": copy_ms,
                         "step_over_floors": None if not floor_ms else ms_per_step / (floor_ms + copy_ms),
                         "floors_what": "normalize_floor_ms = a plain device copy of the table (the normalisation pass's bytes, no "
                                        "arithmetic; torch copy_, batched events); step_over_floors = ms_per_step / (gather_floor_ms + "
                                        "normalize_floor_ms): the whole step against the two measured floors of its two passes, "
                                        "launch boundaries not counted",
                         "normalize_kernel_ms": norm_ms, "finalize_kernel_ms": fin_ms,
                         "finalize_workgroups_in_main": fin_wgs,
                         "finalize_what": ("the split rows (in-degree > 128) are finalized by the last workgroups of the main "
                                           "launch, each row as soon as its tasks have published their candidates: kernel_ms "
                                           "holds that work, finalize_kernel_ms is an empty interval; "
                                           "finalize_as_a_launch = the same step with sngnn_tuning_set(9, 0)") if fin_wgs > 0
                         else "the split rows' finalize is the launch behind the main kernel",
                         "finalize_as_a_launch": fin_ab,
                         "launches_ms_sum": norm_ms + main_batched_ms + fin_ms,
                         "kernel_ms_batched_events": main_batched_ms,
                         "timer": "kernel_ms = max(HIP events on the launch stream around batches of 20 back-to-back "
                                  "launches recorded inside the library: interval / 20, nothing subtracted; the committed "
                                  "rocprofv3 --kernel-trace average of this command) - kernel_ms_source names the one used",
                         "kernel_ms_single_launch_events": main_single_ms,
                         "kernel_frac_single_launch_events": b_alg / (main_single_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "empty_event_interval_ms": empty_ms,
                         "kernel_ms_rocprofv3": None if rocprof_us is None else rocprof_us * 1e-3,
                         "kernel_frac_rocprofv3": None if rocprof_us is None
                         else b_alg / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
            # (the same figure under its round-3 name)
            "roofline_step": {"bound": "hbm", "algorithmic_bytes": b_alg, "ms": ms_per_step,
                              "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS},
        }
        if plus_plus:
            # the whole ++ layer against its own byte model ("++ branch extra")
            b_pp = algorithmic_bytes(e_prime, n, c, plus_plus=True)
            result["roofline_layer"] = {"bound": "hbm", "algorithmic_bytes": b_pp,
                                        "achieved": b_pp / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": b_pp / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "what": "adjacency branch + aggregation + blend over the wall time of a step"}
    if world > 1:
        full = (n * world - n) * c * 4
        exch = {"form": args.exchange, "locality": args.locality,
                "rows_received_per_rank": halo_all / world,
                "bytes_received_per_rank": halo_all / world * c * 4,
                "full_allgather_bytes_per_rank": full,
                "fraction_of_allgather": halo_all / world * c * 4 / max(full, 1),
                "interior_rows_fraction": None if plan is None else 1.0 - plan.n_boundary / max(n, 1),
                "forms_agree_max_abs_diff": exchange_check}
        if rank == 0:
            result["exchange"] = exch
        # the same job on a graph with locality (what a node-range partition is for): second line
        if not strong and args.locality != 0.8:
            R2 = build_rank(0.8, args.exchange)
            dt2, e2, halo2, _ = measure(R2, args.steps, max(args.warmup // 2, 2))
            if rank == 0:
                result["locality_0.8"] = {"value": e2 / (dt2 / args.steps), "unit": "edges/s",
                                          "ms_per_step": dt2 / args.steps * 1e3,
                                          "rows_received_per_rank": halo2 / world,
                                          "bytes_received_per_rank": halo2 / world * c * 4,
                                          "interior_rows_fraction": None if R2["plan"] is None
                                          else 1.0 - R2["plan"].n_boundary / max(n, 1)}
            del R2

    # --- extras on one GPU
    if world == 1 and rank == 0 and not plus_plus:
        # BASELINE.md's other config-4 cases, timed the same way inside this run
        if not args.no_variants and args.workload == "arxiv" and args.channels is None:
            variants = {}
            for name, (vc, vthr) in {"C32_thr0.0": (32, 0.0), "C40_thr0.9": (40, 0.9)}.items():
                vn, vcc, vei, _, vh, _ = make_rank_inputs(args.workload, 0, 1, args.seed, device, vc, args.scale)
                vg = graph if vc == c else Graph(vei, vn, True, True)
                vdt, _ = time_loop(lambda: ops.aggregate_forward(vg, vh, args.top_k, vthr)[0], 5,
                                   min(args.steps, 50), torch.cuda.synchronize)
                vms = vdt / min(args.steps, 50) * 1e3
                _, vmain, _, _ = profile_forward(lib, _lib, ops, vg, vh, args.top_k, vthr, 4, 20)
                vb = algorithmic_bytes(vg.num_edges, vn, vcc)
                vtraffic = None
                try:
                    vtraffic = json.load(open(tpath)).get(f"{args.workload}_k{args.top_k}_{name}")
                except Exception:
                    pass
                variants[name] = {"channels": vcc, "thr": vthr, "ms_per_step": vms,
                                  "edges_per_s": vg.num_edges / (vms * 1e-3), "algorithmic_bytes": vb,
                                  "kernel_ms": vmain,
                                  # EFFECTIVE fractions: B_fwd of the unpruned byte model over the time.  A threshold
                                  # that prunes (the fp16 filter: no fp32 row for an edge that cannot be kept) moves
                                  # fewer bytes than B_fwd - `traffic` is the counter traffic of the committed PMC
                                  # pass of this variant - so this is a speed in roofline units, not a bandwidth
                                  "effective_frac": vb / (vmain * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "effective_frac_step": vb / (vms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": vtraffic}
            # the same call scored on the fly from h (sngnn_tuning_set(2, 2): no normalisation pass, no table - ONE launch
            # with the finalize inside; same selections bit for bit).  Not the library's default for ranking calls: every
            # decision within 2 delta of a cut is re-scored exactly, so its time depends on the DATA (nearly parallel rows
            # - a deep layer's input - re-score everything), where the table form costs the same whatever h holds
            lib.sngnn_tuning_set(2, 2)
            odt, _ = time_loop(lambda: ops.aggregate_forward(graph, h_local, args.top_k, args.thr)[0], 20,
                               args.steps, torch.cuda.synchronize)
            lib.sngnn_tuning_set(2, 0)
            oms = odt / args.steps * 1e3
            variants["C40_thr0.0_on_the_fly"] = {
                "channels": c, "thr": args.thr, "ms_per_step": oms, "edges_per_s": e_prime / (oms * 1e-3),
                "algorithmic_bytes": b_alg, "effective_frac_step": b_alg / (oms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "what": "scored on the fly from h: one launch; data-dependent (see bench.py), not the default for ranking calls"}
            result["variants"] = variants
        hg = h_local.clone().requires_grad_(True)
        gout = torch.randn_like(h_local)
        for _ in range(5):
            ops.aggregate(hg, graph, args.top_k, args.thr).backward(gout)
        torch.cuda.synchronize()
        # (secondary figure: the median of five batches of ten - one stall of the box inside a
        # single 50-call loop once doubled it)
        batches = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                hg.grad = None
                ops.aggregate(hg, graph, args.top_k, args.thr).backward(gout)
            torch.cuda.synchronize()
            batches.append((time.perf_counter() - t0) / 10 * 1e3)
        result["fwd_bwd_ms"] = float(np.median(batches))
        # the same pair replayed from a HIP graph: the eager figure above is bound by the HOST on most
        # boxes (one autograd round trip costs ~200 us of Python / engine time against ~130 us of GPU
        # work: tools/host_overhead.py), this one by the device
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for _ in range(3):
                    hg.grad = None
                    ops.aggregate(hg, graph, args.top_k, args.thr).backward(gout)
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize()
            cg = torch.cuda.CUDAGraph()
            hg.grad = None
            with torch.cuda.graph(cg, stream=side):
                ops.aggregate(hg, graph, args.top_k, args.thr).backward(gout)
            for _ in range(5):
                cg.replay()
            torch.cuda.synchronize()
            batches = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(20):
                    cg.replay()
                torch.cuda.synchronize()
                batches.append((time.perf_counter() - t0) / 20 * 1e3)
            result["fwd_bwd_graphed_ms"] = float(np.median(batches))
            del cg
        except Exception as ex:      # noqa: BLE001  (a secondary figure: never fail the line for it)
            result["fwd_bwd_graphed_ms"] = None
            result["fwd_bwd_graphed_note"] = repr(ex)[:200]
        result["roofline_bwd"] = backward_roofline(ops, graph, h_local, gout, args.top_k, args.thr, e_prime, n, c, args.workload)
        if not args.no_epoch:
            from sngnn_amd.train import epoch_time_ms
            # reference-style eager loop (train.py:73-143) and the same epoch replayed from a HIP
            # graph (sngnn_amd/train.py:GraphedEpoch).  epoch_ms = the reference's THREE forwards
            # (train, validation, test: train.py:134-143); epoch_ms_shared_eval: validation and
            # test read ONE eval-mode forward (bit-identical metrics)
            result["epoch_ms_eager"] = epoch_time_ms(args.workload, x, ei, n, c, args.top_k,
                                                     args.thr, seed=args.seed)
            result["epoch_ms"] = epoch_time_ms(args.workload, x, ei, n, c, args.top_k, args.thr,
                                               seed=args.seed, graphed=True, share_eval_forward=False)
            result["epoch_ms_shared_eval"] = epoch_time_ms(args.workload, x, ei, n, c, args.top_k, args.thr,
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
            # the literal third-party calls too, where they import
            lit = cpu_reference_literal(h_cpu, ei_cpu, args.top_k, args.thr, reps=3)
            if isinstance(lit, str):
                result["cpu_baseline"]["reference"] = None
                result["cpu_baseline"]["reference_note"] = lit
            else:
                result["cpu_baseline"]["reference"] = {
                    "value": e_prime / lit[0], "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "reference",
                    "sample": "3 forward passes with the real torch_scatter.scatter_max / scatter(reduce='mean') "
                              f"at the reference's call sites ({lit[0] * 1e3:.0f} ms each)",
                    "max_abs_diff_vs_port": float((lit[1]["out"] - res["out"]).abs().max())}
            # the benchmarked output is the checked output
            err = (out.cpu() - res["out"]).abs().max().item()
            result["max_abs_err_vs_oracle"] = err
    if world == 1 and rank == 0 and plus_plus:
        result.update(products_training(ops, graph, R, args, e_prime, n, c))
    if world == 1 and rank == 0 and args.preheat_ms > 0:
        # the same W + K measurement from an idle device (the CPU baseline above left it idle for ~a second; a short
        # sleep for the runs that skipped it)
        torch.cuda.synchronize()
        time.sleep(1.0)
        dt_cold, _, _, _ = measure(R, args.steps, args.warmup)
        result["cold_start"] = {"ms_per_step": dt_cold / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
                                "what": "W warmup + K timed steps straight after one second of idle, no preheat"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def backward_roofline(ops, graph, h, gout, top_k, thr, e_prime, n, c, workload="arxiv"):
    """SURVEY.md 8d's B_bwd over the device time of the backward's launches: torch events on the
    launch stream around batches of 10 back-to-back backward calls, nothing subtracted."""
    _, wsel, *_ = ops.aggregate_forward(graph, h, top_k, thr, save_for_backward=True)
    n_sel = int((wsel > -3.0).sum())
    # one backward call as the autograd function makes it: from the kept bits its training forward wrote
    # itself where the library offers that (no packing launch), else from the per-edge weights
    bits = ops.kept_bits_supported(graph, top_k, c)
    if bits:
        _, kb = ops._forward_epilogue(graph, h, None, top_k, thr, True, None, None, True)
        call = lambda: ops.aggregate_backward_bits(graph, h, gout, kb, top_k)      # noqa: E731
    else:
        call = lambda: ops.aggregate_backward(graph, h, gout, wsel, top_k)        # noqa: E731
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    bw = []
    for _ in range(8):
        ev[0].record()
        for _ in range(10):
            call()
        ev[1].record()
        ev[1].synchronize()
        bw.append(ev[0].elapsed_time(ev[1]) / 10)
    bwd_ms = float(np.median(bw[2:]))          # (median of six batches of ten: robust against a stall of the box)
    b_bwd = backward_bytes(e_prime, n_sel, n, c)
    traffic = None
    try:       # counter bytes per backward call of the committed PMC passes (the full-size default workloads only)
        if (e_prime, n, c, top_k) in ((1163820, 169343, 40, 16), (123506252, 2449029, 48, 16)):
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"{workload}_k{top_k}_bwd")
    except Exception:
        traffic = None
    return {"bound": "hbm", "kernels": ("k_bwd_w (node-centric, one launch: the kept bits come from the training forward itself)" if bits else
                        "k_pack_kept + k_bwd_w (node-centric: one backward call as the autograd function makes it, "
                        "with the forward's top_k)" if 2 * graph.num_fused_nodes >= graph.num_nodes else
                        "k_clear_words + k_bwd_t + k_bwd_t_fin + k_bwd_s (the two passes: fewer than half of this graph's "
                        "nodes are small both as target and as source)"),
            "kernel_ms": bwd_ms, "kept_edges": n_sel, "algorithmic_bytes": b_bwd, "traffic": traffic,
            "achieved": b_bwd / (bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": b_bwd / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "timer": "torch events on the launch stream around batches of 10 backward calls (median of 6 batches)"}


def products_training(ops, graph, R, args, e_prime, n, c):
    """Config 5's layer in TRAINING mode on one GPU: forward + backward of the ++ layer
    (adjacency branch over the dense [C, N] w, aggregation, blend) and the Adam step over w -
    models.py:95,130 make w a 460 MB table whose dense gradient and optimizer step SURVEY.md 7
    flags as the dominant cost of a products epoch (train.py:86, :376)."""
    h = R["h_local"].clone().requires_grad_(True)
    w = torch.nn.Parameter(R["wt_local"].t())             # reference-shaped [C, N], column-major storage
    bias = torch.nn.Parameter(R["w_bias"].clone())
    beta = torch.nn.Parameter(R["beta"].clone())
    gout = torch.randn(n, c, device=h.device)
    opt = torch.optim.Adam([w, bias, beta], lr=0.01, weight_decay=5e-4, fused=True)

    def fwd_bwd():
        for p in (h, w, bias, beta):
            p.grad = None
        out = ops.blend(ops.adj_linear(w, bias, graph), ops.aggregate(h, graph, args.top_k, args.thr), beta)
        out.backward(gout)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    fb_ms = timed(fwd_bwd, 5)
    adam_ms = timed(opt.step, 5)
    extra = {"fwd_bwd_ms": fb_ms, "adam_step_w_ms": adam_ms,
             "adam_bytes": int(w.numel()) * 4 * 7,       # read w, grad, m, v; write w, m, v
             "adam_gbs": int(w.numel()) * 4 * 7 / (adam_ms * 1e-3) / 1e9,
             "train_step_ms": fb_ms + adam_ms,
             "roofline_bwd": backward_roofline(ops, graph, R["h_local"], gout, args.top_k, args.thr, e_prime, n, c, "products")}
    return extra


if __name__ == "__main__":
    main()
