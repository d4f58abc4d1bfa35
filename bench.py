#!/usr/bin/env python3
"""bench.py — BSBM Q5 join pipeline (BGP scan -> cross/hash joins -> FILTER) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank
per GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2]/[3]): a synthetic BSBM-shaped store of `--products` products
(285 000 products = 98 M triples, "BSBM-100M"); its three sorted permutations and the typed-value
table are resident in HBM before the timed region starts.

A step = ONE BATCH of `--queries` BSBM Explore Q5 instances (distinct %Product% constants) pushed
through the operator pipeline.  Default: the batch runs as one operator tree (`bsbm.q5_batch_plan`: the
reference's operators — HashJoinExec / FilterExec over the same seven triple patterns — with the
per-instance constant carried as a column, so every pattern partition is streamed once per batch;
results per instance are identical to the reference's per-query plan, tests/test_gpu_parity.py).
`--per-instance` runs the reference's per-query plan (Q5 (Execution Plan).snap) once per instance from
`--threads` host threads instead.  value = solution bindings (rows leaving the top join, before
DISTINCT / ORDER BY / LIMIT) per second over all ranks.

N > 1: the dataset is fixed (BSBM-100M, graph-sharded over the N GPUs) and, by default, the batch grows with N
(`--scaling weak`: N x `--queries` instances per step — the per-GPU join work stays what one GPU does at N = 1, every
row of C still crosses the exchange; `--scaling strong` keeps the batch fixed, and a short run of the OTHER mode is
reported under config.other_scaling either way).  Triples are sharded by hash(subject); each rank
evaluates the batch's constant-subject patterns on its shard (phase A), the resulting table C is re-sharded by
prodFeature with ONE hash repartition per step (rdfgpu_exchange_repartition: counts first, then the rows, RCCL
grouped send / recv), and the candidate join + FILTER pipeline (phase B) runs on the rank's object-sharded copy of
the productFeature triples (sharding.shard_dataset_hybrid): both sides of the join shrink with N.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling is 6290 GB/s


def scan_roofline(rf, device, log2_rows=26, threshold=1000, reps=10, distinct=2000, check=True):
    """BASELINE config 2 at a partition that cannot sit in the 256 MiB Infinity Cache: the Q1 numeric
    scan + FILTER (FilterExec: EBV(GT(ENC_TV(value1@1), 9:c)), projection=[product@0]) over ONE predicate
    partition of 2^log2_rows triples whose objects are `distinct` different xsd:integer literals.
    Bytes: the formula's 17 N + 4 sigma N counts 9 B of typed-value gather per row; with 2000 distinct
    literals (BSBM's value range) the 32 KB table never leaves the caches, so the HBM-roofline fraction is quoted on
    the STREAM bytes 8 N + 4 sigma N (two u32 columns read, survivors written); `distinct` >= 4 M makes the typed
    table (16 B per id) larger than the 32 MiB of L2 — then the gathers are real traffic too."""
    from rdf_fusion_amd import abi
    from rdf_fusion_amd.engine import TV_DTYPE
    from rdf_fusion_amd.plan import PlanBuilder, quad_pattern, col, integer, ENC_TV, GT, EBV
    n = 1 << log2_rows
    rng = np.random.default_rng(7)
    pred, int_base = 1, 2
    subj = np.arange(int_base + distinct, int_base + distinct + n, dtype=np.uint32)
    if distinct == 2000:
        val = np.clip(np.rint(rng.normal(1000, 333, n)), 1, 2000).astype(np.uint32)
        values = np.arange(1, 2001)
    else:                                    # uniform over a large dictionary of integer literals
        val = rng.integers(1, distinct + 1, n).astype(np.uint32)
        values = rng.integers(1, 2001, distinct)                # a random value per literal: survivors are scattered, not clustered by id
    obj = (int_base + val - 1).astype(np.uint32)
    store = rf.GpuQuadStore(device=device)
    store.extend(np.zeros(n, np.uint32), subj, np.full(n, pred, np.uint32), obj)
    tv = np.zeros(int_base + distinct, dtype=TV_DTYPE)
    tv["tag"][1:] = abi.TV_NAMED_NODE
    tv["tag"][int_base:] = abi.TV_INTEGER
    tv["lo"][int_base:] = values
    store.set_typed_values(tv)
    pb = PlanBuilder()
    src = pb.data_source(quad_pattern("product", pred, "value1"))
    desc = pb.build(pb.filter(src, EBV(GT(ENC_TV(col(1)), integer(threshold))), projection=[0]))
    plan = store.plan(desc).enable_kernel_timing(True)
    best, rows = None, 0
    for _ in range(reps):
        plan.execute()
        rows, _ = plan.result_info()
        # the scan + FILTER is two streaming passes and a 16 K-element scan: their summed HIP-event time is the operator's
        ks = [k for k in plan.kernel_stats() if any(x in k[0] for x in ("filter", "device scan", "small_scan", "value_verdict", "value_runs", "run_scan", "run_copy"))]
        ms = sum(k[2] for k in ks)
        if best is None or ms < best[1]:
            ks_best = ks
            best = ("+".join(k[0].replace("void rdfgpu::", "").replace("rdfgpu::", "") for k in ks), ms, sum(k[3] for k in ks),
                    {k[0]: round(k[2] * 1e3, 1) for k in ks})
    expect = int((values[val - 1] > threshold).sum())
    assert rows == expect or not check, (rows, expect)          # full-size parity: exact count against numpy
    plan.close(); store.close()
    name, ms, nbytes, parts = best
    stream = 8 * n + 4 * rows
    run_copy = "run_copy" in name
    # compulsory bytes of the path taken: the streaming form reads both columns and writes the survivors; the run-copy form
    # (sorted slice, few distinct ids) never streams the predicate column: survivors read once + written once
    compulsory = 8 * rows if run_copy else stream
    gbs = compulsory / (ms * 1e-3) / 1e9
    out = {"kernel": name, "kernel_us": parts, "rows": n, "distinct_literals": distinct, "selectivity": round(rows / n, 4), "best_us": round(ms * 1e3, 1),
           "path": "run copy (qualifying runs of the sorted slice)" if run_copy else "streaming (verdict bits, scan, ordered write)",
           "compulsory_bytes": int(compulsory), "stream_bytes": int(stream), "achieved": round(gbs, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
           "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy_ceiling": round(gbs / 6290.0, 4),
           "stream_equivalent_GBps": round(stream / (ms * 1e-3) / 1e9, 1),
           "formula_bytes_17N_4sN": int(17 * n + 4 * rows),
           "note": "frac = compulsory bytes of the path taken / time / 8 TB/s.  Streaming form: 8 N + 4 sigma N (two u32 columns + the survivors; "
                   "the 9 B/row typed gather of the SURVEY formula is cache traffic and NOT counted).  Run-copy form: 8 sigma N (survivors in + "
                   "out); `stream_equivalent_GBps` = (8 N + 4 sigma N) / time says how fast a streaming filter would have to be to match it "
                   "(it may exceed the HBM peak: those bytes are not moved)"}
    t, src_ = pmc_traffic([k[0] for k in ks_best] if run_copy else
                          ["void rdfgpu::filter_bits_kernel", "void rdfgpu::filter_write_kernel"], {"scan_rows": n, "distinct": distinct})
    if t:
        out["traffic"] = t; out["traffic_source"] = src_; out["traffic_frac"] = round(t / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return out


def pmc_traffic(kernels, workload, profiles_dir=None):
    """HBM bytes per launch of `kernels` (one name or several: summed) from the committed rocprofv3 PMC passes of THIS
    command (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 corrections applied by profiles/summarize.py) —
    counters cannot be read from inside the benchmark process, so the newest summary under profiles/ whose recorded
    workload (`_workload`) holds every key / value of `workload` is quoted, with its file name."""
    import glob
    import re
    if isinstance(kernels, str):
        kernels = [kernels]
    profiles_dir = profiles_dir or os.path.join(ROOT, "profiles")
    files = glob.glob(os.path.join(profiles_dir, "*_pmc_fetch_write_per_kernel.json"))
    files.sort(key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        w = d.get("_workload") or {}
        if any(w.get(k) != v for k, v in workload.items()):
            continue
        # only the NEWEST summary of the workload counts: an older one describes kernels that have changed since — and it counts
        # only while the device sources still hash to what it was collected on (a summary without the stamp predates it: stale)
        import rdf_fusion_amd
        if d.get("_source_sha16") != rdf_fusion_amd.kernel_source_sha16():
            return None, f"{os.path.basename(f)} is stale: collected on other kernel sources"
        def pick(kn):        # a kernel class may have several instantiations in the summary (template arguments): the one that ran most often
            m = [v for k, v in d.items() if not k.startswith("_") and k.startswith(kn.split("(")[0]) and isinstance(v, dict)]
            return max(m, key=lambda v: v.get("dispatches", 0)) if m else None
        found = [pick(kn) for kn in kernels]
        if all(e and "hbm_bytes_per_launch" in e for e in found):
            return int(sum(e["hbm_bytes_per_launch"] for e in found)), os.path.join("profiles", os.path.basename(f))
        return None, None
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--products", type=int, default=285_000, help="BSBM scale (285000 products = ~100 M triples)")
    ap.add_argument("--queries", type=int, default=262144, help="Q5 instances per step (the batch)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = N x --queries instances per step over the same sharded graph; strong = --queries instances per step")
    ap.add_argument("--per-instance", action="store_true", help="one reference plan per instance instead of one batched tree")
    ap.add_argument("--threads", type=int, default=4, help="--per-instance: host threads submitting queries")
    ap.add_argument("--cpu-sample", type=int, default=14, help="Q5 instances timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-shard-check", action="store_true", help="N > 1: skip comparing the shards' bindings with an unsharded run on rank 0")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: run exchange and join pipeline of a step back to back instead of pipelining steps")
    ap.add_argument("--no-scan", action="store_true", help="skip the scaled scan+FILTER roofline measurement")
    ap.add_argument("--scan-log2-rows", type=int, default=26)
    ap.add_argument("--no-table-cache", action="store_true",
                    help="the headline itself with RDFGPU_OPT_NO_TABLE_CACHE: every join table is built inside the timed step, like HashJoinExec(CollectLeft) per query")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-start / no-table-cache side measurements")
    ap.add_argument("--allow-host-transport", action="store_true",
                    help="N > 1: when the RCCL communicator cannot be created, measure over the host-staged transport instead of failing")
    args = ap.parse_args()

    import threading
    import torch
    import rdf_fusion_amd as rf
    from rdf_fusion_amd import bsbm, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # rehearsal knob: RDFGPU_BENCH_REHEARSE=1 runs every rank on GPU 0 with the gloo backend (1-GPU boxes)
    rehearse = os.environ.get("RDFGPU_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    xdev = "cpu" if rehearse else "cuda"   # where the exchanged tensors live

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ data, resident in HBM
    t0 = time.time()
    ds = bsbm.generate(args.products)
    # N > 1: subject shard (default graph) + the candidate join's own layout in a named graph (sharding.shard_dataset_hybrid)
    g, s, p, o = sharding.shard_dataset_hybrid(ds, rank, world) if world > 1 else sharding.shard_dataset(ds, rank, world)
    store = rf.GpuQuadStore(device=local_rank)
    store.extend(g, s, p, o)
    store.set_typed_values(ds.typed_values, ds.decimals)
    n_local = len(store)
    del g, s, p, o
    load_s = time.time() - t0

    rng = np.random.default_rng(12345)
    n_batches = args.steps + args.warmup
    Q = args.queries
    if Q > ds.n_products:
        raise SystemExit(f"--queries {Q} exceeds the {ds.n_products} products of this scale (instances of one batch are distinct products)")
    # every step gets its own query mix: Q distinct products, drawn independently per batch (same seed on every rank)
    all_products = np.array([ds.product(i) for i in range(ds.n_products)], dtype=np.uint32)
    def draw(n_inst):
        """one batch: draws of Q distinct products each (a batch of more than Q instances repeats products, as a BSBM driver's
        uniform draws would; instance tags stay distinct)"""
        parts = [all_products[rng.choice(ds.n_products, size=min(Q, n_inst - k), replace=False)] for k in range(0, n_inst, Q)]
        return np.ascontiguousarray(np.concatenate(parts))
    Q_step = Q * world if args.scaling == "weak" else Q      # instances per step over all ranks
    batches = [draw(Q_step) for _ in range(n_batches)]
    # N > 1: a short run of the other scaling mode after the timed region (3 warm-up + 5 timed steps), reported beside the headline
    Q_other = Q if args.scaling == "weak" else Q * world
    other_batches = [draw(Q_other) for _ in range(8)] if world > 1 and not args.per_instance else []
    products = np.concatenate(batches)                   # the instances the latency / CPU-baseline samples are taken from

    kstats = {}
    lock = threading.Lock()

    def mem(stage):
        free_b, total_b = torch.cuda.mem_get_info(local_rank)
        print(f"[bench] device memory after {stage}: {(total_b - free_b) / 2**30:.1f} GiB used of {total_b / 2**30:.0f}", file=sys.stderr, flush=True)

    def account(plan):
        with lock:
            for name, launches, ms, nbytes, rows in plan.kernel_stats():
                k = kstats.setdefault(name, [0, 0.0, 0, 0])
                k[0] += launches; k[1] += ms; k[2] += nbytes; k[3] += rows

    def dev_table(cols):
        """numpy u32 columns -> one device tensor + per-column pointers"""
        n = len(cols[0])
        flat = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint32) for c in cols]) if n else np.zeros((len(cols), 0), np.uint32))
        t = torch.from_numpy(flat.view(np.int32)).cuda()
        return t, [t.data_ptr() + 4 * n * k for k in range(len(cols))], n

    # The query parameters of every step are inputs: resident in HBM before the timed region starts (PARAMS(inst, X),
    # 32 KB per 4096-instance batch; instance tags are 1-based because 0 is null).
    resident = {}

    def params_on_device(batch):
        """PARAMS(inst, X) of one batch in HBM.  N > 1: the rows this rank's shard can answer — the instances whose %Product%
        it owns (the router's job, like sharding the graph itself; the instance tags stay the batch-wide ones)."""
        key = batch.ctypes.data
        if key not in resident:
            inst = np.arange(1, len(batch) + 1, dtype=np.uint32)
            if world > 1:
                mine = sharding.shard_of(batch, world) == rank
                resident[key] = dev_table([inst[mine], batch[mine]])
            else:
                resident[key] = dev_table([inst, batch])
        return resident[key]

    lat_ms = []
    phase_ms = [0.0, 0.0, 0.0, 0.0]    # sharded run, this rank: phase A (constant patterns), exchange (hash repartition of C by feature), phase B, main thread waiting for A + exchange

    # ------------------------------------------------------------------ the step
    if args.per_instance:
        assert world == 1, "--per-instance is a single-GPU mode"
        # Plans are described once, outside the timed region (the ctypes description is the Python stand-in
        # for DataFusion handing the subtree over); timed per query: rdfgpu_plan_compile + execute + count.
        descs = {int(x): bsbm.q5_plan(ds, int(x)) for x in np.unique(np.concatenate(batches))}
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(args.threads) if args.threads > 1 else None

        def one(x, timing):
            t1 = time.perf_counter()
            plan = store.plan(descs[int(x)])
            if timing:
                plan.enable_kernel_timing(True)
            plan.execute()
            n, _ = plan.result_info()
            dt = (time.perf_counter() - t1) * 1e3
            with lock:
                lat_ms.append(dt)
            if timing:
                account(plan)
            plan.close()
            return n

        def step(batch, timing):
            f = lambda x: one(x, timing)
            return sum(pool.map(f, batch)) if pool else sum(f(x) for x in batch)
    elif world == 1:
        if args.no_table_cache:
            store.set_option("NO_TABLE_CACHE", 1)
        plan = store.plan(bsbm.q5_batch_plan(ds))           # compiled once; only the bound PARAMS change

        def step(batch, timing):
            t, ptrs, n = params_on_device(batch)
            plan.bind_table(0, ptrs, n)
            plan.enable_kernel_timing(timing)
            plan.execute()
            rows, _ = plan.result_info()
            if timing:
                account(plan)
            return rows
    else:
        plan_a = store.plan(bsbm.q5_batch_const_plan(ds))
        plan_b = store.plan(bsbm.q5_batch_plan(ds, tables=True, graph=[sharding.candidate_graph(ds)]))
        # The exchange is part of the product: rdfgpu_exchange_repartition behind the C ABI (RCCL over xGMI, grouped send / recv,
        # buffers sized from the exchanged row counts).  torch.distributed only carries the RCCL unique id, the barriers and the
        # final reductions.  Two communicators: the gathered table of batch i stays valid while batch i + 1 is exchanged.
        def host_wire(blocks):     # the caller-supplied wire of the host-staged transport: an all-to-all of byte blocks
            sizes = torch.tensor([len(b) for b in blocks], dtype=torch.int64, device=xdev)
            rsizes = torch.empty_like(sizes)
            dist.all_to_all_single(rsizes, sizes)
            send = torch.from_numpy(np.concatenate([np.frombuffer(b, dtype=np.uint8) for b in blocks] + [np.zeros(0, np.uint8)]).copy()).to(xdev)
            recv = torch.empty(int(rsizes.sum().item()), dtype=torch.uint8, device=xdev)
            dist.all_to_all_single(recv, send, output_split_sizes=rsizes.tolist(), input_split_sizes=sizes.tolist())
            out, at = [], 0
            host = recv.cpu().numpy()
            for n_b in rsizes.tolist():
                out.append(host[at:at + n_b]); at += n_b
            return out
        transport = "RCCL over xGMI (grouped ncclSend / ncclRecv inside librdfgpu.so)"
        comms = []
        if rehearse:           # several ranks on ONE GPU (RCCL refuses that): the host-staged transport, gloo as the wire
            comms = [rf.Comm(rank, world, device=local_rank, host_alltoallv=host_wire) for _ in range(2)]
            transport = "host-staged, torch.distributed (gloo) as the wire: rehearsal on one GPU"
        else:
            why = None
            for _ in range(2):
                ids = [None]
                if rank == 0:
                    try:
                        ids = [rf.Comm.unique_id()]
                    except rf.RdfGpuError as e:
                        why = str(e)
                dist.broadcast_object_list(ids, src=0)
                ok = ids[0] is not None
                if ok:
                    try:
                        comms.append(rf.Comm(rank, world, device=local_rank, unique_id=ids[0]))
                    except rf.RdfGpuError as e:
                        ok, why = False, str(e)
                flag = torch.tensor([1 if ok else 0], device=f"cuda:{local_rank}")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    comms = None
                    break
            if comms is None and not args.allow_host_transport:
                # BASELINE's metric names RCCL: a line measured over PCIe + torch.distributed must not carry it by accident
                print(f"[bench] rank {rank}: the RCCL communicator could not be created ({why}); refusing to measure the sharded path over the "
                      "host-staged transport (pass --allow-host-transport to do that anyway)", file=sys.stderr, flush=True)
                dist.destroy_process_group()
                sys.exit(3)
            if comms is None:   # --allow-host-transport: the run still measures the sharded path, through the host-staged transport
                print(f"[bench] rank {rank}: RCCL communicator unavailable ({why}); falling back to the host-staged exchange", file=sys.stderr, flush=True)
                comms = [rf.Comm(rank, world, device=local_rank, host_alltoallv=host_wire) for _ in range(2)]
                transport = "host-staged, torch.distributed as the wire (the RCCL communicator could not be created)"

        def phase_a(batch, timing):
            """Phase A: the constant-subject patterns of the whole batch on the local shard, joined per instance:
            C(inst, X, prodFeature, origProperty1, origProperty2) for the instances whose %Product% lives here — in HBM."""
            t, ptrs, n = params_on_device(batch)
            plan_a.bind_table(0, ptrs, n)
            plan_a.enable_kernel_timing(timing)
            plan_a.execute()
            if timing:
                account(plan_a)
            return plan_a.result_device()

        def phase_b(tab, timing):
            """Phase B: the batch's join / FILTER pipeline over the local shard of the product-side patterns, probing with
            the gathered C (bound as it arrived: device columns owned by the communicator)."""
            ptrs, rows_all = tab
            plan_b.bind_table(0, ptrs, rows_all)
            plan_b.enable_kernel_timing(timing)
            plan_b.execute()
            rows, _ = plan_b.result_info()
            if timing:
                account(plan_b)
            return rows

        def step(batch, timing):
            """one batch, start to end (the sharded-result check; the timed loop pipelines the same three phases)"""
            ptrs, n = phase_a(batch, timing)
            return phase_b(comms[0].repartition(ptrs, n, 2), timing)

        def run_pipelined(bs, timing):
            """Software pipeline over independent batches: phase A and the repartition of batch i + 1 run on a helper host thread
            (their own plan, communicator and HIP streams) while phase B of batch i runs on the main thread's plan — the step
            costs max(phase A + exchange, phase B), not their sum.  Same work per batch as step()."""
            def produce(i, b, box):
                try:
                    t0 = time.perf_counter()
                    ptrs, n = phase_a(b, timing)
                    t1 = time.perf_counter()
                    box["tab"] = comms[i & 1].repartition(ptrs, n, 2)
                    box["t"] = ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3)
                except BaseException as e:      # surfaces on the main thread
                    box["err"] = e

            def start(i):
                box = {}
                th = threading.Thread(target=produce, args=(i, bs[i], box))
                th.start()
                return th, box

            total = 0
            nxt = start(0) if len(bs) else None
            for i in range(len(bs)):
                t_w = time.perf_counter()
                th, box = nxt
                th.join()
                if "err" in box:
                    raise box["err"]
                waited = (time.perf_counter() - t_w) * 1e3
                nxt = start(i + 1) if i + 1 < len(bs) else None
                t_b = time.perf_counter()
                total += phase_b(box["tab"], timing)
                if timing:
                    phase_ms[0] += box["t"][0]; phase_ms[1] += box["t"][1]; phase_ms[2] += (time.perf_counter() - t_b) * 1e3
                    phase_ms[3] += waited                                      # what phase A + exchange cost beyond the previous phase B
            return total

    if not args.per_instance:
        for b in batches + other_batches:
            params_on_device(b)
        torch.cuda.synchronize()
    # Cold start: the FIRST execution of the batch on a fresh store version builds every join table of the predicate slices
    # (direct / CSR / decoded-value tables, cached per store version) and sizes every operator exactly (one host sync per
    # join); the second execution speculates from the first one's cardinalities and fuses the look-up chains.  Both are
    # measured and reported (config.cold_ms / table_build_ms); the timed steps below are the steady state after them.
    cold = None
    if not args.per_instance:
        torch.cuda.synchronize()
        t1 = time.perf_counter(); step(batches[0], False); torch.cuda.synchronize(); cold_ms = (time.perf_counter() - t1) * 1e3
        t1 = time.perf_counter(); step(batches[0], False); torch.cuda.synchronize(); second_ms = (time.perf_counter() - t1) * 1e3
        t1 = time.perf_counter(); step(batches[0], False); torch.cuda.synchronize(); third_ms = (time.perf_counter() - t1) * 1e3
        cold = {"cold_ms": round(cold_ms, 3), "second_execution_ms": round(second_ms, 3), "third_execution_ms": round(third_ms, 3)}
        mem("cold start")
    overlap = world > 1 and not args.no_overlap
    if overlap:
        run_pipelined(batches[:args.warmup], False)
    else:
        for b in batches[:args.warmup]:
            step(b, False)
    # Per-kernel HIP events cost stream time (two events per launch, ~5 us each: a tenth of a 0.6 ms step).  The timed region
    # therefore brackets ONLY the kernel that took longest when every launch was timed (one untimed step here) — the kernel the
    # `roofline` object is about; the table of all kernels is measured right after the timed region, with every launch bracketed.
    focus = 1 if args.per_instance else 2
    if not args.per_instance:
        step(batches[0], 1)
    kstats.clear(); lat_ms.clear()
    barrier()
    t0 = time.perf_counter()
    local_rows = 0
    if overlap:
        local_rows = run_pipelined(batches[args.warmup:], focus)
    else:
        for b in batches[args.warmup:]:
            local_rows += step(b, focus)
    barrier()
    elapsed = time.perf_counter() - t0
    kstats_timed = dict(kstats)                # the dominant kernel(s), from the timed region
    kstats_steps = args.steps
    if not args.per_instance:                  # every kernel, from three further steps (not part of `value`)
        saved_phase = list(phase_ms)
        kstats.clear()
        extra = batches[args.warmup:args.warmup + min(3, args.steps)]
        for b in extra:
            step(b, 1)
        kstats_steps = len(extra)
        phase_ms[:] = saved_phase
        barrier()

    total_rows = local_rows
    if dist is not None:
        t = torch.tensor([elapsed, float(local_rows)], dtype=torch.float64, device=xdev)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item()); total_rows = int(tsum[1].item())

    # ------------------------------------------------------------------ the other scaling mode, briefly (same code path, other batch size)
    other = None
    if other_batches:
        saved = list(phase_ms)
        run = (lambda bs: run_pipelined(bs, False)) if overlap else (lambda bs: sum(step(b, False) for b in bs))
        run(other_batches[:3])
        barrier()
        t1 = time.perf_counter()
        rows_o = run(other_batches[3:])
        barrier()
        el_o = time.perf_counter() - t1
        t = torch.tensor([el_o, float(rows_o)], dtype=torch.float64, device=xdev)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        n_o = len(other_batches) - 3
        other = {"scaling": "strong" if args.scaling == "weak" else "weak", "instances_per_step": Q_other, "steps": n_o,
                 "ms_per_step": round(float(tmax[0].item()) * 1e3 / n_o, 3),
                 "bindings_per_s": round(float(tsum[1].item()) / float(tmax[0].item()), 2)}
        phase_ms[:] = saved

    # ------------------------------------------------------------------ sharded run: are the shards' bindings the unsharded answer?
    shard_check = None
    if dist is not None and not args.no_shard_check:
        def checksum(cols):      # order-independent (count, sum of per-row mixes mod 2^64) of a binding table
            a, b, c = (np.asarray(x, dtype=np.uint64) for x in cols)
            with np.errstate(over="ignore"):
                mix = a * np.uint64(0x9E3779B97F4A7C15) ^ b * np.uint64(0xC2B2AE3D27D4EB4F) ^ c * np.uint64(0x165667B19E3779F9)
                return len(a), int(mix.sum(dtype=np.uint64))
        probe_batch = batches[args.warmup][:min(Q, 8192)].copy()      # (distinct products: the first draw of the batch)
        step(probe_batch, False)
        n_loc, sum_loc = checksum(plan_b.fetch())
        t = torch.tensor([n_loc, sum_loc - (1 << 64) if sum_loc >= (1 << 63) else sum_loc], dtype=torch.int64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                 # int64 addition wraps: the sum is mod 2^64
        n_all, sum_all = int(t[0].item()), int(t[1].item()) & ((1 << 64) - 1)
        if rank == 0:                                            # the whole graph on one GPU, same batch, one operator tree
            full = rf.GpuQuadStore(device=local_rank)
            full.extend(ds.g, ds.s, ds.p, ds.o)
            full.set_typed_values(ds.typed_values, ds.decimals)
            pf = full.plan(bsbm.q5_batch_plan(ds))
            tt, pp, nn = dev_table([np.arange(1, len(probe_batch) + 1, dtype=np.uint32), probe_batch])
            pf.bind_table(0, pp, nn)
            n_ref, sum_ref = checksum(pf.execute().fetch())
            pf.close(); full.close()
            if (n_ref, sum_ref) != (n_all, sum_all):
                raise RuntimeError(f"sharded bindings differ from the unsharded run: {n_all} rows / {sum_all:#x} vs {n_ref} rows / {sum_ref:#x}")
            shard_check = f"{len(probe_batch)} instances: {n_all} bindings over {world} shards, count and multiset checksum equal to the unsharded run on rank 0"
        barrier()

    mem("timed steps")
    # ------------------------------------------------------------------ roofline of the dominant kernel
    # ONE formula: frac = compulsory bytes of the kernel / its time / 8 TB/s, compulsory = every input the
    # kernel has to read once + everything it has to write (recorded per launch next to the HIP-event time, plan.cpp
    # `timed`); never the bytes of the un-fused logical plan.  `traffic` = PMC FETCH + WRITE of the same kernel (committed
    # rocprofv3 passes), `traffic_over_compulsory` = how much of it is re-reads.
    roofline = None
    pipeline = None
    if kstats_timed:
        name, (launches, ms, nbytes, rows) = max(kstats_timed.items(), key=lambda kv: kv[1][1])
        achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # the committed counters are of a single-GPU batched run: quoted only for the workload they were collected on
        traffic, traffic_src = pmc_traffic(name, {"queries": args.queries, "products": args.products}) if world == 1 and not args.per_instance else (None, None)
        per_launch = nbytes / max(1, launches)
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_GBps": (round(traffic / (ms * 1e-3 / max(1, launches)) / 1e9, 1) if traffic and ms > 0 else None),
                    "traffic_frac": (round(traffic / (ms * 1e-3 / max(1, launches)) / 1e9 / HBM_PEAK_GBS, 4) if traffic and ms > 0 else None),
                    "traffic_over_compulsory": (round(traffic / per_launch, 2) if traffic and per_launch else None),
                    "note": "achieved = compulsory bytes of this kernel (inputs once + output, per launch) / its HIP-event time; a pair-test "
                            "kernel (band_mask_kernel) is bound by VALU compares over |group| x |rows| pairs, not by HBM: its fraction is "
                            "low by construction (DESIGN.md 6).  Only this kernel's launches are bracketed with events inside the timed region "
                            "(every launch bracketed costs ~0.06 ms per step); `kernels` and `pipeline_roofline` are from three further steps "
                            "with every launch bracketed",
                    "launches": launches, "avg_us": round(ms * 1e3 / max(1, launches), 2),
                    "compulsory_bytes_per_launch": int(per_launch)}
        if roofline["frac"] > 1.0:   # a join kernel whose recorded bytes are the SURVEY 8d formula of a logical join it short-cuts: not a roofline number
            roofline["frac"] = None
            roofline["note"] += "; INVALID here: the recorded bytes of this kernel are a logical-join formula, not compulsory bytes"
        # the whole step as ONE fused operator: what HBM has to deliver at least (the store slices and the parameters
        # read once, the bindings written once) over the device time of a step
        if world == 1 and not args.per_instance:
            pr = ds.pred
            cnt = lambda pname: int((ds.p == pr[pname]).sum())
            rows_out = total_rows / args.steps
            in_bytes = 8 * Q + 8 * (cnt("bsbm:productFeature") + cnt("bsbm:productPropertyNumeric1") + cnt("bsbm:productPropertyNumeric2") + cnt("rdfs:label"))
            out_bytes = 12 * rows_out
            dev_ms = sum(v[1] for v in kstats.values()) / kstats_steps
            pipeline = {"compulsory_bytes_per_step": int(in_bytes + out_bytes), "inputs_bytes": int(in_bytes), "output_bytes": int(out_bytes),
                        "kernel_ms_per_step": round(dev_ms, 3),
                        "achieved": round((in_bytes + out_bytes) / (dev_ms * 1e-3) / 1e9, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "frac": round((in_bytes + out_bytes) / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "PARAMS + the four predicate slices (two u32 columns each) read once + 3 u32 columns of bindings written, "
                                "over the summed kernel time of a step: the fraction of the HBM roofline the WHOLE join pipeline reaches"}
    kernel_table = {k: {"launches": v[0], "total_ms": round(v[1], 3), "avg_us": round(v[1] * 1e3 / max(1, v[0]), 1),
                        "compulsory_GBps": round(v[2] / (v[1] * 1e-3) / 1e9, 1) if v[1] > 0 else None}
                    for k, v in sorted(kstats.items(), key=lambda kv: -kv[1][1])}

    # ------------------------------------------------------------------ the fused step with its tables rebuilt INSIDE the step (N = 1)
    # The fair middle between the steady state and the un-fused no-cache step: the same compiled plan, the same fused operators,
    # but every cached join table of the store is dropped (and the store version bumped: ranges are located again) before each
    # step — rdfgpu_store_drop_tables, what any extend / remove does to the caches — so the step itself builds the CSR / direct /
    # decoded-value tables and the band join's entries again before it joins.
    fused_rebuild = None
    if world == 1 and not args.per_instance and not args.no_cold and not args.no_table_cache and cold is not None:
        ms_fr, rows_fr, detail = [], 0, None
        for b in batches[-min(5, len(batches)):]:
            store.drop_tables()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rows_fr += step(b, False)
            ms_fr.append((time.perf_counter() - t2) * 1e3)
            m_fr = plan.metrics()
            detail = {"host_syncs": m_fr.host_syncs, "kernels_launched": m_fr.kernels_launched, "tables_built": m_fr.tables_built,
                      "device_mallocs": m_fr.device_mallocs, "exact_reruns": m_fr.exact_reruns, "device_ms": round(m_fr.elapsed_compute_ms, 3)}
            print(f"[bench] fused-rebuild step: {ms_fr[-1]:.2f} ms wall, {detail}", file=sys.stderr, flush=True)
        med = float(np.median(ms_fr))
        fused_rebuild = {"ms_per_step": round(med, 3), "ms_per_step_mean": round(float(np.mean(ms_fr)), 3), "ms_per_step_min": round(float(np.min(ms_fr)), 3),
                         "bindings_per_s": round(rows_fr / len(ms_fr) / (med * 1e-3), 2), "steps": len(ms_fr), "last_step": detail,
                         "what": "the timed plan (fused: ordered slice join + band join) with every cached join table dropped and the store version bumped "
                                 "before each step: CSR / direct / decoded-value tables and the band join's entries are rebuilt inside the step"}
        cold["fused_rebuild"] = fused_rebuild
        step(batches[0], False)     # (leave the steady state as it was)

    # ------------------------------------------------------------------ the same step without any cached table (N = 1)
    # RDFGPU_OPT_NO_TABLE_CACHE: nothing survives an execution — every HashJoinExec builds its table inside the step, like
    # the reference's HashJoinExec(CollectLeft) does per query (..Q5 (Execution Plan).snap:10-30).  Like for like with a
    # per-query engine; the headline above is the steady state with the slices' tables cached per store version.
    if world == 1 and not args.per_instance and not args.no_cold and not args.no_table_cache and cold is not None:
        plan_nc = store.plan(bsbm.q5_batch_plan(ds)).set_option("NO_TABLE_CACHE", 1)
        def step_nc(batch):
            t, ptrs, n = params_on_device(batch)
            plan_nc.bind_table(0, ptrs, n)
            plan_nc.execute()
            return plan_nc.result_info()[0]
        step_nc(batches[0]); step_nc(batches[1 % len(batches)])
        torch.cuda.synchronize()
        n_nc = min(5, len(batches))
        rows_nc, step_ms, nc_detail = 0, [], []
        for b in batches[-n_nc:]:
            t2 = time.perf_counter()
            rows_nc += step_nc(b)
            step_ms.append((time.perf_counter() - t2) * 1e3)
            m_nc = plan_nc.metrics()
            nc_detail.append({"ms": round(step_ms[-1], 2), "device_ms": round(m_nc.elapsed_compute_ms, 2), "host_syncs": m_nc.host_syncs,
                              "exact_reruns": m_nc.exact_reruns, "device_mallocs": m_nc.device_mallocs, "device_malloc_ms": round(m_nc.device_malloc_ms, 2)})
            print(f"[bench] no-table-cache step: {step_ms[-1]:.1f} ms wall, {m_nc.elapsed_compute_ms:.1f} ms device, {m_nc.host_syncs} syncs, "
                  f"{m_nc.kernels_launched} launches, {m_nc.device_bytes / 2**30:.1f} GiB of intermediates, {m_nc.exact_reruns} exact re-runs, "
                  f"{m_nc.device_mallocs} hipMallocs ({m_nc.device_malloc_ms:.1f} ms)", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        # full-size cross-check of two executions that share no join kernel: the last step's bindings of the un-fused, un-cached plan (partitioned
        # join + streaming joins) against the timed plan's (ordered slice join + band join) on the same batch — row count and an order-independent checksum
        def checksum(cols):
            mix = np.zeros(len(cols[0]), dtype=np.uint64)
            with np.errstate(over="ignore"):
                for k, c in enumerate(cols):
                    mix ^= c.astype(np.uint64) * np.uint64([0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9][k % 3])
                return len(cols[0]), int(mix.sum(dtype=np.uint64))
        chk_nc = checksum(plan_nc.fetch())
        step(batches[-1], False)
        chk_timed = checksum(plan.fetch())
        if chk_nc != chk_timed:
            raise SystemExit(f"[bench] the un-cached plan's bindings differ from the timed plan's on the same batch: {chk_nc} vs {chk_timed}")
        step(batches[0], False)     # (leave the steady state as it was)
        # a side measurement over 5 long steps: the median, with the mean and the fastest beside it.  (Its 0.54 G-row candidate
        # join used to reserve output per full queue — 2.1 M same-address atomics — and ran at 27 ms or at 250-350 ms from step
        # to step; it now counts a partition's matches first and reserves once per partition: 30 ms per step, every step.)
        ms_nc = float(np.median(step_ms))
        mem("no-table-cache steps")
        plan_nc.close()
        mem("closing the no-table-cache plan")
        steady = elapsed * 1e3 / args.steps
        cold["no_table_cache"] = {"ms_per_step": round(ms_nc, 3), "ms_per_step_mean": round(float(np.mean(step_ms)), 3), "ms_per_step_min": round(float(np.min(step_ms)), 3), "bindings_per_s": round(rows_nc / n_nc / (ms_nc * 1e-3), 2), "steps": n_nc,
                                  "per_step": nc_detail,
                                  "result_check": f"the last step's {chk_nc[0]} bindings: row count and order-independent checksum equal to the timed (fused, cached) plan's on the same batch",
                                  "what": "every join table (hash / CSR / direct) built inside the timed step: HashJoinExec-style per-query builds"}
        # the reference's protocol applied to the batch: a FRESH plan per batch — compile, bind, execute, row count, drop (store tables warm)
        fresh_ms = []
        for rep in range(6):
            t, ptrs, n = params_on_device(batches[rep % len(batches)])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pf_ = store.plan(bsbm.q5_batch_plan(ds))
            pf_.bind_table(0, ptrs, n)
            pf_.execute()
            pf_.result_info()
            pf_.close()
            fresh_ms.append((time.perf_counter() - t1) * 1e3)
        cold["fresh_plan_per_batch_ms"] = round(float(np.median(fresh_ms[1:])), 3)
        cold["table_build_ms"] = round(max(0.0, cold["cold_ms"] - cold["second_execution_ms"]), 3)
        gain = ms_nc - steady
        cold["steps_to_amortise"] = (round(cold["table_build_ms"] / gain, 2) if gain > 0 else None)

    # ------------------------------------------------------------------ single-instance latency + CPU baseline, rank 0, N=1
    cpu = None
    single = None
    single_drained = None
    timed_check = None
    if rank == 0 and world == 1:
        sample = [int(x) for x in products[:max(8, args.cpu_sample)]]
        lat, lat_drained = [], []
        for x in sample + [int(y) for y in products[len(sample):len(sample) + 24]]:   # the reference's per-query protocol, one stream:
            d = bsbm.q5_plan(ds, x)                                                     # compile a plan for ONE instance, execute, drain, drop
            t1 = time.perf_counter()                                                    # (bench/benches/bsbm_explore.rs:23-95, utils/mod.rs:8-31)
            pl = store.plan(d).execute()
            pl.result_info()
            t2 = time.perf_counter()
            pl.fetch()
            pl.close()
            t3 = time.perf_counter()
            lat.append((t2 - t1) * 1e3); lat_drained.append((t3 - t1) * 1e3)
        single = round(float(np.median(lat)), 3)
        single_drained = round(float(np.median(lat_drained)), 3)
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_sample > 0:
        from oracle import oracle as orc
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import kat_util as ku
        os_ = orc.OracleStore()
        for comp in (0, 1, 2):   # adopt the device-built permutations: no second 100 M-row sort on the host
            os_.adopt_sorted(comp, store.read_index(comp))
        os_.set_typed_values(ds.typed_values, ds.decimals)
        sample = [int(x) for x in products[:args.cpu_sample]]
        cpu_rows, t_cpu, expected = 0, 0.0, []
        for i, x in enumerate(sample):
            desc = bsbm.q5_plan(ds, x)                     # the reference's per-query plan on the CPU
            t1 = time.perf_counter()
            cols, n, _ = os_.execute(desc)
            t_cpu += time.perf_counter() - t1
            cpu_rows += n
            pl = store.plan(desc).execute()                # full-size parity, per-instance path
            np.testing.assert_array_equal(ku.multiset(pl.fetch(), n), ku.multiset(cols, n))
            pl.close()
            expected.append(np.stack([np.full(n, i + 1, np.uint32), cols[0], cols[1]], axis=1))
        # full-size parity, batched path: the same instances as one batch
        pb = store.plan(bsbm.q5_batch_plan(ds))
        tt, pp, nn = dev_table([np.arange(1, len(sample) + 1, dtype=np.uint32), np.array(sample, dtype=np.uint32)])
        pb.bind_table(0, pp, nn)
        got = pb.execute().fetch()
        np.testing.assert_array_equal(ku.multiset(got), ku.multiset(list(np.concatenate(expected).T)))
        pb.close()
        # full-size parity of the TIMED path itself: the steady-state plan of the timed region (compiled once, tables cached,
        # speculative sizes, fused: ordered slice join + band join) executes the whole batch the sample was drawn from once more;
        # the bindings it leaves in HBM for the sampled instance tags must be the oracle's, and it must have run as the timed
        # steps did (one host sync, no exact re-run, the band join's kernels)
        if not args.per_instance:
            step(batches[0], 1)
            m = plan.metrics()
            ran = {k[0] for k in plan.kernel_stats()}
            got = plan.fetch()
            sel = got[0] <= len(sample)
            np.testing.assert_array_equal(ku.multiset([c[sel] for c in got]), ku.multiset(list(np.concatenate(expected).T)))
            fused = any("band_mask_kernel" in k for k in ran) and any("band_emit_kernel" in k for k in ran)
            if not args.no_table_cache:
                assert fused and m.host_syncs == 1, (sorted(ran), m.host_syncs)
            timed_check = (f"the {len(batches[0])}-instance step of the timed plan (speculative, {m.host_syncs} host sync, {m.kernels_launched} launches, "
                           f"{'band join' if fused else 'un-fused joins'}): its bindings for the first {len(sample)} instance tags ({int(sel.sum())} rows of "
                           f"{len(got[0])}) are multiset-equal to the oracle's per-query results")
        cpu = {"value": round(cpu_rows / t_cpu, 2) if t_cpu > 0 else None, "unit": "bindings/s", "cores": 1, "kind": "port",
               "sample": f"{len(sample)} Q5 instances of this workload ({cpu_rows} bindings, {t_cpu:.1f} s on one host core), each run "
                         "as the reference's per-query plan by the C restatement of the reference's operators "
                         "(oracle/rdf_oracle.c), single thread like the reference's default target_partitions=1; the GPU's "
                         "per-instance and batched results on these instances were compared multiset-equal",
               "queries_per_s": round(len(sample) / t_cpu, 3) if t_cpu > 0 else None}
        # the same port on all of this GPU's host cores (1 thread AND all cores): independent query
        # instances on a thread pool (the C code runs outside the GIL)
        n_threads = max(1, min(16, os.cpu_count() or 1))
        if n_threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            wide = [int(x) for x in products[args.cpu_sample:args.cpu_sample + 2 * n_threads]]
            descs = [bsbm.q5_plan(ds, x) for x in wide]
            t1 = time.perf_counter()
            with ThreadPoolExecutor(n_threads) as pool:
                rows_wide = sum(pool.map(lambda d: os_.execute(d)[1], descs))
            t_wide = time.perf_counter() - t1
            cpu["all_cores"] = {"value": round(rows_wide / t_wide, 2), "unit": "bindings/s", "cores": n_threads,
                                "queries_per_s": round(len(wide) / t_wide, 3),
                                "sample": f"{len(wide)} further instances, {n_threads} threads, {t_wide:.1f} s"}

        # a tuned columnar CPU engine on the same work: the BATCHED plan on pyarrow's multi-threaded
        # Acero hash joins, integer values decoded once per slice — reported next to the port, checked against the GPU
        try:
            from oracle import acero_baseline as ab
        except ImportError:
            ab = None
        if ab is not None:
            prep = ab.prepare(ds)
            ab_batch = 2048
            ab.run(prep, products[:64])                                        # warm the thread pool
            t_ab, rows_ab, runs = 0.0, 0, []
            for r in range(3):
                inst = np.ascontiguousarray(products[64 + r * ab_batch:64 + (r + 1) * ab_batch])
                t1 = time.perf_counter()
                res = ab.run(prep, inst)
                t_ab += time.perf_counter() - t1
                rows_ab += len(res[0])
                runs.append((inst, res))
            inst, res = runs[-1]                                               # parity of the last batch against the GPU
            pb = store.plan(bsbm.q5_batch_plan(ds))
            tt, pp, nn = dev_table([np.arange(1, len(inst) + 1, dtype=np.uint32), inst])
            pb.bind_table(0, pp, nn)
            np.testing.assert_array_equal(ku.multiset(pb.execute().fetch()), ku.multiset(res))
            pb.close()
            import pyarrow
            cpu["tuned_columnar"] = {"value": round(rows_ab / t_ab, 2), "unit": "bindings/s",
                                     "cores": pyarrow.cpu_count(),              # threads of Acero's pool (the box may grant fewer CPUs)
                                     "cpus_granted": len(os.sched_getaffinity(0)),
                                     "queries_per_s": round(3 * ab_batch / t_ab, 1),
                                     "engine": f"pyarrow {pyarrow.__version__} Acero hash joins (multi-threaded) + numpy typed gathers, "
                                               "the batched operator tree; not the reference",
                                     "sample": f"3 batches of {ab_batch} instances, {t_ab:.1f} s; the last batch compared multiset-equal with the GPU"}

    if rank == 0:
        n_q = args.steps * Q_step
        out = {
            "metric": "solution bindings/sec + achieved HBM GB/s, BSBM Q5 at 1/2/4/8 GPUs",
            "value": round(total_rows / elapsed, 2),
            "unit": "bindings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32 ids / i64 typed values",
            "data": "synthetic",
            "config": {"workload": f"BSBM-shaped store, {args.products} products ({ds.n_triples} triples), Explore Q5 "
                                   f"(7 triple patterns -> scans, hash joins, FILTERs), {Q_step} instances per step"
                                   + (f" ({Q} x {world} ranks: the batch grows with N, the graph does not)" if world > 1 and args.scaling == "weak" else "") + ", "
                                   + ("one reference plan per instance" if args.per_instance else "batched into one operator tree (shared scans)")
                                   + ("; every join table built inside the timed step (--no-table-cache)" if args.no_table_cache else
                                      "; steady state: the join tables of the predicate slices are cached per store version (cold start and the "
                                      "no-cache step are under config.cold_start)"),
                       "mode": "per-instance" if args.per_instance else "batched",
                       "triples_per_gpu": n_local, "sharding": ("rdfgpu_shard_of(subject) over N ranks (default graph) + the candidate join's layout in a named graph: productFeature sharded by OBJECT, "
                                    "the three 1:1 star predicates replicated; rdfgpu_exchange_repartition (RCCL over xGMI, behind the C ABI) of the constant-pattern "
                                    "bindings C by prodFeature") if world > 1 else "none",
                       "sharded_result_check": shard_check, "timed_path_check": timed_check, "instances_per_step": Q_step, "other_scaling": other,
                       "exchange_overlapped_with_next_step": bool(world > 1 and not args.no_overlap),
                       "exchange_transport": transport if world > 1 else None,
                       "rank0_phase_ms_per_step": ({"constant_patterns": round(phase_ms[0] / args.steps, 3), "exchange": round(phase_ms[1] / args.steps, 3),
                                                    "join_pipeline": round(phase_ms[2] / args.steps, 3),
                                                    "waiting_for_constants_and_exchange": round(phase_ms[3] / args.steps, 3),
                                                    "note": "constant_patterns + exchange run on a helper thread beside the previous batch's join_pipeline; "
                                                            "the step costs join_pipeline + waiting"} if world > 1 else None),
                       "queries_per_s": round(n_q / elapsed, 2), "bindings": total_rows,
                       "host_threads": args.threads if args.per_instance else 1,
                       "single_instance_latency_ms": single,
                       "single_instance_plan_execute_drain_drop_ms": single_drained,
                       "median_query_latency_ms": round(float(np.median(lat_ms)), 3) if lat_ms else None,
                       "load_seconds": round(load_s, 1), "cold_start": cold},
            "roofline": roofline,
            "pipeline_roofline": pipeline,
            "cpu_baseline": cpu,
            "kernels": kernel_table,
        }
        # like for like only: one query at a time on the GPU against the per-query CPU port; the batched tree against the
        # batched columnar CPU engine
        if cpu and cpu.get("queries_per_s") and single:
            out["config"]["speedup_per_query_vs_cpu_port_1_core"] = round((1e3 / single) / cpu["queries_per_s"], 1)
            if cpu.get("all_cores"):
                out["config"]["speedup_per_query_vs_cpu_port_all_cores"] = round((1e3 / single) / cpu["all_cores"]["queries_per_s"], 1)
        if cpu and cpu.get("tuned_columnar"):
            out["config"]["speedup_batched_vs_batched_columnar_cpu"] = round((n_q / elapsed) / cpu["tuned_columnar"]["queries_per_s"], 1)
        if world == 1 and not args.no_scan:
            # the BGP scan + FILTER kernel on a partition larger than the Infinity Cache (BASELINE config 2)
            mem("the Q5 part")
            out["scan_roofline"] = scan_roofline(rf, local_rank, args.scan_log2_rows)
            # the same scan over a dictionary whose typed-value table (16 B per id) does not fit the 32 MiB of L2
            out["scan_roofline_large_dictionary"] = scan_roofline(rf, local_rank, args.scan_log2_rows, distinct=1 << 22)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
