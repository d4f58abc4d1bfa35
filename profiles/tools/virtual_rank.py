"""One rank's work of an N-way sharded batched Q5 step, on one GPU: phase A (constants of the rank's instances on its shard),
the re-sharded C table (computed here from the full graph and cut to the rank's features: what the repartition delivers), phase B on the shard.
    python profiles/tools/virtual_rank.py <N> <instances>
Everything but xGMI; prints per-phase device + wall times of the steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm, sharding

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
ds = bsbm.generate(285000)
full = rf.GpuQuadStore(); full.extend(ds.g, ds.s, ds.p, ds.o); full.set_typed_values(ds.typed_values, ds.decimals)
g, s, p, o = sharding.shard_dataset_hybrid(ds, 0, N)
shard = rf.GpuQuadStore(); shard.extend(g, s, p, o); shard.set_typed_values(ds.typed_values, ds.decimals)
rng = np.random.default_rng(3)
plan_a = shard.plan(bsbm.q5_batch_const_plan(ds))
plan_c = full.plan(bsbm.q5_batch_const_plan(ds))     # sorted by feature (ordered slice join); cut below into the N sorted runs the repartition delivers
plan_b = shard.plan(bsbm.q5_batch_plan(ds, tables=True, graph=[sharding.candidate_graph(ds)]))

def dev(cols):
    t = [torch.from_numpy(np.ascontiguousarray(c).view(np.int32)).cuda() for c in cols]
    return t, [x.data_ptr() for x in t]

rows = []
for it in range(6):
    prods = np.array([ds.product(i) for i in rng.choice(ds.n_products, Q, replace=True)], dtype=np.uint32)
    params = [np.arange(1, Q + 1, dtype=np.uint32), prods]
    keep, ptrs = dev(params)
    own = sharding.shard_of(prods, N) == 0                               # the router hands a rank the instances whose product it owns
    keep_a, ptrs_a = dev([c[own] for c in params]); n_own = int(own.sum())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan_a.bind_table(0, ptrs_a, n_own); plan_a.execute(); a_rows = plan_a.result_info()[0]
    torch.cuda.synchronize(); t_a = (time.perf_counter() - t0) * 1e3
    plan_c.bind_table(0, ptrs, Q); plan_c.execute()                      # the gathered table (not timed: stands in for the all-gather)
    call = plan_c.fetch()                                                # what the repartition delivers to rank 0: the rows of its features
    mine = sharding.shard_of(call[2], N) == 0
    src = sharding.shard_of(call[1], N)[mine]                            # the rank that owns the instance's product: one sorted run per source rank
    order = np.argsort(src, kind="stable")
    keep_c, c_ptrs = dev([c[mine][order] for c in call]); n_c = int(mine.sum())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan_b.bind_table(0, c_ptrs, n_c); plan_b.execute(); b_rows = plan_b.result_info()[0]
    torch.cuda.synchronize(); t_b = (time.perf_counter() - t0) * 1e3
    rows.append((t_a, plan_a.metrics().elapsed_compute_ms, t_b, plan_b.metrics().elapsed_compute_ms, a_rows, n_c, b_rows))
    print(f"it {it}: phase A {t_a:.3f} ms wall ({rows[-1][1]:.3f} device, {a_rows} rows of C), phase B {t_b:.3f} ms wall ({rows[-1][3]:.3f} device), C {n_c} rows, {b_rows} bindings; A: {plan_a.metrics().kernels_launched} launches {plan_a.metrics().host_syncs} syncs, B: {plan_b.metrics().kernels_launched} launches {plan_b.metrics().host_syncs} syncs {plan_b.metrics().device_mallocs} mallocs ({plan_b.metrics().device_malloc_ms:.2f} ms) {plan_b.metrics().exact_reruns} reruns", flush=True)
best = min(rows[2:], key=lambda r: r[0] + r[2])
print({"ranks": N, "instances": Q, "phase_a_ms": round(best[0], 3), "phase_b_ms": round(best[2], 3), "c_rows": int(best[5]), "c_rows_of_this_rank": int(best[4]),
       "exchange_bytes_per_rank": int(best[4]) * 20, "bindings_of_this_rank": int(best[6])})
plan_b.enable_kernel_timing(True); plan_b.execute()
for k in sorted(plan_b.kernel_stats(), key=lambda k: -k[2]): print("   B", k[0][:60], k[1], round(k[2] * 1e3, 1), "us")
plan_a.enable_kernel_timing(True); plan_a.execute()
for k in sorted(plan_a.kernel_stats(), key=lambda k: -k[2]): print("   A", k[0][:60], k[1], round(k[2] * 1e3, 1), "us")
