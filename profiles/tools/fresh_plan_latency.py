"""The reference's protocol per benchmark iteration: compile a plan for ONE query instance, execute it, read the result, drop the plan
(bench/benches/bsbm_explore.rs:23-95) — here: rdfgpu_plan_compile + execute + fetch + destroy per BSBM Q5 instance, store tables warm."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
ds = bsbm.generate(int(os.environ.get("P", "285000")))
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values, ds.decimals)
rng = np.random.default_rng(11)
prods = [ds.product(int(i)) for i in rng.choice(ds.n_products, 60, replace=False)]
ts, parts = [], []
for k, p in enumerate(prods):
    t0 = time.perf_counter()
    desc = bsbm.q5_plan(ds, p)
    t1 = time.perf_counter()
    plan = st.plan(desc)
    t2 = time.perf_counter()
    plan.execute()
    t3 = time.perf_counter()
    rows = plan.fetch()
    t4 = time.perf_counter()
    m = plan.metrics()
    plan.close()
    t5 = time.perf_counter()
    ts.append((t5 - t1) * 1e3); parts.append(((t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, m.host_syncs, m.kernels_launched, m.device_mallocs, len(rows[0])))
    if k < 3 or k >= 57: print("query %2d: total %.3f ms = compile %.3f + execute %.3f + fetch %.3f + destroy %.3f; syncs %d launches %d mallocs %d rows %d" % ((k, ts[-1]) + parts[-1]), flush=True)
steady = sorted(ts[10:])
print("fresh plan per query, queries 10..59: median %.3f ms, min %.3f, max %.3f" % (steady[len(steady) // 2], steady[0], steady[-1]))
a = np.array([p[:4] for p in parts[10:]])
print("medians: compile %.3f execute %.3f fetch %.3f destroy %.3f" % tuple(np.median(a, axis=0)))
