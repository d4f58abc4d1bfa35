"""First execution of a FRESH plan for the 262 144-instance Q5 batch on a warm store (the reference compiles a plan per query: bsbm_explore.rs:23-95),
repeated: wall time per phase (compile, bind, execute, fetch-free drain, close) and the execution's own metrics."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
ds = bsbm.generate(285000)
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values, ds.decimals)
B = 262144
rng = np.random.default_rng(5)
prods = np.array([ds.product(i) for i in rng.choice(ds.n_products, B, replace=False)], dtype=np.uint32)
t = torch.from_numpy(np.stack([np.arange(1, B + 1, dtype=np.uint32), prods]).view(np.int32)).cuda()
desc = bsbm.q5_batch_plan(ds)
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); plan = st.plan(desc); t1 = time.perf_counter()
    plan.bind_table(0, [t.data_ptr(), t.data_ptr() + 4 * B], B); t2 = time.perf_counter()
    plan.execute(); t3 = time.perf_counter()
    m = plan.metrics()
    plan.close(); t4 = time.perf_counter()
    print("rep", rep, "compile %.3f bind %.3f execute %.3f close %.3f ms | dev_ms %.3f kernels %d syncs %d mallocs %d (%.3f ms) reruns %d" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, m.elapsed_compute_ms, m.kernels_launched, m.host_syncs, m.device_mallocs, m.device_malloc_ms, m.exact_reruns), flush=True)
time.sleep(0.5)
