import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
ds = bsbm.generate(285000)
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values, ds.decimals)
B = 262144
rng = np.random.default_rng(5)
batches = []
for it in range(4):
    prods = np.array([ds.product(i) for i in rng.choice(ds.n_products, B, replace=True)], dtype=np.uint32)
    flat = np.stack([np.arange(1, B + 1, dtype=np.uint32), prods])
    batches.append(torch.from_numpy(flat.view(np.int32)).cuda())
for opts in ([], ["NO_PARTITIONED_JOIN"]):
    plan = st.plan(bsbm.q5_batch_plan(ds)).set_option("NO_TABLE_CACHE", 1)
    for o in opts: plan.set_option(o)
    times = []
    for it in range(30):
        t = batches[it % 4]
        plan.bind_table(0, [t.data_ptr(), t.data_ptr() + 4 * B], B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.execute()
        times.append(round((time.perf_counter() - t0) * 1e3, 1))
        if times[-1] > 100 and it > 0:
            m = plan.metrics()
            print("[nc] outlier", it, times[-1], "device ms", round(m.elapsed_compute_ms, 1), "syncs", m.host_syncs, "launches", m.kernels_launched, "scratch GiB", round(m.device_bytes / 2**30, 1), flush=True)
    print("[nc]", opts, times, flush=True)
    plan.close()
