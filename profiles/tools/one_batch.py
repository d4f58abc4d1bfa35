import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
ds = bsbm.generate(int(os.environ.get("P", "285000")))
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values, ds.decimals)
B = int(os.environ.get("B", "256"))
rng = np.random.default_rng(5)
plan = st.plan(bsbm.q5_batch_plan(ds, topk=bool(int(os.environ.get("TOPK", "0"))))).enable_kernel_timing(bool(int(os.environ.get("TIMING", "1"))))   # TIMING=0: no event brackets (the launch-gap trace)
for it in range(int(os.environ.get("STEPS", "4"))):
    prods = np.array([ds.product(i) for i in rng.choice(ds.n_products, B, replace=False)], dtype=np.uint32)
    flat = np.stack([np.arange(1, B + 1, dtype=np.uint32), prods])
    t = torch.from_numpy(flat.view(np.int32)).cuda()
    plan.bind_table(0, [t.data_ptr(), t.data_ptr() + 4 * B], B)
    t0 = time.perf_counter(); plan.execute(); dt = time.perf_counter() - t0
    m = plan.metrics()
    print("rows", plan.result_info()[0], "wall_ms", round(dt * 1e3, 3), "dev_ms", round(m.elapsed_compute_ms, 3), "kernels", m.kernels_launched, "syncs", m.host_syncs)
for k in plan.kernel_stats(): print("   ", k)
