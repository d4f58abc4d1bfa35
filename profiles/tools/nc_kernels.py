"""Kernel breakdown of one steady-state Q5 batch step with NO cached join table (every HashJoinExec builds inside the step)."""
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
plan = st.plan(bsbm.q5_batch_plan(ds)).set_option("NO_TABLE_CACHE", 1)
plan.bind_table(0, [t.data_ptr(), t.data_ptr() + 4 * B], B)
for it in range(5):
    plan.enable_kernel_timing(it == 4)
    torch.cuda.synchronize(); t0 = time.perf_counter(); plan.execute(); wall = (time.perf_counter() - t0) * 1e3
    print("step", it, round(wall, 2), "ms", flush=True)
for k in sorted(plan.kernel_stats(), key=lambda k: -k[2]):
    print("  %-60s x%-3d %8.3f ms  rows_in %d" % (k[0][:60], k[1], k[2], k[4]))
