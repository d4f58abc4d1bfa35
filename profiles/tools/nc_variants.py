"""The Q5 batch step with NO cached join table under different partition geometries of the partitioned join (PARTITION_ROWS / PARTITION_SLOTS):
step time and the partitioned join's kernel time per setting."""
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
settings = [dict(), dict(PARTITION_SLOTS=2048), dict(PARTITION_ROWS=2048), dict(PARTITION_SLOTS=8192), dict(PARTITION_SLOTS=8192, PARTITION_ROWS=4096), dict(PARTITION_ROWS=512)]
if len(sys.argv) > 1:
    settings = [eval("dict(" + a + ")") for a in sys.argv[1:]]
for s in settings:
    plan = st.plan(bsbm.q5_batch_plan(ds)).set_option("NO_TABLE_CACHE", 1)
    for k, v in s.items():
        plan.set_option(k, v)
    plan.bind_table(0, [t.data_ptr(), t.data_ptr() + 4 * B], B)
    walls = []
    for it in range(6):
        plan.enable_kernel_timing(it == 5)
        torch.cuda.synchronize(); t0 = time.perf_counter(); plan.execute(); walls.append((time.perf_counter() - t0) * 1e3)
    pj = [k for k in plan.kernel_stats() if "part_join" in k[0]]
    print(s, "steps", [round(w, 2) for w in walls[1:5]], "timed step", round(walls[5], 2), "part_join ms", [round(k[2], 3) for k in pj], flush=True)
    del plan
