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
flat = np.stack([np.arange(1, B + 1, dtype=np.uint32), prods])
t = torch.from_numpy(flat.view(np.int32)).cuda()
for rep in range(2):
    plan = st.plan(bsbm.q5_batch_plan(ds)).enable_kernel_timing(True)
    plan.bind_table(0, [t.data_ptr(), t.data_ptr() + 4 * B], B)
    torch.cuda.synchronize(); t0 = time.perf_counter(); plan.execute(); dt = time.perf_counter() - t0
    m = plan.metrics()
    print("first execution of a fresh plan; store tables", "cold" if rep == 0 else "warm", "wall_ms", round(dt * 1e3, 3), "dev_ms", round(m.elapsed_compute_ms, 3), "kernels", m.kernels_launched, "syncs", m.host_syncs, "tables", m.tables_built,
          "hipMallocs", m.device_mallocs, "ms in them", round(m.device_malloc_ms, 3), "bytes", m.device_bytes)
    tot = 0
    for k in sorted(plan.kernel_stats(), key=lambda k: -k[2])[:14]:
        print("   %-60s x%-3d %8.3f ms" % (k[0][:60], k[1], k[2])); tot += k[2]
    print("   sum of listed kernels %.3f ms" % tot)
    plan.close()
