import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
ds = bsbm.generate(int(os.environ.get("P", "285000")))
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values, ds.decimals)
for i in (5, 77, 1234, 99999 % ds.n_products):
    plan = st.plan(bsbm.q5_plan(ds, ds.product(i))).enable_kernel_timing(True)
    t=time.perf_counter(); plan.execute(); dt=time.perf_counter()-t
    m = plan.metrics()
    print("rows", plan.result_info()[0], "wall_ms", round(dt*1e3,3), "dev_ms", round(m.elapsed_compute_ms,3), "kernels", m.kernels_launched, "syncs", m.host_syncs)
    for k in plan.kernel_stats(): print("   ", k)
