import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import lubm
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ds = lubm.generate(U)
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values); st.set_strings(ds.str_offsets, ds.str_heap)
desc = lubm.q9_optional_regex_plan(ds, ".", "")
for rep in range(2):
    plan = st.plan(desc).enable_kernel_timing(True)
    prev = {}
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); plan.execute(); rows, _ = plan.result_info(); dt = (time.perf_counter() - t0) * 1e3
        m = plan.metrics()
        print("plan %d exec %d: %.2f ms rows %d kernels %d syncs %d" % (rep, it, dt, rows, m.kernels_launched, m.host_syncs))
        cur = {k[0]: (k[1], k[2], k[4]) for k in plan.kernel_stats()}
        for name, (l, ms, r) in sorted(cur.items(), key=lambda kv: -kv[1][1]):
            pl, pms, pr = prev.get(name, (0, 0.0, 0))
            if l - pl: print("      %-58s x%d  %.3f ms  rows %d" % (name, l - pl, ms - pms, r - pr))
        prev = cur
