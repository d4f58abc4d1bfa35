import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import lubm
U = 8000
ds = lubm.generate(U)
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values); st.set_strings(ds.str_offsets, ds.str_heap)
for pattern, flags in (("^GraduateStudent1", ""), (".", ""), ("^GraduateStudent1", "")):
    plan = st.plan(lubm.q9_optional_regex_plan(ds, pattern, flags)).enable_kernel_timing(True)
    for it in range(int(os.environ.get("REPS", "4"))):
        torch.cuda.synchronize(); t0 = time.perf_counter(); plan.execute(); rows, _ = plan.result_info(); dt = (time.perf_counter() - t0) * 1e3
        m = plan.metrics()
        print(pattern, "exec", it, "rows", rows, "ms %.2f" % dt, "syncs", m.host_syncs, "reruns", m.exact_reruns, flush=True)
    for k in sorted(plan.kernel_stats(), key=lambda k: -k[2])[:12]:
        print("    %-64s x%-2d %8.3f ms  rows_in %d" % (k[0][:64], k[1], k[2], k[4] if len(k) > 4 else -1))
    plan.close()
