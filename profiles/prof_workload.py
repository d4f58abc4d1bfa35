"""Workload used for the rocprofv3 passes whose summaries are committed next to this file:
  (a) the scaled scan + FILTER (2^26-row partition, BASELINE config 2)  -> filter_kernel<2>
  (b) 32 BSBM-100M Q5 instances (BASELINE config 3)                      -> lds_join_kernel & co
Run:  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 profiles/prof_workload.py
      rocprofv3 --pmc FETCH_SIZE  --output-format csv -d gpurun_out/pmc_fetch -- python3 profiles/prof_workload.py
      rocprofv3 --pmc WRITE_SIZE  --output-format csv -d gpurun_out/pmc_write -- python3 profiles/prof_workload.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
import bench

print(bench.scan_roofline(rf, 0, int(os.environ.get("SCAN_LOG2", "26")), reps=5))
ds = bsbm.generate(int(os.environ.get("P", "285000")))
st = rf.GpuQuadStore()
st.extend(ds.g, ds.s, ds.p, ds.o)
st.set_typed_values(ds.typed_values, ds.decimals)
rng = np.random.default_rng(1)
rows = 0
for i in rng.choice(ds.n_products, 32, replace=False):
    plan = st.plan(bsbm.q5_plan(ds, ds.product(i))).execute()
    rows += plan.result_info()[0]
    plan.close()
print("q5 bindings", rows)
