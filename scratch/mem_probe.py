import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
def mem(stage):
    f, t = torch.cuda.mem_get_info(0)
    print(f"[mem] {stage}: {(t - f) / 2**30:.1f} GiB used", flush=True)
ds = bsbm.generate(285000)
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values, ds.decimals)
mem("store loaded")
B = 262144
rng = np.random.default_rng(5)
plan = st.plan(bsbm.q5_batch_plan(ds)).enable_kernel_timing(True)
for it in range(5):
    prods = np.array([ds.product(i) for i in rng.choice(ds.n_products, B, replace=True)], dtype=np.uint32)
    flat = np.stack([np.arange(1, B + 1, dtype=np.uint32), prods])
    t = torch.from_numpy(flat.view(np.int32)).cuda()
    plan.bind_table(0, [t.data_ptr(), t.data_ptr() + 4 * B], B)
    plan.execute()
    m = plan.metrics()
    mem(f"execution {it}: scratch {m.device_bytes / 2**30:.2f} GiB, rows {plan.result_info()[0]}")
