import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
def mem():
    f, t = torch.cuda.mem_get_info(0)
    return f"{(t - f) / 2**30:.1f} GiB"
ds = bsbm.generate(285000)
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values, ds.decimals)
B = 262144
rng = np.random.default_rng(5)
plan = st.plan(bsbm.q5_batch_plan(ds)).set_option("NO_TABLE_CACHE", 1)
for it in range(6):
    prods = np.array([ds.product(i) for i in rng.choice(ds.n_products, B, replace=True)], dtype=np.uint32)
    flat = np.stack([np.arange(1, B + 1, dtype=np.uint32), prods])
    t = torch.from_numpy(flat.view(np.int32)).cuda()
    plan.bind_table(0, [t.data_ptr(), t.data_ptr() + 4 * B], B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.execute()
    dt = (time.perf_counter() - t0) * 1e3
    m = plan.metrics()
    print(f"[nc] step {it}: {dt:.1f} ms wall, {m.elapsed_compute_ms:.1f} ms device, scratch {m.device_bytes / 2**30:.1f} GiB, syncs {m.host_syncs}, used {mem()}", flush=True)
