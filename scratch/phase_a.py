import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
ds = bsbm.generate(285000)
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values)
Q = 65536
rng = np.random.default_rng(5)
plans = [st.plan(d) for d in bsbm.q5_batch_const_plans(ds)]
class DevCol:
    def __init__(self, ptr, n): self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (int(ptr), False), "version": 2}
caps = [Q * 28, Q * 2, Q * 2]; offs = [0, 3 * caps[0], 3 * (caps[0] + caps[1])]
buf = torch.zeros(3 * sum(caps), dtype=torch.int32, device="cuda")
for it in range(5):
    prods = np.array([ds.product(i) for i in rng.choice(ds.n_products, Q, replace=False)], dtype=np.uint32)
    flat = np.stack([np.arange(1, Q + 1, dtype=np.uint32), prods]); t = torch.from_numpy(flat.view(np.int32)).cuda()
    ptrs = [t.data_ptr(), t.data_ptr() + 4 * Q]
    torch.cuda.synchronize()
    t0 = time.perf_counter(); buf.zero_(); t1 = time.perf_counter()
    ex = []
    for pa in plans:
        a = time.perf_counter(); pa.bind_table(0, ptrs, Q); pa.execute(); ex.append((time.perf_counter() - a) * 1e3)
    t2 = time.perf_counter()
    for pa, cap, off in zip(plans, caps, offs):
        cols, rows = pa.result_device()
        for k in range(3):
            buf[off + k * cap: off + k * cap + rows] = torch.as_tensor(DevCol(cols[k], rows), device="cuda")
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("zero %.3f  exec %s  copies(issue) %.3f  sync %.3f ms" % ((t1 - t0) * 1e3, [round(x, 3) for x in ex], (t3 - t2) * 1e3, (t4 - t3) * 1e3),
          [ (p.metrics().kernels_launched, p.metrics().host_syncs, round(p.metrics().elapsed_compute_ms,3)) for p in plans])
