"""Q5 batch steps with one or two batches in flight (two plans, two host threads): does overlapping steps fill the launch bubbles?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
ds = bsbm.generate(285000)
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values, ds.decimals)
B = 262144
rng = np.random.default_rng(5)
allp = np.array([ds.product(i) for i in range(ds.n_products)], dtype=np.uint32)
batches = []
for it in range(24):
    prods = allp[rng.choice(ds.n_products, B, replace=False)]
    t = torch.from_numpy(np.stack([np.arange(1, B + 1, dtype=np.uint32), prods]).view(np.int32)).cuda()
    batches.append(t)
torch.cuda.synchronize()
def worker(plan, mine, out):
    rows = 0
    for t in mine:
        plan.bind_table(0, [t.data_ptr(), t.data_ptr() + 4 * B], B)
        plan.execute(); rows += plan.result_info()[0]
    out.append(rows)
for n_threads in (1, 2, 3):
    plans = [st.plan(bsbm.q5_batch_plan(ds)) for _ in range(n_threads)]
    for p in plans:                        # warm every plan
        worker(p, batches[:4], [])
    out = []
    ths = [threading.Thread(target=worker, args=(plans[k], batches[4:][k::n_threads], out)) for k in range(n_threads)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(n_threads, "in flight:", round(dt * 1e3 / 20, 3), "ms per step,", round(sum(out) / dt / 1e9, 2), "G bindings/s", flush=True)
    for p in plans: p.close()
