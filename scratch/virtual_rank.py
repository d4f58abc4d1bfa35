"""One rank of a W-rank graph-sharded run, on one GPU: per-step time of phase A / pack / unpack / phase B with the
all-gather replaced by buffers computed beforehand from the other shards (so: everything but xGMI).
  python scratch/virtual_rank.py [W] [Q]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm, sharding
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
ds = bsbm.generate(285000)
rng = np.random.default_rng(5)
allp = np.array([ds.product(i) for i in range(ds.n_products)], dtype=np.uint32)
batches = [np.ascontiguousarray(allp[rng.choice(ds.n_products, Q, replace=False)]) for _ in range(4)]
ex = sharding.BatchExchange(Q, W)
class DevCol:
    def __init__(self, ptr, n): self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (int(ptr), False), "version": 2}
def params(b):
    flat = np.stack([np.arange(1, Q + 1, dtype=np.uint32), b]); t = torch.from_numpy(flat.view(np.int32)).cuda()
    return t, [t.data_ptr(), t.data_ptr() + 4 * Q]
def phase_a(pa, ptrs, buf):
    buf.zero_()
    pa.bind_table(0, ptrs, Q); pa.execute()
    cols, rows = pa.result_device()
    ex.pack(buf, [torch.as_tensor(DevCol(c, rows), device="cuda") if rows else None for c in cols], rows)
    torch.cuda.current_stream().synchronize()
# the other ranks' buffers for every batch
others = [[None] * W for _ in batches]
mine_store = None
for r in range(W):
    g, s, p, o = sharding.shard_dataset(ds, r, W)
    st = rf.GpuQuadStore(); st.extend(g, s, p, o); st.set_typed_values(ds.typed_values, ds.decimals)
    plans = st.plan(bsbm.q5_batch_const_plan(ds))
    if r == 0:
        mine_store, mine_plans = st, plans
        continue
    for bi, b in enumerate(batches):
        t, ptrs = params(b)
        buf = torch.zeros(ex.buf_len, dtype=torch.int32, device="cuda")
        phase_a(plans, ptrs, buf)
        others[bi][r] = buf
    del plans, st
plan_b = mine_store.plan(bsbm.q5_batch_plan(ds, tables=True))
send = torch.zeros(ex.buf_len, dtype=torch.int32, device="cuda")
P = [params(b) for b in batches]
for it in range(3):
    for bi, b in enumerate(batches):
        t, ptrs = P[bi]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        phase_a(mine_plans, ptrs, send)
        t1 = time.perf_counter()
        out = torch.cat([send] + others[bi][1:])              # stands in for all_gather_into_tensor
        torch.cuda.current_stream().synchronize()
        t2 = time.perf_counter()
        keep = ex.unpack(out)
        n = keep.shape[1]
        plan_b.bind_table(0, [keep.data_ptr() + 4 * n * k for k in range(5)], n)
        torch.cuda.current_stream().synchronize()
        t3 = time.perf_counter()
        plan_b.enable_kernel_timing(it == 2 and bi == 0)
        plan_b.execute(); rows, _ = plan_b.result_info()
        t4 = time.perf_counter()
        if it:
            m = plan_b.metrics()
            print("W=%d Q=%d batch %d: phase A+pack %.3f  gather-standin %.3f  unpack+bind %.3f  phase B %.3f (kernels %d, syncs %d)  total %.3f ms  rows %d  buf %.1f MB" % (
                W, Q, bi, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, m.kernels_launched, m.host_syncs, (t4 - t0) * 1e3, rows, ex.buf_len * 4 / 1e6))
        if it == 2 and bi == 0:
            for name, launches, ms, nbytes, krows in plan_b.kernel_stats():
                print("   %-60s launches %d  %.3f ms  rows %d" % (name, launches, ms, krows))
