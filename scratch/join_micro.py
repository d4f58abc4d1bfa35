"""One big join in isolation: TABLE(inst, product, o1, o2) [n rows] JOIN sim1 scan ON product, with / without the window filter."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm
from rdf_fusion_amd.plan import PlanBuilder, quad_pattern
ds = bsbm.generate(int(os.environ.get("P", "285000")))
st = rf.GpuQuadStore(); st.extend(ds.g, ds.s, ds.p, ds.o); st.set_typed_values(ds.typed_values, ds.decimals)
n = int(os.environ.get("N", str(8 << 20)))
rng = np.random.default_rng(1)
cols = np.stack([rng.integers(1, 4097, n), ds.product_base + rng.integers(0, ds.n_products, n),
                 ds.int_base + rng.integers(0, 2000, n), ds.int_base + rng.integers(0, 2000, n)]).astype(np.uint32)
t = torch.from_numpy(cols.view(np.int32)).cuda()
ptrs = [t.data_ptr() + 4 * n * c for c in range(4)]
for mode in os.environ.get("MODES", "none,window,idneq").split(","):
    pb = PlanBuilder()
    tab = pb.table(0, 4)
    sim = pb.data_source(quad_pattern("product", ds.pred["bsbm:productPropertyNumeric1"], "sim"))
    flt = None
    if mode == "window": flt = bsbm._window(5, 2, 120)
    if mode == "idneq": flt = bsbm.ID_NEQ(bsbm.col(5), bsbm.col(2))
    plan = st.plan(pb.build(pb.hash_join(tab, sim, on=[(1, 0)], filter=flt, projection=[0, 1, 2, 3]))).enable_kernel_timing(True)
    plan.bind_table(0, ptrs, n)
    for it in range(4):
        plan.execute()
    print(mode, "rows", plan.result_info()[0])
    for k in plan.kernel_stats(): print("   ", k[0][-40:], k[1], round(k[2] / k[1] * 1e3, 1), "us/launch")
