"""HashJoinExec over bound tables through the general fused join kernel (lds_join_kernel): build sides from 1 K to 1 M rows (LDS table /
HBM hash table in L2), 32 M probe rows, hit rates 0.1 and 1.0 — kernel time against the SURVEY 8d bytes of the operator."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import abi
from rdf_fusion_amd.plan import PlanBuilder
st = rf.GpuQuadStore()
NP = int(os.environ.get("NP", str(32 << 20)))
rng = np.random.default_rng(1)
def dev(cols):
    t = [torch.from_numpy(c.view(np.int32)).cuda() for c in cols]
    return t, [x.data_ptr() for x in t]
for nb in [int(x) for x in os.environ.get("NB", "1000,20000,500000,1000000").split(",")]:
    for hit in (0.1, 1.0):
        bk = rng.permutation(np.arange(1, nb + 1, dtype=np.uint32))
        B = [bk, (bk * 7 + 1).astype(np.uint32)]
        pk = rng.integers(1, int(nb / hit) + 1, NP).astype(np.uint32)
        P = [pk, np.arange(1, NP + 1, dtype=np.uint32)]
        kb, pb_ = dev(B); kp, pp_ = dev(P)
        pb = PlanBuilder()
        desc = pb.build(pb.hash_join(pb.table(0, 2), pb.table(1, 2), on=[(0, 0)], join_type=abi.JOIN_INNER, projection=[1, 3]))
        plan = st.plan(desc)
        plan.bind_table(0, pb_, nb); plan.bind_table(1, pp_, NP)
        best = None
        for rep in range(4):
            plan.enable_kernel_timing(True)
            plan.execute()
            ks = plan.kernel_stats()
            ms = sum(k[2] for k in ks)
            if best is None or ms < best[0]: best = (ms, ks)
        rows = plan.result_info()[0]
        expect = int((pk <= nb).sum())
        assert rows == expect, (rows, expect)
        byt = 8 * nb + 8 * NP + 8 * rows          # keys + payload in, two columns out
        print("build %8d rows, hit %.1f: %8.3f ms  %6.1f GB/s of operator bytes  out %d  kernels %s" % (nb, hit, best[0], byt / best[0] / 1e6, rows, {k[0][-40:]: round(k[2], 3) for k in best[1]}), flush=True)
        plan.close(); del kb, kp
