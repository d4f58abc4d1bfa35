"""KleenePlusClosureExec timing: binary-tree `parent` edges (depth log2 n: the closure is every (node, ancestor) pair).
  python scratch/closure_micro.py [n]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd.plan import PlanBuilder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
child = np.arange(2, n + 2, dtype=np.uint32)
t = [np.zeros(n, np.uint32), child, (child // 2).astype(np.uint32)]
flat = np.stack(t); dev = torch.from_numpy(flat.view(np.int32)).cuda()
ptrs = [dev.data_ptr() + 4 * n * k for k in range(3)]
pb = PlanBuilder(); desc = pb.build(pb.closure(pb.table(0, 3)))
gs = rf.GpuQuadStore(); plan = gs.plan(desc); plan.bind_table(0, ptrs, n)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); plan.execute(); rows, _ = plan.result_info(); dt = time.perf_counter() - t0
    print("GPU: %d inner paths -> %d closure paths in %.1f ms (%.1f M paths/s), host syncs %d" % (n, rows, dt * 1e3, rows / dt / 1e6, plan.metrics().host_syncs))
from oracle import oracle as orc
m = min(n, 200_000)
t0 = time.perf_counter(); cols, k, _ = orc.OracleStore().execute(desc, [[c[:m] for c in t]]); dt = time.perf_counter() - t0
print("CPU port: %d inner paths -> %d closure paths in %.1f ms (%.2f M paths/s)" % (m, k, dt * 1e3, k / dt / 1e6))
