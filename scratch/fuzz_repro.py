"""Regenerates plan `it` of fuzz seed `seed` and reruns it under engine toggles.  python scratch/fuzz_repro.py seed it [small|large]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import abi
from rdf_fusion_amd.plan import PlanBuilder
import test_gpu_fuzz as tf
import kat_util as ku
seed, target = int(sys.argv[1]), int(sys.argv[2]); size = sys.argv[3] if len(sys.argv) > 3 else "small"
tf.configure(*((4000, 64, 40) if size == "small" else (120_000, 6000, 3000)))
rng = np.random.default_rng(1000 + seed)
gs, os_ = tf.make_store(rng)
n0, n1 = (200, 50) if size == "small" else (5000, 2500)
T0 = [rng.integers(0, tf.N_IDS, n0).astype(np.uint32) for _ in range(3)]
T1 = [rng.integers(0, tf.N_SUBJ, n1).astype(np.uint32) for _ in range(2)]
k0, p0 = tf.table_on_device(torch, T0); k1, p1 = tf.table_on_device(torch, T1)
for it in range(target + 1):
    pb = PlanBuilder()
    root = tf.Gen(rng, pb).node((3 if size == "small" else 2) + (it % 3 == 0))
    desc = pb.build(root)
names = {v: k for k, v in vars(abi).items() if k.startswith("NODE_")}
for i, n in enumerate(pb.nodes):
    proj = None if n.n_proj == abi.NO_PROJECTION else [pb.pool[n.proj_off + q] for q in range(n.n_proj)]
    ex = [(e.op, e.u, e.lo, e.tag) for e in pb.exprs[n.expr_off:n.expr_off + n.expr_len]]
    print(i, names.get(n.kind), "L", n.left, "R", n.right, "jt", n.join_type, "keys", [(n.left_keys[k], n.right_keys[k]) for k in range(n.n_keys)], "proj", proj, "w", pb.width[i],
          "slot", n.table_slot if n.kind == abi.NODE_TABLE else "", "expr", ex)
    if n.kind == abi.NODE_DATA_SOURCE:
        print("     scan", [(s.kind, s.var, s.pred, s.from_ if hasattr(s, "from_") else None) for s in n.scan])
print("root", root)
exp, n_exp, _ = os_.execute(desc, [T0, T1]); want = ku.multiset(exp, n_exp)
used = sorted({int(n.table_slot) for n in pb.nodes if n.kind == abi.NODE_TABLE})
for toggle in (None, "RDFGPU_NO_SPECULATION", "RDFGPU_NO_CHAIN_FUSION", "RDFGPU_NO_TABLE_CACHE", "RDFGPU_NO_INDEX_JOIN", "RDFGPU_NO_LDS_JOIN", "RDFGPU_NO_FILTER_FUSION", "RDFGPU_NO_DIRECT_TABLE", "RDFGPU_FORCE_GENERIC_VM", "RDFGPU_NO_JOIN_REORDER", "RDFGPU_NO_FIRST_RUN_SPECULATION"):
    if toggle: gs.set_option(toggle, 1)
    plan = gs.plan(desc)
    for slot in used: plan.bind_table(slot, *[(p0, n0), (p1, n1)][slot])
    res = []
    for rep in range(3):
        got = plan.execute().fetch(); n = plan.result_info()[0]
        ok = n == n_exp and np.array_equal(ku.multiset(got, n), want)
        res.append("ok" if ok else "BAD(%d vs %d)" % (n, n_exp))
    print("%-32s %s" % (toggle, res))
    if toggle: gs.set_option(toggle, 0)
