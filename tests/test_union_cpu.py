"""UnionExec (SPARQL UNION as planned in BSBM Explore - Q4 / Q11 (Execution Plan).snap) in the oracle: bag union of
two inputs, checked against numpy concatenation, and BSBM Q4's two-branch pipeline against plain numpy on the raw
triples.  CPU only."""
import numpy as np

from rdf_fusion_amd import abi, bsbm
from rdf_fusion_amd.plan import PlanBuilder, col, lit_id, ID_NEQ
from oracle import oracle as orc
import kat_util as ku


def test_union_is_concatenation():
    rng = np.random.default_rng(2)
    st = orc.OracleStore()
    for nl, nr in ((0, 0), (0, 5), (7, 0), (300, 200), (1, 1)):
        L = [rng.integers(0, 50, nl).astype(np.uint32) for _ in range(3)]
        R = [rng.integers(0, 50, nr).astype(np.uint32) for _ in range(3)]
        pb = PlanBuilder()
        desc = pb.build(pb.union(pb.table(0, 3), pb.filter(pb.table(1, 3), ID_NEQ(col(0), lit_id(7))), projection=[2, 0]))
        cols, n, _ = st.execute(desc, [L, R])
        keep = (R[0] != 7) & (R[0] != 0)
        exp = [np.concatenate([L[2], R[2][keep]]), np.concatenate([L[0], R[0][keep]])]
        assert n == len(exp[0])
        np.testing.assert_array_equal(cols[0][:n], exp[0])        # left rows first, in order, then the right rows
        np.testing.assert_array_equal(cols[1][:n], exp[1])


def test_bsbm_q4_oracle_equals_numpy():
    ds = bsbm.generate(3000)
    st = orc.OracleStore()
    st.extend(ds.g, ds.s, ds.p, ds.o)
    st.set_typed_values(ds.typed_values, ds.decimals)
    pr, tv = ds.pred, ds.typed_values
    has = lambda pname, o: set(ds.s[(ds.p == pr[pname]) & (ds.o == o)].tolist())
    single = lambda pname: dict(zip(ds.s[ds.p == pr[pname]].tolist(), ds.o[ds.p == pr[pname]].tolist()))
    label, textual, n1, n2 = single("rdfs:label"), single("bsbm:productPropertyTextual1"), single("bsbm:productPropertyNumeric1"), single("bsbm:productPropertyNumeric2")
    rng = np.random.default_rng(9)
    total = 0
    feats = ds.o[ds.p == pr["bsbm:productFeature"]]
    common = np.bincount(feats - ds.feature_base).argsort()[::-1][:6] + ds.feature_base       # frequent features: non-empty answers
    for it in range(12):
        t = ds.type_base + ds.n_types - 1 if it % 2 else int(ds.o[ds.p == pr["rdf:type"]][rng.integers(0, 4 * ds.n_products)])
        f1, f2, f3 = (int(x) for x in rng.choice(common, 3, replace=False))
        thr1, thr2 = int(rng.integers(200, 1200)), int(rng.integers(200, 1200))
        cols, n, _ = st.execute(bsbm.q4_plan(ds, t, f1, f2, f3, thr1, thr2))
        exp = []
        for fb, num, thr in ((f2, n1, thr1), (f3, n2, thr2)):
            for x in sorted(has("rdf:type", t) & has("bsbm:productFeature", f1) & has("bsbm:productFeature", fb)):
                if x in textual and x in num and int(tv["lo"][num[x]]) > thr:
                    exp.append((x, label[x], textual[x]))
        assert sorted(zip(*(c[:n].tolist() for c in cols))) == sorted(exp)
        total += n
    assert total > 10
