"""Independent columnar-CPU cross-check of the oracle's join / filter operators: pyarrow's
Acero hash join and compute kernels on the same u32 tables.  pyarrow is NOT the reference — the reference's
HashJoinExec is DataFusion 52, absent here — but it implements the same SQL join semantics the reference relies on
(null keys never match = NullEqualsNothing; a left join keeps unmatched left rows with null right columns), so an
oracle that agrees with it on random multisets is not agreeing with itself only.  CPU only, no GPU."""
import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
import pyarrow.compute as pc  # noqa: E402

from rdf_fusion_amd import abi  # noqa: E402
from rdf_fusion_amd.engine import TV_DTYPE  # noqa: E402
from rdf_fusion_amd.plan import PlanBuilder, col, integer, ENC_TV, GT, LT, GEQ, LEQ, EBV, ID_NEQ  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import kat_util as ku  # noqa: E402


def rand_table(rng, n, ncols, n_ids, null_frac=0.1):
    cols = [rng.integers(1, n_ids, n).astype(np.uint32) for _ in range(ncols)]
    for c in cols:
        c[rng.random(n) < null_frac] = 0
    return cols


def to_arrow(cols, prefix):
    """0 is the null id (quad_index_data.rs:438-440): becomes an Arrow null"""
    return pa.table({f"{prefix}{k}": pa.array(c, type=pa.uint32(), mask=(c == 0)) for k, c in enumerate(cols)})


def from_arrow(tbl, names):
    return [np.asarray(tbl[n].fill_null(0).to_numpy(zero_copy_only=False), dtype=np.uint32) for n in names]


@pytest.mark.parametrize("nl,nr,n_ids", [(0, 10, 5), (10, 0, 5), (1, 1, 2), (300, 500, 20), (5000, 20_000, 400), (40_000, 3_000, 1500)])
@pytest.mark.parametrize("join_type", ["inner", "left outer"])
def test_hash_join_agrees_with_acero(nl, nr, n_ids, join_type):
    rng = np.random.default_rng(nl * 7 + nr)
    L, R = rand_table(rng, nl, 3, n_ids), rand_table(rng, nr, 2, n_ids)
    os_ = orc.OracleStore()
    for on in ([(0, 0)], [(0, 0), (1, 1)], [(1, 0)]):
        pb = PlanBuilder()
        jt = abi.JOIN_INNER if join_type == "inner" else abi.JOIN_LEFT
        desc = pb.build(pb.hash_join(pb.table(0, 3), pb.table(1, 2), on=on, join_type=jt))
        got, n, _ = os_.execute(desc, [L, R])
        exp = to_arrow(L, "l").join(to_arrow(R, "r"), keys=[f"l{a}" for a, _ in on], right_keys=[f"r{b}" for _, b in on],
                                    join_type=join_type, coalesce_keys=False)
        exp_cols = from_arrow(exp, ["l0", "l1", "l2", "r0", "r1"])
        assert n == exp.num_rows
        np.testing.assert_array_equal(ku.multiset(got, n), ku.multiset(exp_cols, exp.num_rows))


def test_join_with_id_filter_agrees_with_acero():
    """inner join + residual `l0 != r1` (the Q5 `product != X` shape): Acero join then compute filter"""
    rng = np.random.default_rng(11)
    L, R = rand_table(rng, 4000, 3, 300, null_frac=0.05), rand_table(rng, 6000, 2, 300, null_frac=0.05)
    pb = PlanBuilder()
    desc = pb.build(pb.hash_join(pb.table(0, 3), pb.table(1, 2), on=[(1, 0)], filter=ID_NEQ(col(0), col(4)), projection=[0, 4, 2]))
    got, n, _ = orc.OracleStore().execute(desc, [L, R])
    j = to_arrow(L, "l").join(to_arrow(R, "r"), keys=["l1"], right_keys=["r0"], join_type="inner", coalesce_keys=False)
    keep = pc.fill_null(pc.not_equal(j["l0"], j["r1"]), False)       # a null operand => not `true` => dropped
    j = j.filter(keep)
    np.testing.assert_array_equal(ku.multiset(got, n), ku.multiset(from_arrow(j, ["l0", "r1", "l2"]), j.num_rows))


@pytest.mark.parametrize("op,pc_op", [(GT, pc.greater), (LT, pc.less), (GEQ, pc.greater_equal), (LEQ, pc.less_equal)])
def test_integer_filter_agrees_with_arrow_compute(op, pc_op):
    """EBV(op(ENC_TV(col), literal)) over xsd:integer ids (the BSBM Q1 FILTER) vs an Arrow compare on the decoded values"""
    rng = np.random.default_rng(5)
    n_ids = 3000
    tv = np.zeros(n_ids, dtype=TV_DTYPE)
    values = rng.integers(-1000, 1000, n_ids)
    tv["tag"][1:] = abi.TV_INTEGER
    tv["lo"][1:] = values[1:]
    os_ = orc.OracleStore()
    os_.set_typed_values(tv)
    ids = rng.integers(0, n_ids, 50_000).astype(np.uint32)            # 0 = null => dropped
    payload = rng.integers(1, 1 << 30, len(ids)).astype(np.uint32)
    pb = PlanBuilder()
    desc = pb.build(pb.filter(pb.table(0, 2), EBV(op(ENC_TV(col(0)), integer(17))), projection=[1, 0]))
    got, n, _ = os_.execute(desc, [[ids, payload]])
    vals = pa.array(values[ids], type=pa.int64(), mask=(ids == 0))
    keep = np.asarray(pc.fill_null(pc_op(vals, 17), False))
    np.testing.assert_array_equal(ku.multiset(got, n), ku.multiset([payload[keep], ids[keep]]))


def test_topk_distinct_agrees_with_python_sorting():
    """DISTINCT + ORDER BY (term, id) LIMIT k per group in the oracle vs plain Python sorted(set(..))[:k]."""
    from rdf_fusion_amd.plan import PlanBuilder as PB
    rng = np.random.default_rng(8)
    strings = [f"label {i:04d}" for i in rng.permutation(300)]
    tv, _, _ = ku.string_dictionary(strings, lang_every=10 ** 9)
    os_ = orc.OracleStore()
    os_.set_typed_values(tv)
    for n, n_groups, limit in ((0, 3, 5), (1, 1, 5), (5000, 40, 5), (20_000, 1, 7), (3000, 600, 2)):
        g = rng.integers(1, n_groups + 1, n).astype(np.uint32)
        lab = rng.integers(0, 301, n).astype(np.uint32)                   # 0 = unbound: sorts first
        prod = rng.integers(1000, 1040, n).astype(np.uint32)
        for group in (0, None):
            pb = PB()
            desc = pb.build(pb.topk(pb.table(0, 3), keys=[(1, abi.SORT_BY_TERM), (2, abi.SORT_BY_ID)], limit=limit, group=group,
                                    projection=None if group == 0 else [1, 2]))
            cols, m, _ = os_.execute(desc, [[g, lab, prod]])
            exp = []
            for gg in (np.unique(g) if group == 0 else [None]):
                sel = slice(None) if gg is None else g == gg
                key = lambda l: (0, 0) if l == 0 else (int(tv["tag"][l]), int(tv["lo"][l]))
                rows = sorted(set((key(int(l)), int(p), int(l)) for l, p in zip(lab[sel], prod[sel])))[:limit]
                exp += [((int(gg),) if gg is not None else ()) + (l, p) for _, p, l in rows]
            got = sorted(tuple(int(c[r]) for c in cols) for r in range(m))
            assert got == sorted(exp), (n, n_groups, limit, group)


def test_acero_batched_q5_baseline_equals_oracle():
    """bench.py's tuned columnar CPU baseline (oracle/acero_baseline.py) computes the batched Q5 exactly"""
    from rdf_fusion_amd import bsbm
    from oracle import acero_baseline as ab
    ds = bsbm.generate(2500)
    st = orc.OracleStore()
    st.extend(ds.g, ds.s, ds.p, ds.o)
    st.set_typed_values(ds.typed_values, ds.decimals)
    rng = np.random.default_rng(1)
    prep = ab.prepare(ds)
    for batch_size in (1, 150):
        batch = np.array([ds.product(i) for i in rng.choice(ds.n_products, batch_size, replace=False)], dtype=np.uint32)
        cols, n, _ = st.execute(bsbm.q5_batch_plan(ds), [[np.arange(1, batch_size + 1, dtype=np.uint32), batch]])
        np.testing.assert_array_equal(ku.multiset(ab.run(prep, batch)), ku.multiset(cols, n))


def numeric_table(rng, n_ids):
    """ids 1.. : int / integer / float / double / decimal values incl. -0.0, NaN, equal values of different kinds; a string and an IRI"""
    import struct
    tv = np.zeros(n_ids, dtype=TV_DTYPE)
    dec, values = [], [None] * n_ids
    specials = [0.0, -0.0, float("nan"), float("inf"), -float("inf"), 1.5, -1.5, 2.0]
    for i in range(1, n_ids - 2):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            v = int(rng.integers(-50, 50)); tv["tag"][i], tv["lo"][i] = abi.TV_INT, v; values[i] = float(v)
        elif kind == 1:
            v = int(rng.integers(-10 ** 12, 10 ** 12)); tv["tag"][i], tv["lo"][i] = abi.TV_INTEGER, v; values[i] = float(v)
        elif kind == 2:
            v = np.float32(specials[int(rng.integers(0, 8))] if rng.random() < 0.4 else rng.normal() * 10)
            tv["tag"][i], tv["lo"][i] = abi.TV_FLOAT, int(v.view(np.uint32)); values[i] = float(v)
        elif kind == 3:
            v = specials[int(rng.integers(0, 8))] if rng.random() < 0.4 else float(rng.normal() * 1000)
            tv["tag"][i], tv["lo"][i] = abi.TV_DOUBLE, struct.unpack("<q", struct.pack("<d", v))[0]; values[i] = v
        else:
            q = int(rng.integers(-5000, 5000))                      # quarters: exact in binary
            raw = q * 25 * 10 ** 16
            tv["tag"][i], tv["lo"][i] = abi.TV_DECIMAL, len(dec)
            dec.append([raw & ((1 << 64) - 1), (raw >> 64) & ((1 << 64) - 1)]); values[i] = q / 4
    tv["tag"][n_ids - 2], tv["lo"][n_ids - 2] = abi.TV_STRING, 1
    tv["tag"][n_ids - 1], tv["lo"][n_ids - 1] = abi.TV_NAMED_NODE, 2
    return tv, np.array(dec, dtype=np.uint64).astype(np.int64).reshape(-1, 2), values


def total_order_key(v):
    import struct
    if v is None:
        return 0                                                    # unbound / not a number: first
    bits = struct.unpack("<Q", struct.pack("<d", v))[0]
    return (~bits) & ((1 << 64) - 1) if bits >> 63 else bits | (1 << 63)


def test_topk_by_numeric_value_agrees_with_python_sorting():
    """RDFGPU_SORT_BY_DOUBLE (ENC_SORT of a numeric: Double::from(Numeric), IEEE total order, everything else first) in
    the oracle vs Python: mixed numeric kinds, -0.0 < +0.0, NaN last, ties broken by the next key."""
    from rdf_fusion_amd.plan import PlanBuilder as PB
    rng = np.random.default_rng(6)
    tv, dec, values = numeric_table(rng, 400)
    os_ = orc.OracleStore()
    os_.set_typed_values(tv, dec)
    for n, limit in ((0, 5), (3000, 40), (200, 300)):
        price = rng.integers(0, 402, n).astype(np.uint32)                 # incl. unbound (0) and an id beyond the table
        offer = rng.integers(1, 50, n).astype(np.uint32)
        pb = PB()
        desc = pb.build(pb.topk(pb.table(0, 2), keys=[(1, abi.SORT_BY_DOUBLE), (0, abi.SORT_BY_ID), (1, abi.SORT_BY_ID)], limit=limit))
        cols, m, _ = os_.execute(desc, [[offer, price]])
        key = lambda o, p: (total_order_key(values[p] if 0 < p < 400 else None), o, p)
        exp = sorted(set(key(int(o), int(p)) for o, p in zip(offer, price)))[:limit]
        assert [(int(o), int(p)) for o, p in zip(cols[0][:m], cols[1][:m])] == [(o, p) for _, o, p in exp]
