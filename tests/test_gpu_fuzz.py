"""Differential fuzzing of the engine against the oracle: random operator trees (scans with constants / repeated
variables / graph variables, bound tables with nulls, FilterExecs over every expression family, inner / left hash joins
with one or two keys, residual filters and projections, cross / nested-loop joins, UnionExec, KleenePlusClosureExec,
DISTINCT + TopK) over a small random store with every typed-value kind.  Every plan runs three times (first run,
speculative re-execution, fused chain) and a second time from a freshly compiled plan (store-level table caches); each
result is compared multiset-equal with the oracle's.  The reference's own strategy for this layer is example-based
(SURVEY §4); this adds the property-based side."""
import numpy as np
import pytest

import rdf_fusion_amd as rf
from rdf_fusion_amd import abi, xsd
from rdf_fusion_amd.engine import TV_DTYPE
from rdf_fusion_amd.plan import (PlanBuilder, quad_pattern, col, lit_id, integer, double, date_time, ENC_TV, GT, LT, GEQ, LEQ, EQ, NEQ,
                                 ADD, SUB, EBV, ID_EQ, ID_NEQ, AND, OR, NOT, BOUND, IS_COMPATIBLE, REGEX, CONTAINS, LANGMATCHES_LANG)
from oracle import oracle as orc
import kat_util as ku

pytestmark = pytest.mark.gpu

N_IDS, N_SUBJ, PREDS = 64, 40, list(range(41, 47))       # the small configuration; `configure` switches
N_QUADS = 4000
LARGE = False


def configure(n_quads, n_ids, n_subj):
    global N_QUADS, N_IDS, N_SUBJ, PREDS, LARGE
    N_QUADS, N_IDS, N_SUBJ, LARGE = n_quads, n_ids, n_subj, n_quads > 10_000
    PREDS = list(range(n_subj + 1, n_subj + 7))
LANGS = ["", "en", "de-CH"]


def make_store(rng):
    n = N_QUADS
    g = rng.choice([0, 0, 0, 7, 9], n).astype(np.uint32)
    s = rng.integers(1, N_SUBJ + 1, n).astype(np.uint32)
    p = rng.choice(PREDS, n).astype(np.uint32)
    o = rng.integers(1, N_IDS, n).astype(np.uint32)
    tv = np.zeros(N_IDS, dtype=TV_DTYPE)
    dec, words, heap, offsets = [], ["alpha", "Beta", "gamma7", "", "delta beta", "épsilon", "K9", "beta"], bytearray(), np.zeros(N_IDS + 1, np.uint64)
    ranks = {w: r for r, w in enumerate(sorted(set(words)))}
    for i in range(1, N_IDS):
        kind = int(rng.integers(0, 8))
        if kind <= 1:
            tv["tag"][i], tv["lo"][i] = abi.TV_INTEGER, int(rng.integers(-6, 7))
        elif kind == 2:
            tv["tag"][i], tv["lo"][i] = abi.TV_DOUBLE, np.float64(rng.integers(-4, 5) / 2).view(np.int64)
        elif kind == 3:
            w = words[int(rng.integers(0, len(words)))]
            tv["tag"][i], tv["lo"][i], tv["aux"][i] = abi.TV_STRING, ranks[w], int(rng.integers(0, 3))
            tv["flags"][i] = abi.TVF_EMPTY_STRING if w == "" else 0
            heap += w.encode()
        elif kind == 4:
            tv["tag"][i], tv["lo"][i] = abi.TV_NAMED_NODE, i
        elif kind == 5:
            v = xsd.parse_date_time("2008-06-20T00:00:00")[0] + int(rng.integers(-30, 31)) * 3600 * xsd.SCALE
            tv["tag"][i], tv["lo"][i], tv["aux"][i] = abi.TV_DATE_TIME, len(dec), int(rng.integers(0, 2))
            dec.append([v & ((1 << 64) - 1), v >> 64])
        elif kind == 6:
            v = int(rng.integers(-5, 6)) * 10 ** 17
            tv["tag"][i], tv["lo"][i] = abi.TV_DECIMAL, len(dec)
            dec.append([v & ((1 << 64) - 1), (v >> 64) & ((1 << 64) - 1)])
        else:
            tv["tag"][i], tv["lo"][i] = abi.TV_BOOLEAN, int(rng.integers(0, 2))
        offsets[i + 1] = len(heap)
    decimals = np.array(dec, dtype=np.uint64).astype(np.int64).reshape(-1, 2)
    gs, os_ = rf.GpuQuadStore(), orc.OracleStore()
    assert gs.extend(g, s, p, o) == os_.extend(g, s, p, o)
    for st in (gs, os_):
        st.set_typed_values(tv, decimals)
        st.set_strings(offsets, bytes(heap))
    return gs, os_


def table_on_device(torch, cols):
    ts = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.uint32).view(np.int32)).cuda() for c in cols]
    return ts, [t.data_ptr() for t in ts]


class Gen:
    def __init__(self, rng, pb, oracle_run=None):
        """oracle_run(description) -> (columns, rows): when given, a join whose inputs would produce more than MAX_JOIN_ROWS
        rows (joins on a 3-valued graph column cascade into billions) is replaced by its left input"""
        self.rng, self.pb, self.oracle_run = rng, pb, oracle_run

    MAX_JOIN_ROWS = 200_000

    def join_rows(self, l, r, on):
        """exact size of the inner equi-join of two sub-plans on `on` (before any residual filter); None = unknown"""
        if self.oracle_run is None:
            return None
        try:
            (lc, nl), (rc, nr) = self.oracle_run(self.pb.build(l)), self.oracle_run(self.pb.build(r))
        except RuntimeError:                                  # a sub-plan the oracle refuses (the whole plan will be refused too)
            return None
        if nl == 0 or nr == 0:
            return 0
        if not on:
            return nl * nr
        def keyed(cols, n, idx):
            k = np.zeros(n, dtype=np.uint64); ok = np.ones(n, dtype=bool)
            for c in idx:
                k = k * np.uint64(1 << 20) + cols[c][:n].astype(np.uint64); ok &= cols[c][:n] != 0
            return np.unique(k[ok], return_counts=True)
        (ku_l, cl), (ku_r, cr) = keyed(lc, nl, [a for a, _ in on]), keyed(rc, nr, [b for _, b in on])
        common, il, ir = np.intersect1d(ku_l, ku_r, return_indices=True)
        return int((cl[il].astype(np.int64) * cr[ir].astype(np.int64)).sum()) + nl      # + the unmatched rows a left join keeps

    def r(self, n):
        return int(self.rng.integers(0, n))

    def tv_operand(self, w):
        k = self.r(6)
        if k <= 2:
            return ENC_TV(col(self.r(w)))
        if k == 3:
            return integer(self.r(9) - 4)
        if k == 4:
            return double((self.r(9) - 4) / 2)
        return date_time(xsd.parse_date_time("2008-06-20T00:00:00")[0] + (self.r(21) - 10) * 3600 * xsd.SCALE, bool(self.r(2)))

    def expr(self, w, depth=2):
        k = self.r(12 if depth else 8)
        cmp = [GT, LT, GEQ, LEQ, EQ, NEQ][self.r(6)]
        if k == 0:
            return (ID_EQ if self.r(2) else ID_NEQ)(col(self.r(w)), lit_id(1 + self.r(N_IDS - 1)))
        if k == 1:
            return (ID_EQ if self.r(2) else ID_NEQ)(col(self.r(w)), col(self.r(w)))
        if k == 2:
            return EBV(cmp(ENC_TV(col(self.r(w))), integer(self.r(9) - 4)))            # the `col cmp literal` kernel
        if k == 3:
            return EBV(cmp(self.tv_operand(w), self.tv_operand(w)))
        if k == 4:                                                                      # the Q5 window shape
            a, b = self.r(w), self.r(w)
            return AND(EBV(LT(ENC_TV(col(a)), ADD(ENC_TV(col(b)), integer(self.r(5))))), EBV(GT(ENC_TV(col(a)), SUB(ENC_TV(col(b)), integer(self.r(5))))))
        if k == 5:
            return BOUND(col(self.r(w))) if self.r(2) else IS_COMPATIBLE(col(self.r(w)), col(self.r(w)))
        if k == 6:
            pat = ["a", "^b", "beta|K", "[0-9]$", "e.a", "^$"][self.r(6)]
            return EBV(REGEX(ENC_TV(col(self.r(w))), pat, "i" if self.r(3) == 0 else "")) if self.r(3) else EBV(CONTAINS(ENC_TV(col(self.r(w))), "et", self.r(3)))
        if k == 7:
            return EBV(LANGMATCHES_LANG(ENC_TV(col(self.r(w))), ["en", "*", "de", ""][self.r(4)], LANGS))
        if k == 8:
            return NOT(self.expr(w, depth - 1))
        if k == 9:
            return EBV(cmp(ADD(self.tv_operand(w), self.tv_operand(w)), SUB(self.tv_operand(w), self.tv_operand(w))))
        return (AND if self.r(2) else OR)(self.expr(w, depth - 1), self.expr(w, depth - 1))

    def projection(self, w, at_most=5):
        n = 1 + self.r(min(w, at_most))
        return [self.r(w) for _ in range(n)] if self.r(4) == 0 else sorted(self.rng.choice(w, n, replace=False).tolist())

    def leaf(self):
        pb = self.pb
        k = self.r(10)
        if k == 0:
            return pb.table(0, 3)
        if k == 1:
            return pb.table(1, 2)
        names = ["a", "b", "c", "d"]
        var = lambda: names[self.r(4)]
        s = var() if self.r(8) else 1 + self.r(N_SUBJ)
        p = PREDS[self.r(len(PREDS))] if (LARGE or self.r(6)) else var()      # a variable predicate or graph is a 3- / 6-valued join key: small stores only
        o = var() if self.r(5) else 1 + self.r(N_IDS - 1)
        if not any(isinstance(t, str) for t in (s, p, o)):
            o = var()                                                                   # at least one column
        graph = "default" if LARGE else ["default", "default", "all", "named", [0, 9]][self.r(5)]
        gv = "g" if graph != "default" and self.r(2) else None
        return pb.data_source(quad_pattern(s, p, o, graph=graph, graph_variable=gv))

    def node(self, depth):
        pb = self.pb
        if depth == 0 or self.r(5) == 0:
            return self.leaf()
        k = self.r(9 if LARGE else 14) if not (LARGE and self.r(8) == 0) else 11      # large stores: filters, hash joins, TopK (closure / cross / NLJ sizes explode)
        if k <= 1:
            c = self.node(depth - 1)
            w = pb.width[c]
            return pb.filter(c, self.expr(w), projection=self.projection(w) if self.r(2) else None)
        if k <= 8:
            l, r = self.node(depth - 1), self.node(depth - 1)
            wl, wr = pb.width[l], pb.width[r]
            on = [(self.r(wl), self.r(wr)) for _ in range(1 if self.r(4) else 2)]
            w = wl + wr
            size = self.join_rows(l, r, on)
            if size is not None and size > self.MAX_JOIN_ROWS:
                return l
            return pb.hash_join(l, r, on=on, join_type=abi.JOIN_LEFT if self.r(4) == 0 else abi.JOIN_INNER,
                                filter=self.expr(w) if self.r(4) == 0 else None, projection=self.projection(w, 6) if self.r(3) else None)
        if k == 9:
            l, r = self.node(depth - 1), self.node(depth - 1)
            w = min(pb.width[l], pb.width[r])
            return pb.union(pb.projection(l, list(range(w))), pb.projection(r, list(range(w))), projection=self.projection(w) if self.r(2) else None)
        if k == 10:                                                                     # closure over (graph, s, o) of one predicate
            inner = pb.data_source(quad_pattern("s", PREDS[self.r(len(PREDS))], "o", graph="all", graph_variable="g"))
            if self.r(2):
                inner = pb.filter(inner, ID_NEQ(col(1), lit_id(1 + self.r(N_SUBJ))))
            return pb.closure(inner, allow_cross_graph_paths=bool(self.r(2)))
        if k == 11:
            c = self.node(depth - 1)
            w = pb.width[c]
            k1, k2 = self.r(w), self.r(w)
            modes = [abi.SORT_BY_ID, abi.SORT_BY_ID, abi.SORT_BY_DOUBLE, abi.SORT_BY_TERM]      # BY_TERM over mixed kinds: refused by both sides
            keys = [(k1, modes[self.r(4)]), (k2, modes[self.r(3)])]      # the builder appends the output columns as id keys (<= 4 in all)
            return pb.topk(c, keys=keys, limit=1 + self.r(6), group=None, projection=[k1, k2])
        l, r = self.leaf(), self.leaf()                                                  # cross / nested-loop joins: leaves only (size)
        if k == 12:
            fl = pb.filter(l, ID_EQ(col(0), lit_id(1 + self.r(20))))
            size = self.join_rows(fl, r, [])
            return l if size is not None and size > self.MAX_JOIN_ROWS else pb.cross_join(fl, r)
        w = pb.width[l] + pb.width[r]
        size = self.join_rows(l, r, [])
        if size is not None and size > 20 * self.MAX_JOIN_ROWS:                          # the nested loop visits every pair
            return l
        return pb.nested_loop_join(pb.filter(l, ID_NEQ(col(0), lit_id(3))), pb.filter(r, ID_EQ(col(0), lit_id(1 + self.r(20)))),
                                   join_type=abi.JOIN_LEFT if self.r(2) else abi.JOIN_INNER, filter=self.expr(w), projection=self.projection(w, 6))


def _seeds():
    """default: 12 small + 6 large stores; RDFGPU_FUZZ_SEEDS="200-260" hunts with other seeds (small and large each)"""
    import os
    extra = os.environ.get("RDFGPU_FUZZ_SEEDS")
    if extra:
        a, b = (int(x) for x in extra.split("-"))
        return [(s, "small") for s in range(a, b)] + [(s, "large") for s in range(a, b, 2)]
    return [(s, "small") for s in range(12)] + [(s, "large") for s in range(100, 106)]


@pytest.mark.parametrize("seed,size", _seeds())
def test_random_operator_trees(torch_cuda, seed, size):
    # small: 4000 quads over 64 ids (LDS-table joins, everything tiny, many plans); large: 120 k quads over 6000 ids
    # (HBM hash / direct-address / CSR tables cached on store slices, index joins, fused chains on re-execution)
    configure(*((4000, 64, 40) if size == "small" else (120_000, 6000, 3000)))
    rng = np.random.default_rng(1000 + seed)
    gs, os_ = make_store(rng)
    n0, n1 = (200, 50) if size == "small" else (5000, 2500)
    T0 = [rng.integers(0, N_IDS, n0).astype(np.uint32) for _ in range(3)]               # 0 = unbound
    T1 = [rng.integers(0, N_SUBJ, n1).astype(np.uint32) for _ in range(2)]
    k0, p0 = table_on_device(torch_cuda, T0)
    k1, p1 = table_on_device(torch_cuda, T1)
    ran = skipped = rows_total = errors = mutated = 0
    for it in range(150 if size == "small" else 60):
        pb = PlanBuilder()
        sizer = (lambda d: (lambda c, n, _: (c, n))(*os_.execute(d, [T0, T1]))) if size == "small" else None
        root = Gen(rng, pb, sizer).node((3 if size == "small" else 2) + (it % 3 == 0))
        desc = pb.build(root)
        used = sorted({int(n.table_slot) for n in pb.nodes if n.kind == abi.NODE_TABLE})
        bound = [(slot, [(p0, n0), (p1, n1)][slot]) for slot in used]
        try:
            exp, n_exp, _ = os_.execute(desc, [T0, T1])
        except Exception as e:                                                          # e.g. a null start in a closure: both sides refuse
            plan = None
            with pytest.raises(rf.RdfGpuError):
                plan = gs.plan(desc)
                for slot, (ptrs, n) in bound:
                    plan.bind_table(slot, ptrs, n)
                plan.execute()
            errors += 1
            continue
        if n_exp > 1_500_000:
            skipped += 1
            continue
        want = ku.multiset(exp, n_exp)
        try:
            gs.plan(desc).close()
        except rf.RdfGpuError as e:                                                     # documented limits (e.g. > 40 expression nodes): refused
            assert e.status == abi.ERR_UNSUPPORTED, e                                   # loudly at compile time, never computed elsewhere
            skipped += 1
            continue
        for fresh in range(2):
            plan = gs.plan(desc)
            for slot, (ptrs, n) in bound:
                plan.bind_table(slot, ptrs, n)
            for rep in range(3 if fresh == 0 else 1):
                got = plan.execute().fetch()
                assert plan.result_info()[0] == n_exp, (seed, it, fresh, rep)
                np.testing.assert_array_equal(ku.multiset(got, n_exp), want, err_msg=f"seed {seed} plan {it} fresh {fresh} rep {rep}")
            if fresh == 1 and it % 10 == 9:
                # the store changes under a compiled plan (cached slice tables, located ranges, range indexes and verdict
                # tables belong to a store version): the same plan object must answer for the new contents
                k = 60 if size == "small" else 600
                add = [rng.choice([0, 0, 7], k).astype(np.uint32), rng.integers(1, N_SUBJ + 1, k).astype(np.uint32),
                       rng.choice(PREDS, k).astype(np.uint32), rng.integers(1, N_IDS, k).astype(np.uint32)]
                assert gs.extend(*add) == os_.extend(*add)
                drop = [c[: k // 2] for c in add]
                assert gs.remove(*drop) == os_.remove(*drop)
                try:
                    exp2, n2, _ = os_.execute(desc, [T0, T1])
                except RuntimeError:                      # the new contents make the plan one both sides refuse (TopK by term over a date)
                    with pytest.raises(rf.RdfGpuError):
                        plan.execute()
                else:
                    got = plan.execute().fetch()
                    assert plan.result_info()[0] == n2, (seed, it, "after mutation")
                    np.testing.assert_array_equal(ku.multiset(got, n2), ku.multiset(exp2, n2), err_msg=f"seed {seed} plan {it} after store mutation")
                mutated += 1
            plan.close()
        ran += 1
        rows_total += n_exp
    assert ran >= (100 if size == "small" else 30) and rows_total > 0, (ran, skipped, errors)
