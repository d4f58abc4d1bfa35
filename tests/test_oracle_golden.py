"""Pins the CPU oracle against the reference's own known-answer vectors
(tests/golden/reference_kats.json, transcribed from the reference's unit tests)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from rdf_fusion_amd import abi
import kat_util as ku


def _store(quads, batch):
    st = orc.OracleStore(batch_size=batch)
    if quads:
        st.extend(*ku.quad_columns(quads))
    return st


def test_scan_kats(kats):
    for case in kats["scan"]:
        st = _store(case["quads"], case["batch"])
        r = st.scan(ku.instrs(case["instr"]), force_index=abi.GSPO)
        name = case["name"]
        if "n_rows" in case:
            assert r["n_rows"] == case["n_rows"], name
        if "n_cols" in case:
            assert len(r["columns"]) == case["n_cols"], name
        if "columns" in case:
            assert set(r["columns"]) == set(case["columns"]), name
            for k, v in case["columns"].items():
                assert r["columns"][k].tolist() == v, name
        if "batches" in case:
            assert r["batches"] == case["batches"], name
        if "first_batch_rows" in case:
            assert r["batches"][0] == case["first_batch_rows"], name


def test_remove_kats(kats):
    for case in kats["remove"]:
        st = _store(case["insert"], 10)
        removed = st.remove(*ku.quad_columns(case["remove"]))
        assert removed == case["removed"], case["name"]
        assert len(st) == case["remaining"], case["name"]
        r = st.scan(ku.instrs([["S", "a"], ["S", "b"], ["S", "c"], ["S", "d"]]), force_index=abi.GSPO)
        assert r["n_rows"] == 0 and r["batches"] == []


def test_store_kats(kats):
    for case in kats["store"]:
        st = _store(case["quads_gspo"], 10)
        r = st.scan(ku.instrs(case["instr"]))
        if "chosen" in case:
            assert abi.INDEX_NAMES[r["index"]] == case["chosen"], case["name"]
        assert r["order"] == case["order"], case["name"]
        for k, v in case["columns"].items():
            assert r["columns"][k].tolist() == v, case["name"]


def test_predicate_and_kats(kats):
    for case in kats["predicate_and"]:
        out = orc.predicate_and(ku.pred(case["lhs"]), ku.pred(case["rhs"]))
        assert ku.same_pred(out, case["out"]), case


def test_index_choice_kats(kats):
    for case in kats["index_choice"]:
        ins, _ = ku.abi_instrs(case["instr"])
        avail = 0
        for name in case.get("available", ["GSPO", "GPOS", "GOSP"]):
            avail |= 1 << ku.COMPONENTS[name]
        assert abi.INDEX_NAMES[orc.choose_index(ins, avail)] == case["chosen"], case


def test_score_order_kats(kats):
    for case in kats["score_order"]:
        if "greater" in case:
            assert orc.scan_score(ku.abi_instrs(case["greater"])[0]) > orc.scan_score(ku.abi_instrs(case["lesser"])[0]), case["name"]
        else:
            assert orc.scan_score(ku.abi_instrs(case["equal"])[0]) == orc.scan_score(ku.abi_instrs(case["to"])[0]), case["name"]


OPS = {"Eq": abi.OP_EQ, "Gt": abi.OP_GT, "GtEq": abi.OP_GTEQ, "Lt": abi.OP_LT, "LtEq": abi.OP_LTEQ}


def test_pushdown_kats(kats):
    for case in kats["pushdown"]:
        out = orc.pushdown_to_scan_predicate(OPS[case["op"]], case["value"])
        assert ku.same_pred(out, case["out"]), case
    for case in kats["pushdown_display"]:
        cur = None
        for op, value in case["filters"]:
            p = orc.pushdown_to_scan_predicate(OPS[op], value)
            cur = p if cur is None else orc.predicate_and(cur, p)
        assert repr(cur) == case["display"], case


def test_rowgroup_and_dedupe_kats(kats):
    for case in kats["rowgroups"]:
        st = _store([[v] * 4 for v in case["values"]], case["size"])
        sl, _ = st.prune(abi.GSPO, ku.instrs([["T"]] * 4))
        assert [e - s for s, e in sl] == case["groups"], case["src"]
    for case in kats["dedupe"]:
        st = _store([[v] * 4 for v in case["first"]], case["size"])
        st.extend(*ku.quad_columns([[v] * 4 for v in case["second"]]))
        assert len(st) == case["length"]


def test_prune_kats(kats):
    for case in kats["prune"]:
        st = _store(case["quads"], case["size"])
        sl, dropped = st.prune(abi.GSPO, ku.instrs(case["instr"]))
        lens = [e - s for s, e in sl]
        src = case["src"]
        if "group_lens" in case:
            assert lens == case["group_lens"], src
        if "n_groups" in case:
            assert len(lens) == case["n_groups"], src
        if case.get("n_groups_is_all"):
            n = len(case["quads"])
            assert len(lens) == (n + case["size"] - 1) // case["size"], src
        for i in case.get("dropped", []):
            assert dropped & (1 << i), src
        for i in case.get("kept", []):
            assert not dropped & (1 << i), src
        if "rows" in case:
            cols = st.read_index(abi.GSPO)
            rows = [[int(c[i]) for c in cols] for s, e in sl for i in range(s, e)]
            assert rows == case["rows"], src


def test_prune_empty_index():
    st = orc.OracleStore(batch_size=2)
    sl, _ = st.prune(abi.GSPO, ku.instrs([["T"]] * 4))
    assert sl == []


def test_find_range_kats(kats):
    kinds = {"Before": orc.FR_BEFORE, "NotContained": orc.FR_NOT_CONTAINED, "Contained": orc.FR_CONTAINED,
             "After": orc.FR_AFTER}
    for case in kats["find_range"]:
        r, lo, hi = orc.find_range_between(case["values"], case["value"], case["value"])
        exp = case["result"]
        assert r == kinds[exp[0]], case
        if exp[0] == "Contained":
            assert (lo, hi) == (exp[1], exp[2]), case
        if exp[0] == "NotContained":
            assert lo == exp[1], case


def test_numeric_kats(kats):
    """checked add/sub at the type bounds and Decimal -> Double (the reference's xsd in-file tests)"""
    os_ = orc.OracleStore()
    one = [np.array([7], dtype=np.uint32)]
    for name, desc, n_expected in ku.numeric_kat_plans(kats):
        _, n, _ = os_.execute(desc, [one])
        assert n == n_expected, name


def test_typed_value_type_ids(kats):
    """lib/encoding/src/typed_value/encoding.rs (test_type_ids + the enum order :248-268): the ABI's typed-value tags ARE
    the dense-union type ids, and the oracle's."""
    names = {"NamedNode": "TV_NAMED_NODE", "BlankNode": "TV_BLANK_NODE", "String": "TV_STRING", "Boolean": "TV_BOOLEAN", "Float": "TV_FLOAT",
             "Double": "TV_DOUBLE", "Decimal": "TV_DECIMAL", "Int": "TV_INT", "Integer": "TV_INTEGER", "DateTime": "TV_DATE_TIME",
             "Time": "TV_TIME", "Date": "TV_DATE", "Duration": "TV_DURATION", "OtherLiteral": "TV_OTHER", "Null": "TV_NULL"}
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rdfgpu.h")).read()
    for field, type_id in kats["type_ids"]:
        assert getattr(abi, names[field]) == type_id, field
        m = re.search(r"RDFGPU_" + names[field] + r" = (\d+)", header)
        assert m and int(m.group(1)) == type_id, field


def test_join_lowering_kats(kats):
    """SparqlJoinLoweringRule on inputs without shared variables (lib/logical/src/join/rewrite.rs:381-481): the lowered
    plan as the reference prints it, and what it computes: every left row with every right row; a left join against an
    empty right input keeps every left row once, unbound on the right."""
    from rdf_fusion_amd.plan import PlanBuilder, explain_logical_join
    for case in kats["join_lowering"]:
        pb = PlanBuilder()
        l = pb.table(0, len(case["left"]), case["left"]); r = pb.table(1, len(case["right"]), case["right"])
        jt = abi.JOIN_INNER if case["join_type"] == "Inner" else abi.JOIN_LEFT
        node = pb.sparql_join(l, r, jt)
        assert [explain_logical_join(pb, node), "  EmptyRelation: rows=0", "  EmptyRelation: rows=0"] == case["plan"], case["src"]
        desc = pb.build(node)
        os_ = orc.OracleStore()
        a, b = np.array([5, 6, 7], np.uint32), np.array([8, 9], np.uint32)
        cols, n, _ = os_.execute(desc, [[a], [b]])
        assert sorted(zip(cols[0][:n].tolist(), cols[1][:n].tolist())) == [(x, y) for x in (5, 6, 7) for y in (8, 9)], case["src"]
        cols, n, _ = os_.execute(desc, [[a], [b[:0]]])     # the reference's test inputs: EmptyRelation rows=0
        expect = [] if jt == abi.JOIN_INNER else [(5, 0), (6, 0), (7, 0)]
        assert sorted(zip(cols[0][:n].tolist(), cols[1][:n].tolist())) == expect, case["src"]
    # shared, non-nullable variables: an equi-join on all of them (join/rewrite.rs:126-168), left fields then the new right fields
    pb = PlanBuilder()
    node = pb.sparql_join(pb.table(0, 2, ["s", "x"]), pb.table(1, 3, ["x", "y", "s"]))
    assert explain_logical_join(pb, node) == "Inner Join: s = s, x = x" and pb.names[node] == ["s", "x", "y"]


def test_join_row_fixtures(kats):
    """Join ROW results pinned by a fixture the reference holds (testsuite/oxigraph-tests/sparql/nested_anonymous.*):
    a three-pattern BGP = two equi-joins; the oracle's HashJoinExec restatement must give the .srx rows in every
    association order, and on disjoint renamed copies of the data (the fixture's rows once per copy)."""
    import itertools
    for case in kats["join_fixtures"]:
        os_ = orc.OracleStore()
        os_.extend(*ku.quad_columns(case["quads_gspo"]))
        for order in itertools.permutations(range(len(case["patterns"]))):
            pb, root = ku.bgp_plan(case["patterns"], case["select"], order)
            cols, n, _ = os_.execute(pb.build(root))
            assert sorted(zip(*[c[:n].tolist() for c in cols])) == sorted(map(tuple, case["rows"])), (case["name"], order)
        quads, rows = ku.scaled_join_fixture(case, 500)
        os_ = orc.OracleStore()
        os_.extend(*ku.quad_columns(quads))
        pb, root = ku.bgp_plan(case["patterns"], case["select"])
        cols, n, _ = os_.execute(pb.build(root))
        assert sorted(zip(*[c[:n].tolist() for c in cols])) == sorted(map(tuple, rows)), case["name"]


def test_bsbm_plans_equal_the_reference_execution_plan_snapshots(kats):
    """bsbm.q5_plan / q1_plan ARE the operator trees of bench/tests/plans/snapshots/..Q5 / Q1 (Execution Plan).snap below the
    SortExec: node for node — operators, join keys, JoinFilters, projections, the index every DataSourceExec scans
    (IndexPermutations::choose_index through the library's host logic)."""
    from rdf_fusion_amd import bsbm
    from rdf_fusion_amd.plan import explain
    ds = bsbm.generate(300)
    pb, root = bsbm.q5_plan(ds, ds.product(7), builder=True)
    assert explain(pb, root) == kats["plan_snapshots"]["q5"]["lines"]
    pb, root = bsbm.q1_plan(ds, ds.type_base, ds.feature_base + 3, ds.feature_base + 5, 136, builder=True)
    assert explain(pb, root) == kats["plan_snapshots"]["q1"]["lines"]
