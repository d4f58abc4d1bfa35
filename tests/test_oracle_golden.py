"""Pins the CPU oracle against the reference's own known-answer vectors
(tests/golden/reference_kats.json, transcribed from the reference's unit tests)."""
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
