"""CPU-only checks of the product library: it loads, exports every symbol of include/rdfgpu.h,
its host logic (no device access) reproduces the reference's KATs, and it fails loudly — not
silently on a CPU path — when no GPU is present."""
import ctypes as C
import os
import re

import pytest

import rdf_fusion_amd as rf
from rdf_fusion_amd import abi
import kat_util as ku

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = rf.load_library()
    header = open(os.path.join(ROOT, "include", "rdfgpu.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rdfgpu_[a-z_0-9]+)\s*\(", header))
    assert declared == set(abi.EXPORTED_SYMBOLS), declared ^ set(abi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rdfgpu_abi_version() == abi.ABI_VERSION


def test_struct_layouts_match_header():
    assert C.sizeof(abi.TypedValue) == 16
    assert C.sizeof(abi.ScanInstruction) == 16
    assert C.sizeof(abi.ExprNode) == 24
    assert C.sizeof(abi.PlanNode) == 16 + 64 + 4 + 16 + 16 + 6 * 4


def test_predicate_and_kats(kats):
    for case in kats["predicate_and"]:
        out = rf.predicate_and(ku.pred(case["lhs"]), ku.pred(case["rhs"]))
        assert ku.same_pred(out, case["out"]), case


def test_index_choice_and_score_kats(kats):
    for case in kats["index_choice"]:
        ins, _ = ku.abi_instrs(case["instr"])
        avail = 0
        for name in case.get("available", ["GSPO", "GPOS", "GOSP"]):
            avail |= 1 << ku.COMPONENTS[name]
        assert abi.INDEX_NAMES[rf.choose_index(ins, avail)] == case["chosen"], case
    for case in kats["score_order"]:
        if "greater" in case:
            assert rf.scan_score(ku.abi_instrs(case["greater"])[0]) > rf.scan_score(ku.abi_instrs(case["lesser"])[0])
        else:
            assert rf.scan_score(ku.abi_instrs(case["equal"])[0]) == rf.scan_score(ku.abi_instrs(case["to"])[0])


def test_pushdown_kats(kats):
    ops = {"Eq": abi.OP_EQ, "Gt": abi.OP_GT, "GtEq": abi.OP_GTEQ, "Lt": abi.OP_LT, "LtEq": abi.OP_LTEQ}
    for case in kats["pushdown"]:
        assert ku.same_pred(rf.pushdown_to_scan_predicate(ops[case["op"]], case["value"]), case["out"]), case
    for case in kats["pushdown_display"]:
        cur = None
        for op, value in case["filters"]:
            p = rf.pushdown_to_scan_predicate(ops[op], value)
            cur = p if cur is None else rf.predicate_and(cur, p)
        assert repr(cur) == case["display"], case


def test_host_logic_agrees_with_oracle_on_random_instructions():
    import numpy as np
    from oracle import oracle as orc
    rng = np.random.default_rng(7)
    from rdf_fusion_amd.plan import MemIndexScanInstruction as I, MemIndexScanPredicate as P, PlanBuilder
    for _ in range(300):
        ins = []
        for lvl in range(4):
            r = rng.integers(0, 6)
            if r == 0:
                ins.append(I.traverse())
            elif r == 1:
                ins.append(I.traverse(int(rng.integers(0, 50))))
            elif r == 2:
                ins.append(I.scan(f"v{lvl}"))
            elif r == 3:
                a = int(rng.integers(0, 50))
                ins.append(I.scan_with_predicate(f"v{lvl}", P.between(a, a + int(rng.integers(0, 3)))))
            elif r == 4:
                ins.append(I.traverse_with_predicate(P.in_(rng.integers(1, 9, size=int(rng.integers(1, 4))).tolist())))
            else:
                ins.append(I.traverse_with_predicate(P.equal_to("v0")))
        pb = PlanBuilder()
        raw = [pb._instr(i) for i in ins]
        for avail in (0b111, 0b101, 0b011):
            assert rf.choose_index(raw, avail) == orc.choose_index(raw, avail)
        assert rf.scan_score(raw) == orc.scan_score(raw)


def test_no_silent_cpu_fallback():
    """Without a GPU the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rf.RdfGpuError) as e:
        rf.GpuQuadStore()
    assert e.value.status == abi.ERR_NO_DEVICE


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "rdf-fusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle: the product path must not use it"


def test_engine_option_names_match_the_header():
    """include/rdfgpu.h section 4b: option ids, their names and the Python mirror agree; no device needed."""
    import re
    lib = rf.load_library()
    header = open(os.path.join(ROOT, "include", "rdfgpu.h")).read()
    declared = re.findall(r"RDFGPU_OPT_([A-Z0-9_]+?)(?: = 0)?,", header)
    assert declared == abi.OPTION_NAMES
    for i, name in enumerate(abi.OPTION_NAMES):
        assert lib.rdfgpu_option_name(i).decode() == name
    assert lib.rdfgpu_option_name(len(abi.OPTION_NAMES)) is None


def test_environment_is_read_once_not_on_the_execute_path():
    """the RDFGPU_* toggles are process defaults (store.cpp: default_engine_options), copied into the
    store and the plan; nothing that compiles or executes a plan may call getenv."""
    csrc = os.path.join(ROOT, "rdf-fusion_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".cpp", ".hpp", ".hip")):
            continue
        text = open(os.path.join(csrc, name)).read()
        if name == "store.cpp":
            assert text.count("getenv") == 1
        else:
            assert "getenv" not in text, name
