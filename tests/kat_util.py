"""Helpers shared by the golden-vector tests: JSON instruction notation -> plan objects."""
import numpy as np

from rdf_fusion_amd import abi
from rdf_fusion_amd.plan import MemIndexScanInstruction as I, MemIndexScanPredicate as P

COMPONENTS = {"GSPO": abi.GSPO, "GPOS": abi.GPOS, "GOSP": abi.GOSP}


def pred(d):
    if d is None:
        return None
    if d == "false":
        return P.false()
    if "in" in d:
        return P.in_(d["in"])
    if "between" in d:
        return P.between(*d["between"])
    if "equal_to" in d:
        return P.equal_to(d["equal_to"])
    raise ValueError(d)


def instr(j):
    """["T"] | ["T", 7] | ["T", {...}] | ["S", "x"] | ["S", "x", {...}]"""
    if j[0] == "T":
        if len(j) == 1:
            return I.traverse()
        return I.traverse(j[1]) if isinstance(j[1], int) else I.traverse_with_predicate(pred(j[1]))
    if len(j) == 2:
        return I.scan(j[1])
    return I.scan_with_predicate(j[1], pred(j[2]))


def instrs(js):
    return [instr(j) for j in js]


def abi_instrs(js):
    """-> list of 4 abi.ScanInstruction + pool (for the host-logic entry points)"""
    from rdf_fusion_amd.plan import PlanBuilder
    pb = PlanBuilder()
    return [pb._instr(instr(j)) for j in js], pb.pool


def quad_columns(quads):
    a = np.array(quads, dtype=np.uint32).reshape(-1, 4)
    return a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy(), a[:, 3].copy()


def same_pred(p, expected):
    """compare a plan.MemIndexScanPredicate with the JSON expectation"""
    if expected is None:
        return p is None
    if p is None:
        return False
    e = pred(expected)
    return (p.kind, p.ids, p.lo, p.hi) == (e.kind, e.ids, e.lo, e.hi)


def multiset(cols, n_rows=None):
    """rows of a column list as a sorted (n, k) array — multiset comparison of binding tables"""
    if not cols:
        return np.zeros((n_rows or 0, 0), np.uint32)
    m = np.stack([np.asarray(c, dtype=np.uint32) for c in cols], axis=1)
    if len(m) == 0:
        return m
    order = np.lexsort(tuple(m[:, k] for k in reversed(range(m.shape[1]))))
    return m[order]


def kat_literal(v):
    """["int"|"integer"|"decimal"|"double", value] of reference_kats.json -> a typed-value literal expression"""
    from rdf_fusion_amd.plan import int32, integer, decimal, double
    kind, val = v
    return {"int": int32, "integer": integer, "decimal": lambda r: decimal(int(r)), "double": double}[kind](int(val) if kind != "double" else val)


def numeric_kat_plans(kats):
    """(name, plan description over a bound 1-row table, expected row count) for every numeric KAT: the row
    survives iff the reference's assertion holds.  An error is the SPARQL error value: `x = x` is then not true."""
    from rdf_fusion_amd.plan import PlanBuilder, ADD, SUB, EQ, EBV, LT, GT, AND, ENC_TV, col, double  # noqa: F401
    out = []
    for c in kats["numeric_arith"]:
        f = ADD if c["op"] == "add" else SUB
        z = f(kat_literal(c["a"]), kat_literal(c["b"]))
        pb = PlanBuilder()
        if c["expect"] == "error":
            expr, n = EBV(EQ(z, z)), 0
        else:
            expr, n = EBV(EQ(z, kat_literal(c["expect"]))), 1
        out.append((f'{c["src"]} {c["op"]} {c["a"]} {c["b"]}', pb.build(pb.filter(pb.table(0, 1), expr)), n))
    for c in kats["decimal_to_double"]:
        x = kat_literal(["decimal", c["raw"]])
        pb = PlanBuilder()
        if c["tol"] == 0.0:
            expr = EBV(EQ(x, double(c["value"])))            # decimal vs double compares as doubles (numeric.rs:127-201)
        else:
            expr = AND(EBV(LT(x, double(c["value"] + c["tol"]))), EBV(GT(x, double(c["value"] - c["tol"]))))
        out.append((f'decimal->double {c["raw"]}', pb.build(pb.filter(pb.table(0, 1), expr)), 1))
    return out
