"""ctypes binding of the CPU oracle (oracle/librdf_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import rdf_fusion_amd.abi as abi
from rdf_fusion_amd.plan import MemIndexScanPredicate as P

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FR_BEFORE, FR_NOT_CONTAINED, FR_CONTAINED, FR_AFTER = 0, 1, 2, 3


class ScanResult(C.Structure):
    _fields_ = [("n_cols", C.c_uint32), ("vars", C.c_uint32 * 4), ("n_rows", C.c_uint64),
                ("cols", C.POINTER(C.c_uint32) * 4), ("n_batches", C.c_uint32),
                ("batch_rows", C.POINTER(C.c_uint32)), ("chosen_index", C.c_uint32)]


class Table(C.Structure):
    _fields_ = [("n_cols", C.c_uint32), ("n_rows", C.c_uint64),
                ("cols", C.POINTER(C.c_uint32) * abi.MAX_COLUMNS)]


class BoundTable(C.Structure):
    _fields_ = [("cols", C.POINTER(C.c_void_p)), ("n_cols", C.c_uint32), ("n_rows", C.c_uint64)]


def build(force=False):
    so = os.path.join(_HERE, "librdf_oracle.so")
    src = os.path.join(_HERE, "rdf_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        l = C.CDLL(build())
        vp, u64p, u32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        l.orc_store_new.restype = vp
        l.orc_store_new.argtypes = [C.c_uint32]
        l.orc_store_free.argtypes = [vp]
        l.orc_store_extend.restype = C.c_uint64
        l.orc_store_extend.argtypes = [vp, vp, vp, vp, vp, C.c_uint64]
        l.orc_store_remove.restype = C.c_uint64
        l.orc_store_remove.argtypes = [vp, vp, vp, vp, vp, C.c_uint64]
        l.orc_store_clear.argtypes = [vp]
        l.orc_store_len.restype = C.c_uint64
        l.orc_store_len.argtypes = [vp]
        l.orc_store_adopt_sorted.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.c_uint64]
        l.orc_store_read_index.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.c_uint64, u64p]
        l.orc_store_set_typed_values.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
        l.orc_store_set_strings.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
        l.orc_regex_is_match.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        l.orc_regex_is_match.restype = C.c_int
        l.orc_store_set_faithful_decode.argtypes = [vp, C.c_int]
        l.orc_scan_score.restype = C.c_uint64
        l.orc_scan_score.argtypes = [C.POINTER(abi.ScanInstruction)]
        l.orc_choose_index.restype = C.c_uint32
        l.orc_choose_index.argtypes = [C.POINTER(abi.ScanInstruction), C.c_uint32]
        l.orc_predicate_and.argtypes = [C.POINTER(abi.Predicate)] * 3 + [u32p]
        l.orc_pushdown_to_scan_predicate.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(abi.Predicate)]
        l.orc_find_range_between.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, u64p, u64p]
        l.orc_prune.argtypes = [vp, C.c_uint32, C.POINTER(abi.ScanInstruction), u32p, u64p, u64p,
                                C.c_uint32, u32p]
        l.orc_scan.argtypes = [vp, C.POINTER(abi.ScanInstruction), u32p, C.c_int, C.POINTER(ScanResult)]
        l.orc_scan_result_free.argtypes = [C.POINTER(ScanResult)]
        l.orc_plan_execute.argtypes = [vp, C.POINTER(abi.PlanDesc), C.POINTER(BoundTable), C.c_uint32,
                                       C.POINTER(Table), C.POINTER(abi.Metrics)]
        l.orc_table_free.argtypes = [C.POINTER(Table)]
        l.orc_last_error.restype = C.c_char_p
        l.orc_eval_bool.argtypes = [vp, C.POINTER(abi.ExprNode), C.c_uint32, C.POINTER(vp), C.c_uint32,
                                    C.c_uint64, vp]
        l.orc_eval_set_table.argtypes = [C.POINTER(abi.Regex), C.c_uint32]
        l.orc_eval_set_table.restype = None
        l.orc_eval_str.argtypes = [vp, C.POINTER(abi.ExprNode), C.c_uint32, C.POINTER(abi.Regex), C.c_uint32, C.POINTER(vp), C.c_uint32,
                                   C.c_uint64, vp, vp, vp, vp, C.c_uint64]
        l.orc_eval_tv.argtypes = [vp, C.POINTER(abi.ExprNode), C.c_uint32, C.POINTER(vp), C.c_uint32,
                                  C.c_uint64, vp, vp]
        _LIB = l
    return _LIB


def _err():
    return RuntimeError("oracle: " + lib().orc_last_error().decode())


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(C.c_void_p)


class NeedsUnicodeTables(Exception):
    """`\\d \\w \\s \\b` over a subject with non-ASCII characters: the restatement holds only their ASCII members."""


def regex_is_match(pattern, flags, subject):
    """REGEX restatement (regex_oracle.c): True / False, or None for the error value."""
    p, f, s = (x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in (pattern, flags, subject))
    r = lib().orc_regex_is_match(p, len(p), f, len(f), s, len(s))
    if r == -2:
        raise NeedsUnicodeTables(pattern)
    return None if r < 0 else bool(r)


def ntriples_encode(text):
    """The per-triple half of the reference's bulk load, restated (Store::load_from_reader, lib/rdf-fusion/src/store.rs:477-493
    -> MemObjectIdMapping::encode_quad, object_id_mapping.rs:106-116): every term of every triple line is interned, ids in
    insertion order from 1.  Returns (terms: list of bytes, id t + 1 = terms[t]; s, p, o id lists).  Terms are kept as
    written (`<iri>`, `_:b`, `"lex"`, `"lex"@en`, `"lex"^^<dt>`).  Raises ValueError(line number) on a malformed line."""
    data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    ids, terms, cols = {}, [], ([], [], [])
    n_line = 0
    for raw in data.split(b"\n"):
        line = raw.strip(b" \t\r")
        if not line or line.startswith(b"#"):
            continue
        n_line += 1
        p, got = 0, []
        for k in range(3):
            while p < len(line) and line[p:p + 1] in b" \t\r":
                p += 1
            b = p
            c = line[p:p + 1]
            if c == b"<":
                e = line.find(b">", p)
                if e < 0:
                    raise ValueError(n_line)
                p = e + 1
            elif line[p:p + 2] == b"_:":
                while p < len(line) and line[p:p + 1] not in b" \t\r":
                    p += 1
                if k == 2 and p > b + 2 and line[p - 1:p] == b".":
                    p -= 1
            elif c == b'"' and k == 2:
                p += 1
                while p < len(line) and line[p:p + 1] != b'"':
                    p += 2 if line[p:p + 1] == b"\\" else 1
                if p >= len(line):
                    raise ValueError(n_line)
                p += 1
                if line[p:p + 1] == b"@":
                    p += 1
                    while p < len(line) and (line[p:p + 1].isalnum() or line[p:p + 1] == b"-"):
                        p += 1
                elif line[p:p + 3] == b"^^<":
                    e = line.find(b">", p)
                    if e < 0:
                        raise ValueError(n_line)
                    p = e + 1
            else:
                raise ValueError(n_line)
            got.append(line[b:p])
        rest = line[p:].strip(b" \t\r")
        if not rest.startswith(b".") or (rest[1:].strip(b" \t\r") and not rest[1:].strip(b" \t\r").startswith(b"#")):
            raise ValueError(n_line)
        for k, t in enumerate(got):
            if t not in ids:
                ids[t] = len(terms) + 1
                terms.append(t)
            cols[k].append(ids[t])
    return terms, cols[0], cols[1], cols[2]


def decode_terms(ids, typed_values, offsets, heap):
    """ENC_PT restated (MemObjectIdMapping::decode_array, object_id_mapping.rs:331-374; PlainTermType, plain_term/
    encoding.rs:90-127): per id None (null), or (term_type, lexical form, tag, aux) with term_type 0 named node / 1 blank
    node / 2 literal."""
    from rdf_fusion_amd import abi
    out = []
    n_ids = len(typed_values)
    for i in np.asarray(ids, dtype=np.uint32).tolist():
        if i == 0 or i >= n_ids or typed_values["tag"][i] == abi.TV_NULL:
            out.append(None)
            continue
        tag = int(typed_values["tag"][i])
        tt = 0 if tag == abi.TV_NAMED_NODE else 1 if tag == abi.TV_BLANK_NODE else 2
        form = bytes(heap[int(offsets[i]):int(offsets[i + 1])]).decode("utf-8") if i + 1 < len(offsets) else ""
        out.append((tt, form, tag, int(typed_values["aux"][i])))
    return out


def find_range_between(values, lo, hi):
    """MemColumnChunk::find_range_between; None values = nulls (stored as 0)."""
    v = np.array([0 if x is None else x for x in values], dtype=np.uint32)
    a, b = C.c_uint64(), C.c_uint64()
    r = lib().orc_find_range_between(v.ctypes.data_as(C.c_void_p), len(v), lo, hi, C.byref(a), C.byref(b))
    return r, a.value, b.value


def scan_score(instrs):
    return int(lib().orc_scan_score((abi.ScanInstruction * 4)(*instrs)))


def choose_index(gspo, available=0b111):
    return int(lib().orc_choose_index((abi.ScanInstruction * 4)(*gspo), available))


def predicate_and(a, b):
    from rdf_fusion_amd.engine import predicate_and as _pa
    return _pa(a, b, lib_fn=lib().orc_predicate_and)


def pushdown_to_scan_predicate(op, value):
    from rdf_fusion_amd.engine import pushdown_to_scan_predicate as _pd
    return _pd(op, value, lib_fn=lib().orc_pushdown_to_scan_predicate)


class OracleStore:
    def __init__(self, batch_size=8192):
        self._l = lib()
        self._h = C.c_void_p(self._l.orc_store_new(batch_size))
        self.batch_size = batch_size

    def close(self):
        if self._h:
            self._l.orc_store_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extend(self, g, s, p, o):
        (g, gp), (s, sp), (p, pp), (o, op) = _u32(g), _u32(s), _u32(p), _u32(o)
        return self._l.orc_store_extend(self._h, gp, sp, pp, op, len(g))

    def remove(self, g, s, p, o):
        (g, gp), (s, sp), (p, pp), (o, op) = _u32(g), _u32(s), _u32(p), _u32(o)
        return self._l.orc_store_remove(self._h, gp, sp, pp, op, len(g))

    def clear(self):
        self._l.orc_store_clear(self._h)

    def __len__(self):
        return self._l.orc_store_len(self._h)

    def adopt_sorted(self, components, cols):
        ptrs = [_u32(c) for c in cols]
        if self._l.orc_store_adopt_sorted(self._h, components, *[p[1] for p in ptrs], len(ptrs[0][0])):
            raise _err()

    def read_index(self, components):
        n = C.c_uint64()
        self._l.orc_store_read_index(self._h, components, None, None, None, None, 0, C.byref(n))
        cols = [np.empty(n.value, np.uint32) for _ in range(4)]
        self._l.orc_store_read_index(self._h, components, *[c.ctypes.data_as(C.c_void_p) for c in cols],
                                     n.value, C.byref(n))
        return cols

    def set_strings(self, offsets, heap):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        heap = np.frombuffer(bytes(heap), dtype=np.uint8) if not isinstance(heap, np.ndarray) else np.ascontiguousarray(heap, dtype=np.uint8)
        self._l.orc_store_set_strings(self._h, offsets.ctypes.data_as(C.c_void_p), len(offsets) - 1,
                                      heap.ctypes.data_as(C.c_void_p), len(heap))

    def set_typed_values(self, values, decimals=None):
        values = np.ascontiguousarray(values)
        assert values.dtype.itemsize == 16
        dec = np.ascontiguousarray(decimals if decimals is not None else np.zeros((0, 2), np.int64), dtype=np.int64).reshape(-1, 2)   # (lo, hi) per i128
        self._l.orc_store_set_typed_values(self._h, values.ctypes.data_as(C.c_void_p), len(values),
                                           dec.ctypes.data_as(C.c_void_p), len(dec))

    def set_faithful_decode(self, on):
        self._l.orc_store_set_faithful_decode(self._h, int(on))

    # -- scan ---------------------------------------------------------------------------------
    def _build_instrs(self, instructions):
        """instructions: 4 plan.MemIndexScanInstruction -> (ctypes array, pool array)"""
        from rdf_fusion_amd.plan import PlanBuilder
        pb = PlanBuilder()
        arr = (abi.ScanInstruction * 4)(*[pb._instr(i) for i in instructions])
        pool = (C.c_uint32 * max(1, len(pb.pool)))(*pb.pool)
        return arr, pool, pb

    def prune(self, components, instructions):
        """instructions in index order. -> ([(start,end)...], dropped_mask)"""
        arr, pool, _ = self._build_instrs(instructions)
        cap = 1 << 16
        st, en = (C.c_uint64 * cap)(), (C.c_uint64 * cap)()
        dropped = C.c_uint32()
        n = self._l.orc_prune(self._h, components, arr, pool, st, en, cap, C.byref(dropped))
        if n < 0:
            raise _err()
        return [(st[i], en[i]) for i in range(n)], dropped.value

    def scan(self, instructions, force_index=-1):
        """instructions in G,S,P,O order -> dict(columns={var: np.array}, batches=[..], index=..)"""
        arr, pool, pb = self._build_instrs(instructions)
        res = ScanResult()
        if self._l.orc_scan(self._h, arr, pool, force_index, C.byref(res)):
            raise _err()
        names = {v: k for k, v in pb.vars.items()}
        cols = {}
        order = []
        for c in range(res.n_cols):
            a = np.ctypeslib.as_array(res.cols[c], shape=(res.n_rows,)).copy() if res.n_rows else np.zeros(0, np.uint32)
            cols[names[res.vars[c]]] = a
            order.append(names[res.vars[c]])
        batches = [res.batch_rows[i] for i in range(res.n_batches)]
        out = dict(columns=cols, order=order, n_rows=res.n_rows, batches=batches, index=res.chosen_index)
        self._l.orc_scan_result_free(C.byref(res))
        return out

    # -- plans --------------------------------------------------------------------------------
    def execute(self, description, tables=None):
        """description: plan.PlanDescription -> (list of numpy columns, metrics)"""
        out, m = Table(), abi.Metrics()
        bt = None
        keep = []
        n_tables = 0
        if tables:
            n_tables = len(tables)
            bt = (BoundTable * n_tables)()
            for i, cols in enumerate(tables):
                cols = [np.ascontiguousarray(c, dtype=np.uint32) for c in cols]
                ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data_as(C.c_void_p) for c in cols])
                keep.append((cols, ptrs))
                bt[i].cols, bt[i].n_cols, bt[i].n_rows = ptrs, len(cols), (len(cols[0]) if cols else 0)
        if self._l.orc_plan_execute(self._h, C.byref(description.desc), bt, n_tables, C.byref(out), C.byref(m)):
            raise _err()
        cols = [np.ctypeslib.as_array(out.cols[c], shape=(out.n_rows,)).copy() if out.n_rows else np.zeros(0, np.uint32)
                for c in range(out.n_cols)]
        n_rows = out.n_rows
        self._l.orc_table_free(C.byref(out))
        return cols, n_rows, m

    def _program(self, expr):
        """ctypes nodes of an expression, its string-table references resolved like a plan resolves them (and the table handed
        to this thread's evaluator)."""
        from rdf_fusion_amd.plan import PlanBuilder
        pb = PlanBuilder()
        pb._expr(abi.PlanNode(), expr)
        nodes = (abi.ExprNode * max(1, len(pb.exprs)))(*pb.exprs)
        keep = [(bytes(r[0]), bytes(r[1]), int(r[2]) if len(r) > 2 else 0) for r in pb.regexes]
        regexes = (abi.Regex * max(1, len(keep)))(*[abi.Regex(p, f, len(p), len(f), pid, 0) for p, f, pid in keep])
        self._l.orc_eval_set_table(regexes, len(keep))
        return nodes, (keep, regexes)

    def eval_bool(self, expr, cols, n_rows=None):
        nodes, _keep = self._program(expr)
        cols = [np.ascontiguousarray(c, dtype=np.uint32) for c in cols]
        n = n_rows if n_rows is not None else (len(cols[0]) if cols else 1)
        ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data_as(C.c_void_p) for c in cols])
        out = np.empty(n, np.uint8)
        if self._l.orc_eval_bool(self._h, nodes, len(expr.nodes), ptrs, len(cols), n, out.ctypes.data_as(C.c_void_p)):
            raise _err()
        return out

    def eval_str(self, expr, cols, n_rows=None):
        """A string-valued expression row by row: [None (the error value / not a string) | (bytes, language id)]."""
        from rdf_fusion_amd.plan import PlanBuilder
        pb = PlanBuilder()
        node = abi.PlanNode()
        pb._expr(node, expr)                                  # resolves the string-table references like a plan does
        nodes = (abi.ExprNode * max(1, len(pb.exprs)))(*pb.exprs)
        keep = [(bytes(r[0]), bytes(r[1])) for r in pb.regexes]
        regexes = (abi.Regex * max(1, len(keep)))(*[abi.Regex(p, f, len(p), len(f), 0, 0) for p, f in keep])
        cols = [np.ascontiguousarray(c, dtype=np.uint32) for c in cols]
        n = n_rows if n_rows is not None else (len(cols[0]) if cols else 1)
        ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data_as(C.c_void_p) for c in cols])
        state, lang, off = np.zeros(n, np.uint8), np.zeros(n, np.uint32), np.zeros(n + 1, np.uint64)
        cap = 1 << 24
        buf = np.zeros(cap, np.uint8)
        if self._l.orc_eval_str(self._h, nodes, len(pb.exprs), regexes, len(keep), ptrs, len(cols), n, state.ctypes.data_as(C.c_void_p),
                                lang.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), cap):
            raise _err()
        raw = buf.tobytes()
        return [(raw[int(off[i]):int(off[i + 1])], int(lang[i])) if state[i] else None for i in range(n)]

    def eval_tv(self, expr, cols, n_rows=None):
        from rdf_fusion_amd.engine import TV_DTYPE
        nodes, _keep = self._program(expr)
        cols = [np.ascontiguousarray(c, dtype=np.uint32) for c in cols]
        n = n_rows if n_rows is not None else (len(cols[0]) if cols else 1)
        ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data_as(C.c_void_p) for c in cols])
        out = np.zeros(n, TV_DTYPE)
        hi = np.zeros(n, np.int64)
        if self._l.orc_eval_tv(self._h, nodes, len(expr.nodes), ptrs, len(cols), n,
                               out.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p)):
            raise _err()
        return out, hi
