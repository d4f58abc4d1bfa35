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
            try:
                key = ntriples_decode_term(t)          # interned by what the parser hands the dictionary, not by spelling
            except ValueError:
                raise ValueError(n_line)
            if key not in ids:
                ids[key] = len(terms) + 1
                terms.append(t)
            cols[k].append(ids[key])
    return terms, cols[0], cols[1], cols[2]


_NT_ECHAR = {b"t": b"\t", b"b": b"\b", b"n": b"\n", b"r": b"\r", b"f": b"\f", b'"': b'"', b"'": b"'", b"\\": b"\\"}
_XSD = b"http://www.w3.org/2001/XMLSchema#"
_XSD_INTEGERS = {b"integer", b"byte", b"short", b"long", b"unsignedByte", b"unsignedShort", b"unsignedInt", b"unsignedLong",
                 b"positiveInteger", b"negativeInteger", b"nonPositiveInteger", b"nonNegativeInteger"}


def _nt_unescape(raw, echar):
    """ECHAR / UCHAR of the N-Triples grammar -> UTF-8 (what oxttl's lexer hands on).  ValueError on a malformed escape."""
    import re

    def one(m):
        body = m.group(1)
        if body[:1] in (b"u", b"U"):
            want = 4 if body[:1] == b"u" else 8
            if len(body) != 1 + want:
                raise ValueError("bad UCHAR")
            cp = int(body[1:], 16)
            if cp > 0x10FFFF or 0xD800 <= cp <= 0xDFFF:
                raise ValueError("not a Unicode scalar value")
            return chr(cp).encode("utf-8")
        if not echar or body not in _NT_ECHAR:
            raise ValueError("bad ECHAR")
        return _NT_ECHAR[body]

    out = re.sub(rb"\\(u[0-9A-Fa-f]{0,4}|U[0-9A-Fa-f]{0,8}|.|$)", one, raw, flags=re.S)
    return out


def ntriples_decode_term(term):
    """One term as written -> (kind, lexical form, suffix), the canonical form the reference's parser produces (oxttl + oxrdf):
    escapes decoded, the language tag in lower case, the datatype IRI decoded, `"x"^^xsd:string` = the simple literal "x".
    kind: 1 IRI, 2 blank node, 3 simple literal, 4 language-tagged literal, 5 typed literal (abi.NT_*)."""
    t = bytes(term)
    if t.startswith(b"<"):
        return 1, _nt_unescape(t[1:-1], False), b""
    if t.startswith(b"_:"):
        return 2, t[2:], b""
    q = 1                                              # the closing quote: the first unescaped one
    while t[q:q + 1] != b'"':
        q += 2 if t[q:q + 1] == b"\\" else 1
    lex, rest = _nt_unescape(t[1:q], True), t[q + 1:]
    if rest.startswith(b"@"):
        return 4, lex, rest[1:].lower()
    if rest.startswith(b"^^<"):
        dt = _nt_unescape(rest[3:-1], False)
        return (3, lex, b"") if dt == _XSD + b"string" else (5, lex, dt)
    return 3, lex, b""


def _nearest_float(frac, single):
    """frac (a Fraction) rounded to the nearest binary32 / binary64, ties to even — Rust's f32 / f64::from_str on a decimal string."""
    import struct
    from fractions import Fraction
    cand = float(frac)                                   # correctly rounded to binary64 by Python
    if not single:
        return cand
    c32 = np.float32(cand)
    best, best_err = None, None
    for nb in (np.nextafter(c32, np.float32(-np.inf)), c32, np.nextafter(c32, np.float32(np.inf))):
        if not np.isfinite(nb):
            continue
        err = abs(Fraction(float(nb)) - frac)
        even = (struct.unpack("<I", struct.pack("<f", float(nb)))[0] & 1) == 0
        if best is None or err < best_err or (err == best_err and even):
            best, best_err = nb, err
    return float(best)


def ntriples_typed_value(kind, lex, suffix):
    """The typed-value row of one decoded term (lib/model/src/typed_value.rs:349-411 + the xsd FromStr impls), independent of the
    device code: returns (tag, lo, flags, dec_hi, host) — `host` True where the DEVICE is allowed to leave the value to the host's
    parser (off Clinger's fast path, dateTime / durations); for those `lo` is not compared."""
    import re
    import struct
    from fractions import Fraction
    from rdf_fusion_amd import abi
    if kind == 1:
        return abi.TV_NAMED_NODE, 0, 0, 0, False
    if kind == 2:
        return abi.TV_BLANK_NODE, 0, 0, 0, False
    if kind in (3, 4):
        return abi.TV_STRING, 0, abi.TVF_EMPTY_STRING if not lex else 0, 0, False
    if not suffix.startswith(_XSD):
        return abi.TV_OTHER, 0, 0, 0, False
    name = suffix[len(_XSD):]
    if name in _XSD_INTEGERS or name == b"int":
        ok = re.fullmatch(rb"[+-]?[0-9]+", lex) is not None        # i64::from_str / i32::from_str
        v = int(lex) if ok else 0
        lo_b, hi_b = (-2**31, 2**31 - 1) if name == b"int" else (-2**63, 2**63 - 1)
        if not ok or not lo_b <= v <= hi_b:
            return abi.TV_NULL, 0, 0, 0, False
        return (abi.TV_INT if name == b"int" else abi.TV_INTEGER), v, 0, 0, False
    if name == b"boolean":
        if lex in (b"true", b"1"):
            return abi.TV_BOOLEAN, 1, 0, 0, False
        if lex in (b"false", b"0"):
            return abi.TV_BOOLEAN, 0, 0, 0, False
        return abi.TV_NULL, 0, 0, 0, False
    if name == b"decimal":
        m = re.fullmatch(rb"([+-]?)([0-9]*)(?:\.([0-9]*))?", lex)      # (\+|-)?([0-9]+(\.[0-9]*)?|\.[0-9]+), decimal.rs:501
        if not m or (not m.group(2) and not m.group(3)):
            return abi.TV_NULL, 0, 0, 0, False
        frac_digits = (m.group(3) or b"").rstrip(b"0")
        if len(frac_digits) > 18:
            return abi.TV_NULL, 0, 0, 0, False                       # underflow
        scaled = int((m.group(2) or b"0") + frac_digits + b"0" * (18 - len(frac_digits)))
        if m.group(1) == b"-":
            scaled = -scaled
        # the reference accumulates digit by digit in a checked i128, then multiplies by 10^(18 - fractional digits)
        unscaled = int((m.group(2) or b"0") + frac_digits or b"0")
        if not -2**127 <= (-unscaled if m.group(1) == b"-" else unscaled) <= 2**127 - 1 or not -2**127 <= scaled <= 2**127 - 1:
            return abi.TV_NULL, 0, 0, 0, False
        u = scaled & ((1 << 128) - 1)
        to_i64 = lambda x: x - (1 << 64) if x >= (1 << 63) else x
        return abi.TV_DECIMAL, to_i64(u & ((1 << 64) - 1)), 0, to_i64(u >> 64), False
    if name in (b"double", b"float"):
        single = name == b"float"
        tag = abi.TV_FLOAT if single else abi.TV_DOUBLE
        bits = (lambda x: struct.unpack("<I", struct.pack("<f", x))[0]) if single else (lambda x: struct.unpack("<q", struct.pack("<d", x))[0])
        low = lex.lower()
        if low in (b"inf", b"+inf"):
            return tag, bits(float("inf")), 0, 0, False
        if low == b"-inf":
            return tag, bits(float("-inf")), 0, 0, False
        if low == b"nan":
            return tag, 0, 0, 0, None                                # (the payload of a NaN is not pinned: compared as "is NaN")
        m = re.fullmatch(rb"([+-]?)([0-9]*)(?:\.([0-9]*))?(?:[eE]([+-]?[0-9]+))?", lex)
        if not m or (not m.group(2) and not m.group(3)):
            return tag, 0, abi.TVF_NEEDS_HOST, 0, True               # not this grammar: the host's parser decides
        digits = ((m.group(2) or b"") + (m.group(3) or b"")).lstrip(b"0")
        mant = int(digits or b"0")
        scale = int(m.group(4) or b"0") - len(m.group(3) or b"")
        fast = len(digits) <= 19 and mant < (1 << (24 if single else 53)) and abs(scale) <= (10 if single else 22) and abs(int(m.group(4) or b"0")) <= 10000
        if mant == 0:
            fast = len(digits) <= 19 and abs(int(m.group(4) or b"0")) <= 10000
        if not fast:
            return tag, 0, abi.TVF_NEEDS_HOST, 0, True
        val = _nearest_float(Fraction(mant) * Fraction(10) ** scale, single) if mant else 0.0
        if m.group(1) == b"-":
            val = -val
        return tag, bits(val), 0, 0, False
    host_tags = {b"dateTime": abi.TV_DATE_TIME, b"time": abi.TV_TIME, b"date": abi.TV_DATE, b"duration": abi.TV_DURATION,
                 b"yearMonthDuration": abi.TV_DURATION, b"dayTimeDuration": abi.TV_DURATION}
    if name in host_tags:
        return host_tags[name], 0, abi.TVF_NEEDS_HOST, 0, True
    return abi.TV_OTHER, 0, 0, 0, False


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
