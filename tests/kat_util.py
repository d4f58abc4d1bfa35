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
    """["int"|"integer"|"decimal"|"double"|"float", value] or ["string", rank, language id] of reference_kats.json -> a
    typed-value literal expression; "NaN" / "INF" / "-INF" / "MAX" / "MIN" name the IEEE specials of the kind"""
    from rdf_fusion_amd import abi
    from rdf_fusion_amd.plan import int32, integer, decimal, double, float32, lit_tv
    kind, val = v[0], v[1]
    if kind == "string":
        return lit_tv(abi.TV_STRING, int(val), aux=int(v[2]))
    if kind in ("double", "float"):
        fi = np.finfo(np.float64 if kind == "double" else np.float32)
        x = {"NaN": float("nan"), "INF": float("inf"), "-INF": float("-inf"), "MAX": float(fi.max), "MIN": float(fi.min)}.get(val, val)
        return (double if kind == "double" else float32)(x)
    return {"int": int32, "integer": integer, "decimal": lambda r: decimal(int(r))}[kind](int(val))


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
    for c in kats["decimal_to_float"]:       # Float x Decimal compares as floats (numeric.rs:127-201): Float::from(Decimal) == the f32
        from rdf_fusion_amd.plan import float32
        pb = PlanBuilder()
        expr = EBV(EQ(kat_literal(["decimal", c["raw"]]), float32(c["value"])))
        out.append((f'decimal->float {c["raw"]}', pb.build(pb.filter(pb.table(0, 1), expr)), 1))
    for kind in ("float", "double"):         # effective_boolean_value.rs:111-112: `value != 0` — NaN is not equal to 0: EBV(NaN) is true
        pb = PlanBuilder()
        out.append((f"EBV {kind} NaN (code-derived)", pb.build(pb.filter(pb.table(0, 1), EBV(kat_literal([kind, "NaN"])))), 1))
    for c in kats.get("ebv", []):            # Boolean::from(number): the row survives FILTER(value) iff the EBV is true
        pb = PlanBuilder()
        out.append((f'{c["src"]} EBV {c["value"]}', pb.build(pb.filter(pb.table(0, 1), EBV(kat_literal(c["value"])))), 1 if c["ebv"] else 0))
    # PartialOrd of two values: which of `<`, `=`, `>` is true — none of them when the values are incomparable (error)
    for c in kats["compare"]:
        truth = {"Less": (1, 0, 0), "Equal": (0, 1, 0), "Greater": (0, 0, 1), "None": (0, 0, 0)}[c["ordering"]]
        for op, name, n in ((LT, "LT", truth[0]), (EQ, "EQ", truth[1]), (GT, "GT", truth[2])):
            pb = PlanBuilder()
            expr = EBV(op(kat_literal(c["a"]), kat_literal(c["b"])))
            out.append((f'{c["src"]} {name} {c["a"]} {c["b"]}', pb.build(pb.filter(pb.table(0, 1), expr)), n))
    return out


# ---------------------------------------------------------------------------------------------------
# REGEX: random patterns inside the supported subset, rendered for the reference's syntax (Rust `regex`
# crate, what the oracle and the device parse) and for Python's `re` (a third opinion in the tests)
# ---------------------------------------------------------------------------------------------------
REGEX_ALPHABET = ["a", "b", "c", "k", "K", "s", "S", "x", "0", " ", ".", "\n", "é", "€", "K", "ſ", "😀"]
_META = set(".+*?()|[]{}^$\\#&-~")


def _lit(ch):
    return ("\\" + ch) if ch in _META else ch


def _py_class(members, other):
    """A Python character class for a set of ASCII code points (+ every non-ASCII character when `other`)."""
    import re
    if other:
        missing = [c for c in range(128) if c not in members]
        return "[^" + "".join(re.escape(chr(c)) for c in missing) + "]" if missing else "[\\s\\S]"
    return "[" + "".join(re.escape(chr(c)) for c in sorted(members)) + "]" if members else "[^\\s\\S]"


_POSIX = {"alpha": "A-Za-z", "digit": "0-9", "upper": "A-Z", "lower": "a-z", "alnum": "0-9A-Za-z", "xdigit": "0-9A-Fa-f", "word": "0-9A-Za-z_"}
_UNI = {"L": "A-Za-z", "Lu": "A-Z", "Ll": "a-z", "N": "0-9", "Nd": "0-9", "Letter": "A-Za-z", "Pd": "\\-", "Zs": " "}


def _expand(spec):
    out, k = set(), 0
    spec = spec.replace("\\-", "\x00")
    while k < len(spec):
        if k + 2 < len(spec) and spec[k + 1] == "-":
            out |= set(range(ord(spec[k]), ord(spec[k + 2]) + 1)); k += 3
        else:
            out.add(ord("-") if spec[k] == "\x00" else ord(spec[k])); k += 1
    return out


def random_regex(rng, depth=0, perl=False, extended=False):
    """Returns (pattern, flags, python_pattern, python_flags).  `perl`: also draw `\\d \\w \\s \\D \\W \\S` (alone and inside
    classes) and `\\b \\B` — restated / compiled with their ASCII members, so pair them with all-ASCII subjects
    (python_flags then carries re.ASCII).  `extended`: also inline flag groups `(?i:..)` `(?s:..)`, a leading `(?i)` / `(?s)`,
    ASCII POSIX classes, class set operations `&& -- ~~` with nested (negated) classes — and, with `perl`, Unicode general
    categories `\\p{..}`; their Python forms are spelled out sets (a third evaluation of the set algebra)."""
    import re
    flags = "".join(f for f in "ismx" if rng.random() < 0.25)
    # non-ASCII letters under `i` need the Unicode case-folding tables: outside the restated / supported subset
    all_letters = REGEX_ALPHABET
    ascii_letters = [c for c in REGEX_ALPHABET if ord(c) < 0x80]
    state = {"i": "i" in flags}

    def class_operand():
        """one operand of a set operation: (text, ASCII members, includes other non-ASCII)"""
        kind = int(rng.integers(0, 4))
        if kind == 0:
            spec = ["a-f", "c-k", "0-9", "a-z", "b-dx", "aeiou"][int(rng.integers(0, 6))]
            return spec, _expand(spec), False
        if kind == 1:
            spec = ["aeiou", "c", "0-4", "b-y"][int(rng.integers(0, 4))]
            return f"[^{spec}]", set(range(128)) - _expand(spec), True
        if kind == 2:
            name = list(_POSIX)[int(rng.integers(0, len(_POSIX)))]
            return f"[:{name}:]", _expand(_POSIX[name]), False
        spec = ["abc", "x-z", "k"][int(rng.integers(0, 3))]
        return f"[{spec}]q", _expand(spec) | {ord("q")}, False

    def atom(d):
        r = rng.random()
        if extended and r < 0.3:
            k = int(rng.integers(0, 6 if perl else 4))
            if k == 0 and d < 2:                                   # a group with its own flags
                fl = "i" if rng.random() < 0.5 else "s"
                saved = state["i"]
                state["i"] = state["i"] or fl == "i"
                a, b = alt(d + 1)
                state["i"] = saved
                return f"(?{fl}:{a})", f"(?{fl}:{b})"
            if k == 1:
                name = list(_POSIX)[int(rng.integers(0, len(_POSIX)))]
                extra = "_" if rng.random() < 0.3 else ""
                return f"[[:{name}:]{extra}]", f"[{_POSIX[name]}{extra}]"
            if k in (2, 3) and not state["i"]:                    # set operations (case folding of operands: left to the known answers)
                text, members, other = class_operand()
                for _ in range(int(rng.integers(1, 3))):
                    op = ["&&", "--", "~~"][int(rng.integers(0, 3))]
                    t2, m2, o2 = class_operand()
                    text += op + t2
                    if op == "&&":
                        members, other = members & m2, other and o2
                    elif op == "--":
                        members, other = members - m2, other and not o2
                    else:
                        members, other = members ^ m2, other != o2
                if rng.random() < 0.25:
                    text, members, other = "^" + text, set(range(128)) - members, not other
                return f"[{text}]", _py_class(members, other)
            if k >= 4:
                name = list(_UNI)[int(rng.integers(0, len(_UNI)))]
                neg = rng.random() < 0.25
                form = (f"\\P{{{name}}}" if neg else f"\\p{{{name}}}") if len(name) > 1 or rng.random() < 0.5 else (f"\\P{name}" if neg else f"\\p{name}")
                return form, f"[{'^' if neg else ''}{_UNI[name]}]"
        letters = ascii_letters if state["i"] else all_letters
        if perl and r < 0.25:
            k = int(rng.integers(0, 10))
            if k < 6:
                c = ["\\d", "\\w", "\\s", "\\D", "\\W", "\\S"][k]
                return c, c
            if k < 8:
                c = "[" + ("^" if rng.random() < 0.3 else "") + "".join(m for m in ["\\d", "\\s", "\\w", "x", "\\."] if rng.random() < 0.4) + "a]"
                return c, c
            c = "\\b" if k == 8 else "\\B"
            return c + "a", c + "a"              # a repetition operator never lands on the assertion itself
        if r < 0.45:
            ch = letters[int(rng.integers(0, len(letters)))]
            if ch == "\n":
                return "\\n", "\\n"
            if ch == " ":
                return "\\ ", "\\ "          # stays a literal under the x flag too
            return _lit(ch), re.escape(ch)
        if r < 0.6:
            return ".", "."
        if r < 0.8:
            members = [m for m in ["a", "b", "c", "k", "s", "x", "0", "."] if rng.random() < 0.4] or ["a"]
            body = "".join("\\." if m == "." else m for m in members)
            if rng.random() < 0.3:
                body += "a-c"
            neg = "^" if rng.random() < 0.3 else ""
            return f"[{neg}{body}]", f"[{neg}{body}]"
        if d < 2:
            a, b = alt(d + 1)
            grp = "(?:" if rng.random() < 0.5 else "("
            return f"{grp}{a})", f"{grp}{b})"
        return "a", "a"

    def repeat(d):
        a, b = atom(d)
        r = rng.random()
        suffix = ""
        if r < 0.15:
            suffix = "*"
        elif r < 0.3:
            suffix = "+"
        elif r < 0.4:
            suffix = "?"
        elif r < 0.5:
            lo = int(rng.integers(0, 3))
            suffix = "{%d,%d}" % (lo, lo + int(rng.integers(0, 3))) if rng.random() < 0.6 else "{%d,}" % lo if rng.random() < 0.5 else "{%d}" % lo
        return a + suffix, b + suffix

    def cat(d):
        parts = [repeat(d) for _ in range(int(rng.integers(1, 4)))]
        return "".join(p[0] for p in parts), "".join(p[1] for p in parts)

    def alt(d):
        parts = [cat(d) for _ in range(1 if rng.random() < 0.7 else 2)]
        return "|".join(p[0] for p in parts), "|".join(p[1] for p in parts)

    top_alt = rng.random() < 0.2
    lead = ""
    if extended and rng.random() < 0.15:                           # a leading directive: the whole pattern (Python: the same, at the start only)
        lead = "(?i)" if rng.random() < 0.5 else "(?s)"
        state["i"] = state["i"] or lead == "(?i)"
    pat, py = alt(depth) if top_alt else cat(depth)
    if perl and not top_alt and rng.random() < 0.3:
        pat, py = pat + "\\b", py + "\\b"
    if not top_alt:                       # anchors only around a pattern without top-level alternation
        multiline = "m" in flags
        if rng.random() < 0.3:
            pat, py = "^" + pat, "^" + py
        if rng.random() < 0.3:
            pat, py = pat + "$", py + ("$" if multiline else "\\Z")   # Python's bare $ also matches before a final \n
    py_flags = re.ASCII if perl else 0
    for f, v in (("i", re.I), ("s", re.S), ("m", re.M), ("x", re.X)):
        if f in flags:
            py_flags |= v
    if lead:
        pat = lead + pat
        py_flags |= re.I if lead == "(?i)" else re.S
    return pat, flags, py, py_flags


def regex_needs_unicode_fold_care(pattern, flags):
    """Negated classes containing k / s under `i` are outside the device subset (KELVIN SIGN / LONG S)."""
    return "i" in flags and "[^" in pattern


ASCII_ALPHABET = [c for c in REGEX_ALPHABET if ord(c) < 0x80] + ["_", "7", "\t", "-"]


def random_subject(rng, ascii_only=False):
    n = int(rng.integers(0, 9))
    alphabet = ASCII_ALPHABET if ascii_only else REGEX_ALPHABET
    return "".join(alphabet[int(rng.integers(0, len(alphabet)))] for _ in range(n))


def string_dictionary(strings, n_other=5, lang_every=5, lang_id=7):
    """Typed values + string heap for ids 1..len(strings) (simple / language-tagged literals) followed by `n_other`
    non-string ids (integers).  Every `lang_every`-th literal carries language id `lang_id`."""
    from rdf_fusion_amd import abi
    from rdf_fusion_amd.engine import TV_DTYPE
    n_ids = 1 + len(strings) + n_other
    tv = np.zeros(n_ids, dtype=TV_DTYPE)
    order = {s: r for r, s in enumerate(sorted(set(strings)))}
    offsets = np.zeros(n_ids + 1, dtype=np.uint64)
    heap = bytearray()
    for k, s_ in enumerate(strings):
        i = 1 + k
        b = s_.encode("utf-8")
        tv["tag"][i] = abi.TV_STRING
        tv["lo"][i] = order[s_]
        tv["aux"][i] = 0 if k % lang_every else lang_id
        tv["flags"][i] = abi.TVF_EMPTY_STRING if not b else 0
        offsets[i] = len(heap)
        heap += b
        offsets[i + 1] = len(heap)
    for i in range(1 + len(strings), n_ids):
        tv["tag"][i] = abi.TV_INTEGER
        tv["lo"][i] = i
        offsets[i + 1] = len(heap)
    return tv, offsets, bytes(heap)


def bgp_plan(patterns, select, order=None):
    """A basic graph pattern as the reference plans it: one DataSourceExec per triple pattern (default graph), joined
    left-deep in textual order on ALL shared variables (logical_plan_builder_context.rs:239-250 + SparqlJoinLoweringRule,
    join/rewrite.rs:126-168), then the SELECT projection.  `order` = another association order of the same patterns
    (inner joins: the multiset of solutions does not depend on it).  Returns (PlanBuilder, root)."""
    from rdf_fusion_amd.plan import PlanBuilder, quad_pattern
    pb = PlanBuilder()
    order = list(range(len(patterns))) if order is None else list(order)
    node = None
    for i in order:
        s, p, o = patterns[i]
        src = pb.data_source(quad_pattern(s, p, o))
        node = src if node is None else pb.sparql_join(node, src)
    root = pb.projection(node, [pb.names[node].index(v) for v in select])
    return pb, root


def scaled_join_fixture(case, copies):
    """`copies` disjoint renamings of a join fixture's data (every non-predicate, non-query-constant term of copy k shifted
    by k * stride): the expected rows are the fixture's rows once per copy, shifted alike.  A derived vector — it keeps the
    fixture's join shape while the inputs grow past the sizes at which the engine changes its join table form."""
    consts = {t for pat in case["patterns"] for t in pat if isinstance(t, int)}
    preds = {q[2] for q in case["quads_gspo"]}
    stride = max(case["terms"].values()) + 1
    shift = lambda t, k: t if (t in consts or t in preds or t == 0) else t + k * stride
    quads = [[q[0], shift(q[1], k), q[2], shift(q[3], k)] for k in range(copies) for q in case["quads_gspo"]]
    rows = [[shift(v, k) for v in r] for k in range(copies) for r in case["rows"]]
    return quads, rows


def term_dictionary(terms, languages=("",)):
    """Typed values + string heap for ids 1..len(terms); a term is [kind, lexical, datatype-or-language] with kind in
    iri / bnode / literal, the third entry None (simple literal), "@tag" (language-tagged) or "xsd:<type>".  The heap holds
    the lexical form AS WRITTEN of every term (what the host dictionary stores); `languages` numbers the language tags
    (index 0 = no language).  Numeric payloads: int / integer parsed, everything else opaque here."""
    from rdf_fusion_amd import abi
    from rdf_fusion_amd.engine import TV_DTYPE
    languages = list(languages)
    n_ids = 1 + len(terms)
    tv = np.zeros(n_ids, dtype=TV_DTYPE)
    strings = sorted({t[1] for t in terms})
    rank = {s: r for r, s in enumerate(strings)}
    offsets = np.zeros(n_ids + 1, dtype=np.uint64)
    heap = bytearray()
    for k, t in enumerate(terms):
        i = 1 + k
        kind, lex, extra = t[0], t[1], (t[2] if len(t) > 2 else None)
        b = lex.encode("utf-8")
        offsets[i] = len(heap); heap += b; offsets[i + 1] = len(heap)
        if kind == "iri":
            tv["tag"][i] = abi.TV_NAMED_NODE; tv["lo"][i] = rank[lex]
        elif kind == "bnode":
            tv["tag"][i] = abi.TV_BLANK_NODE; tv["lo"][i] = rank[lex]
        elif extra is None or extra.startswith("@"):
            tv["tag"][i] = abi.TV_STRING; tv["lo"][i] = rank[lex]
            if extra:
                if extra[1:] not in languages:
                    languages.append(extra[1:])
                tv["aux"][i] = languages.index(extra[1:])
            tv["flags"][i] = abi.TVF_EMPTY_STRING if not b else 0
        elif extra in ("xsd:int", "xsd:integer"):
            tv["tag"][i] = abi.TV_INT if extra == "xsd:int" else abi.TV_INTEGER; tv["lo"][i] = int(lex)
        elif extra == "xsd:boolean":
            tv["tag"][i] = abi.TV_BOOLEAN; tv["lo"][i] = 1 if lex in ("true", "1") else 0
        else:
            tv["tag"][i] = abi.TV_OTHER; tv["lo"][i] = rank[lex]
    return tv, offsets, bytes(heap), languages
