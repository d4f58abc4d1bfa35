"""Host-side plan description, named after the reference's types.

This is the Python mirror of what the Rust host hands over the C ABI: scan instructions
(`MemIndexScanInstruction`, lib/storage/src/memory/storage/scan_instructions.rs:247-318),
expression programs (the `PhysicalExpr` trees of FilterExec / JoinFilter built from the UDFs in
lib/extensions/src/functions/builtin.rs:101-186) and the operator tree
(DataSourceExec / FilterExec / HashJoinExec / CrossJoinExec / NestedLoopJoinExec as printed in
bench/tests/plans/snapshots/*Q5 (Execution Plan).snap).  It only builds ``rdfgpu_plan_desc``
structs; it computes nothing.
"""
import ctypes as C
import struct

from . import abi


# ----------------------------------------------------------------------------------------------
# scan instructions
# ----------------------------------------------------------------------------------------------
class MemIndexScanPredicate:
    """scan_instructions.rs:157-166"""

    def __init__(self, kind, ids=None, lo=0, hi=0, var=None):
        self.kind, self.ids, self.lo, self.hi, self.var = kind, ids, lo, hi, var

    @staticmethod
    def false():
        return MemIndexScanPredicate(abi.PRED_FALSE)

    @staticmethod
    def in_(ids):
        return MemIndexScanPredicate(abi.PRED_IN, ids=sorted(set(int(i) for i in ids)))

    @staticmethod
    def between(lo, hi):
        return MemIndexScanPredicate(abi.PRED_BETWEEN, lo=int(lo), hi=int(hi))

    @staticmethod
    def equal_to(var):
        return MemIndexScanPredicate(abi.PRED_EQUAL_TO, var=var)

    def __repr__(self):  # Display impl scan_instructions.rs:213-237
        if self.kind == abi.PRED_FALSE:
            return "false"
        if self.kind == abi.PRED_IN:
            return f"== {self.ids[0]}" if len(self.ids) == 1 else "in (" + ", ".join(map(str, self.ids)) + ")"
        if self.kind == abi.PRED_BETWEEN:
            return f"in ({self.lo}..{self.hi})"
        return f"== {self.var}"


class MemIndexScanInstruction:
    """scan_instructions.rs:247-318"""

    def __init__(self, kind, var=None, predicate=None):
        self.kind, self.var, self.predicate = kind, var, predicate

    @staticmethod
    def traverse(object_id=None):
        """`traverse(1)` in the reference tests = Traverse(Some(In{1}))."""
        if object_id is None:
            return MemIndexScanInstruction(abi.TRAVERSE)
        return MemIndexScanInstruction(abi.TRAVERSE, predicate=MemIndexScanPredicate.in_([object_id]))

    @staticmethod
    def traverse_with_predicate(predicate):
        return MemIndexScanInstruction(abi.TRAVERSE, predicate=predicate)

    @staticmethod
    def scan(var):
        return MemIndexScanInstruction(abi.SCAN, var=var)

    @staticmethod
    def scan_with_predicate(var, predicate):
        return MemIndexScanInstruction(abi.SCAN, var=var, predicate=predicate)


def quad_pattern(subject, predicate, obj, graph="default", graph_variable=None):
    """MemQuadStorageSnapshot::plan_pattern_evaluation (snapshot.rs:84-131): int = constant object
    id, str = variable.  `graph`: "default" | "all" | "named" | list of graph ids
    (MemIndexScanInstruction::from_active_graph, scan_instructions.rs:323-355)."""
    P, I = MemIndexScanPredicate, MemIndexScanInstruction
    if graph == "default":
        gp = P.in_([0])
    elif graph == "all":
        gp = None
    elif graph == "named":
        gp = P.between(1, 0xFFFFFFFF)
    else:
        gp = P.in_(list(graph))
    g = I(abi.SCAN, var=graph_variable, predicate=gp) if graph_variable else I(abi.TRAVERSE, predicate=gp)

    def term(t):
        return I.scan(t) if isinstance(t, str) else I.traverse(int(t))

    return [g, term(subject), term(predicate), term(obj)]


# ----------------------------------------------------------------------------------------------
# expressions (postfix programs)
# ----------------------------------------------------------------------------------------------
class Expr:
    def __init__(self, nodes):
        self.nodes = nodes  # list of (op, tag, flags, u, lo, hi)

    def _bin(self, other, op):
        return Expr(self.nodes + other.nodes + [(op, 0, 0, 0, 0, 0)])

    def _un(self, op):
        return Expr(self.nodes + [(op, 0, 0, 0, 0, 0)])


def col(i):
    return Expr([(abi.EX_COLUMN, 0, 0, int(i), 0, 0)])


def lit_id(object_id):
    return Expr([(abi.EX_LIT_ID, 0, 0, int(object_id), 0, 0)])


def lit_tv(tag, lo=0, aux=0, flags=0, hi=0):
    return Expr([(abi.EX_LIT_TV, int(tag), int(flags), int(aux), int(lo), int(hi))])


def lit_bool(v):
    return Expr([(abi.EX_LIT_BOOL, 0, 0, 2 if v is None else int(bool(v)), 0, 0)])


def integer(v):          # plan text "9:120"
    return lit_tv(abi.TV_INTEGER, int(v))


def int32(v):
    return lit_tv(abi.TV_INT, int(v))


def boolean(v):
    return lit_tv(abi.TV_BOOLEAN, int(bool(v)))


def double(v):
    return lit_tv(abi.TV_DOUBLE, struct.unpack("<q", struct.pack("<d", float(v)))[0])


def float32(v):
    return lit_tv(abi.TV_FLOAT, struct.unpack("<I", struct.pack("<f", float(v)))[0])


def decimal(scaled_i128):
    """scaled_i128 = value * 10**18 (decimal.rs:9-21)."""
    v = int(scaled_i128) & ((1 << 128) - 1)
    lo, hi = v & ((1 << 64) - 1), v >> 64
    to_i64 = lambda x: x - (1 << 64) if x >= (1 << 63) else x
    return lit_tv(abi.TV_DECIMAL, to_i64(lo), hi=to_i64(hi))


def _timestamp(tag, scaled_i128, has_timezone):
    v = int(scaled_i128) & ((1 << 128) - 1)
    lo, hi = v & ((1 << 64) - 1), v >> 64
    to_i64 = lambda x: x - (1 << 64) if x >= (1 << 63) else x
    return lit_tv(tag, to_i64(lo), aux=1 if has_timezone else 0, hi=to_i64(hi))


def date_time(scaled_i128, has_timezone):
    """xsd:dateTime literal: Timestamp value * 10**18 (rdf_fusion_amd.xsd.parse_date_time) + timezone presence."""
    return _timestamp(abi.TV_DATE_TIME, scaled_i128, has_timezone)


def date(scaled_i128, has_timezone):
    return _timestamp(abi.TV_DATE, scaled_i128, has_timezone)


def time(scaled_i128, has_timezone):
    return _timestamp(abi.TV_TIME, scaled_i128, has_timezone)


def ENC_TV(e): return e._un(abi.EX_ENC_TV)
def GT(a, b): return a._bin(b, abi.EX_GT)
def LT(a, b): return a._bin(b, abi.EX_LT)
def GEQ(a, b): return a._bin(b, abi.EX_GEQ)
def LEQ(a, b): return a._bin(b, abi.EX_LEQ)
def EQ(a, b): return a._bin(b, abi.EX_EQ)
def NEQ(a, b): return a._bin(b, abi.EX_NEQ)
def ADD(a, b): return a._bin(b, abi.EX_ADD)
def SUB(a, b): return a._bin(b, abi.EX_SUB)
def EBV(e): return e._un(abi.EX_EBV)


def REGEX(e, pattern, flags=""):
    """REGEX(value, "pattern"[, "flags"]) with constant pattern / flags (scalar/strings/regex.rs:47-141)."""
    enc = lambda x: x.encode("utf-8") if isinstance(x, str) else bytes(x)
    return Expr(e.nodes + [(abi.EX_REGEX, 0, 0, (enc(pattern), enc(flags), abi.EX_REGEX), 0, 0)])


def REGEX_VAR(e, pattern_expr, patterns, flags=""):
    """REGEX(value, ?pattern[, "flags"]) with a per-row pattern (regex.rs:59-76): `pattern_expr` = ENC_TV of the pattern column,
    `patterns` = {object id: pattern text} of every distinct pattern literal that can occur (the host's dictionary knows them)."""
    enc = lambda x: x.encode("utf-8") if isinstance(x, str) else bytes(x)
    table = tuple((int(pid), enc(txt)) for pid, txt in sorted(patterns.items()))
    return Expr(e.nodes + pattern_expr.nodes + [(abi.EX_REGEX_VAR, 0, 0, ("var", table, enc(flags)), 0, 0)])


def _string_fn(op):
    def f(e, needle, language_id=0):
        """<fn>(value, "constant"[@lang]) — contains.rs / str_starts.rs / str_ends.rs; language_id = the constant's
        language id in the typed-value table's `aux` numbering (0 = simple literal)."""
        b = needle.encode("utf-8") if isinstance(needle, str) else bytes(needle)
        return Expr(e.nodes + [(op, 0, 0, (b, b"", op), int(language_id), 0)])
    return f


CONTAINS, STRSTARTS, STRENDS = _string_fn(abi.EX_CONTAINS), _string_fn(abi.EX_STRSTARTS), _string_fn(abi.EX_STRENDS)


# String-valued expressions (include/rdfgpu.h, ABI 3).  STR takes the column itself (an object id): over an object-id column
# the reference evaluates STR in the plain-term encoding, i.e. on the lexical form as written (scalar/terms/str.rs:42).
def STR(e):
    return e._un(abi.EX_STR)


def lit_str(text, language_id=0):
    """A string constant with its bytes (what a computed string is compared with); language_id as in the typed-value table."""
    b = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    return Expr([(abi.EX_LIT_STR, 0, 0, (b, b"", abi.EX_LIT_STR), int(language_id), 0)])


def STRLEN(e):
    return e._un(abi.EX_STRLEN)


def SUBSTR(e, start, length=None):
    """SUBSTR(str, start[, length]) — scalar/strings/sub_str.rs; start / length are expressions (e.g. integer(2))."""
    nodes = e.nodes + start.nodes + (length.nodes if length is not None else [])
    return Expr(nodes + [(abi.EX_SUBSTR, 0, 0, 3 if length is not None else 2, 0, 0)])


def UCASE(e):
    return e._un(abi.EX_UCASE)


def LCASE(e):
    return e._un(abi.EX_LCASE)


def STRBEFORE(a, b):
    """STRBEFORE(a, b) — scalar/strings/str_before.rs: the part of a before b's first occurrence (a's language), "" if there is none."""
    return Expr(a.nodes + b.nodes + [(abi.EX_STRBEFORE, 0, 0, 0, 0, 0)])


def STRAFTER(a, b):
    """STRAFTER(a, b) — scalar/strings/str_after.rs."""
    return Expr(a.nodes + b.nodes + [(abi.EX_STRAFTER, 0, 0, 0, 0, 0)])


def lang_matches(language_tag, language_range):
    """LANGMATCHES on two plain strings, as scalar/strings/lang_matches.rs:52-69 evaluates it: "*" matches every
    non-empty tag; otherwise the subtags ('-' separated) are zipped with the longer side padded: a range subtag that is
    missing from, or differs (ASCII case-insensitively) from, the tag's subtag at that position is a mismatch."""
    if language_range == "*":
        return language_tag != ""
    r, l = language_range.split("-"), language_tag.split("-")
    ascii_lower = lambda x: "".join(chr(ord(c) + 32) if "A" <= c <= "Z" else c for c in x)
    for i in range(max(len(r), len(l))):
        if i < len(r) and (i >= len(l) or ascii_lower(r[i]) != ascii_lower(l[i])):
            return False
    return True


def LANGMATCHES_LANG(e, language_range, language_tags):
    """LANGMATCHES(LANG(value), "range") — BSBM explore Q8's FILTER.  `language_tags[i]` = the tag of language id i in the
    typed-value table's `aux` numbering (language_tags[0] must be "": literals without a language).  The host resolves
    the range against that (small) dictionary; the device reads one verdict byte per language id."""
    if not language_tags or language_tags[0] != "":
        raise ValueError("language_tags[0] is the empty tag")
    verdicts = bytes(1 if lang_matches(t, language_range) else 0 for t in language_tags)
    return Expr(e.nodes + [(abi.EX_LANG_IN, 0, 0, (verdicts, b"", abi.EX_LANG_IN), 0, 0)])
def ID_EQ(a, b): return a._bin(b, abi.EX_ID_EQ)
def ID_NEQ(a, b): return a._bin(b, abi.EX_ID_NEQ)
def AND(a, b): return a._bin(b, abi.EX_AND)
def OR(a, b): return a._bin(b, abi.EX_OR)
def NOT(e): return e._un(abi.EX_NOT)
def IS_COMPATIBLE(a, b): return a._bin(b, abi.EX_IS_COMPATIBLE)
def BOUND(e): return e._un(abi.EX_BOUND)
def BOOLEAN_AS_TERM(e): return e._un(abi.EX_BOOL_AS_TV)


# ----------------------------------------------------------------------------------------------
# operator tree
# ----------------------------------------------------------------------------------------------
class PlanDescription:
    """Owns the ctypes arrays behind one ``rdfgpu_plan_desc``."""

    def __init__(self, nodes, exprs, pool, root, n_columns, regexes=()):
        self.n_columns = n_columns  # output width per node (host-side bookkeeping)
        self._nodes = (abi.PlanNode * max(1, len(nodes)))(*nodes)
        self._exprs = (abi.ExprNode * max(1, len(exprs)))(*exprs)
        self._pool = (C.c_uint32 * max(1, len(pool)))(*pool)
        self._regex_bytes = [(bytes(r[0]), bytes(r[1]), int(r[2]) if len(r) > 2 else 0) for r in regexes]      # keeps the char buffers alive
        self._regexes = (abi.Regex * max(1, len(regexes)))(*[abi.Regex(p, f, len(p), len(f), pid, 0) for p, f, pid in self._regex_bytes])
        self.desc = abi.PlanDesc(self._nodes, len(nodes), root, self._exprs, len(exprs), self._pool,
                                 len(pool), 0, self._regexes, len(regexes), 0)
        self.root = root

    @property
    def width(self):
        return self.n_columns[self.root]


class PlanBuilder:
    def __init__(self):
        self.nodes, self.exprs, self.pool, self.width = [], [], [], []
        self.names = []          # per node: the names of its output columns (variables), for plan display
        self.patterns = []       # per node: the quad pattern of a DataSourceExec (four MemIndexScanInstructions) or None
        self.regexes = []
        self._regex_keys = []
        self.vars = {}

    # -- helpers -------------------------------------------------------------------------------
    def _var(self, name):
        return self.vars.setdefault(name, len(self.vars))

    def _instr(self, ins):
        out = abi.ScanInstruction()
        out.kind = ins.kind
        out.var = self._var(ins.var) if ins.kind == abi.SCAN else 0
        p = ins.predicate
        if p is None:
            out.pred = abi.PRED_NONE
        elif p.kind == abi.PRED_IN:
            out.pred, out.a, out.b = abi.PRED_IN, len(self.pool), len(p.ids)
            self.pool.extend(p.ids)
        elif p.kind == abi.PRED_BETWEEN:
            out.pred, out.a, out.b = abi.PRED_BETWEEN, p.lo, p.hi
        elif p.kind == abi.PRED_EQUAL_TO:
            out.pred, out.a = abi.PRED_EQUAL_TO, self._var(p.var)
        else:
            out.pred = abi.PRED_FALSE
        return out

    def _expr(self, node, e):
        if e is None:
            node.expr_off, node.expr_len = 0, 0
            return
        node.expr_off, node.expr_len = len(self.exprs), len(e.nodes)
        for (op, tag, flags, u, lo, hi) in e.nodes:
            if op == abi.EX_REGEX_VAR:
                # one regex entry per announced pattern, contiguous: u = first entry, lo = count
                _, table, fl = u
                u, lo = len(self.regexes), len(table)
                for pid, txt in table:
                    self._regex_keys.append(("var", pid, txt, fl))
                    self.regexes.append((txt, fl, pid))
            elif op in (abi.EX_REGEX, abi.EX_CONTAINS, abi.EX_STRSTARTS, abi.EX_STRENDS, abi.EX_LANG_IN, abi.EX_LIT_STR):
                # u carries (pattern, flags, op): register the plan constant (one entry per function), keep its index
                if u not in self._regex_keys:
                    self._regex_keys.append(u)
                    self.regexes.append(u[:2])
                u = self._regex_keys.index(u)
            self.exprs.append(abi.ExprNode(op, tag, flags, 0, u, lo, hi))

    def _proj(self, node, projection, full):
        if projection is None:
            node.proj_off, node.n_proj = 0, abi.NO_PROJECTION
            return full
        node.proj_off, node.n_proj = len(self.pool), len(projection)
        self.pool.extend(int(c) for c in projection)
        return len(projection)

    def _push(self, node, width, names=None, pattern=None):
        self.nodes.append(node)
        self.width.append(width)
        self.names.append(list(names) if names is not None else [f"c{i}" for i in range(width)])
        self.patterns.append(pattern)
        assert len(self.names[-1]) == width
        return len(self.nodes) - 1

    def _projected(self, full_names, projection):
        return list(full_names) if projection is None else [full_names[int(c)] for c in projection]

    # -- operators -----------------------------------------------------------------------------
    def data_source(self, instructions):
        """DataSourceExec(MemQuadPatternDataSource): four instructions in G,S,P,O order."""
        n = abi.PlanNode(kind=abi.NODE_DATA_SOURCE, left=-1, right=-1)
        seen, width = [], 0
        for i, ins in enumerate(instructions):
            n.scan[i] = self._instr(ins)
            if ins.kind == abi.SCAN and ins.var not in seen:
                seen.append(ins.var)
                width += 1
        n.n_proj = abi.NO_PROJECTION
        return self._push(n, width, seen, list(instructions))

    def filter(self, child, predicate, projection=None):
        n = abi.PlanNode(kind=abi.NODE_FILTER, left=child, right=-1)
        self._expr(n, predicate)
        return self._push(n, self._proj(n, projection, self.width[child]), self._projected(self.names[child], projection))

    def projection(self, child, columns):
        n = abi.PlanNode(kind=abi.NODE_PROJECTION, left=child, right=-1)
        return self._push(n, self._proj(n, columns, self.width[child]), self._projected(self.names[child], columns))

    def hash_join(self, left, right, on, join_type=abi.JOIN_INNER, filter=None, projection=None):
        n = abi.PlanNode(kind=abi.NODE_HASH_JOIN, left=left, right=right, join_type=join_type)
        n.n_keys = len(on)
        for k, (l, r) in enumerate(on):
            n.left_keys[k], n.right_keys[k] = l, r
        self._expr(n, filter)
        return self._push(n, self._proj(n, projection, self.width[left] + self.width[right]),
                          self._projected(self.names[left] + self.names[right], projection))

    def cross_join(self, left, right):
        n = abi.PlanNode(kind=abi.NODE_CROSS_JOIN, left=left, right=right, join_type=abi.JOIN_INNER)
        n.n_proj = abi.NO_PROJECTION
        return self._push(n, self.width[left] + self.width[right], self.names[left] + self.names[right])

    def nested_loop_join(self, left, right, join_type=abi.JOIN_INNER, filter=None, projection=None):
        n = abi.PlanNode(kind=abi.NODE_NESTED_LOOP_JOIN, left=left, right=right, join_type=join_type)
        self._expr(n, filter)
        return self._push(n, self._proj(n, projection, self.width[left] + self.width[right]),
                          self._projected(self.names[left] + self.names[right], projection))

    def table(self, slot, n_cols, names=None):
        n = abi.PlanNode(kind=abi.NODE_TABLE, left=-1, right=-1, table_slot=slot, table_cols=n_cols)
        n.n_proj = abi.NO_PROJECTION
        return self._push(n, n_cols, names)

    def topk(self, left, keys, limit, group=None, projection=None, tie_break=True):
        """DISTINCT + ORDER BY keys ASC LIMIT `limit` (per `group` column if given) — the AggregateExec(first_value) +
        SortExec TopK(fetch) pair above the path in the reference's explore plans.  keys = [(column, abi.SORT_BY_ID |
        abi.SORT_BY_TERM | abi.SORT_BY_DOUBLE), ...].  The AggregateExec groups by the sort expressions AND the raw output
        columns, so (`tie_break`) every output column that is not yet a key by id is appended as one: at most 4 keys."""
        keys = [(int(c), int(how)) for c, how in keys]
        out_cols = list(range(self.width[left])) if projection is None else [int(c) for c in projection]
        if tie_break:
            for c in out_cols:
                if (group is None or c != int(group)) and (c, abi.SORT_BY_ID) not in keys:
                    keys.append((c, abi.SORT_BY_ID))
        if len(keys) > 4:
            raise ValueError("TopK: more than 4 sort keys (ORDER BY keys + output columns)")
        n = abi.PlanNode(kind=abi.NODE_TOPK, left=left, right=-1, n_keys=len(keys), table_cols=int(limit),
                         table_slot=0 if group is None else int(group) + 1)
        for i, (c, how) in enumerate(keys):
            n.left_keys[i], n.right_keys[i] = c, how
        return self._push(n, self._proj(n, projection, self.width[left]), self._projected(self.names[left], projection))

    def union(self, left, right, projection=None):
        """UnionExec: bag union of two inputs with the same columns (SPARQL UNION; Q4 (Execution Plan).snap:11)."""
        if self.width[left] != self.width[right]:
            raise ValueError("UNION inputs differ in width")
        n = abi.PlanNode(kind=abi.NODE_UNION, left=left, right=right)
        return self._push(n, self._proj(n, projection, self.width[left]), self._projected(self.names[left], projection))

    def closure(self, left, allow_cross_graph_paths=False):
        """KleenePlusClosureExec over inner paths (graph, start, end) — the `+` of a SPARQL property path."""
        if self.width[left] != 3:
            raise ValueError("inner paths are (graph, start, end)")
        n = abi.PlanNode(kind=abi.NODE_CLOSURE, left=left, right=-1, join_type=1 if allow_cross_graph_paths else 0)
        return self._push(n, self._proj(n, None, 3), self.names[left])

    def build(self, root):
        return PlanDescription(self.nodes, self.exprs, self.pool, root, list(self.width), list(self.regexes))


    # -- SparqlJoinNode lowering ------------------------------------------------------------------
    def sparql_join(self, left, right, join_type=abi.JOIN_INNER, filter=None):
        """SparqlJoinLoweringRule (lib/logical/src/join/rewrite.rs:71-221) for inputs whose shared variables are
        non-nullable (object-id columns of quad patterns): no shared variable => CrossJoinExec (inner, :74-96) or a
        left join without keys and without a filter (what DataFusion plans as NestedLoopJoinExec, ..Q2 (Execution
        Plan).snap:6-8); otherwise an equi-join on ALL shared variables (:126-168, NullEqualsNothing :89) whose output
        is the left fields followed by the right fields not already present (join/logical.rs:251-279).  Nullable shared
        variables (IS_COMPATIBLE + COALESCE, :183-221, :327-345) are outside this path."""
        ln, rn = self.names[left], self.names[right]
        shared = [v for v in ln if v in rn]
        if not shared:
            if join_type == abi.JOIN_INNER:
                if filter is not None:
                    return self.nested_loop_join(left, right, abi.JOIN_INNER, filter=filter)
                return self.cross_join(left, right)
            return self.nested_loop_join(left, right, abi.JOIN_LEFT, filter=filter)
        on = [(ln.index(v), rn.index(v)) for v in shared]
        keep = list(range(len(ln))) + [len(ln) + i for i, v in enumerate(rn) if v not in ln]
        return self.hash_join(left, right, on, join_type=join_type, filter=filter, projection=keep)


# ----------------------------------------------------------------------------------------------
# plan display, in the reference's format (bench/tests/plans/snapshots/*.snap; DataSourceExec lines as
# MemQuadPatternDataSource::fmt_as prints them, pattern_data_source.rs:80-105)
# ----------------------------------------------------------------------------------------------
_BIN = {abi.EX_GT: "GT", abi.EX_LT: "LT", abi.EX_GEQ: "GEQ", abi.EX_LEQ: "LEQ", abi.EX_EQ: "EQ", abi.EX_ADD: "ADD", abi.EX_SUB: "SUB",
        abi.EX_IS_COMPATIBLE: "IS_COMPATIBLE"}
_UN = {abi.EX_ENC_TV: "ENC_TV", abi.EX_EBV: "EBV", abi.EX_BOUND: "BOUND", abi.EX_BOOL_AS_TV: "BOOLEAN_AS_TERM"}


def format_expr(nodes, names):
    """A postfix program as DataFusion prints the PhysicalExpr tree (`EBV(GT(ENC_TV(value1@1), 9:136))`,
    `product@0 != <object id>`, `.. AND ..`); object-id literals print as the snapshots mask them (<c>)."""
    st = []
    for n in nodes:
        op, tag, u, lo = n.op, n.tag, n.u, n.lo
        if op == abi.EX_COLUMN:
            st.append(f"{names[u]}@{u}")
        elif op == abi.EX_LIT_ID:
            st.append("<c>")
        elif op == abi.EX_LIT_TV:
            st.append(f"{tag}:{lo}")
        elif op == abi.EX_LIT_BOOL:
            st.append({0: "false", 1: "true", 2: "NULL"}[u])
        elif op in _UN:
            st.append(f"{_UN[op]}({st.pop()})")
        elif op in _BIN:
            b, a = st.pop(), st.pop()
            st.append(f"{_BIN[op]}({a}, {b})")
        elif op in (abi.EX_ID_EQ, abi.EX_ID_NEQ, abi.EX_AND, abi.EX_OR):
            b, a = st.pop(), st.pop()
            st.append(f"{a} {({abi.EX_ID_EQ: '=', abi.EX_ID_NEQ: '!=', abi.EX_AND: 'AND', abi.EX_OR: 'OR'})[op]} {b}")
        elif op == abi.EX_NOT:
            st.append(f"NOT {st.pop()}")
        else:
            st.append(f"op{op}({st.pop()})")
    assert len(st) == 1
    return st[0]


def explain(pb, root, choose_index=None):
    """The operator tree under `root`, one line per operator, indented like DataFusion's `displayable(plan).indent()`.
    `choose_index(instructions) -> abi.GSPO | GPOS | GOSP` names the index of a DataSourceExec (the library's
    rdfgpu_choose_index; default: engine.choose_index)."""
    if choose_index is None:
        from .engine import choose_index as _ci
        choose_index = lambda ins: _ci([PlanBuilder()._instr(i) for i in ins])
    lines = []

    def proj(i, full):
        n = pb.nodes[i]
        if n.n_proj == abi.NO_PROJECTION:
            return ""
        cols = [pb.pool[n.proj_off + q] for q in range(n.n_proj)]
        return ", projection=[" + ", ".join(f"{full[c]}@{c}" for c in cols) + "]"

    def term(ins):
        if ins.kind == abi.SCAN:
            return "?" + ins.var
        return "<c>"

    def walk(i, depth):
        n = pb.nodes[i]
        pad = "  " * depth
        if n.kind == abi.NODE_DATA_SOURCE:
            g, s_, p_, o_ = pb.patterns[i]
            idx = abi.INDEX_NAMES[choose_index(pb.patterns[i])]
            graph = f"graph=?{g.var}, " if g.kind == abi.SCAN else ""
            lines.append(f"{pad}DataSourceExec: [{idx}] {graph}subject={term(s_)}, predicate={term(p_)}, object={term(o_)}")
            return
        if n.kind == abi.NODE_FILTER:
            full = pb.names[n.left]
            e = format_expr(pb.exprs[n.expr_off:n.expr_off + n.expr_len], full)
            lines.append(f"{pad}FilterExec: {e}{proj(i, full)}")
            walk(n.left, depth + 1)
            return
        if n.kind in (abi.NODE_HASH_JOIN, abi.NODE_NESTED_LOOP_JOIN):
            ln, rn = pb.names[n.left], pb.names[n.right]
            jt = "Inner" if n.join_type == abi.JOIN_INNER else "Left"
            head = "HashJoinExec: mode=CollectLeft" if n.kind == abi.NODE_HASH_JOIN else "NestedLoopJoinExec:"
            sep = ", " if n.kind == abi.NODE_HASH_JOIN else " "
            text = f"{head}{sep}join_type={jt}"
            if n.kind == abi.NODE_HASH_JOIN:
                text += ", on=[" + ", ".join(f"({ln[n.left_keys[k]]}@{n.left_keys[k]}, {rn[n.right_keys[k]]}@{n.right_keys[k]})" for k in range(n.n_keys)) + "]"
            if n.expr_len:
                # DataFusion prints a JoinFilter over ITS OWN intermediate schema: only the referenced columns, numbered in
                # order of first use, left side first (`simProperty2@1 .. origProperty2@0`)
                ex = pb.exprs[n.expr_off:n.expr_off + n.expr_len]
                used = sorted({e.u for e in ex if e.op == abi.EX_COLUMN})
                renum = {c: k for k, c in enumerate(used)}
                both = ln + rn
                fake = [abi.ExprNode(e.op, e.tag, e.flags, 0, renum[e.u] if e.op == abi.EX_COLUMN else e.u, e.lo, e.hi) for e in ex]
                text += ", filter=" + format_expr(fake, [both[c] for c in used])
            lines.append(f"{pad}{text}{proj(i, ln + rn)}")
            walk(n.left, depth + 1); walk(n.right, depth + 1)
            return
        if n.kind == abi.NODE_CROSS_JOIN:
            lines.append(f"{pad}CrossJoinExec")
            walk(n.left, depth + 1); walk(n.right, depth + 1)
            return
        if n.kind == abi.NODE_TABLE:
            lines.append(f"{pad}BoundTableExec: slot={n.table_slot}, columns=[{', '.join(pb.names[i])}]")
            return
        name = {abi.NODE_PROJECTION: "ProjectionExec", abi.NODE_TOPK: "SortExec: TopK", abi.NODE_UNION: "UnionExec",
                abi.NODE_CLOSURE: "KleenePlusClosureExec"}[n.kind]
        lines.append(f"{pad}{name}")
        walk(n.left, depth + 1)
        if n.kind == abi.NODE_UNION:
            walk(n.right, depth + 1)

    walk(root, 0)
    return lines


def explain_logical_join(pb, node):
    """The LOGICAL form of a join node produced by PlanBuilder.sparql_join, as DataFusion prints it in the reference's
    join-lowering tests (lib/logical/src/join/rewrite.rs:411-437: `Cross Join: ` / `Left Join: `)."""
    n = pb.nodes[node]
    if n.kind == abi.NODE_CROSS_JOIN:
        return "Cross Join: "
    jt = "Inner" if n.join_type == abi.JOIN_INNER else "Left"
    on = ", ".join(f"{pb.names[n.left][n.left_keys[k]]} = {pb.names[n.right][n.right_keys[k]]}" for k in range(n.n_keys)) if n.kind == abi.NODE_HASH_JOIN else ""
    return f"{jt} Join: {on}"
