#!/usr/bin/env python3
"""Writes tests/golden/reference_kats.json: the literal known-answer vectors of the reference's own
unit tests for the scan path, transcribed by hand as DATA (inputs + expected outputs; no source).

Sources (relative to the rdf-fusion tree):
  S  = lib/storage/src/memory/storage/mod.rs            (scan KATs, predicate algebra, index choice)
  D  = lib/storage/src/memory/storage/quad_index_data.rs (row-group build / prune / find_range)
  C  = lib/storage/src/memory/storage/scan.rs            (dynamic-filter scan tests)
  P  = lib/storage/src/memory/storage/predicate_pushdown.rs + pattern_data_source.rs (push-down)
  Q  = lib/storage/src/memory/storage/quad_index.rs      (scan-score ordering)

Instruction notation (index order unless a case says "gspo"):
  ["T"]                traverse, no predicate          ["T", 7]            traverse(In{7})
  ["T", {"in":[..]}]   traverse with In                ["T", {"between":[a,b]}]
  ["S", "x"]           scan binding variable x         ["S", "x", {"in":[..]}] / {"between":[a,b]}
"""
import json
import os

T = lambda *a: ["T", *a]
S = lambda *a: ["S", *a]
q4 = lambda v: [v, v, v, v]

D3 = [[1, 2, 3, 4], [1, 2, 5, 6], [1, 7, 3, 4]]

scan_cases = [
    # name, S line, batch size, quads (GSPO index order), instructions, expected
    dict(name="insert_and_scan_triple", src="S:36", batch=10, quads=[[1, 2, 3, 4]],
         instr=[T(1), T(2), T(3), T(4)], n_rows=1, columns={}, first_batch_rows=1),
    dict(name="scan_returns_sorted_results_on_last_level", src="S:52", batch=10,
         quads=[[1, 2, 3, 4], [1, 2, 3, 3]], instr=[T(1), T(2), T(3), S("d")], columns={"d": [3, 4]}),
    dict(name="scan_returns_sorted_results_on_intermediate_level", src="S:79", batch=10,
         quads=[[1, 2, 3, 4], [1, 2, 2, 4]], instr=[T(1), T(2), S("c"), T(4)], columns={"c": [2, 3]}),
    dict(name="scan_with_no_match", src="S:106", batch=10, quads=[[1, 2, 3, 4]],
         instr=[T(2), S("b"), T(3), T(4)], n_rows=0, batches=[]),
    dict(name="scan_subject_var", src="S:123", batch=10, quads=D3, instr=[T(1), S("b"), T(3), T(4)],
         n_cols=1, n_rows=2),
    dict(name="scan_predicate_var", src="S:145", batch=10, quads=D3, instr=[T(1), T(2), S("c"), T(4)],
         n_cols=1, n_rows=1),
    dict(name="scan_object_var", src="S:167", batch=10, quads=D3, instr=[T(1), T(2), T(3), S("d")],
         n_cols=1, n_rows=1),
    dict(name="scan_multi_vars", src="S:189", batch=10, quads=D3, instr=[T(1), S("b"), T(3), S("d")],
         n_cols=2, n_rows=2),
    dict(name="scan_all_vars", src="S:211", batch=10, quads=D3, instr=[S("a"), S("b"), S("c"), S("d")],
         n_cols=4, n_rows=3),
    dict(name="scan_same_var_appearing_twice", src="S:233", batch=10,
         quads=[[1, 3, 3, 4], [1, 2, 2, 4], [1, 3, 2, 4]],
         instr=[S("a"), S("same"), S("same"), S("d")], n_cols=3, n_rows=2),
    dict(name="scan_considers_predicates", src="S:255", batch=10,
         quads=[[1, 2, 3, 4], [2, 2, 5, 6], [3, 7, 3, 4]],
         instr=[S("a", {"in": [1, 3]}), S("b"), S("c"), S("d")], n_cols=4, n_rows=2),
    dict(name="scan_batches_for_batch_size", src="S:280", batch=10,
         quads=[[1, 2, 3, i + 1] for i in range(25)], instr=[T(1), T(2), T(3), S("d")],
         batches=[10, 10, 5]),
    dict(name="scan_multi_level_batches_coalesce_results", src="S:303", batch=10,
         quads=[[1, 2, i, 3] for i in range(25)], instr=[T(1), T(2), S("c"), T(3)],
         batches=[10, 10, 5]),
    dict(name="test_dynamic_filters", src="C:617", batch=100,
         quads=[[0, 1, 10, 100], [0, 2, 10, 100], [0, 2, 10, 200], [0, 3, 10, 100]],
         # static Between(2,3) AND dynamic Between(1,2) on subject => Between(2,2)
         instr=[T(0), S("subject", {"between": [2, 2]}), T(), T(200)], n_rows=1,
         note="the dynamic filter Between(subject,1,2) is AND-ed into Between(2,3) first (scan.rs:241-261)"),
]

remove_cases = [
    dict(name="delete_triple_removes_it", src="S:327", insert=D3, remove=D3, remaining=0, removed=3),
    dict(name="delete_triple_non_existing_returns_zero", src="S:349", insert=[], remove=[[1, 2, 3, 4]],
         remaining=0, removed=0),
]

# store-level (three permutations; instructions in G,S,P,O order)
store_cases = [
    dict(name="scan_gpos_subject_and_object", src="S:486", quads_gspo=[[1, 2, 3, 4]],
         instr=[T(1), S("subject"), T(3), S("object")], chosen="GPOS",
         columns={"subject": [2], "object": [4]}, order=["subject", "object"]),
    dict(name="insert_quad_then_read", src="lib/storage/tests/memory/mem_quad_storage.rs:29",
         quads_gspo=[[0, 1, 2, 3]], instr=[S("g", {"in": [0]}), S("s"), S("p"), S("o")],
         columns={"g": [0], "s": [1], "p": [2], "o": [3]}, order=["g", "s", "p", "o"],
         note="g = 0 is exported as an Arrow null; schema g nullable, s/p/o non-null"),
]

predicate_and_cases = [  # S:357-436
    dict(lhs={"in": [1, 2, 3]}, rhs="false", out="false"),
    dict(lhs="false", rhs={"between": [1, 2]}, out="false"),
    dict(lhs={"in": [1, 2, 3]}, rhs={"in": [2, 3, 4]}, out={"in": [2, 3]}),
    dict(lhs={"in": [1]}, rhs={"in": [2]}, out="false"),
    dict(lhs={"in": [1, 2, 3]}, rhs={"between": [2, 3]}, out={"in": [2, 3]}),
    dict(lhs={"between": [2, 4]}, rhs={"in": [3, 4, 5]}, out={"in": [3, 4]}),
    dict(lhs={"between": [2, 5]}, rhs={"between": [3, 4]}, out={"between": [3, 4]}),
    dict(lhs={"between": [1, 2]}, rhs={"between": [3, 4]}, out="false"),
    dict(lhs={"in": [1]}, rhs={"equal_to": "x"}, out=None),
]

index_choice_cases = [  # S:439-483 ; store keeps GSPO, GPOS, GOSP ; instructions in G,S,P,O order
    dict(instr=[T(0), T(1), T(2), T(3)], chosen="GSPO"),
    dict(instr=[T(0), T(1), S("predicate"), T(3)], chosen="GOSP"),
    dict(instr=[T(0), S("subject"), T(2), S("object")], chosen="GPOS"),
    # C:674 test_collect_relevant_batches_dynamic_filters_choose_better_index (GSPO + GOSP only)
    dict(instr=[T(0), S("subject"), S("predicate"), S("object", {"between": [1, 200]})],
         available=["GSPO", "GOSP"], chosen="GOSP"),
]

score_order_cases = [  # Q:205-315: instructions in index order
    dict(name="test_in_predicate_better_than_nothing",
         greater=[S("g", {"in": [10]}), T(), T(), T()], lesser=[T(), T(), T(), T()]),
    dict(name="test_in_predicate_following_none_equal_to_nothing",
         equal=[T(), S("g", {"in": [10]}), T(), T()], to=[T(), T(), T(), T()]),
    dict(name="test_in_predicate_better_than_between",
         greater=[S("g", {"in": [10]}), S("s", {"in": [10]}), S("p"), S("o")],
         lesser=[S("g", {"equal_to": "x"}), S("s", {"between": [1, 10]}), S("p"), S("o")]),
]

pushdown_cases = [  # P: predicate_pushdown.rs:324-507
    dict(op="Eq", value=123, out={"in": [123]}),
    dict(op="Gt", value=100, out={"between": [101, 4294967295]}),
    dict(op="GtEq", value=100, out={"between": [100, 4294967295]}),
    dict(op="Lt", value=100, out={"between": [0, 99]}),
    dict(op="LtEq", value=100, out={"between": [0, 100]}),
    dict(op="Gt", value=4294967295, out="false"),
    dict(op="Lt", value=0, out="false"),
]
pushdown_display_cases = [  # pattern_data_source.rs:192-234 (display of the combined scan predicate)
    dict(filters=[["Eq", 1]], display="== 1"),
    dict(filters=[["Gt", 1]], display="in (2..4294967295)"),
    dict(filters=[["Gt", 1], ["Lt", 10]], display="in (2..9)"),
]

rowgroup_cases = [  # D:703-757
    dict(src="D:703", size=4, values=[1, 2, 3], groups=[3]),
    dict(src="D:714", size=2, values=[10, 20, 30, 40, 50], groups=[2, 2, 1]),
    dict(src="D:737", size=3, values=[11, 12, 13, 14, 15, 16], groups=[3, 3]),
]
dedupe_cases = [dict(src="D:748", size=3, first=[1, 2, 3], second=[2, 3, 4], length=4)]

prune_cases = [  # D:857-1200 ; instructions in index order (GSPO)
    dict(src="D:841", size=2, quads=[q4(v) for v in [1, 2, 3, 4]], instr=[T(), T(), T(), T()],
         group_lens=[2, 2], dropped=[]),
    dict(src="D:857", size=2, quads=[q4(v) for v in [10, 20, 30, 40]], instr=[T(30), T(), T(), T()],
         group_lens=[1], dropped=[0], rows=[[30, 30, 30, 30]]),
    dict(src="D:918", size=5,
         quads=[[10, 10, 10, v] for v in range(10, 19)] + [[20, 5, 5, 5], [20, 5, 5, 6], [20, 5, 5, 7]],
         instr=[T(10), T(10), T(), T()], group_lens=[5, 4], dropped=[0, 1]),
    dict(src="D:969", size=5,
         quads=[[10, 10, 10, 10], [10, 10, 10, 11], [10, 10, 10, 12], [10, 10, 10, 13], [11, 10, 10, 14],
                [11, 10, 10, 15]],
         instr=[T(10), T(), T(), T()], group_lens=[4], dropped=[0]),
    dict(src="D:1007", size=5,
         quads=[[0, 10, 10, 10]] + [[0, 11, 10, v] for v in [11, 12, 13, 14, 21, 22, 23, 24, 25, 31]] +
               [[0, 11, 12, 32]],
         instr=[T(0), T(11), T(10), T()], group_lens=[4, 5, 1]),
    dict(src="D:1054", size=5, quads=[[10, 10, 10, 10], [10, 10, 12, 11], [10, 11, 12, 12], [20, 20, 20, 20]],
         instr=[T(10), T(10), T(10), T()], group_lens=[1]),
    dict(src="D:1083", size=5, quads=[[10, 9, 9, 10], [10, 10, 9, 10], [10, 10, 10, 10], [20, 20, 20, 20]],
         instr=[T(10), T(10), T(10), T()], group_lens=[1]),
    dict(src="D:1112", size=2, quads=[q4(v) for v in [1, 2, 3, 4]], instr=[T(99), T(), T(), T()],
         group_lens=[]),
    dict(src="D:1131", size=2, quads=[q4(v) for v in [1, 2, 3, 4]],
         instr=[T({"between": [1, 2]}), T({"between": [1, 2]}), T(), T()], group_lens=[2], dropped=[0],
         kept=[1]),
    dict(src="D:1156", size=2, quads=[q4(v) for v in [1, 2, 3, 4]],
         instr=[T({"between": [1, 1]}), T({"between": [1, 1]}), T(), T()], n_groups=1, dropped=[0, 1]),
    dict(src="D:1182", size=2, quads=[q4(v) for v in [1, 2, 3, 4]], instr=[T({"in": [2, 3]}), T(), T(), T()],
         n_groups_is_all=True),
]

find_range_cases = [  # D:1202-1250 ; None = null
    dict(values=[None, None, None], value=0, result=["Contained", 0, 3]),
    dict(values=[None, None, 3, 5, 7], value=0, result=["Contained", 0, 2]),
    dict(values=[None, None, 3, 5, 7], value=5, result=["Contained", 3, 4]),
    dict(values=[2, 4, 6], value=4, result=["Contained", 1, 2]),
    dict(values=[4, 4, 4, 5], value=4, result=["Contained", 0, 3]),
    dict(values=[1, 3, 5, 7], value=4, result=["NotContained", 2]),
    dict(values=[10, 20, 30], value=2, result=["Before"]),
    dict(values=[10, 20, 30], value=50, result=["After"]),
]

# Checked arithmetic and casts of the xsd value types (inputs and expected results of the reference's in-file
# tests).  Sources: M = lib/model/src/xsd/.  Values: ["int", v] i32, ["integer", v] i64, ["decimal", raw] with
# raw = value * 10^18 as a decimal string (i128: decimal.rs:9-29, STEP = raw 1), ["double", v].
I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1
I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1
I128_MIN, I128_MAX = -(1 << 127), (1 << 127) - 1
dec = lambda raw: ["decimal", str(raw)]
numeric_arith_cases = [
    dict(src="M/integer.rs:420-423", op="add", a=["integer", I64_MIN], b=["integer", 1], expect=["integer", I64_MIN + 1]),
    dict(src="M/integer.rs:420-423", op="add", a=["integer", I64_MAX], b=["integer", 1], expect="error"),
    dict(src="M/integer.rs:426-429", op="sub", a=["integer", I64_MIN], b=["integer", 1], expect="error"),
    dict(src="M/integer.rs:426-429", op="sub", a=["integer", I64_MAX], b=["integer", 1], expect=["integer", I64_MAX - 1]),
    dict(src="M/int.rs:369-372", op="add", a=["int", I32_MIN], b=["int", 1], expect=["int", I32_MIN + 1]),
    dict(src="M/int.rs:369-372", op="add", a=["int", I32_MAX], b=["int", 1], expect="error"),
    dict(src="M/int.rs:375-378", op="sub", a=["int", I32_MIN], b=["int", 1], expect="error"),
    dict(src="M/int.rs:375-378", op="sub", a=["int", I32_MAX], b=["int", 1], expect=["int", I32_MAX - 1]),
    dict(src="M/decimal.rs:755-762", op="add", a=dec(I128_MIN), b=dec(1), expect=dec(I128_MIN + 1)),
    dict(src="M/decimal.rs:755-762", op="add", a=dec(I128_MAX), b=dec(1), expect="error"),
    dict(src="M/decimal.rs:755-762", op="add", a=dec(I128_MAX), b=dec(I128_MIN), expect=dec(-1)),
    dict(src="M/decimal.rs:765-768", op="sub", a=dec(I128_MIN), b=dec(1), expect="error"),
    dict(src="M/decimal.rs:765-768", op="sub", a=dec(I128_MAX), b=dec(1), expect=dec(I128_MAX - 1)),
]
E18 = 10 ** 18
decimal_to_double_cases = [  # M/decimal.rs:1097-1117 (Double::from(Decimal)); tol = the test's own bound, 0 = assert_eq
    dict(raw=str(0), value=0.0, tol=0.0),
    dict(raw=str(1 * E18), value=1.0, tol=0.0),
    dict(raw=str(10 * E18), value=10.0, tol=0.0),
    dict(raw=str(E18 // 10), value=0.1, tol=1.1920928955078125e-07),
    # the reference asserts |x - v| < 1; doubles are 32768 apart at 1.7e20, so that is x == v
    dict(raw=str(I128_MAX), value=1.7014118346046924e20, tol=0.0),
    dict(raw=str(I128_MIN), value=-1.7014118346046924e20, tol=0.0),
]

# Ordering of xsd values (PartialOrd) and their `=`: inputs and expected results of the reference's in-file tests.
# Values: ["double", v] / ["float", v] with v a float or "NaN" / "INF" / "-INF" / "MAX" / "MIN"; ordering "Less" | "Equal" |
# "Greater" | "None" (incomparable: GT / LT / EQ all yield the SPARQL error value).
compare_cases = []
for kind, src in (("double", "M/double.rs"), ("float", "M/float.rs")):
    compare_cases += [
        # fn eq(): assert_eq!(0, 0); assert_ne!(NAN, NAN); assert_eq!(-0., 0.)            (double.rs:283-288, float.rs eq)
        dict(src=src + " eq", a=[kind, 0.0], b=[kind, 0.0], ordering="Equal"),
        dict(src=src + " eq", a=[kind, "NaN"], b=[kind, "NaN"], ordering="None"),
        dict(src=src + " eq", a=[kind, -0.0], b=[kind, 0.0], ordering="Equal"),
        # fn cmp()                                                                          (double.rs:290-310, float.rs cmp)
        dict(src=src + " cmp", a=[kind, 0.0], b=[kind, 0.0], ordering="Equal"),
        dict(src=src + " cmp", a=[kind, "INF"], b=[kind, "MAX"], ordering="Greater"),
        dict(src=src + " cmp", a=[kind, "-INF"], b=[kind, "MIN"], ordering="Less"),
        dict(src=src + " cmp", a=[kind, "NaN"], b=[kind, 0.0], ordering="None"),
        dict(src=src + " cmp", a=[kind, "NaN"], b=[kind, "NaN"], ordering="None"),
        dict(src=src + " cmp", a=[kind, 0.0], b=[kind, -0.0], ordering="Equal"),
    ]
# testsuite/oxigraph-tests/sparql/cmp_langString.{rq,srx}: BIND(?a < ?b AS ?o) over ("a"@fr "b"@fr) and ("a"@en "b"@fr):
# the first row binds ?o = true, the second leaves it unbound (language tags differ: the comparison is an error).
# Strings are ["string", rank of the lexical form in str order, language id]: "a" < "b" => ranks 0 < 1; fr = 1, en = 2.
compare_cases += [
    dict(src="testsuite/oxigraph-tests/sparql/cmp_langString.srx result 1", a=["string", 0, 1], b=["string", 1, 1], ordering="Less"),
    dict(src="testsuite/oxigraph-tests/sparql/cmp_langString.srx result 2", a=["string", 0, 2], b=["string", 1, 1], ordering="None"),
]
# Decimal -> Float (M/decimal.rs fn to_float: Float::from(Decimal)); compared as floats (numeric.rs:127-201: Float x Decimal => Float)
decimal_to_float_cases = [
    dict(raw=str(0), value=0.0, tol=0.0),
    dict(raw=str(1 * E18), value=1.0, tol=0.0),
    dict(raw=str(10 * E18), value=10.0, tol=0.0),
    dict(raw=str(E18 // 10), value=0.1, tol=0.0),                      # assert_eq!(Float::from(Decimal "0.1"), Float::from(0.1))
    # |x - 1.701412e20| < 1 in f32: floats are 1.76e13 apart there, so that is x == the f32 nearest to 1.701412e20
    dict(raw=str(I128_MAX), value=1.701412e20, tol=0.0),
    dict(raw=str(I128_MIN), value=-1.701412e20, tol=0.0),
]
# Boolean::from(number) (lib/model/src/xsd/boolean.rs:139-170: from_integer, from_decimal, from_float, from_double) is the
# xsd:boolean CAST; on every input but NaN it is also the effective boolean value the path computes
# (builtin/native/effective_boolean_value.rs:108-113: `value != 0`).  The tests' NaN assertions (cast: false) are NOT
# transcribed: EBV(NaN) is `NaN != 0` = true there, which tests/test_gpu_parity.py checks separately as a code-derived case.
ebv_cases = []
for kind, one in (("integer", 1), ("decimal", None), ("float", 1.0), ("double", 1.0)):
    src = "M/boolean.rs from_" + kind
    val = (lambda x: [kind, str(int(x) * E18)]) if kind == "decimal" else (lambda x: [kind, x])
    ebv_cases += [dict(src=src, value=val(0), ebv=False), dict(src=src, value=val(1), ebv=True), dict(src=src, value=val(2), ebv=True)]
    if kind in ("float", "double"):
        ebv_cases += [dict(src=src, value=[kind, "INF"], ebv=True)]
# TypedValueEncodingField type ids (lib/encoding/src/typed_value/encoding.rs fn test_type_ids + the enum's declaration
# order :248-268): the dense-union type id of every field round-trips; the ids are the typed-value tags of the ABI.
type_id_cases = [["NamedNode", 1], ["BlankNode", 2], ["String", 3], ["Boolean", 4], ["Float", 5], ["Double", 6], ["Decimal", 7],
                 ["Int", 8], ["Integer", 9], ["DateTime", 10], ["Time", 11], ["Date", 12], ["Duration", 13], ["OtherLiteral", 14], ["Null", 0]]
# SparqlJoinLoweringRule (lib/logical/src/join/rewrite.rs:381-481, insta inline snapshots): joins of inputs WITHOUT shared
# variables.  "plan" = the lowered logical plan as the reference prints it.
join_lowering_cases = [
    dict(src="L/join/rewrite.rs:397-417 join_non_overlapping_variables_produces_cross_join", left=["a"], right=["b"], join_type="Inner",
         plan=["Cross Join: ", "  EmptyRelation: rows=0", "  EmptyRelation: rows=0"]),
    dict(src="L/join/rewrite.rs:419-439 optional_non_overlapping_variables_produces_left_join_with_empty_filter", left=["a"], right=["b"],
         join_type="Left", plan=["Left Join: ", "  EmptyRelation: rows=0", "  EmptyRelation: rows=0"]),
]
# Join ROW results the reference holds as a fixture: testsuite/oxigraph-tests/sparql/nested_anonymous.{rq,ttl,srx}
# (manifest.ttl:129-134).  Data (.ttl):  [ :p1 "t1" ; :p2 [ :p3 :foo ] ] .  [ :p1 "t2" ; :p2 [ :p3 :bar ] ] .
# Query (.rq):  SELECT ?a WHERE { [ :p1 ?a ; :p2 [ :p3 :foo ] ] }  = the BGP  _:x :p1 ?a . _:x :p2 _:y . _:y :p3 :foo
# (blank nodes in a query pattern are variables, quad_pattern/logical.rs:56-68) = two equi-joins (on x, then on y).
# Expected (.srx): exactly one solution, ?a = "t1".
# Terms are numbered in order of first use in the .ttl (any bijection is an equally valid dictionary).
_NA = {":p1": 1, '"t1"': 2, ":p2": 3, ":p3": 4, ":foo": 5, "_:x1": 6, "_:y1": 7, '"t2"': 8, ":bar": 9, "_:x2": 10, "_:y2": 11}
_na = lambda s, p, o: [0, _NA[s], _NA[p], _NA[o]]
join_fixture_cases = [
    dict(name="nested_anonymous", src="testsuite/oxigraph-tests/sparql/nested_anonymous.{rq,ttl,srx} (manifest.ttl:129-134)",
         terms=_NA,
         quads_gspo=[_na("_:x1", ":p1", '"t1"'), _na("_:x1", ":p2", "_:y1"), _na("_:y1", ":p3", ":foo"),
                     _na("_:x2", ":p1", '"t2"'), _na("_:x2", ":p2", "_:y2"), _na("_:y2", ":p3", ":bar")],
         # subject / predicate / object per triple pattern: str = variable, int = constant object id; default graph
         patterns=[["x", _NA[":p1"], "a"], ["x", _NA[":p2"], "y"], ["y", _NA[":p3"], _NA[":foo"]]],
         select=["a"], rows=[[_NA['"t1"']]]),
]
# STR over plain terms: lib/functions/tests/unary.rs:91-125 (the test vector) and lib/functions/tests/snapshots/
# unary__STR(PLAIN_TERM).snap (the answers): [term kind, lexical form, datatype IRI suffix or language tag] -> STR's value.
# (Over an object-id column the reference evaluates STR in this encoding: decide_input_encoding, expr_builder_context.rs:557-582.)
_xsd = lambda lex, dt: ["literal", lex, "xsd:" + dt]
str_cases = [
    dict(term=["iri", "http://example.org/foo"], str="http://example.org/foo"),
    dict(term=["bnode", "myBlank"], str="myBlank"),
    dict(term=["bnode", "4e15c8ff3a72491a92c64e7415a11d90"], str="4e15c8ff3a72491a92c64e7415a11d90"),
    dict(term=["literal", "", None], str=""),
    dict(term=["literal", "Plain Literal", None], str="Plain Literal"),
    dict(term=["literal", "\U0001F916\U0001F980\U0001F916 Bee Boo Boo Ba Bee Bee \U0001F916\U0001F980\U0001F916", None],
         str="\U0001F916\U0001F980\U0001F916 Bee Boo Boo Ba Bee Bee \U0001F916\U0001F980\U0001F916"),
    dict(term=["literal", "\u00c4pfel", "@de-at"], str="\u00c4pfel"),
    dict(term=["literal", "\u043f\u0440\u0438\u0432\u0456\u0442", "@uk-ukr"], str="\u043f\u0440\u0438\u0432\u0456\u0442"),
    dict(term=_xsd("true", "boolean"), str="true"), dict(term=_xsd("false", "boolean"), str="false"),
    dict(term=_xsd("10", "int"), str="10"), dict(term=_xsd("010", "int"), str="010"), dict(term=_xsd("0", "int"), str="0"),
    dict(term=_xsd("10", "integer"), str="10"), dict(term=_xsd("010", "integer"), str="010"), dict(term=_xsd("0", "integer"), str="0"),
    dict(term=_xsd("10", "float"), str="10"), dict(term=_xsd("10.0", "float"), str="10.0"), dict(term=_xsd("0", "float"), str="0"),
    dict(term=_xsd("10", "double"), str="10"), dict(term=_xsd("10.0", "double"), str="10.0"), dict(term=_xsd("0", "double"), str="0"),
    dict(term=_xsd("10", "decimal"), str="10"), dict(term=_xsd("10.0", "decimal"), str="10.0"), dict(term=_xsd("0", "decimal"), str="0"),
]
# testsuite/oxigraph-tests/sparql/small_iri_str.{rq,srx}: ASK { FILTER(STR(<ex:a>) = "ex:a") } => true
str_query_cases = [dict(src="testsuite/oxigraph-tests/sparql/small_iri_str.{rq,srx}", term=["iri", "ex:a"], equals="ex:a", answer=True)]
# bench/tests/plans/snapshots/r#mod__plans__bsbm_explore__BSBM Explore - Q{1,5} (Execution Plan).snap: the operator subtree
# below the SortExec (the path this library executes), line for line; IRIs and the masked object id are written <c>, and the
# `additional_filters=[DynamicFilter ...]` annotations (a run-time superset filter, never changes results) are dropped.
plan_snapshot_cases = dict(
    q5=dict(src="..Q5 (Execution Plan).snap:10-30", lines=[
        "HashJoinExec: mode=CollectLeft, join_type=Inner, on=[(product@0, product@0)], filter=EBV(LT(ENC_TV(simProperty2@1), ADD(ENC_TV(origProperty2@0), 9:170))) AND EBV(GT(ENC_TV(simProperty2@1), SUB(ENC_TV(origProperty2@0), 9:170))), projection=[product@0, productLabel@1]",
        "  CrossJoinExec",
        "    HashJoinExec: mode=CollectLeft, join_type=Inner, on=[(product@0, product@0)], filter=EBV(LT(ENC_TV(simProperty1@1), ADD(ENC_TV(origProperty1@0), 9:120))) AND EBV(GT(ENC_TV(simProperty1@1), SUB(ENC_TV(origProperty1@0), 9:120))), projection=[product@0, productLabel@1]",
        "      CrossJoinExec",
        "        HashJoinExec: mode=CollectLeft, join_type=Inner, on=[(prodFeature@2, prodFeature@1), (product@0, product@0)], projection=[product@0, productLabel@1]",
        "          CrossJoinExec",
        "            FilterExec: product@0 != <c>",
        "              DataSourceExec: [GPOS] subject=?product, predicate=<c>, object=?productLabel",
        "            DataSourceExec: [GSPO] subject=<c>, predicate=<c>, object=?prodFeature",
        "          FilterExec: product@0 != <c>",
        "            DataSourceExec: [GPOS] subject=?product, predicate=<c>, object=?prodFeature",
        "        DataSourceExec: [GSPO] subject=<c>, predicate=<c>, object=?origProperty1",
        "      FilterExec: product@0 != <c>",
        "        DataSourceExec: [GPOS] subject=?product, predicate=<c>, object=?simProperty1",
        "    DataSourceExec: [GSPO] subject=<c>, predicate=<c>, object=?origProperty2",
        "  FilterExec: product@0 != <c>",
        "    DataSourceExec: [GPOS] subject=?product, predicate=<c>, object=?simProperty2",
    ]),
    q1=dict(src="..Q1 (Execution Plan).snap:10-19", lines=[
        "HashJoinExec: mode=CollectLeft, join_type=Inner, on=[(product@0, product@0)], projection=[product@0, label@1]",
        "  HashJoinExec: mode=CollectLeft, join_type=Inner, on=[(product@0, product@0)], projection=[product@0, label@1]",
        "    HashJoinExec: mode=CollectLeft, join_type=Inner, on=[(product@0, product@0)], projection=[product@0, label@1]",
        "      HashJoinExec: mode=CollectLeft, join_type=Inner, on=[(product@0, product@0)], projection=[product@0, label@1]",
        "        DataSourceExec: [GPOS] subject=?product, predicate=<c>, object=?label",
        "        DataSourceExec: [GPOS] subject=?product, predicate=<c>, object=<c>",
        "      DataSourceExec: [GPOS] subject=?product, predicate=<c>, object=<c>",
        "    DataSourceExec: [GPOS] subject=?product, predicate=<c>, object=<c>",
        "  FilterExec: EBV(GT(ENC_TV(value1@1), 9:136)), projection=[product@0]",
        "    DataSourceExec: [GPOS] subject=?product, predicate=<c>, object=?value1",
    ]),
)

out = dict(
    _about="Known-answer vectors transcribed from the reference's unit tests (see make_reference_kats.py).",
    scan=scan_cases, remove=remove_cases, store=store_cases, predicate_and=predicate_and_cases,
    index_choice=index_choice_cases, score_order=score_order_cases, pushdown=pushdown_cases,
    pushdown_display=pushdown_display_cases, rowgroups=rowgroup_cases, dedupe=dedupe_cases,
    prune=prune_cases, find_range=find_range_cases, numeric_arith=numeric_arith_cases,
    decimal_to_double=decimal_to_double_cases, compare=compare_cases, decimal_to_float=decimal_to_float_cases,
    ebv=ebv_cases, type_ids=type_id_cases, join_lowering=join_lowering_cases, join_fixtures=join_fixture_cases, str_plain_term=str_cases, str_queries=str_query_cases,
    plan_snapshots=plan_snapshot_cases)

if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
