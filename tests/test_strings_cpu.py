"""String-valued expressions (STR / STRLEN / SUBSTR / UCASE / LCASE, comparisons of computed strings) in the CPU oracle:
the reference's own vectors (lib/functions/tests/snapshots/unary__STR(PLAIN_TERM).snap, testsuite/oxigraph-tests/sparql/
small_iri_str.*) and random strings against Python's str (a third implementation of the same definitions)."""
import numpy as np
import pytest

from oracle import oracle as orc
from rdf_fusion_amd import abi
from rdf_fusion_amd.plan import (col, integer, int32, double, lit_str, STR, STRLEN, SUBSTR, UCASE, LCASE, STRBEFORE, STRAFTER, ENC_TV, EQ, LT, GT, EBV,
                                 CONTAINS, STRSTARTS, REGEX)
import kat_util as ku


def store_of(terms):
    tv, off, heap, langs = ku.term_dictionary(terms)
    st = orc.OracleStore()
    st.set_typed_values(tv)
    st.set_strings(off, heap)
    return st, langs


def test_str_reference_snapshot(kats):
    """unary__STR(PLAIN_TERM).snap: STR of every kind of term is its lexical form as written."""
    cases = kats["str_plain_term"]
    st, _ = store_of([c["term"] for c in cases])
    ids = np.arange(1, len(cases) + 1, dtype=np.uint32)
    got = st.eval_str(STR(col(0)), [ids])
    assert [g[0].decode("utf-8") for g in got] == [c["str"] for c in cases]
    assert all(g[1] == 0 for g in got)                              # a simple literal, whatever the source's language
    assert st.eval_str(STR(col(0)), [np.array([0, 999], np.uint32)]) == [None, None]   # unbound / unknown id: the error value


def test_small_iri_str_query(kats):
    for c in kats["str_queries"]:
        st, _ = store_of([c["term"]])
        e = EBV(EQ(STR(col(0)), lit_str(c["equals"])))
        assert st.eval_bool(e_with_table(st, e), [np.array([1], np.uint32)])[0] == (1 if c["answer"] else 0)


def e_with_table(st, e):
    return e        # (eval_bool resolves the string table itself, see OracleStore.eval_bool)


WORDS = ["", "a", "Abc", "hello world", "MiXeD Case 123", "äöü ÄÖÜ ß", "日本語のテキスト", "🤖🦀 crab", "tab\tnew\nline", "x" * 70, "naïve café"]


def test_strlen_substr_against_python():
    terms = [["literal", w, None] for w in WORDS] + [["literal", w, "@en"] for w in WORDS[:4]] + [["iri", "http://e/x"], ["literal", "12", "xsd:integer"]]
    st, langs = store_of(terms)
    ids = np.arange(1, len(terms) + 1, dtype=np.uint32)
    lens, _ = st.eval_tv(STRLEN(ENC_TV(col(0))), [ids])
    for k, t in enumerate(terms):
        if t[0] == "literal" and (t[2] is None or t[2].startswith("@")):
            assert lens["tag"][k] == abi.TV_INTEGER and lens["lo"][k] == len(t[1]), t
        else:
            assert lens["tag"][k] == abi.TV_NULL, t                  # strlen.rs: only simple / language-tagged strings
    # STRLEN(STR(x)) counts the lexical form of ANY term
    lens, _ = st.eval_tv(STRLEN(STR(col(0))), [ids])
    assert lens["lo"].tolist() == [len(t[1]) for t in terms]
    for start in (1, 2, 5, 80):
        for length in (None, 0, 1, 3, 200):
            e = SUBSTR(ENC_TV(col(0)), integer(start), None if length is None else int32(length))
            got = st.eval_str(e, [ids])
            for k, t in enumerate(terms):
                if not (t[0] == "literal" and (t[2] is None or t[2].startswith("@"))):
                    assert got[k] is None, t
                    continue
                w = t[1]
                want = w[start - 1:] if length is None else w[start - 1:start - 1 + length]
                assert got[k] == (want.encode("utf-8"), 0 if t[2] is None else langs.index(t[2][1:])), (t, start, length)
    # start < 1, a negative length: errors (usize::try_from / checked_sub in sub_str.rs:88-95)
    assert st.eval_str(SUBSTR(ENC_TV(col(0)), integer(0)), [ids[:3]]) == [None] * 3
    assert st.eval_str(SUBSTR(ENC_TV(col(0)), integer(-1)), [ids[:3]]) == [None] * 3
    assert st.eval_str(SUBSTR(ENC_TV(col(0)), integer(1), integer(-2)), [ids[:3]]) == [None] * 3
    with pytest.raises(RuntimeError):
        st.eval_str(SUBSTR(ENC_TV(col(0)), double(1.0)), [ids[:3]])   # a double position: not restated, refused


def test_case_mapping_and_views():
    ascii_words = [w for w in WORDS if w.isascii()]
    terms = [["literal", w, None] for w in ascii_words] + [["literal", "Grüße", None]]
    st, _ = store_of(terms)
    ids = np.arange(1, len(ascii_words) + 1, dtype=np.uint32)
    assert [g[0].decode() for g in st.eval_str(UCASE(ENC_TV(col(0))), [ids])] == [w.upper() for w in ascii_words]
    assert [g[0].decode() for g in st.eval_str(LCASE(ENC_TV(col(0))), [ids])] == [w.lower() for w in ascii_words]
    assert [g[0].decode() for g in st.eval_str(UCASE(SUBSTR(LCASE(ENC_TV(col(0))), integer(2), integer(4))), [ids])] == [w[1:5].upper() for w in ascii_words]
    with pytest.raises(RuntimeError):                                 # Unicode case tables are not restated: refused, never guessed
        st.eval_str(UCASE(ENC_TV(col(0))), [np.array([len(terms)], np.uint32)])
    # computed strings in predicates
    got = st.eval_bool(EBV(CONTAINS(UCASE(ENC_TV(col(0))), "WORLD")), [ids])
    assert got.tolist() == [1 if "WORLD" in w.upper() else 0 for w in ascii_words]
    got = st.eval_bool(EBV(STRSTARTS(STR(col(0)), "hello")), [ids])
    assert got.tolist() == [1 if w.startswith("hello") else 0 for w in ascii_words]
    got = st.eval_bool(EBV(REGEX(SUBSTR(ENC_TV(col(0)), integer(2)), "^b")), [ids])
    assert got.tolist() == [1 if w[1:2] == "b" else 0 for w in ascii_words]


def test_computed_string_comparisons_follow_str_order():
    rng = np.random.default_rng(5)
    alphabet = list("abcAB09 _é日")
    words = sorted({"".join(rng.choice(alphabet, rng.integers(0, 6))) for _ in range(300)})
    terms = [["literal", w, None] for w in words]
    st, _ = store_of(terms)
    ids = np.arange(1, len(words) + 1, dtype=np.uint32)
    for pivot in ("", "a", "ab", "B0", "é", "日", "zzz"):
        pb = pivot.encode("utf-8")
        for op, f in ((EQ, lambda w: w == pb), (LT, lambda w: w < pb), (GT, lambda w: w > pb)):
            got = st.eval_bool(EBV(op(STR(col(0)), lit_str(pivot))), [ids])
            assert got.tolist() == [1 if f(w.encode("utf-8")) else 0 for w in words], (pivot, op.__name__)
    # different languages never compare (language_string.rs:44-52): the error value
    assert st.eval_bool(EBV(EQ(STR(col(0)), lit_str("a", language_id=3))), [ids[:5]]).tolist() == [2] * 5


def test_strbefore_strafter_against_python():
    """STRBEFORE / STRAFTER (str_before.rs / str_after.rs): the part of the first argument before / behind the first occurrence of the second,
    with the first argument's language; no occurrence => the simple literal ""; the second argument has no language or the first one's
    (string_literal.rs:80-95), anything else — and any argument that is no string literal — is the error value.  Against str.partition."""
    subjects = WORDS + ["a-b-c", "--", "xyzxyz", "ünï-cödé-x"]
    terms = [["literal", w, None] for w in subjects] + [["literal", w, "@en"] for w in subjects] + [["literal", w, "@de"] for w in subjects[:3]] + \
            [["iri", "http://e/x-y"], ["literal", "12", "xsd:integer"]]
    st, langs = store_of(terms)
    ids = np.arange(1, len(terms) + 1, dtype=np.uint32)
    en = langs.index("en")
    for needle, needle_lang in (("-", 0), ("", 0), ("l", 0), ("yz", 0), ("ö", 0), ("zzz", 0), ("-", en), ("a", en)):
        for fn, after in ((STRBEFORE, False), (STRAFTER, True)):
            got = st.eval_str(fn(ENC_TV(col(0)), lit_str(needle, needle_lang)), [ids])
            for k, t in enumerate(terms):
                is_string = t[0] == "literal" and (t[2] is None or t[2].startswith("@"))
                lang = 0 if not is_string or t[2] is None else langs.index(t[2][1:])
                if not is_string or (needle_lang != 0 and needle_lang != lang):
                    assert got[k] is None, (t, needle, needle_lang, fn.__name__)          # not a string literal / incompatible languages: error
                    continue
                head, sep, tail = t[1].partition(needle) if needle != "" else ("", "", t[1])     # the empty needle is found at position 0
                want = ("", 0) if (sep == "" and needle != "") else ((tail if after else head), lang)
                assert (got[k][0].decode("utf-8"), got[k][1]) == want, (t, needle, needle_lang, fn.__name__, got[k])
    # views compose: STRAFTER(STRBEFORE(x, "-c"), "a-") of "a-b-c" is "b"; a case-mapped source is searched in its mapped form
    one = st.eval_str(STRAFTER(STRBEFORE(ENC_TV(col(0)), lit_str("-c")), lit_str("a-")), [np.array([subjects.index("a-b-c") + 1], np.uint32)])
    assert one[0] == (b"b", 0)
    two = st.eval_str(STRBEFORE(UCASE(ENC_TV(col(0))), lit_str("-B")), [np.array([subjects.index("a-b-c") + 1], np.uint32)])
    assert two[0] == (b"A", 0)
    # STRLEN / CONTAINS / comparisons consume the views
    lens, _ = st.eval_tv(STRLEN(STRAFTER(ENC_TV(col(0)), lit_str("-"))), [np.array([subjects.index("a-b-c") + 1], np.uint32)])
    assert lens["lo"][0] == 3
    assert st.eval_bool(EBV(EQ(STRBEFORE(ENC_TV(col(0)), lit_str("-")), lit_str("a"))), [np.array([subjects.index("a-b-c") + 1], np.uint32)])[0] == 1
