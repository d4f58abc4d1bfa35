"""REGEX without a GPU: the oracle's restatement (regex_oracle.c, a Pike VM over code points) against known answers,
the reference's own fixture, and Python's `re` as an independent third opinion on random patterns; and the device
compiler's accept / refuse decisions through the C ABI (rdfgpu_regex_check — host logic only, nothing is matched on
the CPU by the product library)."""
import re

import numpy as np
import pytest

from oracle import oracle as orc
from rdf_fusion_amd import engine
import kat_util as ku

KATS = [  # (pattern, flags, subject, expected)   None = the SPARQL error value
    ("^a$", "", "a", True), ("^a$", "", "b", False),            # testsuite/oxigraph-tests/sparql/regex_variable.{rq,srx}
    ("a.c", "", "abc", True), ("a.c", "", "a\nc", False), ("a.c", "s", "a\nc", True),
    ("(ab|cd)+e", "", "xxcdabe", True), ("(ab|cd)+e", "", "xxce", False),
    ("[a-c]{2,3}x", "", "abx", True), ("[a-c]{2,3}x", "", "ax", False),
    ("k", "i", "K", True), ("k", "i", "K", True), ("S", "i", "ſ", True), ("k", "", "K", False),
    ("^$", "", "", True), ("^$", "", "a", False), ("^b$", "m", "a\nb\nc", True), ("^b$", "", "a\nb\nc", False),
    ("b$", "", "ab\n", False),                                   # `$` is end of text, not "before a final newline"
    ("é+", "", "caféé", True), ("[^a]", "", "a", False), ("[^a]", "", "é", True), (".", "", "😀", True), ("^.$", "", "😀", True),
    ("a*", "", "", True), ("x{2}", "", "axxb", True), ("x{2}", "", "axb", False),
    ("a.c", "q", "a.c", True), ("a.c", "q", "abc", False), ("a b", "x", "ab", True), ("a # comment\n b", "x", "ab", True),
    ("a", "z", "a", None),                                       # invalid flag: error (regex.rs:137)
    ("(a|b)*c|d", "", "bbac", True), ("(a*)*b", "", "aaab", True), ("(a*)*b", "", "aaac", False), ("a|", "", "zzz", True),
    ("\\Aab\\z", "m", "ab", True), ("\\Aab\\z", "m", "x\nab", False), ("a\\.b", "", "a.b", True), ("a\\.b", "", "axb", False),
    ("(?P<n>ab)c", "", "zabc", True), ("a+?b", "", "aaab", True), ("\\x41", "", "A", True),
    # Perl classes and word boundaries (ASCII members; exact on all-ASCII subjects)
    ("\\d+", "", "ab12", True), ("\\d+", "", "abc", False), ("^\\w+$", "", "ab_1Z", True), ("^\\w+$", "", "ab-1", False),
    ("\\s", "", "a\tb", True), ("\\s", "", "ab", False), ("\\D", "", "12", False), ("\\W", "", "a_b", False), ("\\W", "", "a b", True),
    ("^\\S+$", "", "a\x0bb", False), ("[\\d\\s]x", "", " x", True), ("[^\\d]x", "", "1x", False), ("[^\\d]x", "", "ax", True),
    ("\\bfoo\\b", "", "a foo b", True), ("\\bfoo\\b", "", "afoo", False), ("\\bfoo\\b", "", "foo", True), ("\\Bfoo", "", "afoo", True),
    ("\\Bfoo", "", "foo", False), ("\\B", "", "", True), ("\\b", "", "", False), ("\\b", "", "a", True), ("a\\b|\\Bb", "", "ab", True), ("a\\b|\\Bb", "", "a-b", True), ("a\\B|\\Bb", "", "a-b", False),
    ("(\\b|x)+a", "", "-a", True), ("\\w\\b\\w", "", "ab", False), ("^\\b", "", " a", False), ("\\b$", "", "a ", False), ("\\d", "i", "5", True),
    ("grad\\w*\\d{2}\\b", "i", "GraduateStudent42 x", True),
    # inline flags (regex-syntax: a directive holds to the end of the enclosing group; `(?flags:..)` for its own group)
    ("(?i)abc", "", "xABCx", True), ("a(?i)bc", "", "aBC", True), ("a(?i)bc", "", "ABC", False), ("(?i:a)bc", "", "Abc", True), ("(?i:a)bc", "", "ABC", False),
    ("(?i)a(?-i)b", "", "Ab", True), ("(?i)a(?-i)b", "", "AB", False), ("(a(?i)b)c", "", "aBc", True), ("(a(?i)b)c", "", "aBC", False),
    ("(?s)a.b", "", "a\nb", True), ("(?s-s)a.b", "", "a\nb", False), ("(?m)^b$", "", "a\nb\nc", True), ("(?m:^b)", "", "a\nb", True), ("a(?x) b c # d\n e", "", "abce", True),
    ("(?is)A.B", "", "a\nb", True), ("(?U)a+b", "", "aab", True), ("abc", "i", "(?-i)ABC", True), ("(?-i)abc", "i", "ABC", False), ("(?u)a", "", "a", None), ("(?)a", "", "a", None),
    # classes: nesting, POSIX, set operations (left to right; operands are unions)
    ("[a-z&&[^aeiou]]", "", "e", False), ("[a-z&&[^aeiou]]", "", "f", True), ("^[a-z--[aeiou]]+$", "", "xyz", True), ("^[a-z--[aeiou]]+$", "", "xaz", False),
    ("[a-c~~b-d]", "", "a", True), ("[a-c~~b-d]", "", "b", False), ("[a-c~~b-d]", "", "d", True), ("[a-z&&b-y--m]", "", "m", False), ("[a-z&&b-y--m]", "", "n", True), ("[a-z&&b-y--m]", "", "z", False),
    ("^[[:alpha:]]+[[:digit:]]$", "", "ab1", True), ("[[:^alpha:]]", "", "ab", False), ("[[:^alpha:]]", "", "a1", True), ("[[:alpha:][:digit:]_]+!", "", "a_1!", True),
    ("[a[bc]]", "", "c", True), ("[^a[bc]]", "", "c", False), ("[^a[bc]]", "", "d", True), ("[^a[^bc]]", "", "b", True), ("[^a[^bc]]", "", "d", False),
    ("[[:punct:]&&[^!]]", "", "!", False), ("[[:punct:]&&[^!]]", "", "?", True), ("[A-Z&&[a-c]]", "i", "b", True), ("[A-Z&&[a-c]]", "i", "d", False),
    ("[\\w--\\d]", "", "5", False), ("[\\w--\\d]", "", "x", True), ("[^a&&b]", "", "a", True),
    # Unicode general categories: ASCII members (like the Perl classes)
    ("\\p{L}+", "", "12ab", True), ("^\\p{Lu}", "", "ab", False), ("^\\p{Lu}", "", "Ab", True), ("\\pN+", "", "ab", False), ("\\pN", "", "a7", True),
    ("\\P{L}", "", "ab", False), ("\\P{L}", "", "a1", True), ("[\\p{Nd}x]", "", "x", True), ("\\p{^L}", "", "ab", False), ("\\p{gc=Lu}", "", "aB", True),
    ("\\p{Letter}\\p{Decimal_Number}", "", "a1", True), ("\\p{P}", "", "a-b", True), ("\\p{P}", "", "a+b", False), ("\\p{S}", "", "a+b", True), ("\\p{Sc}", "", "$", True),
    ("\\p{Zs}", "", "a b", True), ("\\p{Zs}", "", "a\tb", False), ("\\p{Cc}", "", "a\tb", True), ("\\p{Lu}", "i", "a", True), ("\\p{Lo}", "", "abc", False),
    ("\\p{Greek}", "", "a", None), ("\\p{sc=Latin}", "", "a", None), ("\\p{Foo}", "", "a", None),
]


@pytest.mark.parametrize("pattern,flags,subject,expected", KATS)
def test_oracle_regex_known_answers(pattern, flags, subject, expected):
    assert orc.regex_is_match(pattern, flags, subject) == expected


def test_oracle_regex_agrees_with_python_re():
    rng = np.random.default_rng(2024)
    checked = 0
    for _ in range(4000):
        pat, flags, py, py_flags = ku.random_regex(rng)
        try:
            rx = re.compile(py, py_flags)
        except re.error:
            continue
        for _ in range(6):
            s = ku.random_subject(rng)
            assert orc.regex_is_match(pat, flags, s) == (rx.search(s) is not None), (pat, flags, s)
            checked += 1
    assert checked > 15_000


def test_oracle_perl_classes_agree_with_python_re():
    """`\\d \\w \\s \\b` and their negations against Python's `re` with re.ASCII over all-ASCII subjects."""
    rng = np.random.default_rng(77)
    checked = 0
    for _ in range(4000):
        pat, flags, py, py_flags = ku.random_regex(rng, perl=True)
        try:
            rx = re.compile(py, py_flags)
        except re.error:
            continue
        for _ in range(6):
            s = ku.random_subject(rng, ascii_only=True)
            if s == "" and "\\B" in pat:
                continue                      # Python (< 3.14) never matches \B against the empty string; the regex crate does
            assert orc.regex_is_match(pat, flags, s) == (rx.search(s) is not None), (pat, flags, s)
            checked += 1
    assert checked > 15_000


def test_oracle_extended_syntax_agrees_with_python_re():
    """Inline flag groups, a leading flag directive, POSIX classes, class set operations and (over ASCII subjects) Unicode general
    categories: the oracle against Python's `re` on the same patterns with the set algebra spelled out."""
    rng = np.random.default_rng(99)
    checked = used = 0
    for k in range(6000):
        perl = k % 2 == 0
        pat, flags, py, py_flags = ku.random_regex(rng, perl=perl, extended=True)
        if "x" in flags and ("[:" in pat or "&&" in pat or "--" in pat or "~~" in pat or "#" in py):
            continue                            # (Python's verbose mode treats '#' / spaces inside the spelled-out classes its own way)
        try:
            rx = re.compile(py, py_flags)
        except re.error:
            continue
        used += any(t in pat for t in ("(?", "[:", "&&", "--", "~~", "\\p", "\\P"))
        for _ in range(5):
            s = ku.random_subject(rng, ascii_only=perl)
            if s == "" and "\\B" in pat:
                continue
            assert orc.regex_is_match(pat, flags, s) == (rx.search(s) is not None), (pat, flags, py, s)
            checked += 1
    assert checked > 15_000 and used > 1500


def test_oracle_perl_classes_refuse_non_ascii_subjects():
    for pat in ("\\d", "\\w+", "a\\b", "[\\s]"):
        with pytest.raises(orc.NeedsUnicodeTables):
            orc.regex_is_match(pat, "", "caf\u00e9 1")
        assert orc.regex_is_match(pat, "", "cafe 1a") in (True, False)
    assert orc.regex_is_match("é", "", "café") is True          # no Perl class: non-ASCII subjects stay exact


def test_device_compiler_accepts_the_subset_and_refuses_the_rest():
    for pattern, flags, _, expected in KATS:
        if expected is None and flags != "z":
            continue
        if pattern == "(\\b|x)+a":
            with pytest.raises(engine.RdfGpuError, match="word boundary under a repetition"):
                engine.regex_check(pattern, flags)       # a loop whose body can be just the assertion: outside the device subset
            continue
        assert engine.regex_check(pattern, flags) >= 0
    for _ in range(2000):
        pat, flags, _, _ = ku.random_regex(np.random.default_rng(_), perl=True)
        try:
            assert 0 <= engine.regex_check(pat, flags) <= 64
        except engine.RdfGpuError as e:
            assert "64 positions" in str(e) or "word boundary under a repetition" in str(e) or ku.regex_needs_unicode_fold_care(pat, flags), (pat, flags, str(e))
    for k in range(3000):                                        # the extended syntax: accepted unless it is one of the documented refusals
        pat, flags, _, _ = ku.random_regex(np.random.default_rng(10_000 + k), perl=k % 2 == 0, extended=True)
        try:
            assert 0 <= engine.regex_check(pat, flags) <= 64
        except engine.RdfGpuError as e:
            assert any(t in str(e) for t in ("64 positions", "word boundary under a repetition", "non-ASCII case partner", "non-ASCII literal under", "anchor that is not")) \
                or ku.regex_needs_unicode_fold_care(pat, flags), (pat, flags, str(e))
    assert engine.regex_check("a", "z") == 0                     # invalid flag: a program that only yields the error value
    rng = np.random.default_rng(7)
    n_ok = 0
    for _ in range(3000):
        pat, flags, _, _ = ku.random_regex(rng)
        try:
            assert 0 <= engine.regex_check(pat, flags) <= 64
            n_ok += 1
        except engine.RdfGpuError as e:
            assert "64 positions" in str(e) or ku.regex_needs_unicode_fold_care(pat, flags), (pat, flags, str(e))
    assert n_ok > 2000
    for bad in ("\\p{Greek}", "\\p{sc=Latin}", "\\p{Foo}", "[\\b]", "\\b*", "\\b+a", "\\<a", "[[:alfa:]]", "[a&&]", "(?u)a", "(?R)a", "(?)a", "(?i", "[^k]+(?i)[^k]", "a|^b", "x^", "[é]", "a{", "*a", "(", "a)", "\\"):
        with pytest.raises(engine.RdfGpuError):
            engine.regex_check(bad, "")
    with pytest.raises(engine.RdfGpuError):
        engine.regex_check("é", "i")                            # non-ASCII letter under `i` needs the Unicode fold tables


def test_oracle_string_functions_agree_with_python():
    """CONTAINS / STRSTARTS / STRENDS in the oracle (through a FILTER plan) against Python's substring tests."""
    from rdf_fusion_amd.plan import PlanBuilder, col, ENC_TV, EBV, CONTAINS, STRSTARTS, STRENDS
    rng = np.random.default_rng(4)
    strings = [ku.random_subject(rng) for _ in range(400)] + ["", "abc", "€uro"]
    tv, offsets, heap = ku.string_dictionary(strings)
    os_ = orc.OracleStore()
    os_.set_typed_values(tv)
    os_.set_strings(offsets, heap)
    ids = rng.integers(0, len(tv), 3000).astype(np.uint32)
    payload = np.arange(len(ids), dtype=np.uint32) + 1
    is_str = (ids >= 1) & (ids <= len(strings))
    lang = np.array([0] + [0 if k % 5 else 7 for k in range(len(strings))] + [0] * 5)
    for fn, py in ((CONTAINS, lambda s_, n: n in s_), (STRSTARTS, str.startswith), (STRENDS, str.endswith)):
        for needle in ("", "a", "ab", "€", "k ", "abc"):
            for const_lang in (0, 7, 9):
                pb = PlanBuilder()
                desc = pb.build(pb.filter(pb.table(0, 2), EBV(fn(ENC_TV(col(0)), needle, const_lang)), projection=[1]))
                got, n, _ = os_.execute(desc, [[ids, payload]])
                ok = np.array([bool(is_str[r]) and (const_lang == 0 or lang[ids[r]] == const_lang) and py(strings[ids[r] - 1], needle)
                               for r in range(len(ids))])
                np.testing.assert_array_equal(np.sort(got[0][:n]), payload[ok])
