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


def test_device_compiler_accepts_the_subset_and_refuses_the_rest():
    for pattern, flags, _, expected in KATS:
        if expected is None and flags != "z":
            continue
        assert engine.regex_check(pattern, flags) >= 0
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
    for bad in ("\\d+", "\\w", "\\bfoo", "\\p{L}", "[[:alpha:]]", "[a&&b]", "(?i)a", "a|^b", "x^", "[é]", "a{", "*a", "(", "a)", "\\"):
        with pytest.raises(engine.RdfGpuError):
            engine.regex_check(bad, "")
    with pytest.raises(engine.RdfGpuError):
        engine.regex_check("é", "i")                            # non-ASCII letter under `i` needs the Unicode fold tables
