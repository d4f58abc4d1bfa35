"""LANGMATCHES(LANG(x), "range") — the FILTER of BSBM explore Q8 (`EBV(LANGMATCHES(LANG(ENC_TV(text)), 3:{value:EN,language:}))`,
Q8 (Execution Plan).snap:18).  The host-side range matching restates scalar/strings/lang_matches.rs:52-69; the oracle's
operator restates scalar/terms/lang.rs:45-58 + the verdict lookup.  The reference holds no unit test for either, so the
known answers below are the SPARQL 1.1 §17.4.2.7 / RFC 4647 §3.3.1 ones (basic filtering) plus the corners the
reference's code decides its own way (an empty tag against an empty range).  CPU only."""
import numpy as np

from rdf_fusion_amd import abi
from rdf_fusion_amd.engine import TV_DTYPE
from rdf_fusion_amd.plan import PlanBuilder, col, ENC_TV, EBV, NOT, LANGMATCHES_LANG, lang_matches
from oracle import oracle as orc

KATS = [("en", "en", True), ("en-US", "en", True), ("en", "en-US", False), ("fr", "en", False), ("EN", "en", True),
        ("en-us", "EN-US", True), ("ena", "en", False), ("e", "en", False), ("de-CH-1996", "de-ch", True),
        ("de-CH-1996", "de-1996", False), ("", "*", False), ("de", "*", True), ("", "en", False),
        ("", "", True),              # split('-') of "" is [""] on both sides: equal (lang_matches.rs:58-68)
        ("en", "", False), ("en-", "en", True), ("en", "en-", False), ("EN", "ÉN", False), ("İ", "i", False)]
LANGUAGES = ["", "en", "en-US", "de", "fr-CA", "EN-gb", "zh-Hant-TW"]


def test_lang_matches_known_answers():
    for tag, rng, exp in KATS:
        assert lang_matches(tag, rng) is exp, (tag, rng)


def lang_table():
    """ids: 1..6 strings with language ids 1..6, 7 simple literal, 8 IRI, 9 blank node, 10 integer, 11 boolean, 12 string with an
    unknown language id (beyond the table)"""
    tv = np.zeros(13, dtype=TV_DTYPE)
    for i in range(1, 7):
        tv["tag"][i], tv["aux"][i], tv["lo"][i] = abi.TV_STRING, i, i
    tv["tag"][7] = abi.TV_STRING
    tv["tag"][8], tv["tag"][9], tv["tag"][10], tv["tag"][11] = abi.TV_NAMED_NODE, abi.TV_BLANK_NODE, abi.TV_INTEGER, abi.TV_BOOLEAN
    tv["tag"][12], tv["aux"][12] = abi.TV_STRING, 40
    return tv


def expected_rows(ids, rng, negate=False):
    tv = lang_table()
    out = []
    for r, i in enumerate(ids):
        if i == 0 or i >= len(tv) or tv["tag"][i] in (abi.TV_NAMED_NODE, abi.TV_BLANK_NODE):
            continue                                        # LANG(unbound / IRI / blank) is an error: dropped either way
        lang = int(tv["aux"][i]) if tv["tag"][i] == abi.TV_STRING else 0
        if lang >= len(LANGUAGES):
            continue
        if lang_matches(LANGUAGES[lang], rng) != negate:
            out.append(r)
    return out


def test_oracle_lang_in_filter():
    st = orc.OracleStore()
    st.set_typed_values(lang_table())
    ids = np.array(list(range(0, 15)) * 3, dtype=np.uint32)
    row = np.arange(len(ids), dtype=np.uint32)
    for rng in ("en", "EN", "*", "de", "en-us", "zh-hant", "fr-CA-x", ""):
        for negate in (False, True):
            e = EBV(LANGMATCHES_LANG(ENC_TV(col(0)), rng, LANGUAGES))
            pb = PlanBuilder()
            desc = pb.build(pb.filter(pb.table(0, 2), NOT(e) if negate else e, projection=[1]))
            cols, n, _ = st.execute(desc, [[ids, row]])
            assert sorted(cols[0][:n].tolist()) == expected_rows(ids.tolist(), rng, negate), (rng, negate)
