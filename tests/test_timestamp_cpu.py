"""xsd:dateTime / xsd:date / xsd:time comparisons (the FILTERs of BSBM explore Q7 / Q8 / Q10 / Q11 compare dates):
the oracle's restatement of `PartialOrd for Timestamp` (lib/model/src/xsd/date_time.rs:1617-1654) and of the W3C
timeOnTimeline function (rdf_fusion_amd/xsd.py) against the reference's own known answers — the `equals` and `cmp`
unit tests of date_time.rs:2696-2830, transcribed — and against a direct Python statement of the rule on random
values.  CPU only."""
import numpy as np
import pytest

from rdf_fusion_amd import abi, xsd
from rdf_fusion_amd.engine import TV_DTYPE
from rdf_fusion_amd.plan import PlanBuilder, col, ENC_TV, GT, LT, GEQ, LEQ, EQ, NEQ, EBV
from oracle import oracle as orc

DT, D, T = "dateTime", "date", "time"
PARSE = {DT: (xsd.parse_date_time, abi.TV_DATE_TIME), D: (xsd.parse_date, abi.TV_DATE), T: (xsd.parse_time, abi.TV_TIME)}

# (kind, a, b, relation that the reference asserts TRUE, or "!rel" for asserted FALSE)  date_time.rs:2697-2757, 2798-2813
REFERENCE_KATS = [
    (DT, "2002-04-02T12:00:00-01:00", "2002-04-02T17:00:00+04:00", "=="),
    (DT, "2002-04-02T12:00:00-05:00", "2002-04-02T23:00:00+06:00", "=="),
    (DT, "2002-04-02T12:00:00-05:00", "2002-04-02T17:00:00-05:00", "!="),
    (DT, "2002-04-02T12:00:00-05:00", "2002-04-02T12:00:00-05:00", "=="),
    (DT, "2002-04-02T23:00:00-04:00", "2002-04-03T02:00:00-01:00", "=="),
    (DT, "1999-12-31T24:00:00-05:00", "2000-01-01T00:00:00-05:00", "=="),
    (DT, "2005-04-04T24:00:00-05:00", "2005-04-04T00:00:00-05:00", "!="),
    (D, "2004-12-25Z", "2004-12-25+07:00", "!="),
    (D, "2004-12-25-12:00", "2004-12-26+12:00", "=="),
    (T, "08:00:00+09:00", "17:00:00-06:00", "!="),
    (T, "21:30:00+10:30", "06:00:00-05:00", "=="),
    (T, "24:00:00+01:00", "00:00:00+01:00", "=="),
    (T, "05:00:00-03:00", "10:00:00+02:00", "=="),
    (T, "23:00:00-03:00", "02:00:00Z", "!="),
    (D, "2004-12-25Z", "2004-12-25-05:00", "<"),
    (D, "2004-12-25-12:00", "2004-12-26+12:00", "!<"),
    (D, "2004-12-25Z", "2004-12-25+07:00", ">"),
    (D, "2004-12-25-12:00", "2004-12-26+12:00", "!>"),
    (T, "12:00:00", "23:00:00+06:00", "!<"),
    (T, "11:00:00-05:00", "17:00:00Z", "<"),
    (T, "23:59:59", "24:00:00", "!<"),
    (T, "08:00:00+09:00", "17:00:00-06:00", "!>"),
]
OPS = {"==": EQ, "!=": NEQ, "<": LT, ">": GT, "<=": LEQ, ">=": GEQ}


def timestamp_table(values):
    """values: [(tag, scaled i128, has_tz)] -> (typed-value table with ids 1.., i128 side table)"""
    tv = np.zeros(len(values) + 1, dtype=TV_DTYPE)
    dec = np.zeros((len(values), 2), dtype=np.int64)
    for i, (tag, v, tz) in enumerate(values):
        u = int(v) & ((1 << 128) - 1)
        dec[i] = np.array([u & ((1 << 64) - 1), u >> 64], dtype=np.uint64).astype(np.int64)
        tv["tag"][i + 1], tv["lo"][i + 1], tv["aux"][i + 1] = tag, i, int(tz)
    return tv, dec


def surviving_rows(store, op, a_ids, b_ids):
    pb = PlanBuilder()
    desc = pb.build(pb.filter(pb.table(0, 3), EBV(op(ENC_TV(col(0)), ENC_TV(col(1)))), projection=[2]))
    cols, n, _ = store.execute(desc, [[np.asarray(a_ids, np.uint32), np.asarray(b_ids, np.uint32), np.arange(len(a_ids), dtype=np.uint32)]])
    return set(int(x) for x in cols[0][:n])


def test_time_on_timeline_known_values():
    # XSD 1.1 part 2, E.3.6: the time line's second 0 is 0001-01-01T00:00:00 (proleptic Gregorian)
    assert xsd.time_on_timeline(1, 1, 1, 0, 0, 0) == 0
    assert xsd.time_on_timeline(1970, 1, 1, 0, 0, 0, 0) == 62_135_596_800        # the Unix epoch
    assert xsd.time_on_timeline(2000, 3, 1, 0, 0, 0, 0) - xsd.time_on_timeline(2000, 2, 28, 0, 0, 0, 0) == 2 * 86400   # leap year
    assert xsd.time_on_timeline(1900, 3, 1, 0, 0, 0, 0) - xsd.time_on_timeline(1900, 2, 28, 0, 0, 0, 0) == 86400
    # the reference's Time::MIN / Time::MAX test constants (date_time.rs:327-340): midnight of the reference day
    # 1972-12-31 at +14:00 and at -14:00
    assert xsd.time_on_timeline(hour=0, tz_minutes=840) == 62_230_154_400
    assert xsd.time_on_timeline(hour=0, tz_minutes=-840) == 62_230_255_200
    assert xsd.parse_date_time("2002-04-02T12:00:00.5Z") == (xsd.parse_date_time("2002-04-02T12:00:00Z")[0] + 5 * 10 ** 17, True)
    assert xsd.parse_date("2004-12-25")[1] is False and xsd.parse_time("08:00:00+09:00")[1] is True


def test_reference_equals_and_cmp_known_answers():
    values, pairs = [], []
    for kind, a, b, rel in REFERENCE_KATS:
        parse, tag = PARSE[kind]
        values += [(tag,) + parse(a), (tag,) + parse(b)]
        pairs.append((len(values) - 1, len(values), rel))
    tv, dec = timestamp_table(values)
    st = orc.OracleStore()
    st.set_typed_values(tv, dec)
    for row, (ia, ib, rel) in enumerate(pairs):
        negated, op = rel.startswith("!") and rel != "!=", rel.lstrip("!") if rel != "!=" else "!="
        got = bool(surviving_rows(st, OPS[op], [ia], [ib]))
        assert got == (not negated), REFERENCE_KATS[row]


def python_partial_cmp(a, b):
    """date_time.rs:1617-1654 stated directly on Python ints; None = incomparable"""
    (ta, va, za), (tb, vb, zb) = a, b
    if ta != tb:
        return None
    cmp = lambda x, y: (x > y) - (x < y)
    if za == zb:
        return cmp(va, vb)
    shift = 14 * 3600 * xsd.SCALE
    fits = lambda x: -(1 << 127) <= x < (1 << 127)
    other = vb if za else va
    if not fits(other + shift) or not fits(other - shift):
        return None
    plus, minus = (cmp(va, vb + shift), cmp(va, vb - shift)) if za else (cmp(va + shift, vb), cmp(va - shift, vb))
    return plus if plus == minus else None


def random_timestamps(rng, n):
    """a few clusters so that equal values, values within 14 h of each other and the i128 edges all occur"""
    base = [xsd.parse_date_time("2008-06-20T00:00:00Z")[0], xsd.parse_date("2004-12-25")[0], xsd.parse_time("12:00:00")[0], (1 << 127) - 1, -(1 << 127)]
    out = []
    for _ in range(n):
        b = base[int(rng.integers(0, len(base)))]
        delta = int(rng.integers(-20, 21)) * 3600 * xsd.SCALE + int(rng.integers(0, 2)) * int(rng.integers(0, 10 ** 18))
        v = min(max(b + (0 if abs(b) > (1 << 126) and rng.random() < 0.5 else delta), -(1 << 127)), (1 << 127) - 1)
        out.append((int(rng.choice([abi.TV_DATE_TIME, abi.TV_DATE, abi.TV_TIME], p=[0.6, 0.2, 0.2])), v, bool(rng.integers(0, 2))))
    return out


def test_oracle_agrees_with_the_rule_on_random_values():
    rng = np.random.default_rng(14)
    values = random_timestamps(rng, 400)
    tv, dec = timestamp_table(values)
    # one integer and one null id ride along: a timestamp against anything else is an error (typed_value.rs:222-242)
    tv = np.concatenate([tv, np.zeros(1, dtype=TV_DTYPE)])
    tv["tag"][-1], tv["lo"][-1] = abi.TV_INTEGER, 5
    st = orc.OracleStore()
    st.set_typed_values(tv, dec)
    a = rng.integers(0, len(tv), 6000).astype(np.uint32)
    b = rng.integers(0, len(tv), 6000).astype(np.uint32)
    truth = {"==": lambda o: o == 0, "!=": lambda o: o != 0, "<": lambda o: o < 0, ">": lambda o: o > 0, "<=": lambda o: o <= 0, ">=": lambda o: o >= 0}
    seen_none = seen_mixed = 0
    for name, op in OPS.items():
        got = surviving_rows(st, op, a, b)
        exp = set()
        for r, (x, y) in enumerate(zip(a, b)):
            if 1 <= x <= len(values) and 1 <= y <= len(values):
                o = python_partial_cmp(values[x - 1], values[y - 1])
                seen_none += o is None
                seen_mixed += values[x - 1][2] != values[y - 1][2] and o is not None
                if o is not None and truth[name](o):
                    exp.add(r)
        assert got == exp, name
    assert seen_none > 100 and seen_mixed > 100


def test_reference_plan_snapshot_literal():
    """`10:{value:6334951680000.0000000000000000,offset:}` is how BSBM Explore - Q10 (Execution Plan).snap:26 prints
    "2008-06-20T00:00:00"^^xsd:dateTime: tag 10, Decimal digits 63349516800 then 18 zeros, no timezone."""
    v, tz = xsd.parse_date_time("2008-06-20T00:00:00")
    assert (v, tz) == (63349516800 * 10 ** 18, False)
    assert str(v) == "6334951680000" + "0" * 16
    assert abi.TV_DATE_TIME == 10


def test_bsbm_q10_oracle_equals_numpy():
    """Q10's join / FILTER pipeline (Q10 (Execution Plan).snap:12-27) in the oracle vs plain numpy on the raw triples."""
    from rdf_fusion_amd import bsbm
    ds = bsbm.generate(1500)
    st = orc.OracleStore()
    st.extend(ds.g, ds.s, ds.p, ds.o)
    st.set_typed_values(ds.typed_values, ds.decimals)
    pr, tv = ds.pred, ds.typed_values

    def objects(pname):                       # subject -> object for a functional predicate
        m = ds.p == pr[pname]
        return dict(zip(ds.s[m].tolist(), ds.o[m].tolist()))
    vendor, publisher, days, price, valid = (objects(n) for n in ("bsbm:vendor", "dc:publisher", "bsbm:deliveryDays", "bsbm:price", "bsbm:validTo"))
    country = objects("bsbm:country")
    stamp = lambda i: (int(ds.decimals[int(tv["lo"][i])][1]) << 64) | (int(ds.decimals[int(tv["lo"][i])][0]) & ((1 << 64) - 1))
    after = "2004-03-01T06:00:00"
    lim, _ = xsd.parse_date_time(after)
    total = 0
    for i in range(0, 120):
        x, c = ds.product(i), ds.country_base + i % ds.n_countries
        exp = []
        for off in ds.s[(ds.p == pr["bsbm:product"]) & (ds.o == x)].tolist():
            if vendor[off] != publisher[off] or country.get(vendor[off]) != c or int(tv["lo"][days[off]]) > 7:
                continue
            d = valid[off]
            v, tz = stamp(d), int(tv["aux"][d])
            later = v > lim if not tz else (v > lim + 14 * 3600 * xsd.SCALE and v > lim - 14 * 3600 * xsd.SCALE)
            if later:
                exp.append((off, price[off]))
        cols, n, _ = st.execute(bsbm.q10_plan(ds, x, c, max_days=7, after=after))
        assert sorted(zip(cols[0][:n].tolist(), cols[1][:n].tolist())) == sorted(exp)
        total += n
    assert total > 20
