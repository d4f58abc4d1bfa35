"""Synthetic BSBM-shaped triple stores and the reference's Q1 / Q5 physical plans.

The reference benchmarks on datasets produced by the Java ``bsbm-tools`` generator
(bench/src/benchmarks/bsbm/requirements.rs:9-44), which cannot run here (no Java, no network).
This module generates a store of the same SHAPE (≈350 triples per product,
productFeature fan-out U{9..28}, integer numeric properties ~N(1000,333) clipped to 1..2000,
20 offers and 10 reviews per product) with dense object ids from 1, the default graph only, and a
typed-value table (tag 9 xsd:integer for numerics, tag 6 doubles for prices, tag 3 strings,
tag 1 IRIs), deterministically from a seed.

``q1_plan`` / ``q5_plan`` build the operator trees the reference's planner produces for
bench/tests/query_results/queries/explore-q{1,5}.sparql, transcribed from
bench/tests/plans/snapshots/*Q1 / Q5 (Execution Plan).snap (DataFusion's dynamic-filter push-down,
a superset filter on raw ids, is not reproduced: it never changes results).
"""
from dataclasses import dataclass, field

import numpy as np

from . import abi, xsd
from .engine import TV_DTYPE
from .plan import (PlanBuilder, quad_pattern, col, lit_id, integer, date_time, ENC_TV, GT, LT, LEQ, ADD, SUB, EBV, AND,
                   ID_NEQ)

PREDICATES = ["rdf:type", "rdfs:label", "rdfs:comment", "bsbm:producer", "bsbm:productFeature",
              "bsbm:productPropertyNumeric1", "bsbm:productPropertyNumeric2", "bsbm:productPropertyNumeric3",
              "bsbm:productPropertyNumeric4", "bsbm:productPropertyNumeric5",
              "bsbm:productPropertyTextual1", "bsbm:productPropertyTextual2", "bsbm:productPropertyTextual3",
              "bsbm:productPropertyTextual4", "bsbm:productPropertyTextual5", "dc:publisher", "dc:date",
              "bsbm:product", "bsbm:vendor", "bsbm:price", "bsbm:validFrom", "bsbm:validTo",
              "bsbm:deliveryDays", "bsbm:offerWebpage", "bsbm:reviewFor", "rev:reviewer", "dc:title",
              "rev:text", "bsbm:rating1", "bsbm:rating2", "bsbm:rating3", "bsbm:rating4", "bsbm:reviewDate",
              "bsbm:country"]


@dataclass
class BsbmDataset:
    n_products: int
    g: np.ndarray
    s: np.ndarray
    p: np.ndarray
    o: np.ndarray
    typed_values: np.ndarray          # TV_DTYPE, index = object id
    pred: dict                        # predicate name -> object id
    product_base: int                 # products are ids product_base .. product_base + n_products - 1
    feature_base: int
    n_features: int
    type_base: int
    n_types: int
    int_base: int                     # integer literal v (1..2000) has id int_base + v - 1
    n_ids: int
    class_ids: dict = field(default_factory=dict)
    decimals: np.ndarray = None       # (n, 2) int64: the i128 side table of the typed values (timestamps of the date literals)
    country_base: int = 0             # vendor countries are ids country_base .. country_base + n_countries - 1
    n_countries: int = 0

    @property
    def n_triples(self):
        return len(self.s)

    def product(self, i):
        return self.product_base + int(i)


def generate(n_products, seed=None, offers_per_product=20, reviews_per_product=10):
    """Returns a BsbmDataset with ≈350 * n_products triples."""
    P = int(n_products)
    rng = np.random.default_rng(P if seed is None else seed)
    next_id = [1]

    def take(n):
        b = next_id[0]
        next_id[0] += int(n)
        return b

    pred = {name: take(1) for name in PREDICATES}
    cls = {name: take(1) for name in ["bsbm:Product", "bsbm:Offer", "rev:Review", "bsbm:Producer", "bsbm:Vendor"]}
    n_leaf = max(4, P // 150)
    n_types = n_leaf + max(1, n_leaf // 4) + max(1, n_leaf // 16) + 1     # leaf, two inner levels, root
    type_base = take(n_types)
    n_features = int(np.clip(P // 6, 20, 47_000))
    feature_base = take(n_features)
    n_producers = max(1, P // 50)
    producer_base = take(n_producers)
    n_vendors = max(1, P // 100)
    vendor_base = take(n_vendors)
    n_reviewers = max(1, P // 2)
    reviewer_base = take(n_reviewers)
    product_base = take(P)
    n_offers = P * offers_per_product
    offer_base = take(n_offers)
    n_reviews = P * reviews_per_product
    review_base = take(n_reviews)
    int_base = take(2000)                 # "1"^^xsd:integer .. "2000"^^xsd:integer
    n_words = 20_000
    word_base = take(n_words)             # shared textual literals
    label_base = take(P)                  # one rdfs:label literal per product
    n_prices = 100_000
    price_base = take(n_prices)           # xsd:double price literals
    n_dates = 4000
    date_base = take(n_dates)             # xsd:date literals: 2000-01-01 + i days (dc:date)
    datetime_base = take(n_dates)         # xsd:dateTime literals (validFrom / validTo / reviewDate), every 7th with a timezone
    n_countries = 10
    country_base = take(n_countries)
    n_ids = next_id[0]

    S, Pc, O = [], [], []

    def emit(s, p, o):
        s = np.asarray(s, dtype=np.uint32)
        S.append(s)
        Pc.append(np.full(len(s), p, dtype=np.uint32))
        O.append(np.asarray(o, dtype=np.uint32))

    prod = np.arange(product_base, product_base + P, dtype=np.uint32)
    # --- products ---------------------------------------------------------------------------------
    emit(prod, pred["rdfs:label"], label_base + np.arange(P))
    emit(prod, pred["rdfs:comment"], word_base + rng.integers(0, n_words, P))
    leaf = rng.integers(0, n_leaf, P)
    lvl1 = n_leaf + leaf // 4 % max(1, n_leaf // 4)
    lvl2 = n_leaf + max(1, n_leaf // 4) + leaf // 16 % max(1, n_leaf // 16)
    root = np.full(P, n_types - 1)
    for t in (leaf, lvl1, lvl2, root):     # forward-chained type ancestors (`-fc`, requirements.rs:28)
        emit(prod, pred["rdf:type"], type_base + t)
    emit(prod, pred["bsbm:producer"], producer_base + rng.integers(0, n_producers, P))
    fan = rng.integers(9, 29, P)                                  # U{9..28}
    fs = np.repeat(prod, fan)
    fo = feature_base + rng.integers(0, n_features, fan.sum())
    emit(fs, pred["bsbm:productFeature"], fo)                     # duplicates collapse in the store (set semantics)
    for k, prob in ((1, 1.0), (2, 1.0), (3, 1.0), (4, 0.7), (5, 0.5)):
        keep = rng.random(P) < prob
        v = np.clip(np.rint(rng.normal(1000, 333, P)), 1, 2000).astype(np.int64)
        emit(prod[keep], pred[f"bsbm:productPropertyNumeric{k}"], int_base + v[keep] - 1)
    for k, prob in ((1, 1.0), (2, 1.0), (3, 1.0), (4, 0.7), (5, 0.8)):
        keep = rng.random(P) < prob
        emit(prod[keep], pred[f"bsbm:productPropertyTextual{k}"], word_base + rng.integers(0, n_words, keep.sum()))
    emit(prod, pred["dc:publisher"], producer_base + rng.integers(0, n_producers, P))
    emit(prod, pred["dc:date"], date_base + rng.integers(0, n_dates, P))
    # --- offers -----------------------------------------------------------------------------------
    off = np.arange(offer_base, offer_base + n_offers, dtype=np.uint32)
    emit(off, pred["rdf:type"], np.full(n_offers, cls["bsbm:Offer"]))
    emit(off, pred["bsbm:product"], np.repeat(prod, offers_per_product))
    ov = vendor_base + rng.integers(0, n_vendors, n_offers)
    emit(off, pred["bsbm:vendor"], ov)
    emit(off, pred["bsbm:price"], price_base + rng.integers(0, n_prices, n_offers))
    emit(off, pred["bsbm:validFrom"], datetime_base + rng.integers(0, n_dates, n_offers))
    emit(off, pred["bsbm:validTo"], datetime_base + rng.integers(0, n_dates, n_offers))
    emit(off, pred["bsbm:deliveryDays"], int_base + rng.integers(1, 22, n_offers) - 1)
    emit(off, pred["bsbm:offerWebpage"], word_base + rng.integers(0, n_words, n_offers))
    emit(off, pred["dc:publisher"], ov)
    emit(off, pred["dc:date"], date_base + rng.integers(0, n_dates, n_offers))
    # --- reviews ----------------------------------------------------------------------------------
    rev = np.arange(review_base, review_base + n_reviews, dtype=np.uint32)
    emit(rev, pred["rdf:type"], np.full(n_reviews, cls["rev:Review"]))
    emit(rev, pred["bsbm:reviewFor"], np.repeat(prod, reviews_per_product))
    rr = reviewer_base + rng.integers(0, n_reviewers, n_reviews)
    emit(rev, pred["rev:reviewer"], rr)
    emit(rev, pred["dc:title"], word_base + rng.integers(0, n_words, n_reviews))
    emit(rev, pred["rev:text"], word_base + rng.integers(0, n_words, n_reviews))
    for k in (1, 2, 3, 4):
        keep = rng.random(n_reviews) < 0.7
        emit(rev[keep], pred[f"bsbm:rating{k}"], int_base + rng.integers(1, 11, keep.sum()) - 1)
    emit(rev, pred["bsbm:reviewDate"], datetime_base + rng.integers(0, n_dates, n_reviews))
    emit(rev, pred["dc:publisher"], rr)
    emit(rev, pred["dc:date"], date_base + rng.integers(0, n_dates, n_reviews))

    # --- vendors (their own generator: the draws above stay what they were) -----------------------
    rng_v = np.random.default_rng([P, 1])
    emit(np.arange(vendor_base, vendor_base + n_vendors), pred["bsbm:country"], country_base + rng_v.integers(0, n_countries, n_vendors))

    s = np.concatenate(S)
    p = np.concatenate(Pc)
    o = np.concatenate(O)
    g = np.zeros(len(s), dtype=np.uint32)      # default graph only
    del S, Pc, O

    # --- typed-value table (index = object id) ----------------------------------------------------
    tv = np.zeros(n_ids, dtype=TV_DTYPE)
    tv["tag"][1:] = abi.TV_NAMED_NODE           # IRIs unless overwritten below
    tv["lo"][1:] = np.arange(1, n_ids)          # rank of the IRI in str order (any total order works)
    ints = slice(int_base, int_base + 2000)
    tv["tag"][ints] = abi.TV_INTEGER
    tv["lo"][ints] = np.arange(1, 2001)
    for base, n in ((word_base, n_words), (label_base, P)):
        sl = slice(base, base + n)
        tv["tag"][sl] = abi.TV_STRING           # simple literals; lo = rank of the lexical form
        tv["lo"][sl] = np.arange(n)
        tv["aux"][sl] = 0
    pr = slice(price_base, price_base + n_prices)
    tv["tag"][pr] = abi.TV_DOUBLE
    tv["lo"][pr] = (rng.random(n_prices) * 10_000.0).view(np.int64)
    # dates: Timestamp = timeOnTimeline seconds * 10^18 in the i128 side table + "has a timezone" (include/rdfgpu.h)
    day0 = xsd.time_on_timeline(2000, 1, 1)
    stamps = []
    for i in range(n_dates):                                     # "2000-01-01" + i days, no timezone
        stamps.append((int(day0 + 86400 * i) * xsd.SCALE, 0))
    for i in range(n_dates):                                     # T06:00:00 local; every 7th one is ...T06:00:00+02:00
        tz = i % 7 == 3
        stamps.append((int(day0 + 86400 * i + 6 * 3600 - (7200 if tz else 0)) * xsd.SCALE, int(tz)))
    dec = np.zeros((len(stamps), 2), dtype=np.int64)
    for i, (v, _) in enumerate(stamps):
        dec[i] = np.array([v & ((1 << 64) - 1), v >> 64], dtype=np.uint64).astype(np.int64)
    for base, tag, first in ((date_base, abi.TV_DATE, 0), (datetime_base, abi.TV_DATE_TIME, n_dates)):
        tv["tag"][base:base + n_dates] = tag
        tv["lo"][base:base + n_dates] = first + np.arange(n_dates)
        tv["aux"][base:base + n_dates] = [z for _, z in stamps[first:first + n_dates]]

    return BsbmDataset(P, g, s, p, o, tv, pred, product_base, feature_base, n_features, type_base, n_types,
                       int_base, n_ids, cls, dec, country_base, n_countries)


# --------------------------------------------------------------------------------------------------
# The reference's physical plans (bench/tests/plans/snapshots/*.snap)
# --------------------------------------------------------------------------------------------------
def q5_plan(ds, product_id, w1=120, w2=170, topk=False, builder=False):
    """BSBM Explore Q5 ("similar products"), 7 triple patterns, as planned by the reference:
    J3(J2(J1(label x features(X), productFeature), numeric1), numeric2) — Q5 (Execution Plan).snap:10-30.
    Output: (product, productLabel) bindings before DISTINCT / ORDER BY / LIMIT."""
    X = int(product_id)
    pr = ds.pred
    pb = PlanBuilder()
    not_x = lambda: ID_NEQ(col(0), lit_id(X))                       # FilterExec: product@0 != <object id>
    label = pb.filter(pb.data_source(quad_pattern("product", pr["rdfs:label"], "productLabel")), not_x())
    x_feat = pb.data_source(quad_pattern(X, pr["bsbm:productFeature"], "prodFeature"))
    c1 = pb.cross_join(label, x_feat)                               # product, productLabel, prodFeature
    pf = pb.filter(pb.data_source(quad_pattern("product", pr["bsbm:productFeature"], "prodFeature")), not_x())
    j1 = pb.hash_join(c1, pf, on=[(2, 1), (0, 0)], projection=[0, 1])
    node = j1
    for k, w in ((1, w1), (2, w2)):
        x_num = pb.data_source(quad_pattern(X, pr[f"bsbm:productPropertyNumeric{k}"], f"origProperty{k}"))
        cx = pb.cross_join(node, x_num)                             # product, productLabel, origPropertyK
        sim = pb.filter(pb.data_source(quad_pattern("product", pr[f"bsbm:productPropertyNumeric{k}"], f"simProperty{k}")), not_x())
        # EBV(LT(ENC_TV(sim), ADD(ENC_TV(orig), 9:w))) AND EBV(GT(ENC_TV(sim), SUB(ENC_TV(orig), 9:w)))
        flt = AND(EBV(LT(ENC_TV(col(4)), ADD(ENC_TV(col(2)), integer(w)))),
                  EBV(GT(ENC_TV(col(4)), SUB(ENC_TV(col(2)), integer(w)))))
        node = pb.hash_join(cx, sim, on=[(0, 0)], filter=flt, projection=[0, 1])
    if builder:              # (PlanBuilder, root) for plan display: plan.explain
        return pb, node
    return pb.build(q5_topk(pb, node, False) if topk else node)


def _window(sim, orig, w):
    """EBV(LT(ENC_TV(sim), ADD(ENC_TV(orig), 9:w))) AND EBV(GT(ENC_TV(sim), SUB(ENC_TV(orig), 9:w)))"""
    return AND(EBV(LT(ENC_TV(col(sim)), ADD(ENC_TV(col(orig)), integer(w)))),
               EBV(GT(ENC_TV(col(sim)), SUB(ENC_TV(col(orig)), integer(w)))))


def _q5_batch_constants(pb, ds, params):
    """C(inst, X, prodFeature, origProperty1, origProperty2): the three constant-subject patterns of every instance
    of PARAMS(inst, X).  `PARAMS JOIN (?s p ?v) ON X = ?s` is what B separate `<X> p ?v` scans return, tagged with
    the instance.  The three patterns share the subject, so after the first join the other two are look-ups by X on
    rows that already carry it (inner joins commute: the same C as joining three per-instance tables on inst) — index
    joins on the store's predicate slices, fused into one kernel on re-execution."""
    pr = ds.pred
    scan = lambda pname: pb.data_source(quad_pattern("s", pr[pname], "v"))           # (s, v)
    c = pb.hash_join(params, scan("bsbm:productFeature"), on=[(1, 0)], projection=[0, 1, 3])             # (inst, X, f)
    c = pb.hash_join(c, scan("bsbm:productPropertyNumeric1"), on=[(1, 0)], projection=[0, 1, 2, 4])      # + orig1
    return pb.hash_join(c, scan("bsbm:productPropertyNumeric2"), on=[(1, 0)], projection=[0, 1, 2, 3, 5])  # + orig2


def q5_batch_const_plan(ds):
    """Phase A of a graph-sharded batched Q5: table 0 = PARAMS(inst, X), one row per query instance (inst = 1-based
    position in the batch, X = %Product%); output C(inst, X, prodFeature, origProperty1, origProperty2) for the
    instances whose %Product% is a subject of THIS shard (all three patterns have the subject X, so an instance's
    constants live on one shard).  The union of the shards' C tables is the C of the whole graph."""
    pb = PlanBuilder()
    return pb.build(_q5_batch_constants(pb, ds, pb.table(0, 2)))


def q5_batch_plan(ds, w1=120, w2=170, tables=None, topk=False, graph="default"):
    """BSBM Q5 for a BATCH of instances in one operator tree (shared scans: every triple-pattern partition
    is streamed once per batch instead of once per query).  Same operators as q5_plan; the per-instance
    constant becomes a column: the table C(inst, X, prodFeature, origProperty1, origProperty2) is either bound
    (slot 0 — the graph-sharded case, after the all-gather of q5_batch_const_plan's outputs) or, when `tables` is
    None, computed in-plan from PARAMS(inst, X) bound at slot 0.
    `product != X` is applied once, where the instance meets the product (all later joins are on product).
    Inner joins commute, so the batch is free to order them by cost: constants first, then the candidate
    join, then the two selective numeric windows, and the 1:1 label lookup last.
    `graph`: the active graph of the product-side patterns (sharding.shard_dataset_hybrid keeps them in a named graph).
    Output: (inst, product, productLabel) — per instance exactly q5_plan's bindings (as a multiset)."""
    pr = ds.pred
    pb = PlanBuilder()
    c = _q5_batch_constants(pb, ds, pb.table(0, 2)) if tables is None else pb.table(0, 5)
    quad = lambda s_, p_, o_: quad_pattern(s_, p_, o_, graph=graph)
    pf = pb.data_source(quad("product", pr["bsbm:productFeature"], "prodFeature"))      # (product, prodFeature)
    # candidates: JOIN (product, f) ON f, product != X   ->  (inst, product, orig1, orig2)
    node = pb.hash_join(c, pf, on=[(2, 1)], filter=ID_NEQ(col(5), col(1)), projection=[0, 5, 3, 4])
    # the two numeric windows cut the candidates down before anything else is looked up
    for k, w, orig in ((1, w1, 2), (2, w2, 3)):
        sim = pb.data_source(quad("product", pr[f"bsbm:productPropertyNumeric{k}"], f"simProperty{k}"))
        node = pb.hash_join(node, sim, on=[(1, 0)], filter=_window(5, orig, w), projection=[0, 1, 2, 3])
    label = pb.data_source(quad("product", pr["rdfs:label"], "productLabel"))           # (product, label)
    out = pb.hash_join(node, label, on=[(1, 0)], projection=[0, 1, 5])                           # (inst, product, label)
    return pb.build(q5_topk(pb, out, True) if topk else out)


def q5_topk(pb_or_desc_builder, node, batched):
    """The operators above the join pipeline in the reference's Q5 plan (Q5 (Execution Plan).snap:5-9): DISTINCT
    (AggregateExec gby = sort keys, first_value) + SortExec TopK(fetch=5) ORDER BY ENC_SORT(ENC_PT(productLabel)) ASC,
    product ASC — per instance when `batched` (input columns (inst, product, label)), else over (product, label)."""
    pb = pb_or_desc_builder
    if batched:
        return pb.topk(node, keys=[(2, abi.SORT_BY_TERM), (1, abi.SORT_BY_ID)], limit=5, group=0)
    return pb.topk(node, keys=[(1, abi.SORT_BY_TERM), (0, abi.SORT_BY_ID)], limit=5)


def q1_plan(ds, type_id, feature1, feature2, threshold, builder=False):
    """BSBM Explore Q1: four chained single-key hash joins on ?product and the numeric FILTER
    (Q1 (Execution Plan).snap:10-19).  Output: (product, label)."""
    pr = ds.pred
    pb = PlanBuilder()
    label = pb.data_source(quad_pattern("product", pr["rdfs:label"], "label"))
    ptype = pb.data_source(quad_pattern("product", pr["rdf:type"], int(type_id)))
    j = pb.hash_join(label, ptype, on=[(0, 0)], projection=[0, 1])
    for f in (feature1, feature2):
        pf = pb.data_source(quad_pattern("product", pr["bsbm:productFeature"], int(f)))
        j = pb.hash_join(j, pf, on=[(0, 0)], projection=[0, 1])
    num = pb.data_source(quad_pattern("product", pr["bsbm:productPropertyNumeric1"], "value1"))
    flt = pb.filter(num, EBV(GT(ENC_TV(col(1)), integer(threshold))), projection=[0])
    j = pb.hash_join(j, flt, on=[(0, 0)], projection=[0, 1])
    if builder:
        return pb, j
    return pb.build(j)


def q1_scan_filter_plan(ds, threshold):
    """BASELINE config 2: the single-pattern scan + numeric FILTER of Q1
    (FilterExec: EBV(GT(ENC_TV(value1@1), 9:c)), projection=[product@0] over the GPOS slice)."""
    pb = PlanBuilder()
    num = pb.data_source(quad_pattern("product", ds.pred["bsbm:productPropertyNumeric1"], "value1"))
    return pb.build(pb.filter(num, EBV(GT(ENC_TV(col(1)), integer(threshold))), projection=[0]))


def q1_instance(ds, rng):
    """Query constants drawn so the query has answers: type/features of a random product."""
    i = int(rng.integers(0, ds.n_products))
    x = ds.product(i)
    sel = ds.s == x
    types = ds.o[sel & (ds.p == ds.pred["rdf:type"])]
    feats = np.unique(ds.o[sel & (ds.p == ds.pred["bsbm:productFeature"])])
    f = rng.choice(feats, size=2, replace=False) if len(feats) >= 2 else np.array([feats[0], feats[0]])
    return int(types.min()), int(f[0]), int(f[1]), int(rng.integers(1, 501))


def q10_plan(ds, product_id, country_id, max_days=3, after="2008-06-20T00:00:00", topk=False):
    """BSBM Explore Q10 below its DISTINCT / ORDER BY: six chained hash joins on ?offer / ?vendor with the FilterExecs
    `EBV(LEQ(ENC_TV(deliveryDays), 9:3))` and `EBV(GT(ENC_TV(date), 10:{value:6334951680000.0000000000000000,offset:}))`
    (BSBM Explore - Q10 (Execution Plan).snap:12-27; the dateTime literal is timeOnTimeline("2008-06-20T00:00:00") =
    63349516800 s, no timezone).  Output: (offer, price)."""
    pr = ds.pred
    pb = PlanBuilder()
    node = pb.hash_join(pb.data_source(quad_pattern("offer", pr["bsbm:product"], int(product_id))),
                        pb.data_source(quad_pattern("offer", pr["bsbm:vendor"], "vendor")), on=[(0, 0)], projection=[0, 2])
    node = pb.hash_join(node, pb.data_source(quad_pattern("offer", pr["dc:publisher"], "vendor")), on=[(0, 0), (1, 1)], projection=[0, 1])
    node = pb.hash_join(node, pb.data_source(quad_pattern("vendor", pr["bsbm:country"], int(country_id))), on=[(1, 0)], projection=[0])
    days = pb.filter(pb.data_source(quad_pattern("offer", pr["bsbm:deliveryDays"], "deliveryDays")),
                     EBV(LEQ(ENC_TV(col(1)), integer(max_days))), projection=[0])
    node = pb.hash_join(node, days, on=[(0, 0)], projection=[0])
    node = pb.hash_join(node, pb.data_source(quad_pattern("offer", pr["bsbm:price"], "price")), on=[(0, 0)], projection=[0, 2])
    valid = pb.filter(pb.data_source(quad_pattern("offer", pr["bsbm:validTo"], "date")),
                      EBV(GT(ENC_TV(col(1)), date_time(*xsd.parse_date_time(after)))), projection=[0])
    node = pb.hash_join(node, valid, on=[(0, 0)], projection=[0, 1])
    if topk:   # DISTINCT + ORDER BY xsd:double(str(?price)), ?offer, ?price LIMIT 10  (Q10 (Execution Plan).snap:6-8)
        node = pb.topk(node, keys=[(1, abi.SORT_BY_DOUBLE), (0, abi.SORT_BY_ID), (1, abi.SORT_BY_ID)], limit=10)
    return pb.build(node)


def q4_plan(ds, type_id, feature1, feature2, feature3, threshold1, threshold2, topk=False):
    """BSBM Explore Q4 below its DISTINCT / ORDER BY / OFFSET: a UnionExec of two five-join pipelines that differ in the
    second feature and the numeric property filtered (`EBV(GT(ENC_TV(p1), 9:457))` / `EBV(GT(ENC_TV(p2), 9:488))`),
    Q4 (Execution Plan).snap:11-38.  Output: (product, label, propertyTextual)."""
    pr = ds.pred
    pb = PlanBuilder()

    def branch(feature_b, k, threshold):
        node = pb.hash_join(pb.data_source(quad_pattern("product", pr["rdfs:label"], "label")),
                            pb.data_source(quad_pattern("product", pr["rdf:type"], int(type_id))), on=[(0, 0)], projection=[0, 1])
        for f in (feature1, feature_b):
            node = pb.hash_join(node, pb.data_source(quad_pattern("product", pr["bsbm:productFeature"], int(f))), on=[(0, 0)], projection=[0, 1])
        node = pb.hash_join(node, pb.data_source(quad_pattern("product", pr["bsbm:productPropertyTextual1"], "propertyTextual")),
                            on=[(0, 0)], projection=[0, 1, 3])
        num = pb.filter(pb.data_source(quad_pattern("product", pr[f"bsbm:productPropertyNumeric{k}"], f"p{k}")),
                        EBV(GT(ENC_TV(col(1)), integer(threshold))), projection=[0])
        return pb.hash_join(node, num, on=[(0, 0)], projection=[0, 1, 2])
    node = pb.union(branch(feature2, 1, threshold1), branch(feature3, 2, threshold2))
    if topk:    # AggregateExec(gby = the three sort keys, first_value) + SortExec TopK(fetch=15): ORDER BY label, product, propertyTextual
        node = pb.topk(node, keys=[(1, abi.SORT_BY_TERM), (0, abi.SORT_BY_ID), (2, abi.SORT_BY_ID)], limit=15)
    return pb.build(node)
