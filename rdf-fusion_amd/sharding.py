"""Graph shards across the GPUs of one node.

Triples are partitioned by hash(subject) mod world_size: every variable-subject pattern on the
star variable (?product in BSBM Q1/Q5) is co-partitioned, so all product = product joins are local.
Constant-subject patterns (<ProductX> p ?v, 1–30 rows) live on one shard only; their bindings are
all-gathered (counts first, then padded rows — an all-gatherv) so that every rank can run the rest
of the plan locally with those bindings bound as RDFGPU_NODE_TABLE inputs.  Object ids are global
(assigned once by the host dictionary) and the typed-value table is replicated.

The reference has no multi-process code at all; this file has no counterpart there.
"""
import numpy as np

from . import abi
from .plan import PlanBuilder, quad_pattern, col, lit_id, integer, ENC_TV, GT, LT, ADD, SUB, EBV, AND, ID_NEQ

MAX_FEATURES = 56          # BSBM productFeature fan-out is U{9..28}; padded exchange record
RECORD = 64                # u32 per query: [n_feat, n_o1, n_o2, pad, feats[56], o1[2], o2[2]]


def shard_of(subject_ids, world_size):
    """hash(subject) mod G — a multiplicative hash so that dense id ranges spread evenly."""
    x = np.asarray(subject_ids, dtype=np.uint64)
    h = (x * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)
    return (h % np.uint64(world_size)).astype(np.int64)


def shard_dataset(ds, rank, world_size):
    """Columns (g, s, p, o) of this rank's shard."""
    if world_size == 1:
        return ds.g, ds.s, ds.p, ds.o
    keep = shard_of(ds.s, world_size) == rank
    return ds.g[keep], ds.s[keep], ds.p[keep], ds.o[keep]


def candidate_graph(ds):
    """The named graph of shard_dataset_hybrid's second copy (an id beyond the dictionary: no term, no typed value)."""
    return int(ds.n_ids) + 7


def shard_dataset_hybrid(ds, rank, world_size):
    """The layout of the sharded batched Q5 (bench.py --gpus N).  Default graph: this rank's SUBJECT shard of every triple
    (star joins on ?product and the constant-subject patterns <X> p ?v are local).  Named graph candidate_graph(ds): what
    the candidate join and its star look-ups read, laid out for THAT join — the productFeature triples whose OBJECT
    (the feature, the join key) hashes to this rank, and all triples of the three 1:1 star predicates the candidate
    filter reads (numeric1, numeric2, label: 0.9 % of BSBM's triples, replicated).  With it the C table is re-sharded by
    feature (rdfgpu_exchange_repartition) and BOTH sides of the candidate join shrink with the number of ranks; with the
    subject shard alone every rank has to look at every row of C."""
    g, s, p, o = shard_dataset(ds, rank, world_size)
    pr = ds.pred
    pf = ds.p == pr["bsbm:productFeature"]
    if world_size > 1:
        pf = pf & (shard_of(ds.o, world_size) == rank)
    star = (ds.p == pr["bsbm:productPropertyNumeric1"]) | (ds.p == pr["bsbm:productPropertyNumeric2"]) | (ds.p == pr["rdfs:label"])
    keep = pf | star
    g2 = np.full(int(keep.sum()), candidate_graph(ds), dtype=np.uint32)
    return (np.concatenate([g, g2]), np.concatenate([s, ds.s[keep]]), np.concatenate([p, ds.p[keep]]), np.concatenate([o, ds.o[keep]]))


def q5_const_plans(ds, product_id):
    """Three single-pattern plans for the constant-subject side of Q5."""
    X = int(product_id)
    pr = ds.pred
    out = []
    for pname, var in (("bsbm:productFeature", "prodFeature"), ("bsbm:productPropertyNumeric1", "origProperty1"),
                       ("bsbm:productPropertyNumeric2", "origProperty2")):
        pb = PlanBuilder()
        out.append(pb.build(pb.data_source(quad_pattern(X, pr[pname], var))))
    return out


def q5_local_plan(ds, product_id, w1=120, w2=170):
    """Q5 with the constant-subject scans replaced by bound tables 0 (features), 1 (orig1), 2 (orig2):
    the same operator tree as bsbm.q5_plan otherwise."""
    X = int(product_id)
    pr = ds.pred
    pb = PlanBuilder()
    not_x = lambda: ID_NEQ(col(0), lit_id(X))
    label = pb.filter(pb.data_source(quad_pattern("product", pr["rdfs:label"], "productLabel")), not_x())
    c1 = pb.cross_join(label, pb.table(0, 1))
    pf = pb.filter(pb.data_source(quad_pattern("product", pr["bsbm:productFeature"], "prodFeature")), not_x())
    node = pb.hash_join(c1, pf, on=[(2, 1), (0, 0)], projection=[0, 1])
    for k, w in ((1, w1), (2, w2)):
        cx = pb.cross_join(node, pb.table(k, 1))
        sim = pb.filter(pb.data_source(quad_pattern("product", pr[f"bsbm:productPropertyNumeric{k}"], f"simProperty{k}")), not_x())
        flt = AND(EBV(LT(ENC_TV(col(4)), ADD(ENC_TV(col(2)), integer(w)))),
                  EBV(GT(ENC_TV(col(4)), SUB(ENC_TV(col(2)), integer(w)))))
        node = pb.hash_join(cx, sim, on=[(0, 0)], filter=flt, projection=[0, 1])
    return pb.build(node)


def pack_record(feats, o1, o2):
    rec = np.zeros(RECORD, dtype=np.int32)
    if len(feats) > MAX_FEATURES or len(o1) > 2 or len(o2) > 2:    # never clip silently: the batched flow (all-gatherv sized from counts) has no such limit
        raise ValueError(f"fixed-size exchange record: {len(feats)} features / {len(o1)} / {len(o2)} values do not fit ({MAX_FEATURES} / 2 / 2)")
    nf, n1, n2 = len(feats), len(o1), len(o2)
    rec[0], rec[1], rec[2] = nf, n1, n2
    rec[4:4 + nf] = np.asarray(feats[:nf], dtype=np.uint32).view(np.int32)
    rec[60:60 + n1] = np.asarray(o1[:n1], dtype=np.uint32).view(np.int32)
    rec[62:62 + n2] = np.asarray(o2[:n2], dtype=np.uint32).view(np.int32)
    return rec


def unpack_records(gathered):
    """gathered: [world, Q, RECORD] int32 -> per query (feats, o1, o2) from whichever ranks hold rows
    (a product's triples live on exactly one shard, so at most one rank contributes)."""
    g = np.asarray(gathered).astype(np.int32)
    world, q, _ = g.shape
    out = []
    for i in range(q):
        feats, o1, o2 = [], [], []
        for r in range(world):
            rec = g[r, i]
            feats.extend(rec[4:4 + rec[0]].view(np.uint32).tolist())
            o1.extend(rec[60:60 + rec[1]].view(np.uint32).tolist())
            o2.extend(rec[62:62 + rec[2]].view(np.uint32).tolist())
        out.append((np.array(feats, np.uint32), np.array(o1, np.uint32), np.array(o2, np.uint32)))
    return out


def run_q5_batch_sharded(ds, products, run_const, run_local, all_gather, pmap=None):
    """One step of the sharded Q5 batch.

    run_const(desc) -> 1-column numpy result of a tiny constant-subject plan on this rank
    run_local(desc, [feats, o1, o2]) -> number of bindings the local plan produced on this rank
    all_gather(int32 array [Q, RECORD]) -> int32 array [world, Q, RECORD]
    pmap(fn, items) -> list: optional parallel map (host threads; every plan owns its own HIP stream)
    Returns the number of bindings this rank produced."""
    pmap = pmap or (lambda fn, items: [fn(i) for i in items])
    recs = np.zeros((len(products), RECORD), dtype=np.int32)

    def phase_a(i):
        f, a, b = [run_const(d) for d in q5_const_plans(ds, products[i])]
        recs[i] = pack_record(f, a, b)
    pmap(phase_a, range(len(products)))
    gathered = all_gather(recs)
    tables = unpack_records(gathered)
    return sum(pmap(lambda i: run_local(q5_local_plan(ds, products[i]), list(tables[i])), range(len(products))))


# --------------------------------------------------------------------------------------------------
# Staged execution over graph shards.  A "table" is whatever the executor and the exchange agree on: numpy columns with
# a CPU executor and torch.distributed (the CPU tests), (device pointers, rows) with the HIP library and rdfgpu_comm (bench.py).
# --------------------------------------------------------------------------------------------------
def run_q5_batch_sharded_tables(ds, params, execute, allgatherv):
    """One graph-sharded step of a BATCH of BSBM Q5 instances (bench.py --gpus N):
    phase A   C(inst, X, prodFeature, origProperty1, origProperty2) for the instances whose %Product% lives on THIS shard
              (bsbm.q5_batch_const_plan over the local shard: all three constant-subject patterns share the subject);
    exchange  all-gatherv of C — every rank needs every instance's constants; sizes come from the row counts, nothing padded;
    phase B   the batch's join / FILTER pipeline over the local shard of the product-side patterns, probing with all of C.
    `execute(desc, [table, ...]) -> table`, `allgatherv(table) -> table`.  Returns this rank's bindings."""
    from . import bsbm
    c_local = execute(bsbm.q5_batch_const_plan(ds), [params])
    c_all = allgatherv(c_local)
    return execute(bsbm.q5_batch_plan(ds, tables=True), [c_all])


def run_q5_batch_hybrid(ds, params, execute, repartition):
    """The same step over shard_dataset_hybrid's layout: phase A as above (default graph = subject shard); the exchange is a
    hash REPARTITION of C by prodFeature (column 2) — a rank receives the rows of the features it holds the candidate
    triples of; phase B reads the named graph.  The union of the ranks' bindings is the unsharded answer: a binding
    belongs to exactly one (instance, shared feature, product) triple, and a feature lives on exactly one rank."""
    from . import bsbm
    c_local = execute(bsbm.q5_batch_const_plan(ds), [params])
    c_mine = repartition(c_local, 2)
    return execute(bsbm.q5_batch_plan(ds, tables=True, graph=[candidate_graph(ds)]), [c_mine])


def run_stages(stages, execute, repartition):
    """Plans chained through hash repartitions (lubm.q9_sharded_stages): stage k's output, re-sharded by `key column`, is
    stage k + 1's bound table 0.  `repartition(table, key_col) -> table`."""
    table = None
    for desc, key_col in stages:
        table = execute(desc, [] if table is None else [table])
        if key_col is not None:
            table = repartition(table, key_col)
    return table


class NumpyExchange:
    """The two exchange steps on numpy tables over a torch.distributed process group (gloo in the CPU tests): the
    protocol of exchange.hip — counts first, then the rows, rank order — with pickled objects as the wire."""

    def __init__(self, dist, world, rank):
        self.dist, self.world, self.rank = dist, world, rank

    def allgatherv(self, cols):
        mine = [np.ascontiguousarray(c, dtype=np.uint32) for c in cols]
        got = [None] * self.world
        self.dist.all_gather_object(got, mine)
        return [np.concatenate([g[k] for g in got]) for k in range(len(mine))]

    def repartition(self, cols, key_col):
        cols = [np.ascontiguousarray(c, dtype=np.uint32) for c in cols]
        dest = shard_of(cols[key_col], self.world)
        blocks = [[c[dest == d] for c in cols] for d in range(self.world)]
        got = [None] * self.world
        self.dist.all_gather_object(got, blocks)          # got[r][d] = what rank r sends to rank d
        return [np.concatenate([got[r][self.rank][k] for r in range(self.world)]) for k in range(len(cols))]
