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
    nf, n1, n2 = min(len(feats), MAX_FEATURES), min(len(o1), 2), min(len(o2), 2)
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


class BatchExchange:
    """The exchange step of a graph-sharded BATCH of Q5 instances (bench.py --gpus N, tests/test_sharding_cpu.py):
    every rank holds C(inst, X, prodFeature, origProperty1, origProperty2) — bsbm.q5_batch_const_plan — for the
    instances whose %Product% is a subject of ITS shard; all ranks need all of it.  One fixed-size int32 buffer per
    rank, zero padded, ONE all-gather: a padding row has inst = 0 = null, and a null key never joins
    (NullEqualsNothing), so the gathered buffer is bound as it is — an all-gatherv without the count exchange.
    Works on torch tensors of any device."""
    N_COLS = 5

    def __init__(self, n_instances, world, fanout_max=28, fanout_mean=18.5):
        # Hash sharding gives a rank Binomial(Q, 1/world) of a batch's instances: 10 % + 256 instances of head room is
        # > 15 standard deviations at every batch size.  C holds U{9..28} rows per instance: small batches get the
        # worst case, large ones the mean + 13 % (the sum of >= 4096 fan-outs is within 1 % of its mean).
        # pack() refuses a table that does not fit.
        self.world = world
        self.inst_cap = min(n_instances, int(n_instances / world * 1.1) + 256)
        per_inst = fanout_max if self.inst_cap <= 4096 else min(fanout_max, int(fanout_mean * 1.13) + 1)
        self.cap = self.inst_cap * per_inst                  # rows per rank
        self.buf_len = self.N_COLS * self.cap                # int32 elements per rank

    def pack(self, buf, cols, rows):
        """writes this rank's C (N_COLS int32 tensors of `rows` elements) into the zeroed send buffer, column-major"""
        if rows > self.cap:
            raise RuntimeError(f"exchange buffer too small: {rows} rows > {self.cap}")
        for k in range(self.N_COLS):
            if rows:
                buf[k * self.cap:k * self.cap + rows] = cols[k]

    def unpack(self, gathered):
        """(world * buf_len,) gathered buffer -> one contiguous (N_COLS, world * cap) tensor: a column per variable"""
        return gathered.view(self.world, self.N_COLS, self.cap).permute(1, 0, 2).contiguous().view(self.N_COLS, self.world * self.cap)
