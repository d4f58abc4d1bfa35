"""A tuned columnar CPU baseline for the batched BSBM Q5 — test / measurement infrastructure, like
everything under oracle/; never on the product path.

The C port in rdf_oracle.c runs the reference's per-query plan the way the reference does (one operator at a time, a
typed decode per row and operator).  This module answers "what does a good columnar CPU engine make of the same work":
the BATCHED operator tree of bsbm.q5_batch_plan (shared scans, constants first, candidate join, the two numeric
windows, label last) on pyarrow's multi-threaded Acero hash join, with the integer values decoded once per predicate
slice by a numpy gather.  It is not the reference (that is DataFusion 52, absent here) and is reported next to the
port, never instead of it."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

from rdf_fusion_amd import abi


def prepare(ds):
    """per-predicate slices as Arrow tables (the CPU engine's resident data; not timed)"""
    pr, tv = ds.pred, ds.typed_values

    def pattern(pname):
        m = ds.p == pr[pname]
        so = np.unique((ds.s[m].astype(np.uint64) << np.uint64(32)) | ds.o[m].astype(np.uint64))     # a quad store is a set
        return (so >> np.uint64(32)).astype(np.uint32), (so & np.uint64(0xFFFFFFFF)).astype(np.uint32)

    def numeric(pname):
        s, o = pattern(pname)
        assert (tv["tag"][o] == abi.TV_INTEGER).all()
        return s, tv["lo"][o].astype(np.int64)
    pf_s, pf_o = pattern("bsbm:productFeature")
    n1_s, n1_v = numeric("bsbm:productPropertyNumeric1")
    n2_s, n2_v = numeric("bsbm:productPropertyNumeric2")
    lb_s, lb_o = pattern("rdfs:label")
    return {"pf": pa.table({"product": pf_s, "feature": pf_o}), "n1": pa.table({"product": n1_s, "v": n1_v}),
            "n2": pa.table({"product": n2_s, "v": n2_v}), "label": pa.table({"product": lb_s, "label": lb_o})}


def run(prep, batch, w1=120, w2=170):
    """(inst, product, label) of a batch of Q5 instances: numpy columns"""
    params = pa.table({"inst": np.arange(1, len(batch) + 1, dtype=np.uint32), "X": np.asarray(batch, dtype=np.uint32)})
    ren = lambda t, names: t.rename_columns(names)
    c = params.join(ren(prep["pf"], ["Xf", "f"]), keys="X", right_keys="Xf")                                  # inst, X, f
    for k in (1, 2):                                                                                          # + orig1, orig2
        o = params.join(ren(prep[f"n{k}"], ["Xn", f"orig{k}"]), keys="X", right_keys="Xn").select(["inst", f"orig{k}"])
        c = c.join(ren(o, [f"inst{k}", f"orig{k}"]), keys="inst", right_keys=f"inst{k}")
    c = c.join(ren(prep["pf"], ["product", "f2"]), keys="f", right_keys="f2")                                   # candidates
    c = c.filter(pc.not_equal(c["product"], c["X"])).select(["inst", "product", "orig1", "orig2"])
    for k, w in ((1, w1), (2, w2)):
        c = c.join(ren(prep[f"n{k}"], [f"p{k}", f"sim{k}"]), keys="product", right_keys=f"p{k}")
        keep = pc.and_(pc.less(c[f"sim{k}"], pc.add(c[f"orig{k}"], w)), pc.greater(c[f"sim{k}"], pc.subtract(c[f"orig{k}"], w)))
        c = c.filter(keep).select(["inst", "product", "orig1", "orig2"])
    c = c.join(ren(prep["label"], ["pl", "label"]), keys="product", right_keys="pl")
    return [np.asarray(c[n].combine_chunks()) for n in ("inst", "product", "label")]
