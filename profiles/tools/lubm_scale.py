"""BASELINE config 5 at scale: LUBM-shaped store (rdf_fusion_amd/lubm.py), Q9 + OPTIONAL + REGEX through the C ABI.
The C oracle cannot follow to this size; checked instead: (1) the WHOLE answer (triangle, REGEX, OPTIONAL: count + order-independent
checksum over all five columns) against a recomputation on the host by another algorithm (numpy merge joins over the raw triples, Python `re`
over the name pool), (2) every sampled result row is a triangle of the raw triples, (3) the bindings are the same with the engine's index
joins / direct tables / LDS joins switched off — different join algorithms, same answer.
  python profiles/tools/lubm_scale.py [universities]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import lubm

U = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.perf_counter(); ds = lubm.generate(U); t_gen = time.perf_counter() - t0
print("generated %d universities: %d triples, %d ids, %.1f s" % (U, ds.n_triples, ds.n_ids, t_gen), flush=True)
st = rf.GpuQuadStore()
t0 = time.perf_counter(); n = st.extend(ds.g, ds.s, ds.p, ds.o); torch.cuda.synchronize(); t_load = time.perf_counter() - t0
st.set_typed_values(ds.typed_values); st.set_strings(ds.str_offsets, ds.str_heap)
free, total = torch.cuda.mem_get_info()
print("loaded %d quads in %.1f s (3 sorted permutations); HBM in use %.1f GB" % (n, t_load, (total - free) / 1e9), flush=True)


def checksum(cols):
    mix = np.zeros(len(cols[0]), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for k, c in enumerate(cols):
            mix ^= c.astype(np.uint64) * np.uint64([0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x27D4EB2F165667C5, 0x85EBCA77C2B2AE63][k])
        return len(cols[0]), int(mix.sum(dtype=np.uint64))


key = lambda a, b: (a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)
edges = {}
for pname in ("ub:advisor", "ub:teacherOf", "ub:takesCourse"):      # sorted (s, o) keys of the triangle's predicates, for the membership check
    m = ds.p == ds.pred[pname]
    edges[pname] = np.sort(key(ds.s[m], ds.o[m]))


def host_triangle():
    """Q9's triangle by another algorithm on the host (numpy: class flags, argsort + searchsorted merge joins over the raw triples):
    rows (x, y, z) with x advisor y, x a Student, y a Faculty, y teacherOf z, x takesCourse z, z a Course."""
    S, P_, O = ds.s, ds.p, ds.o
    pr, cl = ds.pred, ds.cls
    def members(cls_id):
        f = np.zeros(ds.n_ids + 1, dtype=bool)
        f[S[(P_ == pr["rdf:type"]) & (O == cls_id)]] = True
        return f
    is_student, is_faculty, is_course = members(cl["ub:Student"]), members(cl["ub:Faculty"]), members(cl["ub:Course"])
    m = P_ == pr["ub:advisor"]
    ax, ay = S[m], O[m]
    keep = is_student[ax] & is_faculty[ay]
    ax, ay = ax[keep], ay[keep]
    m = P_ == pr["ub:teacherOf"]
    ty, tz = S[m], O[m]
    order = np.argsort(ty, kind="stable"); ty, tz = ty[order], tz[order]
    lo, hi = np.searchsorted(ty, ay, "left"), np.searchsorted(ty, ay, "right")
    cnt = (hi - lo).astype(np.int64)
    bx, by = np.repeat(ax, cnt), np.repeat(ay, cnt)
    pos = np.repeat(lo - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) + np.arange(int(cnt.sum()), dtype=np.int64)
    bz = tz[pos]
    tc = np.sort(key(S[P_ == pr["ub:takesCourse"]], O[P_ == pr["ub:takesCourse"]]))
    k = key(bx, bz)
    at = np.minimum(np.searchsorted(tc, k), len(tc) - 1)
    hit = (tc[at] == k) & is_course[bz]
    return bx[hit], by[hit], bz[hit]


def host_answer(tri, pattern, flags):
    """... + (x name n) FILTER REGEX(n, pattern, flags) OPTIONAL (x emailAddress e): Python's `re` over the lexical forms of the distinct name
    ids the triangle's students carry, numpy joins for the rest; (count, checksum) of (x, y, z, n, e) with e = 0 where unbound."""
    import re
    x, y, z = tri
    def one_to_many(keys, pred_name):                # rows of `keys` joined with (key <pred> value): (row index, value)
        m = ds.p == ds.pred[pred_name]
        ks, vs = ds.s[m], ds.o[m]
        order = np.argsort(ks, kind="stable"); ks, vs = ks[order], vs[order]
        lo, hi = np.searchsorted(ks, keys, "left"), np.searchsorted(ks, keys, "right")
        cnt = (hi - lo).astype(np.int64)
        rows = np.repeat(np.arange(len(keys), dtype=np.int64), cnt)
        pos = np.repeat(lo - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) + np.arange(int(cnt.sum()), dtype=np.int64)
        return rows, vs[pos], cnt
    rows, names, _ = one_to_many(x, "ub:name")
    x, y, z = x[rows], y[rows], z[rows]
    rx = re.compile(pattern, re.I if "i" in flags else 0)
    uniq = np.unique(names)
    off, heap = ds.str_offsets, ds.str_heap
    ok_ids = np.array([i for i in uniq.tolist() if rx.search(heap[int(off[i]):int(off[i + 1])].decode())], dtype=np.uint32)
    keep = np.isin(names, ok_ids)
    x, y, z, names = x[keep], y[keep], z[keep], names[keep]
    rows, mails, cnt = one_to_many(x, "ub:emailAddress")
    unbound = cnt == 0
    X = np.concatenate([x[rows], x[unbound]]); Y = np.concatenate([y[rows], y[unbound]]); Z = np.concatenate([z[rows], z[unbound]])
    N = np.concatenate([names[rows], names[unbound]]); E = np.concatenate([mails, np.zeros(int(unbound.sum()), np.uint32)])
    return checksum([X, Y, Z, N, E]), int(unbound.sum())


t0 = time.perf_counter(); tri = host_triangle(); t_tri = time.perf_counter() - t0
print("host triangle: %d rows (%.0f s)" % (len(tri[0]), t_tri), flush=True)
out = {"universities": U, "triples": int(n), "ids": int(ds.n_ids), "generate_seconds": round(t_gen, 1), "load_seconds": round(t_load, 2),
       "hbm_in_use_GB": round((total - free) / 1e9, 1), "queries": []}
for pattern, flags in (("^GraduateStudent1", ""), ("student[0-9]*7$", "i"), (".", "")):
    desc = lubm.q9_optional_regex_plan(ds, pattern, flags)
    plan = st.plan(desc).enable_kernel_timing(True)
    times = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); plan.execute(); rows, _ = plan.result_info(); times.append((time.perf_counter() - t0) * 1e3)
    base = checksum(plan.fetch())
    kern = sorted(plan.kernel_stats(), key=lambda k: -k[2])[:4]
    for toggle in ("RDFGPU_NO_INDEX_JOIN", "RDFGPU_NO_DIRECT_TABLE", "RDFGPU_NO_LDS_JOIN"):
        st.set_option(toggle, 1)
        alt = st.plan(desc).execute()
        assert checksum(alt.fetch()) == base, (pattern, toggle)
        alt.close()
        st.set_option(toggle, 0)
    got = plan.fetch()
    sel = np.random.default_rng(0).choice(len(got[0]), min(len(got[0]), 200_000), replace=False) if len(got[0]) else np.zeros(0, np.int64)
    for pname, a, b in (("ub:advisor", 0, 1), ("ub:teacherOf", 1, 2), ("ub:takesCourse", 0, 2)):
        k = key(got[a][sel], got[b][sel])
        pos = np.searchsorted(edges[pname], k)
        assert (edges[pname][np.minimum(pos, len(edges[pname]) - 1)] == k).all()
    t0 = time.perf_counter(); host, host_unbound = host_answer(tri, pattern, flags); t_host = time.perf_counter() - t0
    assert host == base, (pattern, host, base)       # the WHOLE answer — triangle, REGEX, OPTIONAL — recomputed on the host at full size
    assert host_unbound == int((got[4] == 0).sum())
    q = {"regex": pattern, "flags": flags, "bindings": rows, "ms": [round(x, 2) for x in times], "bindings_per_s": round(rows / (min(times) * 1e-3)),
         "host_recomputation_at_full_size": {"count_and_checksum_equal": True, "rows": host[0], "seconds": round(t_host + t_tri, 1),
                                             "how": "numpy class flags + argsort / searchsorted merge joins over the raw triples, Python re over the name pool"},
         "optional_unbound": int((got[4] == 0).sum()), "same_answer_without": ["index joins", "direct tables", "LDS joins"],
         "top_kernels": [(k[0], k[1], round(k[2], 3)) for k in kern]}
    print(json.dumps(q), flush=True)
    out["queries"].append(q)
    plan.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/lubm_scale_%d.json" % U, "w"), indent=1)
