"""The two-key join that closes LUBM Q9's triangle — (student x, advisor y, course z of y) JOIN (x takesCourse z) ON (x, z) —
at LUBM-8000 scale (BASELINE config 5), with the BUILD INSIDE THE TIMED REGION: HashJoinExec(CollectLeft) per query, as the
reference plans it (lib/logical/src/join/rewrite.rs:126-168).  The build side is a join output (no store slice), so the join
runs radix-partitioned with LDS-staged hash tables (part_join.hip).  Reported: HIP-event time of the whole operator (partition
passes of both sides + join kernel), the SURVEY 8d hash-join bytes (partition passes in the time, not in the bytes), the
fraction of 8 TB/s; beside it the same join against the store slice's cached table and through one HBM hash table; all three
must agree on count and checksum.
  python profiles/tools/lubm_join_bench.py [universities]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import lubm, abi
from rdf_fusion_amd.plan import PlanBuilder, quad_pattern

U = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
t0 = time.perf_counter(); ds = lubm.generate(U); t_gen = time.perf_counter() - t0
st = rf.GpuQuadStore()
n = st.extend(ds.g, ds.s, ds.p, ds.o)
st.set_typed_values(ds.typed_values)
print("LUBM-%d: %d quads, generated in %.1f s" % (U, n, t_gen), flush=True)
pr, cl = ds.pred, ds.cls

# the build side: (x, y, z) = advisor JOIN Student JOIN Faculty JOIN teacherOf — a join OUTPUT, kept in HBM
pb = PlanBuilder()
node = pb.hash_join(pb.data_source(quad_pattern("x", pr["ub:advisor"], "y")), pb.data_source(quad_pattern("x", pr["rdf:type"], cl["ub:Student"])), on=[(0, 0)], projection=[0, 1])
node = pb.hash_join(node, pb.data_source(quad_pattern("y", pr["rdf:type"], cl["ub:Faculty"])), on=[(1, 0)], projection=[0, 1])
node = pb.hash_join(node, pb.data_source(quad_pattern("y", pr["ub:teacherOf"], "z")), on=[(1, 0)], projection=[0, 1, 3])
prefix = st.plan(pb.build(node)).execute()
cols, n_build = prefix.result_device()
print("build side: %d rows (a join output in HBM)" % n_build, flush=True)

pb = PlanBuilder()
desc = pb.build(pb.hash_join(pb.table(0, 3), pb.data_source(quad_pattern("x", pr["ub:takesCourse"], "z")), on=[(0, 0), (2, 1)], projection=[0, 1, 2]))


def checksum(plan):
    c = [np.asarray(x, dtype=np.uint64) for x in plan.fetch()]
    with np.errstate(over="ignore"):
        mix = c[0] * np.uint64(0x9E3779B97F4A7C15) ^ c[1] * np.uint64(0xC2B2AE3D27D4EB4F) ^ c[2] * np.uint64(0x165667B19E3779F9)
        return len(c[0]), int(mix.sum(dtype=np.uint64))


def run(options, reps=4):
    plan = st.plan(desc)
    for o in options:
        plan.set_option(o, 1)
    plan.bind_table(0, cols, n_build)
    best = None
    for _ in range(reps):
        plan.enable_kernel_timing(True)
        torch.cuda.synchronize(); t1 = time.perf_counter(); plan.execute(); rows, _ = plan.result_info(); wall = (time.perf_counter() - t1) * 1e3
        ks = plan.kernel_stats()
        dev = sum(k[2] for k in ks)
        if best is None or dev < best["kernel_ms"]:
            best = {"wall_ms": round(wall, 3), "kernel_ms": round(dev, 3), "rows": rows,
                    "kernels": {k[0]: round(k[2], 3) for k in sorted(ks, key=lambda k: -k[2])[:6]},
                    "formula_bytes": int(sum(k[3] for k in ks if "join_kernel" in k[0] or "gjoin_build" in k[0]))}
    best["check"] = checksum(plan)
    plan.close()
    return best


res = {"universities": U, "quads": int(n), "build_rows": int(n_build), "probe_rows": int((ds.p == pr["ub:takesCourse"]).sum())}
res["partitioned_build_in_timed_region"] = run(["NO_TABLE_CACHE"])
res["hash_partitioned_both_sides_build_in_timed_region"] = run(["NO_TABLE_CACHE", "NO_RANGE_PARTITION"])
res["hbm_hash_build_in_timed_region"] = run(["NO_TABLE_CACHE", "NO_PARTITIONED_JOIN"])
res["cached_slice_table_steady_state"] = run(["NO_PARTITIONED_JOIN"])      # the slice's cached hash table, probed with the 98 M rows
res["default_planner_steady_state"] = run([])                                # what the planner picks with everything allowed (since round 3: the partitioned join)
assert res["partitioned_build_in_timed_region"]["check"] == res["hbm_hash_build_in_timed_region"]["check"] == res["cached_slice_table_steady_state"]["check"] == \
       res["hash_partitioned_both_sides_build_in_timed_region"]["check"] == res["default_planner_steady_state"]["check"]
# The answer at this size by another algorithm on the host (numpy: flag arrays, argsort + searchsorted merge joins over the raw
# triples, u64 pair keys) — the C oracle stops at a few universities (tests/), this is the full LUBM-8000 join: same count and the
# same order-independent checksum as every device path above.
t1 = time.perf_counter()
S, P_, O = np.asarray(ds.s), np.asarray(ds.p), np.asarray(ds.o)
n_ids = int(max(S.max(), O.max())) + 1
def members(cls_id):
    f = np.zeros(n_ids, dtype=bool)
    m = (P_ == pr["rdf:type"]) & (O == cls_id)
    f[S[m]] = True
    return f
is_student, is_faculty = members(cl["ub:Student"]), members(cl["ub:Faculty"])
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
assert len(bx) == n_build, (len(bx), n_build)
m = P_ == pr["ub:takesCourse"]
tc = np.sort((S[m].astype(np.uint64) << np.uint64(32)) | O[m].astype(np.uint64))
key = (bx.astype(np.uint64) << np.uint64(32)) | bz.astype(np.uint64)
at = np.minimum(np.searchsorted(tc, key), len(tc) - 1)
hit = tc[at] == key
with np.errstate(over="ignore"):
    mix = bx[hit].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) ^ by[hit].astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F) ^ bz[hit].astype(np.uint64) * np.uint64(0x165667B19E3779F9)
host = [int(hit.sum()), int(mix.sum(dtype=np.uint64))]
res["numpy_cross_check_at_full_size"] = {"rows": host[0], "checksum": host[1], "seconds": round(time.perf_counter() - t1, 1),
                                         "how": "flag arrays + argsort / searchsorted merge joins over the raw triples on the host; (takesCourse is a set: no duplicate pairs)"}
assert host == list(res["partitioned_build_in_timed_region"]["check"]), (host, res["partitioned_build_in_timed_region"]["check"])
print("numpy cross-check at full size: %d rows, checksum equal (%.0f s)" % (host[0], time.perf_counter() - t1), flush=True)
p = res["partitioned_build_in_timed_region"]
p["GBps"] = round(p["formula_bytes"] / (p["kernel_ms"] * 1e-3) / 1e9, 1)
p["frac_of_8TBps"] = round(p["GBps"] / 8000.0, 4)
p["bytes_formula"] = "SURVEY 8d: (4(k + p_b) + 8) N_b + (4(k + p_p) + 8) N_p + 4 c_o N_o, k = 2; partition passes (rocPRIM radix sort of the build side's 16-B records; the probe slice is sorted by a join key and read in place, its partitions are key ranges) are in kernel_ms, not in the bytes"
print(json.dumps(res), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/lubm_join_%d.json" % U, "w"), indent=1)
