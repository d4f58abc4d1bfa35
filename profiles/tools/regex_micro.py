"""REGEX / CONTAINS filter throughput: N rows over D distinct strings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
from rdf_fusion_amd import abi
from rdf_fusion_amd.engine import TV_DTYPE
from rdf_fusion_amd.plan import PlanBuilder, col, ENC_TV, EBV, REGEX, CONTAINS, STRSTARTS
D = int(os.environ.get("D", "4000000")); N = int(os.environ.get("N", "50000000"))
rng = np.random.default_rng(0)
kinds = np.array(["GraduateStudent", "UndergraduateStudent", "FullProfessor", "Lecturer"])
k = rng.integers(0, 4, D); num = rng.integers(0, 100000, D)
strs = np.char.add(np.char.add(kinds[k], num.astype(str)), np.char.add("@dept", rng.integers(0, 500, D).astype(str)))
enc = np.char.encode(strs, "utf-8")
lens = np.char.str_len(strs).astype(np.uint64)
offsets = np.zeros(D + 2, dtype=np.uint64); offsets[2:] = np.cumsum(lens)
heap = b"".join(enc.tolist())
tv = np.zeros(D + 1, dtype=TV_DTYPE); tv["tag"][1:] = abi.TV_STRING; tv["lo"][1:] = np.argsort(np.argsort(strs))
st = rf.GpuQuadStore(); st.set_typed_values(tv); st.set_strings(offsets, heap)
ids = rng.integers(1, D + 1, N).astype(np.uint32)
t = torch.from_numpy(ids.view(np.int32)).cuda()
for name, e in (("regex ^Grad.*[0-9]7@", EBV(REGEX(ENC_TV(col(0)), "^Grad.*[0-9]7@", ""))), ("regex prof", EBV(REGEX(ENC_TV(col(0)), "prof", "i"))),
                ("contains dept42", EBV(CONTAINS(ENC_TV(col(0)), "dept42"))), ("strstarts Full", EBV(STRSTARTS(ENC_TV(col(0)), "Full")))):
    pb = PlanBuilder()
    plan = st.plan(pb.build(pb.filter(pb.table(0, 1), e))).enable_kernel_timing(True)
    plan.bind_table(0, [t.data_ptr()], N)
    plan.execute()
    first = {k[0][-28:]: round(k[2] * 1e3, 1) for k in plan.kernel_stats()}
    for _ in range(3): plan.execute()
    ks = [k for k in plan.kernel_stats() if "filter_kernel" in k[0]]
    us = ks[0][2] / ks[0][1] * 1e3
    print(f"{name:24s} rows {plan.result_info()[0]:9d}  {us:9.1f} us  {N / us / 1e3:7.2f} G rows/s  string bytes {float(lens.mean()) * N / us / 1e3:7.1f} GB/s   {ks[0][0][-30:]}  first run (us): {first}")
