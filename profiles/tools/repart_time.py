"""Send-side cost of rdfgpu_exchange_repartition on one GPU (one-rank RCCL communicator: count, scan, stable scatter, send/recv to self)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rdf_fusion_amd as rf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_850_000
rng = np.random.default_rng(1)
cols = [rng.integers(1, 1 << 30, n).astype(np.uint32) for _ in range(5)]
cols[2] = np.sort(rng.integers(1, 50_000, n).astype(np.uint32))
t = [torch.from_numpy(c.view(np.int32)).cuda() for c in cols]
ptrs = [x.data_ptr() for x in t]
comm = rf.Comm(0, 1, device=0, unique_id=rf.Comm.unique_id())
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, rows = comm.repartition(ptrs, n, 2)
    torch.cuda.synchronize(); print(f"repartition of {n} rows x 5 columns: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
comm.close()
