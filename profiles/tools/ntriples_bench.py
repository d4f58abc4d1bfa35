"""N-Triples -> ids on the device: throughput on a synthetic file (2 M triples by default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import rdf_fusion_amd as rf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(1)
s = rng.integers(0, n // 20 + 1, n); p = rng.integers(0, 40, n); o = rng.integers(0, n // 4 + 1, n); lit = rng.random(n) < 0.4
lines = [f'<http://example.org/product{a}> <http://example.org/vocabulary/p{b}> ' + (f'"{c}"^^<http://www.w3.org/2001/XMLSchema#integer>' if l else f'<http://example.org/thing{c}>') + " ."
         for a, b, c, l in zip(s.tolist(), p.tolist(), o.tolist(), lit.tolist())]
text = ("\n".join(lines) + "\n").encode()
torch.cuda.init(); torch.zeros(1, device="cuda")
rf.NTriples(text[:10_000].rsplit(b"\n", 1)[0] + b"\n").close()
for _ in range(3):
    t0 = time.perf_counter(); nt = rf.NTriples(text); dt = time.perf_counter() - t0
    print({"triples": nt.n_triples, "distinct_terms": nt.n_terms, "text_MB": round(len(text) / 1e6, 1), "seconds": round(dt, 4),
           "M_triples_per_s": round(nt.n_triples / dt / 1e6, 1), "GB_per_s": round(len(text) / dt / 1e9, 2)}, flush=True)
    nt.close()
from oracle import oracle as orc
t0 = time.perf_counter(); orc.ntriples_encode(text[:len(text) // 20].rsplit(b"\n", 1)[0] + b"\n"); dt = time.perf_counter() - t0
print({"cpu_restatement_python_M_triples_per_s": round(n / 20 / dt / 1e6, 3)})
