"""bench.py pieces that run without a GPU: which committed PMC summary a run quotes, and the contract of its flags."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pmc_traffic_quotes_the_newest_summary_of_the_same_workload():
    b = load_bench()
    wl = {"queries": 262144, "products": 285000}
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_fetch_write_per_kernel.json"))
    newest = max((f for f in files if all(json.load(open(os.path.join(ROOT, "profiles", f))).get("_workload", {}).get(k) == v for k, v in wl.items())),
                 key=lambda f: [int(x) for x in __import__("re").findall(r"\d+", f)])
    d = json.load(open(os.path.join(ROOT, "profiles", newest)))
    kernels = [k for k in d if k != "_workload"][:2]
    t, src = b.pmc_traffic(kernels[0], wl)
    assert src == os.path.join("profiles", newest) and t == d[kernels[0]]["hbm_bytes_per_launch"]
    t2, _ = b.pmc_traffic(kernels, wl)                                 # several kernels of one operator: summed
    assert t2 == sum(d[k]["hbm_bytes_per_launch"] for k in kernels)
    t14, src14 = b.pmc_traffic("void rdfgpu::lds_join_kernel<2, 0, 4, 3, true>", {"queries": 65536, "products": 285000})   # another batch size has its own counters
    assert src14.endswith("r01_v14_pmc_fetch_write_per_kernel.json") and t14 > 0
    assert b.pmc_traffic(kernels[0], {"queries": 12345, "products": 285000}) == (None, None)   # never quoted for a workload they were not collected on
    assert b.pmc_traffic("no such kernel", wl) == (None, None)
    assert b.pmc_traffic(kernels + ["no such kernel"], wl) == (None, None)


def test_bench_flags_follow_the_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"',
                '"config"', '"roofline"', '"cpu_baseline"', '"workload"'):
        assert key in src, key
    assert "oracle" in src and "cpu_baseline" in src                 # the oracle is used for the CPU baseline / parity only
