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
    kernel = "void rdfgpu::lds_join_kernel<2, 0, 4, 3, true>"
    t, src = b.pmc_traffic(kernel, 262144, 285000)
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_fetch_write_per_kernel.json"))
    newest = max((f for f in files if json.load(open(os.path.join(ROOT, "profiles", f))).get("_workload", {}).get("queries") == 262144),
                 key=lambda f: [int(x) for x in __import__("re").findall(r"\d+", f)])
    assert src == os.path.join("profiles", newest) and t == json.load(open(os.path.join(ROOT, "profiles", newest)))[kernel]["hbm_bytes_per_launch"]
    t14, src14 = b.pmc_traffic(kernel, 65536, 285000)                 # another batch size has its own counters
    assert src14.endswith("r01_v14_pmc_fetch_write_per_kernel.json") and t14 != t
    assert b.pmc_traffic(kernel, 12345, 285000) == (None, None)       # never quoted for a workload they were not collected on
    assert b.pmc_traffic("no such kernel", 262144, 285000) == (None, None)


def test_bench_flags_follow_the_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"',
                '"config"', '"roofline"', '"cpu_baseline"', '"workload"'):
        assert key in src, key
    assert "oracle" in src and "cpu_baseline" in src                 # the oracle is used for the CPU baseline / parity only
