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


def test_pmc_traffic_is_quoted_only_for_the_same_workload_and_the_same_kernel_sources(tmp_path):
    """roofline.traffic comes from a committed rocprofv3 counter summary: the newest one of the SAME workload, and only while the
    device sources still hash to what it was collected on (profiles/summarize.py stamps `_source_sha16`) — counters measured on
    other kernel code, or on another workload, are never quoted."""
    import rdf_fusion_amd
    b = load_bench()
    now = rdf_fusion_amd.kernel_source_sha16()
    wl = {"queries": 262144, "products": 285000}
    k1, k2 = "void rdfgpu::band_emit_kernel", "void rdfgpu::band_mask_kernel"        # what bench.py asks for: its kernel-class names (prefixes)
    long1, long2 = k1 + "<3>", k2 + "<2, 1, true>"                                     # what a summary is keyed by: the names as rocprofv3 prints them

    def write(name, sha, workload, k1_bytes=100, k2_bytes=20):
        d = {long1: {"hbm_bytes_per_launch": k1_bytes}, long2: {"hbm_bytes_per_launch": k2_bytes}, "_workload": workload}
        if sha is not None:
            d["_source_sha16"] = sha
        json.dump(d, open(tmp_path / name, "w"))

    write("r03_v1_pmc_fetch_write_per_kernel.json", now, wl, 100, 20)
    assert b.pmc_traffic(k1, wl, str(tmp_path)) == (100, os.path.join("profiles", "r03_v1_pmc_fetch_write_per_kernel.json"))
    assert b.pmc_traffic([k1, k2], wl, str(tmp_path))[0] == 120                                   # several kernels of one operator: summed
    assert b.pmc_traffic(long1 + "(rdfgpu::BandArgs)", wl, str(tmp_path))[0] == 100               # (an argument list on the asked name is ignored)
    assert b.pmc_traffic(k1, {"queries": 12345, "products": 285000}, str(tmp_path)) == (None, None)   # another workload: never
    assert b.pmc_traffic("no such kernel", wl, str(tmp_path)) == (None, None)
    assert b.pmc_traffic([k1, "no such kernel"], wl, str(tmp_path)) == (None, None)
    write("r03_v2_pmc_fetch_write_per_kernel.json", now, wl, 777, 1)                              # the newest summary of the workload wins
    assert b.pmc_traffic(k1, wl, str(tmp_path))[0] == 777
    write("r03_v3_pmc_fetch_write_per_kernel.json", "0123456789abcdef", wl, 5, 5)                 # ... unless it was collected on other kernel code:
    t, why = b.pmc_traffic(k1, wl, str(tmp_path))                                                 # then nothing is quoted (not the older one either:
    assert t is None and "stale" in why and "r03_v3" in why                                        # it describes kernels that have changed since)
    write("r03_v4_pmc_fetch_write_per_kernel.json", None, wl, 9, 9)                               # a summary without the stamp predates it: stale
    t, why = b.pmc_traffic(k1, wl, str(tmp_path))
    assert t is None and "stale" in why
    # the summaries committed by earlier rounds carry no stamp (and the kernels have changed): none of THEM may be quoted any more — a
    # workload is answered by a committed summary of the current sources (this round's set, while no kernel has changed since) or not at all
    committed = [f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_fetch_write_per_kernel.json")]
    for f in committed:
        d = json.load(open(os.path.join(ROOT, "profiles", f)))
        if d.get("_source_sha16") != now and d.get("_workload"):
            t, src = b.pmc_traffic(k1, d["_workload"])
            assert t is None or (os.path.basename(src) != f and json.load(open(os.path.join(ROOT, src))).get("_source_sha16") == now), (f, src)


def test_bench_flags_follow_the_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"',
                '"config"', '"roofline"', '"cpu_baseline"', '"workload"'):
        assert key in src, key
    assert "oracle" in src and "cpu_baseline" in src                 # the oracle is used for the CPU baseline / parity only
