"""The scaled scan + FILTER alone (BASELINE config 2 on a 2^26-row partition), for rocprofv3 passes (round_profiles.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import rdf_fusion_amd as rf
import bench
print(bench.scan_roofline(rf, 0, int(os.environ.get("SCAN_LOG2", "26")), reps=5, distinct=int(os.environ.get("DISTINCT", "2000"))))
