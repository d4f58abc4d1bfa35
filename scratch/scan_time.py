"""kernel times of the scaled scan + FILTER only (no count check): experiments"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import rdf_fusion_amd as rf
import bench
try:
    r = bench.scan_roofline(rf, 0, 26, reps=5, check=not os.environ.get('EXP_FLOOR'))
    print(r["kernel_us"], r["best_us"])
except AssertionError as e:
    print("count differs (expected for the floor variant)", e)
