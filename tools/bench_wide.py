"""fields/s of pflib.find_peptides_batch (dicts) and find_peptides_records (byte tables) on the same 1 024 fields as uint16 and - scaled
by 300 - as uint32 (FSQ_PIXELS_U32): do wide pixels run at the rate of 16-bit ones?   usage: python3 tools/bench_wide.py [fields=1024]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluorosequencingimageanalysis_amd import pflib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
imgs = bench.make_fields(range(1000, 1000 + n), (512, 512), 500)
for name, stack in (("uint16", imgs), ("uint32 (x 300)", imgs.astype(np.uint32) * 300)):
    pflib.find_peptides_batch(stack[:256])
    pflib.find_peptides_records(stack[:256])
    best_d = best_r = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        d = pflib.find_peptides_batch(stack)
        t1 = time.perf_counter()
        npk = sum(len(x) for x in d)
        del d
        t2 = time.perf_counter()
        rec, counts, fmt = pflib.find_peptides_records(stack)
        t3 = time.perf_counter()
        best_d, best_r = min(best_d, t1 - t0), min(best_r, t3 - t2)
    print("%-15s dicts %.0f fields/s (%.3f s, %d peaks) | records %.0f fields/s (%.3f s, %d-byte records)"
          % (name, n / best_d, best_d, npk, n / best_r, best_r, rec.shape[1]), flush=True)
