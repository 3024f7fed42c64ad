"""fields/s of the dict-materialising pflib surface on host fields (what bench.py's extras report), plus a time split.
usage: python3 tools/bench_batch.py [n_fields=1024] [size=512] [spots=500] [fields per chunk=128]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluorosequencingimageanalysis_amd import pflib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    spots = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    if os.environ.get("DICT_SLICE"):
        pflib.DICT_SLICE_PEAKS = int(os.environ["DICT_SLICE"])
    if len(sys.argv) > 4:
        pflib.CHUNK_PIXELS = int(sys.argv[4]) * size * size
        pflib.WINDOW_PIXELS = 8 * pflib.CHUNK_PIXELS
    imgs = bench.make_fields(range(1000, 1000 + n), (size, size), spots)
    pflib.find_peptides_batch(imgs[:256])
    for rep in range(3):
        t0 = time.perf_counter()
        d = pflib.find_peptides_batch(imgs)
        t1 = time.perf_counter()
        npk = sum(len(x) for x in d)
        del d
        t2 = time.perf_counter()
        rec, counts, fmt = pflib.find_peptides_records(imgs)
        t3 = time.perf_counter()
        dd = pflib.records_to_dicts(rec, counts, fmt)
        t4 = time.perf_counter()
        del dd
        print("dicts %.0f fields/s (%.3f s, %d peaks) | records %.0f fields/s (%.3f s) | records_to_dicts alone %.3f s"
              % (n / (t1 - t0), t1 - t0, npk, n / (t3 - t2), t3 - t2, t4 - t3), flush=True)


if __name__ == "__main__":
    main()
