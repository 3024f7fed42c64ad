"""Inside a real find_peptides_batch call: every chunk's records -> dicts call made twice on the same input (the first result
dropped before the second call, so that it re-uses the first's memory), CPU time of each.  usage: python3 tools/builder_twice.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluorosequencingimageanalysis_amd import pflib  # noqa: E402

imgs = bench.make_fields(range(5000, 6024), (512, 512), 500)
inner = pflib._records_to_dicts
rows_log = []


def twice(rows, *a, **k):
    c0 = time.thread_time()
    d = inner(rows, *a, **k)
    c1 = time.thread_time()
    del d
    c2 = time.thread_time()
    d = inner(rows, *a, **k)
    c3 = time.thread_time()
    rows_log.append((len(rows), (c1 - c0) * 1e3, (c2 - c1) * 1e3, (c3 - c2) * 1e3))
    return d


pflib.find_peptides_batch(imgs[:256])
pflib._records_to_dicts = twice
out = pflib.find_peptides_batch(imgs)
for n, a, f, b in rows_log:
    print("%6d peaks: first build %5.1f ms (%4.0f ns/peak), freeing it %5.1f ms, second build %5.1f ms (%4.0f ns/peak)" % (n, a, a / n * 1e6, f, b, b / n * 1e6))
