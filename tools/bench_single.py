"""Latency of the reference's own call, pflib.find_peptides(image), image after image (one 512 x 512 field, ~500 spots per call), and of
small stacks through find_peptides_batch.   usage: python3 tools/bench_single.py [calls=40]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluorosequencingimageanalysis_amd import pflib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
imgs = [synth.make_field(9000 + i, (512, 512), 500) for i in range(8)]
pflib.find_peptides(imgs[0])
pflib.find_peptides(imgs[1])
t0 = time.perf_counter()
tot = 0
for i in range(n):
    tot += len(pflib.find_peptides(imgs[i % 8]))
dt = time.perf_counter() - t0
print("find_peptides(image): %.1f ms per call (%d calls, %d peaks per image)" % (1e3 * dt / n, n, tot // n), flush=True)
for m in (2, 4, 8):
    stack = np.stack(imgs[:m])
    pflib.find_peptides_batch(stack)
    t0 = time.perf_counter()
    for i in range(10):
        pflib.find_peptides_batch(stack)
    dt = time.perf_counter() - t0
    print("find_peptides_batch(%d fields): %.1f ms per call = %.1f ms per field" % (m, 1e3 * dt / 10, 1e3 * dt / 10 / m), flush=True)
