"""What slows the dict builder inside find_peptides_batch (590 ns per peak) against the same builder called alone (320 ns per peak)?
The same records -> dicts call under one condition at a time.  usage: python3 tools/builder_factors.py [fields=512]"""
import concurrent.futures
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluorosequencingimageanalysis_amd import engine as E, pflib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
imgs = bench.make_fields(range(7000, 7000 + n), (512, 512), 500)
rec, counts, fmt = pflib.find_peptides_records(imgs)
npk = len(rec)
torch = E._torch()


def build(src=rec):
    t0 = time.perf_counter()
    d = pflib.records_to_dicts(src, counts, fmt)
    dt = time.perf_counter() - t0
    del d
    return dt


def report(what, dt):
    print("%-62s %6.1f ms  %4.0f ns per peak" % (what, dt * 1e3, dt / npk * 1e9), flush=True)


build(); build()
report("alone, main thread", min(build() for _ in range(3)))
pool = concurrent.futures.ThreadPoolExecutor(1)
report("alone, worker thread", min(pool.submit(build).result() for _ in range(3)))
old = sys.getswitchinterval()
sys.setswitchinterval(pflib.BATCH_SWITCH_INTERVAL)
report("worker thread, switch interval %g s" % pflib.BATCH_SWITCH_INTERVAL, min(pool.submit(build).result() for _ in range(3)))
sys.setswitchinterval(old)
pinned = torch.from_numpy(rec).pin_memory()
report("worker thread, records in pinned memory", min(pool.submit(build, pinned.numpy()).result() for _ in range(3)))
# a previous result kept alive while the next one is built (what a caller that keeps its results does)
keep = pflib.records_to_dicts(rec, counts, fmt)
report("worker thread, the previous result still alive", min(pool.submit(build).result() for _ in range(3)))
del keep
# fresh memory: with earlier results alive the interpreter's allocator has to get new arenas from the system, and every page of them
# is touched for the first time (in find_peptides_batch every chunk's dicts go into memory the call has not used before)
import resource  # noqa: E402


def faults():
    return resource.getrusage(resource.RUSAGE_SELF).ru_minflt


f0 = faults(); dt = build(); f1 = faults()
report("warm (memory of the previous run re-used), %d page faults" % (f1 - f0), dt)
keep = [pflib.records_to_dicts(rec, counts, fmt) for _ in range(2)]
f0 = faults()
t0 = time.perf_counter()
fresh = pflib.records_to_dicts(rec, counts, fmt)
dt = time.perf_counter() - t0
f1 = faults()
report("into fresh memory (two earlier results alive), %d page faults" % (f1 - f0), dt)
del keep, fresh
# other threads of the process busy in the GPU library (no interpreter lock held), as the lanes are
stop = threading.Event()
small = imgs[:64]


def lane():
    torch.cuda.set_device(0)
    runner = E.PathRunner(64, 512, 512)
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)
    with torch.cuda.stream(torch.cuda.Stream()):
        d = E.to_device_u16(small)
        while not stop.is_set():
            runner.run(d, prm)


ths = [threading.Thread(target=lane, daemon=True) for _ in range(3)]
for t in ths:
    t.start()
time.sleep(1.0)
report("worker thread, three lanes running fsq_find_peptides", min(pool.submit(build).result() for _ in range(3)))
sys.setswitchinterval(pflib.BATCH_SWITCH_INTERVAL)
report("... and the short switch interval", min(pool.submit(build).result() for _ in range(3)))
sys.setswitchinterval(old)
stop.set()
for t in ths:
    t.join()
