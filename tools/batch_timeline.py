"""Timeline of one warm pflib.find_peptides_batch call over n fields: when each lane's library call (fsq_find_peptides) of a chunk
starts and ends, and when the worker builds the chunk's dicts.  usage: python3 tools/batch_timeline.py [fields=1024]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluorosequencingimageanalysis_amd import engine as E, pflib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
imgs = bench.make_fields(range(5000, 5000 + n), (512, 512), 500)
marks = []
lock = threading.Lock()


def wrap(obj, name, label):
    inner = getattr(obj, name)

    def f(*a, **k):
        t0, c0 = time.perf_counter(), time.thread_time()
        r = inner(*a, **k)
        with lock:           # (thread_time: CPU time of this thread - far below the wall time means it waited, e.g. for the interpreter)
            marks.append((t0, time.perf_counter(), "%s (cpu %.1f ms)" % (label, (time.thread_time() - c0) * 1e3), threading.current_thread().name))
        return r
    setattr(obj, name, f)


wrap(E.PathRunner, "run", "gpu call")
wrap(pflib, "_records_to_dicts", "dicts")
pflib.find_peptides_batch(imgs[:256])
out = pflib.find_peptides_batch(imgs)
del out
marks.clear()
t0 = time.perf_counter()
out = pflib.find_peptides_batch(imgs)
t1 = time.perf_counter()
print("%d fields in %.3f s = %.0f fields/s" % (n, t1 - t0, n / (t1 - t0)))
busy = {}
for a, b, label, th in sorted(marks):
    print("%7.1f .. %7.1f ms  %-24s %s" % ((a - t0) * 1e3, (b - t0) * 1e3, label, th))
    busy[label.split(" (")[0]] = busy.get(label.split(" (")[0], 0.0) + (b - a)
print({k: "%.1f ms" % (v * 1e3) for k, v in busy.items()})
t2 = time.perf_counter()
del out
print("freeing the result: %.1f ms" % ((time.perf_counter() - t2) * 1e3))
