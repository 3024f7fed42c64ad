"""Does a large numpy allocation get transparent huge pages on this host?  Page faults of touching 64 MB as one array against 680
arrays of 96 KB (numpy advises MADV_HUGEPAGE for allocations of 4 MB and more).  usage: python3 tools/thp_check.py"""
import resource
import time

import numpy as np


def faults():
    return resource.getrusage(resource.RUSAGE_SELF).ru_minflt


for name in ("enabled", "defrag"):
    try:
        print(name, open("/sys/kernel/mm/transparent_hugepage/" + name).read().strip())
    except OSError as e:
        print(name, e)
f0, t0 = faults(), time.perf_counter()
a = np.empty(64 << 20, np.uint8)
a[::4096] = 1
f1, t1 = faults(), time.perf_counter()
print("one 64 MB array:   %6d page faults, %.1f ms" % (f1 - f0, (t1 - t0) * 1e3))
del a
f0, t0 = faults(), time.perf_counter()
b = [np.empty(96 << 10, np.uint8) for _ in range(680)]
for x in b:
    x[::4096] = 1
f1, t1 = faults(), time.perf_counter()
print("680 x 96 KB arrays: %6d page faults, %.1f ms" % (f1 - f0, (t1 - t0) * 1e3))
