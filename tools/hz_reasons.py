"""Why fits leave the fast Jacobian kernel (need a library built with -DFSQ_DEBUG_HZ: FSQ_HIP_LIB=.../prof/libfsq_hip_hz.so):
counts per guarded-range / undecidable-decision site (KA_HZ codes of csrc/fsq_fit_rounds.hip) on copies of the bench's fields.
usage: FSQ_HIP_LIB=... python3 tools/hz_reasons.py [fields=256]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluorosequencingimageanalysis_amd import _native as N, engine as E, pflib  # noqa: E402

NAMES = ["perturbed centre range", "perturbed sigma range", "exp argument range", "step h range", "height / amplitude range", "no pivot (NaN norms)",
         "pivot choice not settled by the tracked norms", "Householder norm range", "reflector head range", "update numerator exponents", "1 - t^2 is NaN",
         "sign of 1 - t^2 not settled", "tracked error bound too large", "tiny numerators (exponent floor)", "re-computation test not settled"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
imgs = bench.make_fields(range(n), (512, 512), 500)
eng = E.Engine(n, 512, 512)
prm = E.detect_params(5, pflib.default_correlation_matrix, 2)
total = eng.run(E.to_device_u16(imgs), prm, 0.7, 4, N.MODE_REF, True)
L = N.lib()
L.fsq_debug_hz.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 32)()
N.check(L.fsq_debug_hz(buf, 1), "fsq_debug_hz")
v = list(buf)
print("%d fits, %d went through the slow kernel; flags raised per site (a fit can raise several):" % (total, L.fsq_fit_last_slow_count()))
for k, name in enumerate(NAMES):
    print("  %2d %-48s %d" % (k, name, v[k]))
