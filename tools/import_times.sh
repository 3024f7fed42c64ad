#!/bin/bash
# interpreter start-up costs on the GPU box's host: importing numpy / torch with the BLAS / OpenMP pools at their default size
# (a thread per core) and limited to one thread
cd "$GRAFT_REPO_ROOT"
t() { python3 - <<PY
import time, os
t0 = time.perf_counter()
import $1
print("%-6s %-28s %.3f s (cpu_count %d)" % ("$1", "$2", time.perf_counter() - t0, os.cpu_count()))
PY
}
t numpy default; t numpy default
OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 t numpy "one thread"
OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 t numpy "one thread"
t torch default; t torch default
OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 t torch "one thread"
OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 t torch "one thread"
