#!/bin/bash
# round 4: batch / IO / distributed GPU tests of the records-shipping file layer + the CLI rate for a few worker counts
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_cli; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_batch_io.py tests/test_gpu_distributed.py tests/test_gpu_tracking.py -q -m gpu -x > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 600 python3 tools/bench_cli.py 256 12 16 > $O/cli.log 2>&1 || { tail -20 $O/cli.log; exit 1; }
grep -v amdgpu.ids $O/cli.log | tail -4
timeout -k 10 600 python3 tools/bench_cli.py 256 > $O/cli2.log 2>&1 || { tail -20 $O/cli2.log; exit 1; }
grep -v amdgpu.ids $O/cli2.log | tail -2
