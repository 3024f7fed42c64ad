#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run6; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_register.py tests/test_gpu_configs.py::test_config3_cycle_stack_registration_and_fitting -q -m gpu -x > $O/pytest_reg.log 2>&1; echo "register tests rc=$?"; tail -15 $O/pytest_reg.log
timeout -k 10 300 python3 tools/bench_register.py > $O/bench_register.log 2>&1; cat $O/bench_register.log | tail -6
FSQ_REGISTER_Z2Z=1 timeout -k 10 300 python3 tools/bench_register.py > $O/bench_register_z2z.log 2>&1; tail -4 $O/bench_register_z2z.log
timeout -k 10 300 python3 bench.py --config 3 --steps 20 --warmup 3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; tail -c 600 $O/bench_cfg3.json
