#!/bin/bash
# round 4, first GPU call: the new parity / distributed tests, the bench launcher with two ranks (gloo rehearsal), a bench line
# and the kA / kB phase split of the current kernels
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_first; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_fit.py tests/test_gpu_stream.py -q -m gpu -x > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
FSQ_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --fields 128 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench2.json 2> $O/bench2.err || { tail -20 $O/bench2.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench2.json'));print('launcher:', d['n_gpus'], d['ranks'], d['backend'], d['value'])"
FSQ_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --fields 128 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --materialise > $O/bench2m.json 2> $O/bench2m.err || { tail -20 $O/bench2m.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench2m.json'));print('materialise:', d['n_gpus'], d['value'], d['config']['workload'][-120:])"
timeout -k 10 400 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench.json'));print('bench:', d['value'],d['ms_per_step'],d['roofline']['frac'])"
export FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/variants/libfsq_prof.so
timeout -k 10 300 python3 tools/time_fit.py 256 0 > $O/phase.log 2>&1 && tail -24 $O/phase.log
