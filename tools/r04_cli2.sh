#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_cli; mkdir -p $O
timeout -k 10 900 python3 tools/bench_cli.py 1024 16 16 > $O/cli1024.log 2>&1 || { tail -20 $O/cli1024.log; exit 1; }
grep -v amdgpu.ids $O/cli1024.log | tail -3
timeout -k 10 600 python3 -c "
import cProfile, pstats, sys, os, tempfile, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
from PIL import Image
from fluorosequencingimageanalysis_amd import basic_image_script as cli, pflib
imgs = bench.make_fields(range(3000, 3512), (512, 512), 500)
tmp = tempfile.mkdtemp(prefix='fsq_prof_cli_')
for i in range(512): Image.fromarray(imgs[i]).save(os.path.join(tmp, 'field%04d.tif' % i), format='TIFF')
pflib.IO_WORKERS = 16
cli.main(['-L', os.path.join(tmp, 'log0.txt'), tmp])      # warm: workers started, PNGs converted
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
res = cli.main(['-L', os.path.join(tmp, 'log.txt'), tmp])
pr.disable(); dt = time.perf_counter() - t0
print('second run (converted PNGs exist, workers warm): %d images in %.2f s = %.1f images/s' % (len(res), dt, len(res)/dt))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
" > $O/cli_prof.log 2>&1 || { tail -20 $O/cli_prof.log; exit 1; }
grep -v amdgpu.ids $O/cli_prof.log | head -60
