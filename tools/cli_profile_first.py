"""cProfile of the parent process over a FIRST command-line run (TIFFs not yet converted) with the GPU already warm:
what the process that drives the GPU spends its time on.  usage: python3 tools/cli_profile_first.py [images=512] [workers=16]"""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluorosequencingimageanalysis_amd import basic_image_script as cli, pflib  # noqa: E402


def main():
    from PIL import Image
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    pflib.IO_WORKERS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    imgs = bench.make_fields(range(3000, 3000 + m), (512, 512), 500)
    tmp0, tmp = tempfile.mkdtemp(prefix="fsq_prof0_"), tempfile.mkdtemp(prefix="fsq_prof_")
    try:
        for d in (tmp0, tmp):
            for i in range(m):
                Image.fromarray(imgs[i]).save(os.path.join(d, "field%04d.tif" % i), format="TIFF")
        cli.main(["-L", os.path.join(tmp0, "log.txt"), tmp0])      # warm: torch, GPU, workers, runner
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        res = cli.main(["-L", os.path.join(tmp, "log.txt"), tmp])
        pr.disable()
        dt = time.perf_counter() - t0
        print("first run over fresh TIFFs, process warm: %d images in %.2f s = %.1f images/s" % (len(res), dt, len(res) / dt))
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        shutil.rmtree(tmp0, ignore_errors=True)
        pflib.shutdown_io_workers()


if __name__ == "__main__":
    main()
