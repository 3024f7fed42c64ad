"""images/s of the command line (basic_image_script: TIFF -> PNG conversion, read, fit, protocol-0 pickle + CSV per image) over a
temporary directory of synthetic 16-bit TIFFs, for a given number of host I/O worker processes.
usage: python3 tools/bench_cli.py [images=256] [io workers ...]"""
import os
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
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    workers = [int(x) for x in sys.argv[2:]] or [None]
    imgs = bench.make_fields(range(3000, 3000 + m), (512, 512), 500)
    print("cpu_count %s, affinity %d" % (os.cpu_count(), len(os.sched_getaffinity(0))), flush=True)
    for w in workers:
        tmp = tempfile.mkdtemp(prefix="fsq_bench_cli_")
        try:
            for i in range(m):
                Image.fromarray(imgs[i]).save(os.path.join(tmp, "field%04d.tif" % i), format="TIFF")
            pflib.IO_WORKERS = w
            t0 = time.perf_counter()
            res = cli.main(["-L", os.path.join(tmp, "log.txt"), tmp])
            dt = time.perf_counter() - t0
            print("io workers %s: %d images in %.2f s = %.1f images/s" % (w, len(res), dt, len(res) / dt), flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
            pflib.shutdown_io_workers()


if __name__ == "__main__":
    main()
