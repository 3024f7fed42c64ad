"""Where a cold command-line run spends its time: wall-clock marks at the stages of pflib.image_batch (worker start, first image
back, window read, GPU pass, files written) for m synthetic TIFFs.  usage: python3 tools/cli_phases.py [images=256] [workers=16]"""
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
    pflib.IO_WORKERS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    imgs = bench.make_fields(range(3000, 3000 + m), (512, 512), 500)
    tmp = tempfile.mkdtemp(prefix="fsq_cli_phases_")
    marks = []

    def mark(what):
        marks.append((time.perf_counter(), what))

    def wrap(name):
        inner = getattr(pflib, name)

        def f(*a, **k):
            mark(name + " >")
            r = inner(*a, **k)
            mark(name + " <")
            return r
        setattr(pflib, name, f)
    for name in ("_io_pool", "find_peptides_records", "_BatchRunner", "_warm_gpu"):
        wrap(name)
    read_all = pflib._read_all

    def read_all_marked(paths, pool):
        for k, item in enumerate(read_all(paths, pool)):
            if k in (0, 15, len(paths) - 1):
                mark("image %d read" % k)
            yield item
    pflib._read_all = read_all_marked
    try:
        for i in range(m):
            Image.fromarray(imgs[i]).save(os.path.join(tmp, "field%04d.tif" % i), format="TIFF")
        mark("start")
        res = cli.main(["-L", os.path.join(tmp, "log.txt"), tmp])
        mark("end (%d images)" % len(res))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        pflib.shutdown_io_workers()
    t0 = marks[0][0]
    for t, what in marks:
        print("%8.3f s  %s" % (t - t0, what))


if __name__ == "__main__":
    main()
