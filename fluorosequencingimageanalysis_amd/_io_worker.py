"""Host worker process of the file layer (pflib.image_batch): reads pickled requests (function name, arguments) from stdin,
runs pflib._read_job / pflib._save_records_job (/ _save_job) and writes the pickled result to stdout.  Started as
`python -m fluorosequencingimageanalysis_amd._io_worker` by pflib._IoPool - a plain child process with its own interpreter,
independent of what the parent's __main__ is (multiprocessing's spawn would re-import it) and of the GPU state of the parent
(nothing here touches the GPU)."""
import pickle
import struct
import sys


def main():
    import logging
    logging.getLogger().addHandler(logging.NullHandler())      # failures travel back to the parent, which logs them
    from . import pflib
    try:        # what the jobs import on first use, now: the parent is still busy with its own start-up
        from PIL import Image, ImageDraw, ImageOps, PngImagePlugin, TiffImagePlugin  # noqa: F401
        import _compat_pickle  # noqa: F401
        Image.init()
    except Exception:       # noqa: BLE001 - the job that needs it reports it
        pass
    jobs = {"read": pflib._read_job, "save": pflib._save_job, "save_records": pflib._save_records_job}
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                     # stray prints must not corrupt the reply stream
    while True:
        head = inp.read(8)
        if len(head) < 8:
            return
        (n,) = struct.unpack("<q", head)
        name, args = pickle.loads(inp.read(n))
        try:
            res = jobs[name](*args)
        except BaseException as e:              # noqa: BLE001 - the jobs catch their own errors; this is the safety net
            res = ("__worker_error__", "%s: %s" % (type(e).__name__, e))
        blob = pickle.dumps(res, protocol=pickle.HIGHEST_PROTOCOL)
        out.write(struct.pack("<q", len(blob)))
        out.write(blob)
        out.flush()


if __name__ == "__main__":
    main()
