"""Container-only loader: imports the Python-2 reference under Python 3.

TEST INFRASTRUCTURE ONLY (never imported by the product).  Reads the reference's
source files from /root/reference at run time, applies a purely mechanical
load-time transform in memory (expandtabs + lib2to3 + numpy-2 shims, the list in
SURVEY.md section 8c) and exec()s the result.  Nothing from the reference is
written to disk.  Used only by oracle/gen_golden.py to produce tests/golden/*.npz.

Run with
  NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"
so numpy.exp/sin/cos are glibc's (the C restatement calls the same libm).
"""
import ctypes
import os
import sys
import types
import warnings

import numpy

warnings.filterwarnings("ignore")
REF = os.environ.get("FSQ_REFERENCE", "/root/reference")


def _refactor_tool():
    from lib2to3.refactor import RefactoringTool, get_fixers_from_package
    return RefactoringTool(get_fixers_from_package("lib2to3.fixes"))


_RT = None


def load(name, rel, patches=(), inject=None):
    global _RT
    if _RT is None:
        _RT = _refactor_tool()
    src = open(os.path.join(REF, rel)).read().expandtabs(8)
    out = str(_RT.refactor_string(src if src.endswith("\n") else src + "\n", rel))
    for a, b in patches:
        assert a in out, (rel, a)
        out = out.replace(a, b)
    m = types.ModuleType(name)
    m.__file__ = os.path.join(REF, rel)
    sys.modules[name] = m
    m.__dict__.update(inject or {})
    exec(compile(out, m.__file__, "exec"), m.__dict__)
    return m


class Ref:
    """Namespace with the loaded reference modules: mp (mpfit module), gf, pf, pc."""


def load_reference(textbook_qrsolv=False):
    numpy.rank = numpy.ndim
    for n, t in (("float", float), ("int", int), ("object", object), ("bool", bool)):
        if n not in numpy.__dict__:
            setattr(numpy, n, t)
    mp_patches = []
    if textbook_qrsolv:
        mp_patches.append(("x = numpy.diagonal(r)\n", "x = numpy.diagonal(r).copy()\n"))
    mp = load("agpy.mpfit.mpfit", "agpy/mpfit/mpfit.py", mp_patches)
    pkg, sub = types.ModuleType("agpy"), types.ModuleType("agpy.mpfit")
    pkg.__path__ = []
    sub.mpfit = mp.mpfit
    pkg.mpfit = sub
    sys.modules.update({"agpy": pkg, "agpy.mpfit": sub})
    gf = load("gaussfitter", "agpy/gaussfitter.py",
              [("elif params == [] or len(params)==0:", "elif len(params)==0:")])
    import scipy
    misc = types.ModuleType("scipy.misc")
    misc.imread = None
    sys.modules["scipy.misc"] = scipy.misc = misc
    sk = types.ModuleType("skimage")
    sk.exposure = types.ModuleType("skimage.exposure")
    sys.modules.update({"skimage": sk, "skimage.exposure": sk.exposure})
    libm = ctypes.CDLL("libm.so.6")
    libm.round.restype = ctypes.c_double
    libm.round.argtypes = [ctypes.c_double]
    # (Python 3: pickle.dump needs a binary file - the reference opens it in Python 2's text mode 'w', pflib.py:635)
    pf = load("pflib", "pflib.py", [("pickle.dump(psfs, open(output_path, 'w'))", "pickle.dump(psfs, open(output_path, 'wb'))")],
              inject={"round": lambda x: libm.round(float(x))})
    pc = load("phase_correlate", "phase_correlate.py",
              [("np.array(ref_image, dtype=np.float64, copy=False)",
                "np.asarray(ref_image, dtype=np.float64)"),
               ("np.array(reg_image, dtype=np.float64, copy=False)",
                "np.asarray(reg_image, dtype=np.float64)")])
    r = Ref()
    r.mp, r.gf, r.pf, r.pc = mp, gf, pf, pc
    return r


def load_flexlibrary(ref=None):
    """flexlibrary.py on top of load_reference() - only the Spot photometry methods are used.
    photutils and stepfitting_library are named at import time only; they are stood in for by empty modules (nothing of
    theirs is called by the methods recorded in the goldens)."""
    ref = ref or load_reference()
    for name in ("photutils", "stepfitting_library"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    import scipy.ndimage
    if "scipy.ndimage.measurements" not in sys.modules:
        meas = types.ModuleType("scipy.ndimage.measurements")
        meas.center_of_mass = scipy.ndimage.center_of_mass
        sys.modules["scipy.ndimage.measurements"] = meas
    libm = ctypes.CDLL("libm.so.6")
    libm.round.restype = ctypes.c_double
    libm.round.argtypes = [ctypes.c_double]
    # Python-2 round() (half away from zero) for the bin coordinates of the tracking code (flexlibrary.py:850, 880)
    # (Python 2: `/` between ints is floor division - Spot's (size - 1) / 2 radius feeds numpy slices)
    ref.fl = load("flexlibrary", "flexlibrary.py",
                  [("(self.size - 1) / 2", "(self.size - 1) // 2"), ("(size - 1) / 2", "(size - 1) // 2"),
                   # (Python 3: a pickle is read from a binary file - the reference opens it in Python 2's text mode, :547)
                   ("pickle.load(open(psf_pkl_filepath))", "pickle.load(open(psf_pkl_filepath, 'rb'))")],
                  inject={"round": lambda x: libm.round(float(x))})
    return ref
