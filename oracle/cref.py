"""ctypes loader of the C/OpenMP restatement oracle/bulkscan_null_ref.c (test infrastructure, like bulklmm_oracle.py: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "bulkscan_null_ref.c")
LIB = os.path.join(HERE, "libblmm_oracle_c.so")


def _cpu_tag() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build(force: bool = False) -> str:
    """-march=native: the library is rebuilt when it was built for another CPU model (the .so travels to the GPU box)."""
    tag = LIB + ".cpu"
    same_cpu = os.path.exists(tag) and open(tag).read() == _cpu_tag()
    if not force and same_cpu and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    r = subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"], capture_output=True, text=True)
    if r.returncode != 0:   # e.g. a gcc without -march=native support for this CPU
        r = subprocess.run(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed on oracle/bulkscan_null_ref.c:\n" + r.stderr[-2000:])
    open(tag, "w").write(_cpu_tag())
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        vp, i64 = C.c_void_p, C.c_int64
        lib.blmm_ref_bulkscan_null.argtypes = [vp, i64, i64, vp, i64, vp, i64, C.c_int, vp, C.c_double, C.c_double, C.c_int, C.c_int, vp, vp, C.c_int]
        lib.blmm_ref_bulkscan_null_at.argtypes = [vp, i64, i64, vp, i64, vp, i64, C.c_int, vp, C.c_double, C.c_double, C.c_int, C.c_int, vp, vp, C.c_int,
                                                  vp, C.c_int]
        lib.blmm_ref_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def bulkscan_null(Y, G, K, Covar=None, addIntercept=True, prior_variance=1.0, prior_sample_size=0.0, reml=False, optim_interval=1,
                  nthreads=0, h2_override=None, skip_search=False):
    """(L p x m, h2 m) of src/bulkscan.jl:212-314 computed by the C restatement; nthreads = 0: OpenMP's default.
    h2_override (m values): L is evaluated at these heritabilities (as bulklmm_oracle.bulkscan_null(..., h2_override=...)); the
    returned h2 is still the restatement's own Brent estimate unless skip_search."""
    lib = load()
    Y = np.asfortranarray(np.asarray(Y, dtype=np.float64).reshape(np.shape(Y)[0], -1))
    G = np.asfortranarray(np.asarray(G, dtype=np.float64))
    K = np.asfortranarray(np.asarray(K, dtype=np.float64))
    n, m = Y.shape
    p = G.shape[1]
    cov, ncov = None, 0
    if Covar is not None:
        cov = np.asfortranarray(np.asarray(Covar, dtype=np.float64))
        ncov = cov.shape[1]
    else:
        addIntercept = True
    L = np.empty((p, m), order="F")
    h2 = np.empty(m)
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)   # noqa: E731
    ov = None
    if h2_override is not None:
        ov = np.ascontiguousarray(np.asarray(h2_override, dtype=np.float64).ravel())
        if ov.shape[0] != m:
            raise ValueError("h2_override: one value per trait")
    rc = lib.blmm_ref_bulkscan_null_at(ptr(Y), n, m, ptr(G), p, ptr(cov), ncov, int(bool(addIntercept)), ptr(K), float(prior_variance),
                                       float(prior_sample_size), int(bool(reml)), int(optim_interval), ptr(L), ptr(h2), int(nthreads),
                                       ptr(ov), int(bool(skip_search)))
    if rc == -8:
        raise ZeroDivisionError("Dividing by zeros: the input vector can not contain any zeros!")
    if rc != 0:
        raise ValueError("blmm_ref_bulkscan_null: bad arguments")
    return L, h2
