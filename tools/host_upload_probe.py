import sys, os, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, bulklmm_jl_amd as B
from common import make_data
Y, G, K, _ = make_data(n=79, p=7321, m=35554, seed=1)
Y = np.asfortranarray(Y); G = np.asfortranarray(G); K = np.asfortranarray(K)
ctx = B.Context(0)
for i in range(4):
    t0 = time.perf_counter(); r = B.bulkscan_reduced(Y, G, K, method="null-exact", threshold=5.0, ctx=ctx); t1 = time.perf_counter()
    print("reduced call", round((t1 - t0) * 1e3, 3), "ms")
# raw H2D rate of the same arrays
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
d = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(d), Y.nbytes)
for i in range(3):
    t0 = time.perf_counter(); hip.hipMemcpy(d, Y.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(Y.nbytes), 1); t1 = time.perf_counter()
    print("hipMemcpy H2D pageable", Y.nbytes / 1e6, "MB", round((t1 - t0) * 1e3, 3), "ms")
