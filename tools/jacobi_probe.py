"""Sweeps / cycles of the LDS Jacobi eigensolver on the BXD kinship (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bulklmm_jl_amd as B
from common import make_data
Y, G, K, _ = make_data(p=512, m=64, seed=1)
ctx = B.Context(0)
for it in range(4):
    from bulklmm_jl_amd import api as A, _lib as L
    _, _, st = A._bulkscan_call(L.BLMM_NULL_EXACT, Y, G, K, None, None, True, None, 1.0, 0.0, False, 1, "eigen", 0, ctx, return_status=True)
    n = K.shape[0]
    N = n + (n & 1)
    rounds = st.jacobi_sweeps * (N - 1)
    print("sweeps", st.jacobi_sweeps, "cycles", st.jacobi_cycles, "ticks", st.jacobi_ticks_100mhz, "rounds", rounds,
          "cycles/round", st.jacobi_cycles / rounds, "ns/round", st.jacobi_ticks_100mhz * 10 / rounds, "t_eigen_ms", st.t_eigen_ms)
