"""CPU checks of numerical claims the device design rests on, in NumPy on the reference's own kinship fixture:
the rank of the weight family { w(h2) = 1 ./ (delta * lambda + 1) } over the whole heritability axis and over the segments the
low-rank form of the null-exact scan cuts it into (kernels_lowrank.hip: lr_segments)."""
import re
from pathlib import Path

import numpy as np

from common import bxd_kinship

ROOT = Path(__file__).resolve().parent.parent


def family_rank(lam, h_lo, h_hi, tol=4e-15, npts=400):
    h = np.linspace(h_lo, h_hi, npts)
    W = 1.0 / (1.0 + np.outer(np.abs(lam), h / (1.0 - h)))
    W /= np.linalg.norm(W, axis=0)
    s = np.linalg.svd(W, compute_uv=False)
    return int((s > tol * s[0]).sum())


def default_edges():
    src = (ROOT / "bulklmm.jl_amd" / "csrc" / "kernels_lowrank.hip").read_text()
    m = re.search(r"static const double def\[7\] = \{([^}]*)\}", src)
    vals = [float(x) for x in m.group(1).split(",")]
    assert vals[0] == 0.0 and vals[-1] >= 1.0 and all(a < b for a, b in zip(vals, vals[1:]))
    return vals


def test_segments_of_the_heritability_axis_halve_the_rank_of_the_weight_family():
    """One basis for h2 in [0, 1) needs rank 21-24 on the BXD kinship spectrum (six K steps of the f64 MFMA per accumulator);
    every default segment needs at most 12 (three K steps) -- the reason the traits are grouped by segment."""
    lam = np.linalg.eigvalsh(bxd_kinship())
    whole = family_rank(lam, 0.0, 1.0 - 1e-9)
    assert 20 <= whole <= 24
    edges = default_edges()
    ranks = [family_rank(lam, a, min(b, 1.0 - 1e-9)) for a, b in zip(edges, edges[1:])]
    assert max(ranks) <= 12 and min(ranks) >= 8, ranks
