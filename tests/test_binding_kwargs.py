"""Drop-in check of the host bindings against the reference's own signatures, read as TEXT: every keyword argument of
bulkscan / bulkscan_null / bulkscan_null_grid / bulkscan_alt_grid (src/bulkscan.jl:81-124,188-219,321-346,428-451) and of
the four scan methods (src/scan.jl:94-199) must be accepted by the Julia binding (bulklmm.jl_amd/julia/BulkLMMHIP.jl, the
method with the same positional arity) and by the Python mirror (bulklmm.jl_amd/api.py).  Skipped where the reference tree
is absent (the GPU box)."""
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def julia_methods(path):
    """{name: [(n_positional, {kwargs}), ...]} for every `function name(...)` of a Julia file."""
    txt = open(path).read()
    txt = re.sub(r"#=.*?=#", "", txt, flags=re.S)
    txt = "\n".join(line.split("#", 1)[0] if '"' not in line.split("#", 1)[0] or line.count('"') % 2 == 0 else line
                    for line in txt.splitlines())
    out = {}
    for mt in re.finditer(r"^function\s+([A-Za-z_][\w!]*)\s*\(", txt, flags=re.M):
        i = mt.end()
        depth, j = 1, i
        while depth and j < len(txt):
            depth += txt[j] in "([{"
            depth -= txt[j] in ")]}"
            j += 1
        sig = txt[i:j - 1]
        # split at top-level ';' and ','
        parts, cur, d, semi = [[]], "", 0, False
        for ch in sig:
            if ch in "([{":
                d += 1
            elif ch in ")]}":
                d -= 1
            if d == 0 and ch in ",;":
                parts[-1].append(cur.strip())
                cur = ""
                if ch == ";":
                    parts.append([])
                continue
            cur += ch
        parts[-1].append(cur.strip())
        pos = [a for a in parts[0] if a]
        kws = set()
        for a in (parts[1] if len(parts) > 1 else []):
            if a:
                kws.add(re.split(r"::|=", a, maxsplit=1)[0].strip())
        out.setdefault(mt.group(1), []).append((len(pos), kws))
    return out


REF_FUNCS = {"bulkscan.jl": ["bulkscan", "bulkscan_null", "bulkscan_null_grid", "bulkscan_alt_grid"], "scan.jl": ["scan"]}


def test_reference_signatures_parse():
    ref = julia_methods(os.path.join(REF, "bulkscan.jl"))
    assert {"method", "h2_grid", "nb", "nt_blas", "weights", "prior_variance", "prior_sample_size", "reml", "optim_interval",
            "decomp_scheme", "output_pvals", "chisq_df"} <= ref["bulkscan"][0][1]
    sc = julia_methods(os.path.join(REF, "scan.jl"))
    assert len(sc["scan"]) == 4 and all({"profileLL", "markerID", "h2_grid", "method"} <= kw for _, kw in sc["scan"])


def test_julia_binding_accepts_every_reference_keyword():
    mine = julia_methods(os.path.join(ROOT, "bulklmm.jl_amd", "julia", "BulkLMMHIP.jl"))
    for fname, funcs in REF_FUNCS.items():
        ref = julia_methods(os.path.join(REF, fname))
        for f in funcs:
            assert f in mine, f
            for npos, kws in ref[f]:
                cands = [k for n, k in mine[f] if n == npos]
                assert cands, f"{f}: no method with {npos} positional arguments in BulkLMMHIP.jl"
                missing = kws - set().union(*cands)
                assert not missing, f"{f}/{npos}: BulkLMMHIP.jl lacks keyword(s) {sorted(missing)}"


def test_python_mirror_accepts_every_reference_keyword(blmm):
    for fname, funcs in REF_FUNCS.items():
        ref = julia_methods(os.path.join(REF, fname))
        for f in funcs:
            params = set(inspect.signature(getattr(blmm, f)).parameters)
            for _, kws in ref[f]:
                missing = kws - params
                assert not missing, f"api.{f} lacks keyword(s) {sorted(missing)}"
