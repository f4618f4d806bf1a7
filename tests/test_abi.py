"""CPU tests of the drop-in boundary: the C-ABI library builds, loads and exports every symbol include/*.h declares
(no compute calls without a GPU), and the host mirror's argument checking raises the reference's messages."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "bulklmm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(blmm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(blmm):
    lib = blmm.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(blmm.EXPORTS) == syms
    assert lib.blmm_version() == 210


def header_struct(name):
    """[(field, ctype), ...] of `typedef struct <name> { ... } <name>;` in include/bulklmm_hip.h."""
    txt = open(os.path.join(ROOT, "include", "bulklmm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    body = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), txt, flags=re.S).group(1)
    ctype = {"int32_t": C.c_int32, "int64_t": C.c_int64, "double": C.c_double, "uint64_t": C.c_uint64}
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, names = decl.split(None, 1)
        out += [(nm.strip(), ctype[ty]) for nm in names.split(",")]
    return out


def test_struct_layouts_match_header(blmm):
    """The ctypes mirrors are checked field by field (name, type, order) against the header the C callers compile."""
    from bulklmm_jl_amd import _lib as L
    for name in ("blmm_opts", "blmm_status", "blmm_multi_opts"):
        assert [(f, t) for f, t in getattr(L, name)._fields_] == header_struct(name), name
    assert C.sizeof(L.blmm_opts) == 6 * 4 + 2 * 8
    o = L.blmm_opts()
    blmm.load().blmm_default_opts(C.byref(o))
    assert (o.method, o.reml, o.add_intercept, o.decomp_scheme, o.optim_interval) == (1, 0, 1, 0, 1)
    assert (o.prior_variance, o.prior_sample_size) == (1.0, 0.0)  # bulkscan defaults, src/bulkscan.jl:81-92


def test_error_strings(blmm):
    lib = blmm.load()
    assert lib.blmm_err_string(-2).decode() == "Dimension mismatch."
    assert lib.blmm_err_string(-3).decode() == "Heritability of 1 is not allowed."
    assert lib.blmm_err_string(-6).decode() == "Can only handle one trait."
    assert lib.blmm_err_string(-8).decode() == "Dividing by zeros: the input vector can not contain any zeros!"


def test_no_gpu_fails_loudly(blmm):
    """Without a GPU the product path must refuse to run (there is no CPU fallback)."""
    lib = blmm.load()
    if lib.blmm_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(blmm.BulkLMMError):
        blmm.Context(0)
    with pytest.raises(blmm.BulkLMMError):
        blmm.bulkscan_null(np.zeros((5, 2)), np.zeros((5, 3)), np.eye(5))


def test_more_than_2048_individuals_is_refused_before_anything_is_uploaded(blmm):
    """The one capability gap against the reference (LAPACK's eigen takes any n, src/transform_helpers.jl:21-34): the host mirror
    refuses n > 2048 with the library's message and code BEFORE it creates a context or uploads data (this test runs without a GPU)."""
    n = 2049
    Y = np.zeros((n, 1)); G = np.zeros((n, 2)); K = np.eye(n)
    for call in (lambda: blmm.bulkscan(Y, G, K), lambda: blmm.bulkscan_null(Y, G, K), lambda: blmm.bulkscan_reduced(Y, G, K),
                 lambda: blmm.scan(Y[:, 0], G, K), lambda: blmm.bulkscan_alt_exact(Y, G, K), lambda: blmm.transform_rotation(Y, G, K)):
        with pytest.raises(blmm.BulkLMMError) as e:
            call()
        assert e.value.code == -10 and "2048" in e.value.msg


def test_product_path_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "bulklmm.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_host_mirror_argument_checks(blmm):
    Y = np.zeros((6, 2)); G = np.zeros((6, 3)); K = np.eye(6)
    with pytest.raises(blmm.BulkLMMError) as e:
        blmm.scan(np.zeros((6, 2)), G, K)
    assert e.value.msg == "Can only handle one trait."
    with pytest.raises(blmm.BulkLMMError) as e:
        blmm.scan(Y[:, 0], G, K, addIntercept=False)
    assert e.value.msg == "Intercept has to be added when no other covariate is given."
    with pytest.raises(blmm.BulkLMMError) as e:
        blmm.scan(Y[:, 0], G, K, assumption="nope")
    assert e.value.msg == "Assumption keyword is not supported. Please enter null or alt."
    with pytest.raises(blmm.BulkLMMError):
        blmm.bulkscan(Y, G, K, method="nope")
    with pytest.raises(blmm.BulkLMMError) as e:
        blmm.bulkscan_null(Y[:-1], G, K)
    assert e.value.msg == "Dimension mismatch."


def test_trait_sharding_helper(blmm):
    """ONE partition everywhere (blocks of ceil(m/R)): the torch.distributed host path (sharding.trait_shard,
    bench.py) and the C ABI's multi-GPU entry point (blmm_multi_shard) must agree."""
    import ctypes as C
    lib = blmm.load()
    for m, w in [(35554, 8), (10, 3), (7, 8), (0, 2), (20000, 8), (1, 1), (4096, 5)]:
        parts = [blmm.trait_shard(m, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == m
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        blk = -(-m // w)
        assert all(b - a in (blk, max(0, m - r * blk)) or b == a for r, (a, b) in enumerate(parts))
        assert blmm.shard_sizes(m, w) == [b - a for a, b in parts]
        for r in range(w):
            lo, hi = C.c_int64(-1), C.c_int64(-1)
            lib.blmm_multi_shard(m, r, w, C.byref(lo), C.byref(hi))
            assert (lo.value, hi.value) == parts[r], (m, w, r)
    assert [b - a for a, b in (blmm.trait_shard(35554, r, 8) for r in range(8))] == [4445] * 7 + [4439]


def test_readers_for_the_reference_file_formats(blmm, tmp_path):
    """readGenoProb[_ExcludeComplements], readBXDpheno, readBXDgeno (src/readData.jl:41-96,159-165) and Helium .he: host
    code behind the C ABI, no GPU needed.  Files in the shape of the BXD CSVs: quoted header, quoted id column, pheno with a
    trailing column that is dropped, genotype probabilities in complementary column pairs."""
    rng = np.random.default_rng(5)
    n, pm = 7, 6
    prob = rng.random((n, pm))
    geno = np.empty((n, 2 * pm))
    geno[:, 0::2] = prob
    geno[:, 1::2] = 1.0 - prob
    gfile = tmp_path / "geno.csv"
    with open(gfile, "w") as f:
        f.write(",".join(['"id"'] + [f'"m{j}_{ab}"' for j in range(pm) for ab in "BD"]) + "\n")
        for i in range(n):
            f.write(",".join([f'"BXD,{i}"'] + [repr(float(x)) for x in geno[i]]) + "\r\n")   # a comma inside a quoted id, CRLF
    assert np.array_equal(blmm.readGenoProb(str(gfile)), geno)
    assert np.array_equal(blmm.readGenoProb_ExcludeComplements(str(gfile)), prob)
    assert np.array_equal(blmm.readBXDgeno(str(gfile)), prob)                  # [:, 2:2:end] of the raw table
    ph = rng.standard_normal((n, 5))
    pfile = tmp_path / "pheno.csv"
    with open(pfile, "w") as f:
        f.write("id,t1,t2,t3,t4,t5,extra\n")
        for i in range(n):
            f.write(",".join([f"s{i}"] + [repr(float(x)) for x in ph[i]] + ["0"]) + "\n")
    assert np.array_equal(blmm.readBXDpheno(str(pfile)), ph)
    sys_path_tests = os.path.join(ROOT, "tests")
    import sys
    sys.path.insert(0, sys_path_tests)
    from common import GOLDEN, bxd_kinship
    K = blmm.readhe(os.path.join(GOLDEN, "bxd_kinship_ref.he"))
    assert K.shape == (79, 79) and np.array_equal(np.round(K, 12), bxd_kinship())
    with pytest.raises(blmm.BulkLMMError):
        blmm.readBXDpheno(str(tmp_path / "missing.csv"))
    bad = tmp_path / "bad.csv"
    open(bad, "w").write("id,a,b,x\ns1,1.0,2.0,0\ns2,1.0,0\n")                 # ragged
    with pytest.raises(blmm.BulkLMMError):
        blmm.readBXDpheno(str(bad))
