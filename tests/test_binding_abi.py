"""The Julia binding (bulklmm.jl_amd/julia/BulkLMMHIP.jl) cannot be executed here -- there is no Julia in the image -- so what
a first run would trip over is checked as TEXT against the C header the library is compiled from:

  * the isbits structs BlmmOpts / BlmmStatus / BlmmMultiOpts: field names, types and order against include/bulklmm_hip.h, and
    the offsets / sizes that Julia's C-compatible layout gives them against `offsetof` / `sizeof` printed by a C program;
  * every `ccall((:sym, libblmm), Ret, (Args...), ...)`: the symbol is exported, the return type and the argument-type tuple
    agree position by position with the C prototype, and as many values are passed as the tuple has types.
No GPU, no reference tree needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "bulklmm.jl_amd", "julia", "BulkLMMHIP.jl")
HDR = os.path.join(ROOT, "include", "bulklmm_hip.h")


def _header():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.sub(r"^\s*#.*$", "", txt, flags=re.M)


def c_structs():
    """{name: [(field, ctype), ...]}"""
    out = {}
    for mt in re.finditer(r"typedef struct (\w+)\s*\{(.*?)\}\s*\1\s*;", _header(), flags=re.S):
        fields = []
        for decl in mt.group(2).split(";"):
            decl = decl.strip()
            if decl:
                ty, names = decl.split(None, 1)
                fields += [(nm.strip(), ty) for nm in names.split(",")]
        out[mt.group(1)] = fields
    return out


def c_prototypes():
    """{name: (ret, [arg types])} with types normalised ('const double*', 'int64_t', ...)."""
    out = {}
    for mt in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(blmm_\w+)\s*\(([^;{}]*?)\)\s*;", _header()):
        ret, name, args = mt.group(1).strip(), mt.group(2), mt.group(3).strip()
        if "typedef" in ret or "struct" in ret:
            continue

        def norm(a, drop_name=True):
            a = " ".join(a.replace("*", " * ").split())
            toks = a.split(" ")
            if drop_name and len(toks) > 1 and re.fullmatch(r"[A-Za-z_]\w*", toks[-1]) and toks[-1] not in ("int", "double", "void", "float"):
                toks = toks[:-1]
            return " ".join(toks).replace(" *", "*").replace("* ", "*")
        argl = [] if args in ("", "void") else [norm(a) for a in args.split(",")]
        out[name] = (norm(ret, drop_name=False), argl)
    return out


def julia_text():
    txt = open(JL).read()
    txt = re.sub(r"#=.*?=#", "", txt, flags=re.S)
    lines = []
    for line in txt.splitlines():      # strip comments (a '#' outside a string literal)
        q, cut = False, None
        for i, ch in enumerate(line):
            if ch == '"' and (i == 0 or line[i - 1] != "\\"):
                q = not q
            elif ch == "#" and not q:
                cut = i
                break
        lines.append(line if cut is None else line[:cut])
    return "\n".join(lines)


def julia_structs():
    out = {}
    for mt in re.finditer(r"(?:mutable\s+)?struct\s+(\w+)\s*;?(.*?)\bend\b", julia_text(), flags=re.S):
        body = re.sub(r"\b\w+\(\)\s*=\s*new\(.*?\)", "", mt.group(2), flags=re.S)    # inner constructor
        out[mt.group(1)] = re.findall(r"(\w+)::([\w{}]+)", body)
    return out


def split_top(s):
    parts, cur, d = [], "", 0
    for ch in s:
        if ch in "([{":
            d += 1
        elif ch in ")]}":
            d -= 1
        if ch == "," and d == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def julia_ccalls():
    """[(symbol, ret, [arg types], n_values_passed)]"""
    txt = julia_text()
    out = []
    for mt in re.finditer(r"ccall\(\(:(\w+),\s*libblmm\)\s*,", txt):
        i = mt.end()
        depth, j = 1, i
        while depth and j < len(txt):
            depth += txt[j] in "([{"
            depth -= txt[j] in ")]}"
            j += 1
        parts = split_top(txt[i:j - 1])
        ret, tup, vals = parts[0], parts[1], parts[2:]
        assert tup.startswith("(") and tup.endswith(")"), (mt.group(1), tup)
        types = split_top(tup[1:-1])
        out.append((mt.group(1), ret, types, len(vals)))
    return out


JL2C = {"Int32": "int32_t", "Int64": "int64_t", "Float64": "double", "UInt64": "uint64_t",
        "Ptr{Float64}": "double*", "Ptr{Int64}": "int64_t*", "Ptr{Int32}": "int32_t*"}
SIZE = {"int32_t": 4, "int64_t": 8, "double": 8, "uint64_t": 8, "double*": 8, "int64_t*": 8, "int32_t*": 8}
STRUCTS = {"BlmmOpts": "blmm_opts", "BlmmStatus": "blmm_status", "BlmmMultiOpts": "blmm_multi_opts", "BlmmReduced": "blmm_reduced"}

OPAQUE = {"Ptr{Cvoid}"}
ALLOWED = {
    "blmm_ctx*": OPAQUE, "const blmm_ctx*": OPAQUE, "blmm_multi*": OPAQUE, "const blmm_multi*": OPAQUE, "blmm_table*": OPAQUE,
    "const blmm_table*": OPAQUE, "void*": OPAQUE,
    "blmm_ctx**": {"Ref{Ptr{Cvoid}}"}, "blmm_multi**": {"Ref{Ptr{Cvoid}}"}, "blmm_table**": {"Ref{Ptr{Cvoid}}"},
    "const double*": {"Ptr{Float64}"}, "double*": {"Ptr{Float64}"}, "float*": {"Ptr{Float32}"},
    "const int32_t*": {"Ptr{Int32}"}, "int32_t*": {"Ptr{Int32}"}, "const int*": {"Ptr{Int32}", "Ptr{Cint}"},
    "int64_t*": {"Ptr{Int64}", "Ref{Int64}"},
    "int64_t": {"Int64"}, "uint64_t": {"UInt64"}, "int": {"Cint"}, "double": {"Float64", "Cdouble"},
    "const blmm_opts*": {"Ref{BlmmOpts}"}, "blmm_opts*": {"Ref{BlmmOpts}"},
    "blmm_status*": {"Ref{BlmmStatus}", "Ptr{Cvoid}"},      # Ptr{Cvoid}: C_NULL, or the per-device array of bulkscan_multi
    "const blmm_multi_opts*": {"Ref{BlmmMultiOpts}"}, "const char*": {"Cstring"}, "const blmm_reduced*": {"Ref{BlmmReduced}"},
    "const int64_t*": {"Ptr{Int64}"},
}
RET = {"int": {"Cint"}, "void": {"Cvoid"}, "const char*": {"Cstring"}, "int64_t": {"Int64"}, "void*": {"Ptr{Cvoid}"}}


def test_struct_fields_match_the_header():
    cs, js = c_structs(), julia_structs()
    for jname, cname in STRUCTS.items():
        assert jname in js, jname
        got = [(f, JL2C[t]) for f, t in js[jname]]
        assert got == cs[cname], f"{jname} vs {cname}:\n{got}\n{cs[cname]}"


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_struct_layout_matches_offsetof(tmp_path):
    """Julia lays an isbits struct out like C (fields in order, natural alignment): the offsets that rule gives the Julia
    declaration must be the ones the C compiler gives the header's struct."""
    cs, js = c_structs(), julia_structs()
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HDR}"', "int main(void) {"]
    for cname in STRUCTS.values():
        for f, _ in cs[cname]:
            src.append(f'  printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
        src.append(f'  printf("{cname}.sizeof %zu\\n", sizeof({cname}));')
    src += ["  return 0;", "}"]
    (tmp_path / "lay.c").write_text("\n".join(src))
    subprocess.run(["gcc", "-o", str(tmp_path / "lay"), str(tmp_path / "lay.c")], check=True)
    real = dict(line.split() for line in subprocess.run([str(tmp_path / "lay")], capture_output=True, text=True, check=True).stdout.splitlines())
    for jname, cname in STRUCTS.items():
        off, align = 0, 1
        for f, t in js[jname]:
            sz = SIZE[JL2C[t]]
            off = (off + sz - 1) // sz * sz
            assert int(real[f"{cname}.{f}"]) == off, (jname, f, off, real[f"{cname}.{f}"])
            off += sz
            align = max(align, sz)
        assert int(real[f"{cname}.sizeof"]) == (off + align - 1) // align * align, jname


def test_every_ccall_matches_its_prototype():
    protos = c_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 25
    seen = set()
    for sym, ret, types, nvals in calls:
        assert sym in protos, f"ccall of {sym}: not declared in include/bulklmm_hip.h"
        cret, cargs = protos[sym]
        seen.add(sym)
        assert ret in RET[cret], f"{sym}: return type {ret} for C `{cret}`"
        assert len(types) == len(cargs), f"{sym}: {len(types)} argument types for {len(cargs)} C parameters"
        assert nvals == len(types), f"{sym}: {nvals} values passed for {len(types)} argument types"
        for k, (jt, ct) in enumerate(zip(types, cargs)):
            assert ct in ALLOWED, (sym, ct)
            assert jt in ALLOWED[ct], f"{sym}: argument {k + 1} is {jt} for C `{ct}`"
    # the binding covers the entry points a Julia caller of the reference's API needs
    need = {"blmm_create", "blmm_kinship", "blmm_kinship_rounded", "blmm_bulkscan", "blmm_bulkscan_multi", "blmm_scan_perms",
            "blmm_scan_perms_f32", "blmm_scan_alt", "blmm_lod2log10p", "blmm_last_log10p", "blmm_get_thresholds", "blmm_lod_threshold",
            "blmm_lod_colmax", "blmm_host_alloc", "blmm_host_free", "blmm_host_register", "blmm_host_unregister", "blmm_read_csv",
            "blmm_read_he", "blmm_create_multi", "blmm_destroy_multi", "blmm_multi_ndev", "blmm_set_log10p_output",
            "blmm_bulkscan_reduced", "blmm_last_lod_colmax", "blmm_last_lod_threshold", "blmm_last_lod_columns", "blmm_last_dims",
            "blmm_set_tuning"}
    assert need <= seen, sorted(need - seen)


def test_library_exports_what_the_binding_calls():
    import ctypes
    lib_path = os.path.join(ROOT, "bulklmm.jl_amd", "csrc", "libbulklmm_hip.so")
    if not os.path.exists(lib_path):
        pytest.skip("library not built")
    lib = ctypes.CDLL(lib_path)
    for sym, *_ in julia_ccalls():
        getattr(lib, sym)
