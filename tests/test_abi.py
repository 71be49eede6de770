"""CPU tests of the C-ABI library: it loads, exports every symbol include/pstat.h declares, its structs
have the layout the bindings assume, and it fails loudly (no CPU fallback).  No compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ps():
    import polymer_stats_amd as ps
    ps._lib.load()
    return ps


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pstat.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pstat_[A-Za-z_0-9]+)\s*\(", text)))


def test_exports_every_declared_symbol(ps):
    lib = ps._lib.load()
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"libpstat.so does not export {n}"
    assert sorted(ps._lib.SYMBOLS) == names, "binding's symbol list is out of date"
    assert lib.pstat_abi_version() == 6


def test_struct_layout_matches_header(ps, tmp_path):
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pstat.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(pstat_params), offsetof(pstat_params, steps_per_adjust),'
                   'offsetof(pstat_params, seed), offsetof(pstat_params, rng), sizeof(pstat_summary),'
                   'offsetof(pstat_summary, num_chains), sizeof(pstat_launch_info));'
                   'printf("%zu %zu %zu %zu\\n", offsetof(pstat_params, move_set), offsetof(pstat_params, bend_mod),'
                   'offsetof(pstat_params, use_x0), offsetof(pstat_summary, extra_avg));'
                   'printf("%zu %zu %d %d\\n", offsetof(pstat_params, cutoff_radius), offsetof(pstat_summary, nan_rejects),'
                   'PSTAT_NRED, PSTAT_NQ);return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    P, S, LI = ps._lib.Params, ps._lib.Summary, ps._lib.LaunchInfo
    assert got == [C.sizeof(P), P.steps_per_adjust.offset, P.seed.offset, P.rng.offset, C.sizeof(S),
                   S.num_chains.offset, C.sizeof(LI), P.move_set.offset, P.bend_mod.offset, P.use_x0.offset,
                   S.extra_avg.offset, P.cutoff_radius.offset, S.nan_rejects.offset, ps.NRED, ps.NQ]


def test_defaults_are_the_references(ps):
    import math
    p = ps.default_params()
    # mcmc_eap_chain.jl:19-153
    assert (p.E0, p.K1, p.K2, p.mu, p.kT, p.Fz, p.Fx, p.b) == (0.0, 1.0, 0.0, 1e-2, 1.0, 0.0, 0.0, 1.0)
    assert p.phi_step == 3 * math.pi / 8 and p.theta_step == 3 * math.pi / 16
    assert (p.adj_lb, p.adj_ub, p.adj_scale, p.steps_per_adjust) == (0.15, 0.55, 1.1, 2500)
    assert p.n == 100 and p.chain_type == ps.DIELECTRIC and p.energy_type == ps.NONINTERACTING
    assert p.rng == ps.RNG_MWC64X and p.precision == ps.F64 and p.uniform_bits == 0      # Float64 like the reference
    # mcmc_clustering_eap_chain.jl:36-43,87-90,146-148
    assert p.move_set == ps.MOVES_SINGLE and (p.bend_mod, p.bend_angle, p.cluster_prob) == (0.0, 0.0, 0.5)
    assert p.use_x0 == 0 and (p.dx0_phi, p.dx0_theta) == (2 * math.pi, 0.1) and p.cutoff_radius == 7.5


def test_strerror_and_invalid_arguments(ps):
    lib = ps._lib.load()
    assert lib.pstat_strerror(0) == b"ok"
    assert b"device" in lib.pstat_strerror(-2)
    h = C.c_void_p()
    for bad in (dict(n=0), dict(kT=-1.0), dict(chain_type=7), dict(energy_type=9), dict(num_chains=0),
                dict(precision=5), dict(phi_step=0.0), dict(rng=3), dict(uniform_bits=1), dict(uniform_bits=53, precision=0), dict(move_set=2),
                dict(move_set=1, cluster_prob=1.5), dict(move_set=1, do_flips=1), dict(bend_mod=1.0),
                dict(use_x0=1, x0_theta=float("nan")), dict(energy_type=3),
                dict(energy_type=3, move_set=1, cutoff_radius=0.0)):
        p = ps.default_params(**bad)
        rc = lib.pstat_create(C.byref(p), 1, None, C.byref(h))
        assert rc == -1, (bad, rc)
        assert lib.pstat_last_error() != b""
    assert lib.pstat_create(None, 1, None, C.byref(h)) == -1
    assert lib.pstat_advance(None, 10) == -1


def test_no_cpu_fallback(ps):
    """Without a GPU every compute entry point refuses; with one this test is skipped."""
    lib = ps._lib.load()
    if lib.pstat_device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(ps.PstatError) as ei:
        ps.Ensemble(ps.default_params(n=10))
    assert ei.value.code == -2 and "no CPU path" in str(ei.value)


def test_product_never_touches_the_oracle():
    """The package may not import, load or name anything under oracle/."""
    pkg = os.path.join(ROOT, "polymer_stats_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                code = "\n".join(l for l in text.splitlines() if "oracle" in l.lower()
                                 and not l.strip().startswith(("//", "#", "*", "/*", '"""')))
                assert "liboracle" not in text and "import oracle" not in text and "from oracle" not in text, f
                assert "eap_oracle" not in code, (f, code)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libpstat.so")]).decode()
    assert "oracle" not in out
    # tools/ measure and profile the product: they may not use the checker either.  bench.py may, in its cpu_baseline leg only.
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            text = open(os.path.join(ROOT, "tools", f), errors="ignore").read()
            assert "from oracle" not in text and "import oracle" not in text and "liboracle" not in text, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("from oracle") == 1 and bench.index("from oracle") > bench.index("def cpu_baseline")
    assert bench.index("from oracle") < bench.index("def parity_vs_cpu")


def test_julia_hosts_mirror_the_structs(ps):
    """The (unexecuted) Julia hosts declare PstatParams / PstatSummary by hand: same field names, order and
    widths as the ctypes mirror that test_struct_layout_matches_header pins against the C header."""
    import re
    jl_type = {C.c_double: "Cdouble", C.c_int64: "Int64", C.c_uint64: "UInt64", C.c_int32: "Int32"}
    for name in ("mcmc_eap_chain.jl", "mcmc_clustering_eap_chain.jl"):
        src = open(os.path.join(ROOT, "julia", name)).read()
        body = re.search(r"struct PstatParams\n(.*?)\nend", src, re.S).group(1)
        got = re.findall(r"(\w+)::(\w+)", body)
        want = [(f, jl_type[t]) for f, t in ps._lib.Params._fields_]
        assert got == want, name
        body = re.search(r"struct PstatSummary\n(.*?)\nend", src, re.S).group(1)
        got = re.findall(r"(\w+)::([\w{},]+)", body)
        want = []
        for f, t in ps._lib.Summary._fields_:
            want.append((f, jl_type[t] if t in jl_type else "NTuple{%d,Cdouble}" % (C.sizeof(t) // 8)))
        assert got == want, name
        # the positional constructor call passes exactly one argument per field
        i = src.index("  PstatParams(pargs[") + len("  PstatParams(")
        depth, args = 1, 1
        while depth > 0:
            c = src[i]
            if c in "([":
                depth += 1
            elif c in ")]":
                depth -= 1
            elif c == "," and depth == 1:
                args += 1
            elif c == "#":
                i = src.index("\n", i)
            i += 1
        assert args == len(ps._lib.Params._fields_), (name, args)
        # every ccall names a symbol the library exports
        for sym in set(re.findall(r"ccall\(\(:(\w+), LIBPSTAT\)", src)):
            assert sym in ps._lib.SYMBOLS, (name, sym)


def build_c_client(tmp_path):
    exe = tmp_path / "abi_smoke"
    pkg = os.path.join(ROOT, "polymer_stats_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(exe),
                           "-L", pkg, "-lpstat", "-lm", "-Wl,-rpath," + pkg])
    return exe


def test_plain_c_client_links_and_fails_loudly_without_a_gpu(ps, tmp_path):
    """The boundary is a C ABI: a C11 translation unit that includes only include/pstat.h links against libpstat.so and
    runs.  Without a GPU it gets PSTAT_ERR_NO_DEVICE and a message (exit code 2 of the client); with one it runs the
    ensemble (tests/test_gpu_host.py checks the numbers)."""
    exe = build_c_client(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    if ps._lib.load().pstat_device_count() > 0:
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 2, r.stdout + r.stderr
        assert "no HIP device" in r.stdout and "no CPU path" in r.stdout
