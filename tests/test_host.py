"""CPU tests of the host mirror of the reference's interface (polymer_stats_amd/mcmc_eap_chain.py,
julia_fmt.py): option table, error behaviour, output formatting, reduction -> summary arithmetic."""
import math

import numpy as np
import pytest

import polymer_stats_amd as ps
from polymer_stats_amd import mcmc_eap_chain as host
from polymer_stats_amd.julia_fmt import jl_float, jl_row, jl_vector

# (long flag, short alias, default) read off mcmc_eap_chain.jl:19-153
REFERENCE_TABLE = [
    ("--E0", "-e", 0.0), ("--chain-type", "-T", "dielectric"), ("--K1", "-J", 1.0), ("--K2", "-K", 0.0),
    ("--mu", "-m", 1e-2), ("--energy-type", "-u", "noninteracting"), ("--kT", "-k", 1.0),
    ("--ensemble-type", "-E", "force"), ("--Fz", "-F", 0.0), ("--Fx", "-G", 0.0), ("--rz", "-z", 0.0),
    ("--rx", "-x", 0.0), ("--mlen", "-b", 1.0), ("--num-monomers", "-n", 100), ("--num-steps", "-N", 100000),
    ("--num-inits", "-M", 1), ("--force-init", "-I", False), ("--phi-step", "-p", 3 * math.pi / 8),
    ("--do-flips", None, False), ("--theta-step", "-q", 3 * math.pi / 16), ("--chain-frac-step", "-f", 0.15),
    ("--step-adjust-lb", "-L", 0.15), ("--step-adjust-ub", "-U", 0.55), ("--step-adjust-scale", "-A", 1.1),
    ("--steps-per-adjust", "-S", 2500), ("--acc", "-a", "metropolis"), ("--umbrella-sampling", "-B", False),
    ("--update-freq", None, 15.0), ("--verbose", "-v", 3), ("--prefix", "-P", "eap-mcmc"), ("--postfix", "-Q", ""),
    ("--stepout", "-s", 500), ("--numeric-type", None, "float64"), ("--profile", "-Z", False),
]


def test_option_table_matches_reference():
    assert len(REFERENCE_TABLE) == 34
    d = host.parse_args([])
    for long, short, default in REFERENCE_TABLE:
        key = long[2:]
        assert key in d, key
        assert d[key] == default and type(d[key]) is type(default), (key, d[key], default)
    # every short alias parses to the same key
    for long, short, default in REFERENCE_TABLE:
        if short is None:
            continue
        val = {bool: None, int: "7", float: "0.25", str: "polar"}[type(default)]
        argv = [short] if val is None else [short, val]
        got = host.parse_args(argv)[long[2:]]
        assert got == (True if val is None else type(default)(val)), (short, got)


def test_julia_float_formatting():
    cases = [(500.0, "500.0"), (35.02, "35.02"), (1e-5, "1.0e-5"), (0.0001, "0.0001"), (1e6, "1.0e6"),
             (999999.0, "999999.0"), (123456789012345678.0, "1.2345678901234568e17"), (-6.2875452, "-6.2875452"),
             (1.5e-7, "1.5e-7"), (0.1 + 0.2, "0.30000000000000004"), (0.0, "0.0"), (-0.0, "-0.0"), (1e21, "1.0e21"),
             (2.5e-5, "2.5e-5"), (float("inf"), "Inf"), (float("-inf"), "-Inf")]
    for x, want in cases:
        assert jl_float(x) == want, (x, jl_float(x), want)
    assert jl_float(float("nan")) == "NaN"
    # round trip through the parser the consumers use (a Julia literal is also a Python literal here)
    rng = np.random.default_rng(0)
    for x in np.concatenate([rng.normal(size=200) * 10.0 ** rng.integers(-12, 12, 200), [1e-300, 1e300]]):
        assert float(jl_float(x)) == x
    assert jl_vector([1.0, -2.5, 1e-7]) == "[1.0, -2.5, 1.0e-7]"
    assert jl_row([500, 1.25, -3.0]) == "500.0,1.25,-3.0"    # `step` prints as a Float64 (hcat promotes)


def test_summary_lines_shape():
    avg = np.arange(1.0, 17.0)
    sas = [host.Averager(avg[6], 0), host.Averager(avg[13], 0), host.Averager(avg[14], 0), host.Averager(avg[15], 0)]
    vas = [host.Averager(avg[0:3], 0), host.Averager(avg[3:6], 0), host.Averager(avg[7:10], 0), host.Averager(avg[10:13], 0)]
    lines = host.summary_lines(sas, vas, 0.6, host.default_pargs(**{"num-monomers": 10, "mlen": 2.0}))
    assert [l.split("=")[0].strip() for l in lines] == ["<r>", "<r/nb>", "<rj2>", "<r2>", "<p>", "<pj2>", "<p2>", "<U>", "<U2>", "AR"]
    assert lines[0] == "<r>    =   [1.0, 2.0, 3.0]"
    assert lines[1] == "<r/nb> =   [0.05, 0.1, 0.15]"
    assert lines[3] == "<r2>   =   7.0" and lines[9] == "AR     =   0.6"
    # the consumers split on '=' and eval the right-hand side (scripts/aggregate_mcmc.jl:71)
    for l in lines:
        rhs = l.split("=")[1]
        assert np.all(np.isfinite(np.array(eval(rhs), dtype=float)))


def test_reference_error_branches():
    with pytest.raises(host.ReferenceError_, match="acceptance criteria has not yet been implemented"):
        host.mcmc(10, host.default_pargs(acc="kawasaki"))
    with pytest.raises(host.ReferenceError_, match="numeric-type 'float16' not understood"):
        host.mcmc(10, host.default_pargs(**{"numeric-type": "float16"}))
    with pytest.raises(host.ReferenceError_, match="chain-type is not understood."):
        host.params_from_pargs(host.default_pargs(**{"chain-type": "rod"}), 1, 0, 0)
    with pytest.raises(host.ReferenceError_, match="energy-type is not understood."):
        host.params_from_pargs(host.default_pargs(**{"energy-type": "cutoff"}), 1, 0, 0)
    with pytest.raises(host.ReferenceError_, match="not implemented for the HPC env"):
        host.main(["--profile"])
    p = host.params_from_pargs(host.default_pargs(**{"energy-type": "Ising", "chain-type": "polar", "mlen": 2.0,
                                                     "do-flips": True}), 128, 64, 0)
    assert (p.energy_type, p.chain_type, p.b, p.do_flips, p.num_chains, p.chain_id0) == (ps.ISING, ps.POLAR, 2.0, 1, 128, 64)


def test_summary_from_reduction_is_pooled_mean_and_stderr():
    rng = np.random.default_rng(1)
    C = 37
    m = rng.normal(size=(C, 17)) + 5.0          # per-chain means: 16 observables + acceptance ratio
    red = np.zeros(ps.NRED)
    red[0] = C
    red[1:18] = m.sum(0)
    red[1 + ps.NQ:1 + ps.NQ + 17] = (m ** 2).sum(0)
    s = ps.summary_from_reduction(red, 1000)
    np.testing.assert_allclose(np.array(s.avg), m[:, :16].mean(0), rtol=1e-13)
    np.testing.assert_allclose(np.array(s.stderr), m[:, :16].std(0, ddof=1) / np.sqrt(C), rtol=1e-9)
    assert s.acceptance_ratio == pytest.approx(m[:, 16].mean())
    assert s.num_chains == C and s.steps_per_chain == 1000 and s.attempted_updates == C * 1000.0


# ------------------------------------------------------------------ clustering main's host twin
from polymer_stats_amd import mcmc_clustering_eap_chain as chost

# (long flag, short alias, default) read off mcmc_clustering_eap_chain.jl:14-152
CLUSTER_TABLE = [
    ("--E0", "-e", 0.0), ("--chain-type", "-T", "dielectric"), ("--K1", "-J", 1.0), ("--K2", "-K", 0.0),
    ("--mu", "-m", 1e-2), ("--bend-mod", "-a", 0.0), ("--bend-angle", "-g", 0.0), ("--energy-type", "-u", "Ising"),
    ("--cutoff-radius", None, 7.5), ("--kT", "-k", 1.0), ("--Fz", "-F", 0.0), ("--Fx", "-G", 0.0),
    ("--mlen", "-b", 1.0), ("--num-monomers", "-n", 100), ("--num-steps", "-N", 1000000),
    ("--phi-step", "-p", 3 * math.pi / 8), ("--theta-step", "-q", 3 * math.pi / 16), ("--cluster-prob", None, 0.5),
    ("--step-adjust-lb", "-L", 0.15), ("--step-adjust-ub", "-U", 0.40), ("--step-adjust-scale", "-A", 1.1),
    ("--steps-per-adjust", "-S", 2500), ("--umbrella-sampling", "-B", False), ("--update-freq", None, 15.0),
    ("--verbose", "-v", 3), ("--prefix", "-P", "eap-mcmc"), ("--postfix", "-Q", ""), ("--stepout", "-s", 500),
    ("--numeric-type", None, "float64"), ("--burn-in", None, 50000),
    ("--burn-schedule", None, "[1000; 100; 10; 2; 1]"), ("--dx0", None, "[2*pi, 1e-1]"), ("--profile", "-Z", False),
]


def test_cluster_option_table_matches_reference():
    d = chost.parse_args([])
    assert d["x0"] is None                      # no default: EAPChain(pargs) then draws uniform angles
    for long, short, default in CLUSTER_TABLE:
        key = long[2:]
        assert d[key] == default and type(d[key]) is type(default), (key, d[key], default)
        if short is not None:
            val = {bool: None, int: "7", float: "0.25", str: "polar"}[type(default)]
            got = chost.parse_args([short] if val is None else [short, val])[key]
            assert got == (True if val is None else type(default)(val)), (short, got)
    assert len(CLUSTER_TABLE) + 1 == 34         # the reference's table has 34 entries (x0 is the 34th)


def test_cluster_julia_vector_literals_and_params():
    assert chost.julia_vector("[2*pi, 1e-1]") == [2 * math.pi, 0.1]
    assert chost.julia_vector("[1000; 100; 10; 2; 1]") == [1000.0, 100.0, 10.0, 2.0, 1.0]
    assert chost.julia_vector("[]") == [] and chost.julia_vector("[π/2; -0.3]") == [math.pi / 2, -0.3]
    for bad in ("1, 2", "[__import__('os')]", "[a]", "[1; f(2)]"):
        with pytest.raises((ValueError, SyntaxError)):
            chost.julia_vector(bad)
    p = chost.params_from_pargs(chost.default_pargs(**{"x0": "[0.3; 1.2]", "bend-mod": 2.0, "bend-angle": 0.1,
                                                       "cluster-prob": 0.25, "energy-type": "noninteracting"}), 128, 64, 0)
    assert (p.move_set, p.use_x0, p.x0_phi, p.x0_theta, p.dx0_phi, p.dx0_theta) == (ps.MOVES_CLUSTER, 1, 0.3, 1.2, 2 * math.pi, 0.1)
    assert (p.bend_mod, p.bend_angle, p.cluster_prob, p.energy_type, p.adj_ub) == (2.0, 0.1, 0.25, ps.NONINTERACTING, 0.40)
    assert chost.params_from_pargs(chost.default_pargs(), 1, 0, 0).energy_type == ps.ISING
    with pytest.raises(host.ReferenceError_, match="Invalid input for 'x0'"):
        chost.params_from_pargs(chost.default_pargs(x0="[1; 2; 3]"), 1, 0, 0)
    with pytest.raises(host.ReferenceError_, match="energy-type is not understood."):
        chost.params_from_pargs(chost.default_pargs(**{"energy-type": "x"}), 1, 0, 0)
    p = chost.params_from_pargs(chost.default_pargs(**{"energy-type": "cutoff", "cutoff-radius": 3.5}), 1, 0, 0)
    assert (p.energy_type, p.cutoff_radius) == (ps.CUTOFF, 3.5)
    assert chost.params_from_pargs(chost.default_pargs(**{"energy-type": "interacting"}), 1, 0, 0).energy_type == ps.INTERACTING
    p = chost.params_from_pargs(chost.default_pargs(x0="[" + "; ".join(["0.1"] * 200) + "]"), 1, 0, 0)
    assert p.use_x0 == 0          # the per-monomer form is applied after creation (restart_from_x0)
    with pytest.raises(host.ReferenceError_):
        chost.main(["--profile"])


def test_cluster_output_shapes():
    assert chost.traj_header(2) == "step,r1,r2,r3,p1,p2,p3,U,phi1,theta1,phi2,theta2,mux1,muy1,muz1,mux2,muy2,muz2"
    assert chost.ROLL_HEADER.split(",")[-2:] == ["Ealign", "psi"] and len(chost.ROLL_HEADER.split(",")) == 19
    avg = np.arange(1.0, 17.0)
    sas = [host.Averager(avg[6], 0), host.Averager(avg[13], 0), host.Averager(avg[14], 0), host.Averager(avg[15], 0),
           host.Averager(33.5, 0), host.Averager(1.25, 0)]
    vas = [host.Averager(avg[0:3], 0), host.Averager(avg[3:6], 0), host.Averager(avg[7:10], 0), host.Averager(avg[10:13], 0)]
    lines = chost.summary_lines(sas, vas, 0.3, chost.default_pargs(**{"num-monomers": 10}))
    assert [l.split("=")[0].strip() for l in lines] == ["<r>", "<r/nb>", "<rj2>", "<r2>", "<p>", "<pj2>", "<p2>", "<U>",
                                                        "<U2>", "<cos2(θ)>", "<ψ>", "AR"]
    assert lines[9] == "<cos2(θ)>   =   33.5" and lines[10] == "<ψ>    =   1.25" and lines[11] == "AR     =   0.3"
    # dipoles written to the trajectory file (inc/dipole_response.jl:7-29)
    pa = chost.default_pargs(E0=2.0, K1=1.0, K2=0.25)
    mu = chost._dipoles(pa, np.array([0.0]), np.array([0.0]))
    np.testing.assert_allclose(mu, [[0.0, 0.0, 2.0]])           # n = z: mu = K1 E0 z
    mu = chost._dipoles(pa, np.array([0.0]), np.array([math.pi / 2]))
    np.testing.assert_allclose(mu, [[0.0, 0.0, 0.5]], atol=1e-15)   # n = x: mu = K2 E0 z


# ------------------------------------------------------------------ seed contract
def test_default_seed_is_fresh_entropy_and_echoed(capsys):
    """The reference never seeds its RNG (its sweeps launch one command 25x and use the scatter,
    run/interacting-compare-with-clustering_2021-09-28.jl:26-27): --seed defaults to fresh entropy, the seed drawn
    is echoed on stderr at -v >= 2 and kept in pargs so that every shard of the run uses the same one."""
    for mod in (host, chost):
        assert mod.parse_args([])["seed"] is None
        assert mod.parse_args(["--seed", "17"])["seed"] == 17
    seeds = {host.fresh_seed() for _ in range(64)}
    assert len(seeds) == 64 and all(0 <= s < 2 ** 63 for s in seeds)
    pa = host.default_pargs(verbose=2)
    s1 = host.resolve_seed(pa)
    err = capsys.readouterr().err
    assert f"seed: {s1}" in err and f"--seed {s1}" in err
    assert host.resolve_seed(pa) == s1 and capsys.readouterr().err == ""     # drawn once per run
    pb = host.default_pargs(verbose=1)
    s2 = host.resolve_seed(pb)
    assert s2 != s1 and capsys.readouterr().err == ""                        # quiet below -v 2
    pc = host.default_pargs(seed=5)
    assert host.resolve_seed(pc) == 5
    # both mains' parameter builders resolve it (shards are built from one pargs => one seed)
    p1 = host.params_from_pargs(pa, 4, 0, 0)
    p2 = host.params_from_pargs(pa, 4, 4, 0)
    assert p1.seed == p2.seed == s1
    pk = chost.default_pargs()
    q1 = chost.params_from_pargs(pk, 4, 0, 0)
    assert q1.seed == pk["seed"] and pk["seed"] is not None and pk["seed"] != s1


def test_rank_launcher_stops_the_survivors_when_a_rank_fails(tmp_path):
    """tools/rank_spawn.py (bench.py --gpus N, tools/phase_scan.py --gpus N): a rank that dies must not leave the others
    waiting in a rendezvous -- they are terminated and the launcher returns non-zero promptly; on success rank 0's stdout
    is relayed and the other ranks' stdout goes to stderr."""
    import os
    import subprocess
    import sys
    import time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, time\n"
        "r = int(os.environ['RANK']); mode = sys.argv[1]\n"
        "assert os.environ['WORLD_SIZE'] in ('3', '8') and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and int(os.environ['MASTER_PORT']) > 0\n"
        "if mode == 'fail' and r == 1: sys.exit(7)\n"
        "if mode == 'killed' and r == 5: time.sleep(1.0); os.kill(os.getpid(), 9)\n"
        "if mode in ('fail', 'killed'): time.sleep(600)\n"
        "print('hello from', r)\n")
    drv = ("import sys; sys.path.insert(0, %r); from rank_spawn import spawn_ranks; rc, out = spawn_ranks(%r, [sys.argv[1]], int(sys.argv[2])); "
           "sys.stdout.write(out); sys.exit(rc)") % (os.path.join(ROOT, "tools"), str(script))
    ok = subprocess.run([sys.executable, "-c", drv, "ok", "3"], capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0 and ok.stdout == "hello from 0\n" and "hello from 2" in ok.stderr
    t0 = time.time()
    bad = subprocess.run([sys.executable, "-c", drv, "fail", "3"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and bad.stdout == "" and "rank 1 exited with code 7" in bad.stderr
    assert time.time() - t0 < 30          # did not sit out the survivors' 600 s
    # eight ranks -- the driver's scaling run: one port, rank 0's stdout relayed, the others' on stderr; and a rank KILLED
    # mid-run (signal, not an exit code) stops the other seven promptly
    ok8 = subprocess.run([sys.executable, "-c", drv, "ok", "8"], capture_output=True, text=True, timeout=60)
    assert ok8.returncode == 0 and ok8.stdout == "hello from 0\n" and all(f"hello from {r}" in ok8.stderr for r in range(1, 8))
    t0 = time.time()
    k8 = subprocess.run([sys.executable, "-c", drv, "killed", "8"], capture_output=True, text=True, timeout=60)
    assert k8.returncode == 1 and k8.stdout == "" and "rank 5 exited with code -9" in k8.stderr
    assert time.time() - t0 < 15
