"""End-to-end drop-in test on the GPU: the Python twin of the reference's command line writes the two
CSV files and the ten stdout lines in the reference's formats (mcmc_eap_chain.jl:256-259,329-348,
386-395), and the numbers in them agree with the CPU oracle run under the same options."""
import io
import contextlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cli_outputs_match_reference_formats_and_oracle(tmp_path, oracle):
    from polymer_stats_amd import mcmc_eap_chain as host
    prefix = str(tmp_path / "run")
    argv = ["-n", "20", "-e", "1.0", "-J", "1.0", "-F", "0.5", "-N", "6000", "-s", "1500", "-v", "0",
            "--num-chains", "2048", "--seed", "3", "--prefix", prefix]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert host.main(argv) == 0
    lines = buf.getvalue().strip().splitlines()
    assert [l.split("=")[0].strip() for l in lines] == ["<r>", "<r/nb>", "<rj2>", "<r2>", "<p>", "<pj2>", "<p2>", "<U>", "<U2>", "AR"]
    vals = {l.split("=")[0].strip(): np.array(eval(l.split("=")[1]), dtype=float) for l in lines}   # as aggregate_mcmc.jl:71 does
    np.testing.assert_allclose(vals["<r/nb>"], vals["<r>"] / 20.0, rtol=1e-12)
    assert vals["<r2>"] == pytest.approx(vals["<rj2>"].sum(), rel=1e-9)

    traj = open(prefix + "_trajectory.csv").read().strip().splitlines()
    roll = open(prefix + "_rolling.csv").read().strip().splitlines()
    assert traj[0] == "step,r1,r2,r3,p1,p2,p3,U"
    assert roll[0] == "step,r1,r2,r3,r1sq,r2sq,r3sq,rsq,p1,p2,p3,p1sq,p2sq,p3sq,psq,U,Usq"
    assert len(traj) == 1 + 4 and len(roll) == 1 + 4
    assert [r.split(",")[0] for r in roll[1:]] == ["1500.0", "3000.0", "4500.0", "6000.0"]   # Float64-formatted step
    last = np.array([float(x) for x in roll[-1].split(",")])
    np.testing.assert_allclose(last[1:4], vals["<r>"], rtol=1e-12)
    np.testing.assert_allclose(last[15], vals["<U>"], rtol=1e-12)
    assert all(len(r.split(",")) == 8 for r in traj[1:]) and all(len(r.split(",")) == 17 for r in roll[1:])

    # same options through the oracle (faithful mode), pooled over independent chains
    P = oracle.make_params(n=20, E0=1.0, K1=1.0, Fz=0.5, num_steps=6000, seed=3, stepout=1500)
    sums, norm, nacc = oracle.run_many(P, 10 ** 6, 192, nthreads=8, mode="faithful")
    m = sums / norm[:, None]
    se = m.std(0, ddof=1) / np.sqrt(m.shape[0])
    got = np.r_[vals["<r>"], vals["<rj2>"], vals["<r2>"], vals["<p>"], vals["<pj2>"], vals["<p2>"], vals["<U>"], vals["<U2>"]]
    z = (got - m.mean(0)) / (se * 1.03 + 1e-12)      # GPU side: 2048 chains, its own error is ~0.3 of the oracle's
    assert np.all(np.abs(z) < 4.5), z
    assert abs(float(vals["AR"]) - nacc.mean() / 6000) < 0.01


def test_cli_multi_init_and_umbrella_run(tmp_path):
    from polymer_stats_amd import mcmc_eap_chain as host
    prefix = str(tmp_path / "m")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        host.main(["-n", "12", "-e", "0.8", "-F", "0.3", "-N", "2000", "-M", "2", "-I", "-B", "-s", "1000", "-v", "0",
                   "--num-chains", "256", "--prefix", prefix, "-T", "polar", "-m", "0.7", "-u", "Ising",
                   "--rng", "xoshiro128++", "--precision", "f64"])
    assert len(buf.getvalue().strip().splitlines()) == 10
    roll = open(prefix + "_rolling.csv").read().strip().splitlines()
    assert [r.split(",")[0] for r in roll[1:]] == ["1000.0", "2000.0", "1000.0", "2000.0"]   # step restarts per init


def test_clustering_cli_outputs_and_oracle(tmp_path, oracle):
    """The clustering main end to end: burn-in ladder, twelve stdout lines, the two CSV files with the
    clustering main's extra columns (mcmc_clustering_eap_chain.jl:253-259,312-335,389-400); numbers
    against the oracle's literal restatement run under the same options."""
    from polymer_stats_amd import mcmc_clustering_eap_chain as host
    prefix = str(tmp_path / "cl")
    n, N, burn = 12, 6000, 1500
    argv = ["-n", str(n), "-e", "1.0", "-F", "0.5", "-u", "noninteracting", "-a", "0.5", "-g", "0.2", "--cluster-prob", "0.5",
            "-N", str(N), "--burn-in", str(burn), "--burn-schedule", "[10; 1]", "-s", "2000", "-v", "0", "-U", "0.55",
            "--num-chains", "2048", "--seed", "4", "--prefix", prefix]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert host.main(argv) == 0
    lines = buf.getvalue().strip().splitlines()
    keys = [l.split("=")[0].strip() for l in lines]
    assert keys == ["<r>", "<r/nb>", "<rj2>", "<r2>", "<p>", "<pj2>", "<p2>", "<U>", "<U2>", "<cos2(θ)>", "<ψ>", "AR"]
    vals = {k: np.array(eval(l.split("=")[1]), dtype=float) for k, l in zip(keys, lines)}
    traj = open(prefix + "_trajectory.csv").read().strip().splitlines()
    roll = open(prefix + "_rolling.csv").read().strip().splitlines()
    assert traj[0] == host.traj_header(n) and roll[0] == host.ROLL_HEADER
    assert [r.split(",")[0] for r in roll[1:]] == ["2000.0", "4000.0", "6000.0"]      # only the production run's rows
    assert all(len(r.split(",")) == 8 + 5 * n for r in traj[1:]) and all(len(r.split(",")) == 19 for r in roll[1:])
    last = np.array([float(x) for x in roll[-1].split(",")])
    np.testing.assert_allclose(last[17:19], [vals["<cos2(θ)>"], vals["<ψ>"]], rtol=1e-12)
    row = np.array([float(x) for x in traj[-1].split(",")])
    ang = row[8:8 + 2 * n].reshape(n, 2)
    mus = row[8 + 2 * n:].reshape(n, 3)
    np.testing.assert_allclose(mus.sum(0), row[4:7], rtol=1e-5, atol=1e-6)             # p = sum of the mu_i columns
    np.testing.assert_allclose(np.cos(ang[:, 1]).sum(), row[3], rtol=1e-5, atol=1e-5)  # r3 = b sum cos(theta_i)

    P = oracle.make_params(n=n, E0=1.0, Fz=0.5, bend_mod=0.5, bend_angle=0.2, cluster_prob=0.5, num_steps=N, seed=4,
                           burn_in=burn, burn_sched=[10.0, 1.0], adj_ub=0.55)
    sums, norm, nacc = oracle.run_many(P, 10 ** 6, 384, nthreads=8, mode="cluster")
    m = sums / norm[:, None]
    se = m.std(0, ddof=1) / np.sqrt(m.shape[0])
    got = np.r_[vals["<r>"], vals["<rj2>"], vals["<r2>"], vals["<p>"], vals["<pj2>"], vals["<p2>"], vals["<U>"], vals["<U2>"]]
    z = (got - m.mean(0)) / (se * 1.1 + 1e-12)
    assert np.all(np.abs(z) < 4.5), z
    assert abs(float(vals["AR"]) - nacc.mean() / N) < 0.01


def test_cli_device_sharding_is_invisible(tmp_path):
    """--devices shards the chains by global chain id over several handles and merges their additive reduction
    vectors on the host: the same job on one handle or split over two (here both on device 0) prints the
    same numbers."""
    from polymer_stats_amd import mcmc_eap_chain as host
    outs = []
    for tag, dev in (("a", "0"), ("b", "0,0")):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            host.main(["-n", "30", "-e", "1.0", "-F", "0.6", "-N", "4000", "-s", "2000", "-v", "0", "--num-chains", "1000",
                       "--seed", "12", "--devices", dev, "--precision", "f64", "--prefix", str(tmp_path / tag)])
        outs.append({l.split("=")[0].strip(): np.array(eval(l.split("=")[1]), dtype=float)
                     for l in buf.getvalue().strip().splitlines()})
    for k in outs[0]:
        np.testing.assert_allclose(outs[1][k], outs[0][k], rtol=1e-11, atol=1e-12, err_msg=k)
    ra = open(tmp_path / "a_rolling.csv").read().splitlines()
    rb = open(tmp_path / "b_rolling.csv").read().splitlines()
    assert len(ra) == len(rb) == 3
    np.testing.assert_allclose([float(x) for x in rb[-1].split(",")], [float(x) for x in ra[-1].split(",")], rtol=1e-11, atol=1e-12)
    assert open(tmp_path / "a_trajectory.csv").read() == open(tmp_path / "b_trajectory.csv").read()   # chain 0 is chain 0


def _run_main(host, argv):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert host.main(argv) == 0
    return buf.getvalue()


def test_default_seed_gives_independent_runs_and_explicit_seed_reproduces(tmp_path):
    """Seed contract (reference: unseeded RNG; its sweeps repeat one command and use the scatter,
    run/interacting-compare-with-clustering_2021-09-28.jl:26-27, run/K1_E0-kT-phase.jl:19,30)."""
    from polymer_stats_amd import mcmc_eap_chain as host
    from polymer_stats_amd import mcmc_clustering_eap_chain as chost
    base = ["-n", "16", "-e", "1.0", "-F", "0.5", "-N", "2000", "-s", "0", "-v", "0", "--num-chains", "128"]
    a = _run_main(host, base + ["--prefix", str(tmp_path / "a")])
    b = _run_main(host, base + ["--prefix", str(tmp_path / "b")])
    assert a.splitlines()[0].startswith("<r>") and a.splitlines()[0] != b.splitlines()[0]
    c = _run_main(host, base + ["--seed", "11", "--prefix", str(tmp_path / "c")])
    d = _run_main(host, base + ["--seed", "11", "--prefix", str(tmp_path / "d")])
    assert c == d
    cb = ["-n", "12", "-e", "1.0", "-F", "0.5", "-N", "1500", "--burn-in", "300", "-s", "0", "-v", "0", "--num-chains", "64"]
    e = _run_main(chost, cb + ["--prefix", str(tmp_path / "e")])
    f = _run_main(chost, cb + ["--prefix", str(tmp_path / "f")])
    assert e.splitlines()[0] != f.splitlines()[0]
    g = _run_main(chost, cb + ["--seed", "3", "--prefix", str(tmp_path / "g")])
    h = _run_main(chost, cb + ["--seed", "3", "--prefix", str(tmp_path / "h")])
    assert g == h


def test_numeric_type_wide_merge_matches_float64_and_warns(tmp_path, capsys):
    """--numeric-type (mcmc_eap_chain.jl:186-197): the merge over chains runs in the wide type; the result agrees with
    the Float64 device reduction to rounding, and stderr says what was done."""
    from polymer_stats_amd import mcmc_eap_chain as host
    base = ["-n", "16", "-e", "1.0", "-F", "0.5", "-N", "3000", "-s", "0", "-v", "2", "--num-chains", "256", "--seed", "9"]
    a = _run_main(host, base + ["--prefix", str(tmp_path / "a")])
    capsys.readouterr()
    b = _run_main(host, base + ["--numeric-type", "float128", "--prefix", str(tmp_path / "b")])
    err = capsys.readouterr().err
    assert "--numeric-type float128" in err and "Float64 on the device" in err
    for la, lb in zip(a.splitlines(), b.splitlines()):
        va, vb = (np.array(eval(l.split("=")[1]), dtype=float) for l in (la, lb))
        np.testing.assert_allclose(vb, va, rtol=1e-12, atol=1e-13)


def test_plain_c_client_runs_the_ensemble(tmp_path):
    """tests/c/abi_smoke.c: a C program on include/pstat.h alone -- create, burn-in, reset, advance, summary, destroy --
    lands on the freely-jointed-chain closed form (BASELINE configs[0])."""
    import subprocess
    from test_abi import build_c_client
    exe = build_c_client(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "chains = 2048" in r.stdout and "steps = 30000" in r.stdout and "nan_rejects = 0" in r.stdout


def test_cli_uniform_bits_option_reaches_the_device(tmp_path):
    """--uniform-bits (ours, like --precision): 0 and 53 are the same f64 run; 23 is the f32-style Metropolis draw (the same
    stream, so the same output unless an eps fell into [u, u + 2^-23)); 53 with --precision f32 is refused by pstat_create."""
    from polymer_stats_amd import _lib, mcmc_clustering_eap_chain as chost, mcmc_eap_chain as host
    outs = {}
    for bits in ("0", "53", "23"):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            assert host.main(["-n", "16", "-e", "1.0", "-F", "0.4", "-N", "3000", "-v", "0", "--num-chains", "128", "--seed", "9",
                              "--prefix", str(tmp_path / ("u" + bits)), "--uniform-bits", bits]) == 0
        outs[bits] = buf.getvalue()
    assert outs["0"] == outs["53"] and len(outs["23"].splitlines()) == 10
    with pytest.raises(_lib.PstatError) as ei:
        host.main(["-n", "16", "-N", "100", "-v", "0", "--num-chains", "64", "--prefix", str(tmp_path / "bad"), "--precision", "f32",
                   "--uniform-bits", "53"])
    assert "uniform_bits" in str(ei.value)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert chost.main(["-n", "16", "-e", "1.0", "-N", "800", "-v", "0", "--num-chains", "64", "--seed", "9",
                           "--prefix", str(tmp_path / "c"), "--uniform-bits", "23"]) == 0
    assert len(buf.getvalue().splitlines()) == 12
