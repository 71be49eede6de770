"""The sweep runner on the GPU: the cases of a run/*.jl-style sweep in batched launches (polymer_stats_amd/sweep.py).
Every case's .out must hold what the single-case host prints for the same options, seed and chains (the batching is
invisible), cases are partitioned over ranks without anything depending on the partition, finished cases are kept
(the run scripts' `isfile(outfile)`), and the files reach the closed-form fixture through the aggregator's CSV."""
import contextlib
import io
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(main, pargs) -> list[str]:
    """The single-case host on the options the sweep planned for that case (its own prefix, so nothing is overwritten)."""
    p = {k: v for k, v in pargs.items() if not k.startswith("_")}
    p["prefix"] = pargs["prefix"] + "_single"
    p["stepout"] = 0 if main.__name__.endswith("mcmc_eap_chain") else p["stepout"]
    if main.__name__.endswith("clustering_eap_chain"):
        sas, vas, ar = main.run(p)
    else:
        sas, vas, ar = main.mcmc(p["num-steps"], p)
    return main.summary_lines(sas, vas, ar, p)


def _values(lines):
    from polymer_stats_amd.aggregate_mcmc import julia_value
    return np.array([x for l in lines for x in julia_value(l.split("=")[1])])


@pytest.mark.parametrize("main_name,fixed,cases", [
    ("mcmc_eap_chain", ["--chain-type", "dielectric", "--num-steps", "3000", "-v", "0", "--num-inits", "2"],
     [dict(E0=1.0, K1=1.0, Fz=fz, n=n, b=b) for n in (20, 50) for fz in (0.0, 1.0) for b in (1.0, 0.5)]),
    ("mcmc_eap_chain", ["--chain-type", "polar", "--energy-type", "interacting", "--num-steps", "1500", "-v", "0", "--burn-in", "300",
                        "--burn-schedule", "[10; 1]"],
     [dict(E0=e0, mu=0.5, Fz=0.5, n=16, kT=kt) for e0 in (0.5, 1.0) for kt in (1.0, 2.0)]),
    ("mcmc_clustering_eap_chain", ["--chain-type", "dielectric", "--energy-type", "Ising", "--num-steps", "2500", "--burn-in", "400",
                                   "--burn-schedule", "[100; 10; 1]", "-v", "0", "--stepout", "250"],
     [dict(E0=e0, K1=1.0, K2=0.0, kT=kt, Fz=0, Fx=0, n=n, b=1, kappa=kappa, run=1)
      for n in (24, 100) for e0, kt, kappa in ((1.0, 1.0, 0.0), (2.0, 0.1, 0.0), (0.4, 10.0, 0.5))]),
    ("mcmc_clustering_eap_chain", ["--energy-type", "cutoff", "--cutoff-radius", "3.0", "--num-steps", "800", "--burn-in", "100", "-v", "0"],
     [dict(E0=1.0, K1=k1, n=20, Fz=0.3) for k1 in (0.5, 1.0)]),
], ids=["fixed-force-ni", "fixed-force-interacting", "clustering-ising", "clustering-cutoff"])
def test_batched_sweep_writes_what_the_single_case_host_prints(tmp_path, main_name, fixed, cases):
    from polymer_stats_amd import sweep as sw
    res = sw.run_sweep(main_name, fixed, cases, str(tmp_path), num_chains=32, seed=77)
    assert sorted(res["ran"]) == sorted(p["_name"] for p in sw.plan(main_name, fixed, cases, str(tmp_path), num_chains=32, seed=77))
    assert res["launches"] == len({c["n"] for c in cases}) and not res["skipped"]             # one ensemble per chain length
    main = sw.MAINS[main_name]
    for p in sw.plan(main_name, fixed, cases, str(tmp_path), num_chains=32, seed=77):
        got = open(p["_out"]).read()
        assert got.endswith("\n") and len(got.splitlines()) == (12 if main is sw.cluster_main else 10)
        want = _single(main, p)
        assert [l.split("=")[0] for l in got.splitlines()] == [l.split("=")[0] for l in want]
        np.testing.assert_allclose(_values(got.splitlines()), _values(want), rtol=1e-11, atol=1e-11, err_msg=p["_name"])


def test_finished_cases_are_kept_and_any_partition_gives_the_same_files(tmp_path):
    from polymer_stats_amd import sweep as sw
    fixed = ["--energy-type", "Ising", "--num-steps", "1200", "--burn-in", "200", "-v", "0"]
    cases = [dict(E0=e0, kT=kt, n=n, run=r) for r in (1, 2) for n in (12, 30) for e0 in (0.5, 1.5) for kt in (0.5, 2.0)]
    kw = dict(num_chains=16, seed=5, name="E0,kT,n,run:int")
    one, three = tmp_path / "one", tmp_path / "three"
    a = sw.run_sweep("mcmc_clustering_eap_chain", fixed, cases, str(one), **kw)
    assert len(a["ran"]) == 16 and a["launches"] == 2
    parts = [sw.run_sweep("mcmc_clustering_eap_chain", fixed, cases, str(three), rank=r, world=3, **kw) for r in (2, 0, 1)]
    assert sorted(sum((p["ran"] for p in parts), [])) == sorted(a["ran"]) and [len(p["ran"]) for p in parts] == [5, 6, 5]
    for nm in a["ran"]:
        assert (one / (nm + ".out")).read_text() == (three / (nm + ".out")).read_text(), nm
    # chunked into several ensembles per chain length: still the same files
    four = tmp_path / "four"
    c = sw.run_sweep("mcmc_clustering_eap_chain", fixed, cases, str(four), max_chains=48, **kw)
    assert c["launches"] == 6
    for nm in a["ran"]:
        assert (one / (nm + ".out")).read_text() == (four / (nm + ".out")).read_text(), nm
    # a second pass runs nothing and touches nothing; a removed file alone is redone, identically
    victim = one / (a["ran"][3] + ".out")
    text, stamp = victim.read_text(), {f: os.stat(one / f).st_mtime_ns for f in os.listdir(one)}
    victim.unlink()
    b = sw.run_sweep("mcmc_clustering_eap_chain", fixed, cases, str(one), **kw)
    assert b["ran"] == [a["ran"][3]] and len(b["skipped"]) == 15 and victim.read_text() == text
    assert all(os.stat(one / f).st_mtime_ns == t for f, t in stamp.items() if f != victim.name)
    assert not [f for f in os.listdir(one) if ".tmp" in f]


def test_csv_option_writes_every_cases_two_files(tmp_path):
    from polymer_stats_amd import sweep as sw, mcmc_clustering_eap_chain as cm
    cases = [dict(E0=1.0, Fz=fz, n=10) for fz in (0.0, 1.0)]
    sw.run_sweep("mcmc_eap_chain", ["--num-steps", "2000", "--stepout", "500", "-v", "0"], cases, str(tmp_path / "f"), num_chains=8,
                 seed=1, write_csv=True)
    for p in sw.plan("mcmc_eap_chain", ["--num-steps", "2000", "--stepout", "500", "-v", "0"], cases, str(tmp_path / "f"), num_chains=8, seed=1):
        roll = open(p["prefix"] + "_rolling.csv").read().splitlines()
        traj = open(p["prefix"] + "_trajectory.csv").read().splitlines()
        assert roll[0].startswith("step,r1,r2,r3") and [r.split(",")[0] for r in roll[1:]] == ["500.0", "1000.0", "1500.0", "2000.0"]
        assert len(traj) == 5 and all(len(r.split(",")) == 8 for r in traj)
        out = _values(open(p["_out"]).read().splitlines())
        np.testing.assert_allclose([float(x) for x in roll[-1].split(",")[1:4]], out[0:3], rtol=1e-12, atol=1e-12)   # <r> = last rolling row
    fixed = ["--energy-type", "noninteracting", "--num-steps", "600", "--burn-in", "100", "--stepout", "300", "-v", "0"]
    sw.run_sweep("mcmc_clustering_eap_chain", fixed, cases, str(tmp_path / "c"), num_chains=8, seed=1, write_csv=True)
    for p in sw.plan("mcmc_clustering_eap_chain", fixed, cases, str(tmp_path / "c"), num_chains=8, seed=1):
        traj = open(p["prefix"] + "_trajectory.csv").read().splitlines()
        assert traj[0] == cm.traj_header(10) and len(traj) == 3 and all(len(r.split(",")) == 8 + 5 * 10 for r in traj)
        assert open(p["prefix"] + "_rolling.csv").read().splitlines()[0] == cm.ROLL_HEADER


def test_cli_with_two_ranks_and_the_aggregate_reaches_the_closed_form(tmp_path, golden):
    """BASELINE configs[1]'s force sweep the way run/noninteracting-compare-with-clustering_2021-09-24.jl writes it: one
    .out per Fz, read back through the aggregator's CSV, against tests/golden (the analytic values the reference says
    it was verified against, README.md:32)."""
    work, agg = tmp_path / "w", tmp_path / "agg.csv"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_sweep.py"), str(work), "--gpus", "2", "--share-gpu", "--num-chains", "1024",
           "--seed", "9", "--axis", "E0=1", "--axis", "K1=1", "--axis", "K2=0", "--axis", "kT=1", "--axis", "Fz=0,1,5", "--axis", "Fx=0",
           "--axis", "n=100", "--axis", "b=1", "--aggregate", str(agg),
           "--", "--chain-type", "dielectric", "--num-steps", "20000", "--burn-in", "4000", "-v", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0 of 2: 2 cases" in r.stderr and "rank 1 of 2: 1 cases" in r.stderr
    rows = agg.read_text().splitlines()
    head = rows[0].split(",")
    assert head[:8] == ["E0", "K1", "K2", "kT", "Fz", "Fx", "n", "b"] and len(rows) == 4
    tab = np.array([[float(x) for x in row.split(",")] for row in rows[1:]])
    col = {h: tab[:, i] for i, h in enumerate(head)}
    assert list(col["Fz"]) == [0.0, 1.0, 5.0] and set(col["n"]) == {100.0}
    for i, key in enumerate(["cfg2_n100_E0_1_K1_1_Fz0", "cfg2_n100_E0_1_K1_1_Fz1", "cfg2_n100_E0_1_K1_1_Fz5"]):
        g = golden[key]["avg"]
        # 1 024 chains x 20 000 steps, ~200 steps of autocorrelation: standard errors ~0.02 on <r_z>, ~0.13 on <r_x^2> (variance
        # 2 <r_x^2>^2), ~0.03 on <U>; bounds at >= 5 of those (tests/test_gpu_validation.py runs the same configuration at
        # 5 sigma with MEASURED errors -- this test is about the files)
        for name, gk, rel, ab in (("r3", "r3", 0, 0.15), ("p3", "p3", 0, 0.15), ("U", "U", 5e-3, 0.2), ("r1sq", "r1sq", 0, 0.8),
                                  ("rsquared", "rsq", 1e-2, 1.0), ("Usquared", "Usq", 1e-2, 1.0), ("psquared", "psq", 1e-2, 1.0)):
            assert col[name][i] == pytest.approx(g[gk], rel=rel or None, abs=ab), (key, name, col[name][i], g[gk])
        assert col["lambda3"][i] == pytest.approx(col["r3"][i] / 100.0, rel=1e-12, abs=1e-15)
    # again: everything is there, nothing runs
    r = subprocess.run(cmd[:cmd.index("--aggregate")] + cmd[cmd.index("--"):], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 cases run" in r.stderr and "3 already there" in r.stderr


def test_cli_explicit_case_list_one_chain_per_case_and_options(tmp_path):
    """An explicit case list (run/Ising_2025-12-18.jl spells its 23 cases out), ONE chain per case -- literally the reference's
    sweep --, the fast path and the named generator, CSV files on, then --overwrite; through the command line, one rank."""
    import json
    cases = [dict(E0=1, K2=1, K1=0.0, kT=1, Fz=0, Fx=0, n=30, b=1, run=r) for r in (1, 2)] + \
            [dict(E0=5, K2=0.1, K1=0.0, kT=0.1, Fz=1, Fx=0, n=30, b=0.5, run=1), dict(E0=1, K2=1, K1=0.0, kT=1, Fz=0, Fx=1, n=12, b=2.0, run=1)]
    spec = tmp_path / "cases.json"
    spec.write_text(json.dumps(cases))
    work = tmp_path / "w"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_sweep.py"), str(work), "--main", "mcmc_clustering_eap_chain", "--cases", str(spec),
           "--num-chains", "1", "--seed", "21", "--precision", "f32", "--rng", "xoshiro128++", "--csv",
           "--", "--chain-type", "dielectric", "--energy-type", "Ising", "--bend-mod", "0.0", "--num-steps", "3000", "--burn-in", "500",
           "-v", "2", "--stepout", "1000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "4 cases run in 2 ensembles" in r.stderr and "cluster_kernel<float>" in r.stderr
    names = sorted(f for f in os.listdir(work) if f.endswith(".out"))
    assert names == sorted(["E0-0001000_K1-0000000_K2-0001000_kT-0001000_Fz-0000000_Fx-0000000_n-0030000_b-0001000_run-001.out",
                            "E0-0001000_K1-0000000_K2-0001000_kT-0001000_Fz-0000000_Fx-0000000_n-0030000_b-0001000_run-002.out",
                            "E0-0005000_K1-0000000_K2-0000100_kT-0000100_Fz-0001000_Fx-0000000_n-0030000_b-0000500_run-001.out",
                            "E0-0001000_K1-0000000_K2-0001000_kT-0001000_Fz-0000000_Fx-0001000_n-0012000_b-0002000_run-001.out"])
    first = {f: (work / f).read_text() for f in names}
    assert first[names[0]] != first[names[1]] or "run-001" not in names[0]            # two runs of one case: different seeds
    assert first[sorted(names)[0]].count("\n") == 12
    for f in names:
        roll = (work / f.replace(".out", "_rolling.csv")).read_text().splitlines()
        assert [x.split(",")[0] for x in roll[1:]] == ["1000.0", "2000.0", "3000.0"] and roll[0].endswith("Ealign,psi")
        assert len((work / f.replace(".out", "_trajectory.csv")).read_text().splitlines()) == 4
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and "0 cases run" in r2.stderr
    r3 = subprocess.run(cmd[:3] + ["--overwrite"] + cmd[3:], capture_output=True, text=True, timeout=300)
    assert r3.returncode == 0 and "4 cases run" in r3.stderr
    assert {f: (work / f).read_text() for f in names} == first                       # same seed, same files
    # an option the main does not have is the main's own parser error, before anything runs
    bad = subprocess.run(cmd[:-2] + ["--no-such-option", "1"], capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "no-such-option" in bad.stderr


def test_a_case_does_not_depend_on_what_shares_its_ensemble_even_past_the_resident_slots():
    """What INTEGRATION.md 3a promises about co-batching, at the size where it gets hard: 1 300 cases x 64 chains are more
    chain blocks than the chip has resident workgroups, so pstat_advance cuts the launch into time segments (and, for an
    f32 sweep of many chains, may pick another state home) -- choices that depend on how many cases are co-batched.  Per
    case: the f64 TRAJECTORIES (angles, generator, counters, step sizes) are identical whatever shares the ensemble; the
    running sums of the f64 clustering main agree to ~1e-10 (its cached n-hat is mapped at a reflection and re-derived at
    a segment boundary), those of the f64 sweep to rounding (1e-13: the 128-step blocks in which records are folded into
    the sums start at the segment boundaries)."""
    import polymer_stats_amd as ps
    for main_kw, exact in ((dict(), True), (dict(move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, bend_mod=0.2), False)):
        mk = lambda i: ps.default_params(n=48, E0=0.2 + 0.001 * i, K1=1.0, K2=0.1, Fz=0.3, kT=0.8 + 0.0005 * i, seed=70000 + i,
                                         num_chains=64, precision=ps.F64, **main_kw)
        big = [mk(i) for i in range(1300)]
        with ps.Ensemble(big) as e:
            info = e.launch_info()
            assert info.blocks > info.blocks_per_cu * info.num_cus, (info.blocks, info.blocks_per_cu, info.num_cus)
            e.advance(4400)
            picked = {i: ([e.chain_state(i * 64 + k) for k in (0, 33, 63)], e.rolling(i)[0]) for i in (0, 649, 1299)}
        for i, (states, avg) in picked.items():
            with ps.Ensemble([mk(i), mk((i + 7) % 1300)]) as small:          # the same case in other company: one launch, one segment
                small.advance(4400)
                for k, a in zip((0, 33, 63), states):
                    b = small.chain_state(k)
                    for key in ("theta", "phi", "rng"):
                        assert np.array_equal(a[key], b[key]), (i, k, key)
                    assert a["nacc_total"] == b["nacc_total"] and a["phi_step"] == b["phi_step"]
                    np.testing.assert_allclose(a["sums"], b["sums"], rtol=1e-13 if exact else 1e-10, atol=1e-9 if exact else 1e-8)
                np.testing.assert_allclose(avg, small.rolling(0)[0], rtol=1e-10, atol=1e-10)
