"""Host logic of the sweep runner (polymer_stats_amd/sweep.py, tools/run_sweep.py) and of the aggregator twin: the case
lists, file names and CSV of the reference's run/*.jl and scripts/aggregate_mcmc.jl.  No GPU needed."""
import os

import numpy as np
import pytest

from polymer_stats_amd import aggregate_mcmc as ag
from polymer_stats_amd import sweep as sw


def test_axes_follow_the_reference_grids():
    # run/K1_E0-kT-phase.jl:21-24
    E0s, kTs = sw.axis_values("0.0:0.2:5.0"), sw.axis_values("10^(-2:0.2:2)")
    assert len(E0s) == 26 and E0s[0] == 0.0 and E0s[-1] == 5.0 and E0s[3] == pytest.approx(0.6, abs=1e-15)
    assert len(kTs) == 21 and kTs[0] == pytest.approx(1e-2) and kTs[-1] == pytest.approx(1e2) and kTs[10] == pytest.approx(1.0)
    assert sw.axis_values("1:25") == list(range(1, 26)) and all(isinstance(v, int) for v in sw.axis_values("1:25"))
    assert sw.axis_values("0.1,1,10") == [0.1, 1, 10] and sw.axis_values("100") == [100]
    assert sw.axis_values("0:0.05:1")[-1] == 1.0 and len(sw.axis_values("0:0.05:1")) == 21   # run/noninteracting-compare...:21
    Fzs = sw.axis_values("0.0:0.05:1.0,1.5:0.5:5.0")                  # vcat(0.0:0.05:1.0, 1.5:0.5:5.0)
    assert len(Fzs) == 21 + 8 and Fzs[20] == 1.0 and Fzs[21] == 1.5 and Fzs[-1] == 5.0
    with pytest.raises(ValueError):
        sw.axis_values("1:0:5")
    with pytest.raises((ValueError, SyntaxError)):
        sw.axis_values("__import__('os')")


def test_case_product_order_skip_and_names():
    # run/interacting_dielectric_study.jl:19-33: `for b in bs, n in ns, ... K1 in K1s, K2 in K2s; if K1==K2 continue`
    axes = [("b", [0.5, 1.0, 2.0]), ("n", [100, 200]), ("Fx", [0.0, 0.5, 1.0, 2.0]), ("Fz", [0.0, 0.5, 1.0, 2.0]), ("kT", [1.0]),
            ("E0", [1e-1, 1.0, 10.0]), ("K1", [0.0, 1e-1, 0.5, 1.0, 2.0]), ("K2", [0.0, 1e-1, 0.5, 1.0, 2.0])]
    cases = [c for c in sw.product_cases(axes) if not sw.skip_case("K1==K2", c)]
    assert len(cases) == 3 * 2 * 4 * 4 * 3 * 20
    assert cases[0] == dict(b=0.5, n=100, Fx=0.0, Fz=0.0, kT=1.0, E0=0.1, K1=0.0, K2=0.1)      # innermost axis varies fastest
    assert cases[1]["K2"] == 0.5 and cases[4]["K1"] == 0.1 and cases[-1]["b"] == 2.0
    spec = sw.name_spec("E0,K1,K2,kT,Fz,Fx,n,b", None)
    assert sw.case_name(cases[0], spec) == "E0-0000100_K1-0000000_K2-0000100_kT-0001000_Fz-0000000_Fx-0000000_n-0100000_b-0000500"
    # default token order = the reference's; run is %03d (run/Ising_2025-12-18.jl:11-15), raw on request (run/K1_E0-kT-phase.jl:15)
    c = dict(run=7, b=1, n=100, Fx=0, Fz=-0.5, kT=0.01, E0=5, K1=0.0, K2=1, kappa=0.25)
    assert sw.case_name(c, sw.name_spec(None, list(c))) == \
        "E0-0005000_K1-0000000_K2-0001000_kT-0000010_Fz--000500_Fx-0000000_n-0100000_b-0001000_kappa-0000250_run-007"
    assert sw.case_name(c, sw.name_spec("E0,run:raw", None)) == "E0-0005000_run-7"
    assert sw.fmt(0.0005) == "0000000" and sw.fmt(0.0015) == "0000002" and sw.fmt(1e-2 * 10 ** 0.2) == "0000016"   # round(Int, .) ties to even
    assert sw.skip_case("K1 == K2 || n > 100", dict(K1=1, K2=0, n=200)) and not sw.skip_case("K1==K2 && n>100", dict(K1=1, K2=0, n=200))
    with pytest.raises(ValueError):
        sw.skip_case("K3 == 1", dict(K1=1))
    with pytest.raises(ValueError):
        sw.skip_case("__import__('os').system('true')", dict(K1=1))


def test_plan_builds_the_mains_own_options(tmp_path):
    cases = [dict(E0=1, K1=0.0, K2=1, kT=0.1, Fz=0, Fx=1, n=100, b=0.5, kappa=0.0, run=r) for r in (1, 2)] + \
            [dict(E0=1, K1=0.0, K2=1, kT=1, Fz=0, Fx=0, n=200, b=1, kappa=0.0, run=1)]
    fixed = ["--chain-type", "dielectric", "--energy-type", "Ising", "--num-steps", "2500000", "--burn-in", "100000", "-v", "2",
             "--stepout", "250"]                                                              # run/Ising_2025-12-18.jl:113
    pl = sw.plan("mcmc_clustering_eap_chain", fixed, cases, str(tmp_path), num_chains=8, seed=100)
    assert [p["seed"] for p in pl] == [100, 101, 102] and [p["num-monomers"] for p in pl] == [100, 100, 200]
    assert pl[0]["mlen"] == 0.5 and pl[0]["kT"] == 0.1 and pl[0]["Fx"] == 1.0 and pl[0]["bend-mod"] == 0.0 and pl[0]["energy-type"] == "Ising"
    assert pl[0]["num-steps"] == 2500000 and pl[0]["burn-in"] == 100000 and pl[0]["num-chains"] == 8
    assert pl[0]["prefix"] == str(tmp_path / "E0-0001000_K1-0000000_K2-0001000_kT-0000100_Fz-0000000_Fx-0001000_n-0100000_b-0000500_kappa-0000000_run-001")
    assert pl[1]["_out"].endswith("_run-002.out")
    assert sw._signature(pl[0]) == sw._signature(pl[1]) != sw._signature(pl[2])               # one ensemble per chain length
    # any option of the main can be an axis, integer ones included; its key is then the option's own name
    pl2 = sw.plan("mcmc_clustering_eap_chain", fixed, [dict(E0=1, n=50.0, **{"num-steps": 4000, "cluster-prob": 0.25})], str(tmp_path), seed=1)
    assert pl2[0]["num-steps"] == 4000 and pl2[0]["cluster-prob"] == 0.25 and pl2[0]["num-monomers"] == 50
    assert pl2[0]["_name"] == "E0-0001000_n-0050000_num-steps-4000000_cluster-prob-0000250"
    with pytest.raises(ValueError, match="share the file name"):
        sw.plan("mcmc_clustering_eap_chain", fixed, cases, str(tmp_path), name="E0,K1", seed=1)
    with pytest.raises(SystemExit):                      # the fixed-force main has no --bend-mod: its own parser says so
        sw.plan("mcmc_eap_chain", [], [dict(E0=1, kappa=0.5)], str(tmp_path), seed=1)
    with pytest.raises(KeyError):
        sw.plan("mcmc_eap_chain", [], [dict(E0=1)], str(tmp_path), name="E0,K1", seed=1)


def test_aggregator_reads_names_and_lines_like_the_reference(tmp_path):
    from polymer_stats_amd import mcmc_clustering_eap_chain as cm, mcmc_eap_chain as fm
    from polymer_stats_amd.mcmc_eap_chain import Averager
    d = tmp_path / "w"
    d.mkdir()
    v = lambda *x: Averager(np.array(x, dtype=float), None)
    sas = [Averager(x, None) for x in (57.25, 1.5e-7, -3.0, 9.0, 0.4, 1.2)]
    vas = [v(1, 2, 3), v(4, 5, 6), v(0.1, 0.2, 1e7), v(7, 8, 9)]
    p = dict(mlen=0.5, **{"num-monomers": 100})
    (d / "E0-0001000_K1-0000000_K2-0001000_kT-0000100_Fz--000500_Fx-0000000_n-0100000_b-0000500_kappa-0000250_run-002.out").write_text(
        "\n".join(cm.summary_lines(sas, vas, 0.25, p)) + "\n")
    (d / "notes.txt").write_text("not an output file\n")
    out = tmp_path / "agg.csv"
    assert ag.main([str(out), str(d), "*.out", "dielectric", "3D", "true", "true"]) == 0
    rows = out.read_text().splitlines()
    assert rows[0] == "E0,K1,K2,kT,Fz,Fx,n,b,kappa," + ",".join(ag.OUT_3D)
    assert rows[1] == "1.0,0.0,1.0,0.1,-0.5,0.0,100.0,0.5,0.25,1.0,2.0,3.0,0.02,0.04,0.06,4.0,5.0,6.0,57.25,0.1,0.2,1.0e7,7.0,8.0,9.0," \
                      "1.5e-7,-3.0,9.0,0.4,1.2,0.25"
    # the fixed-force main's ten lines: 20 values, no Ealign / psi
    d2 = tmp_path / "w2"
    d2.mkdir()
    (d2 / "E0-0001000_mu-0000500_kT-0001000_Fz-0000000_Fx-0000000_n-0100000_b-0001000.out").write_text(
        "\n".join(fm.summary_lines(sas[:4], vas, float("nan"), dict(mlen=1.0, **{"num-monomers": 100}))) + "\n")
    assert ag.main([str(out), str(d2), "E0-*", "polar"]) == 0
    rows = out.read_text().splitlines()
    assert rows[0].split(",")[:7] == ["E0", "mu", "kT", "Fz", "Fx", "n", "b"] and len(rows[0].split(",")) == 7 + 20
    assert rows[1].endswith(",9.0,NaN") and len(rows[1].split(",")) == 27
    assert ag.main([str(out)]) == 1 and ag.main([str(out), str(d), "*.out", "rubber"]) == 1
    assert ag.main([str(out), str(d), "*.out", "dielectric", "2D"]) == 1
    # a file name with fewer tokens than the header has input columns would put every value under the wrong heading (the
    # reference does so silently, scripts/aggregate_mcmc.jl:54,67-74): refused, nothing misaligned is written
    d3 = tmp_path / "w3"
    d3.mkdir()
    (d3 / "E0-0001000_Fz-0000500_n-0100000.out").write_text(
        "\n".join(fm.summary_lines(sas[:4], vas, 0.5, dict(mlen=1.0, **{"num-monomers": 100}))) + "\n")
    assert ag.main([str(tmp_path / "bad.csv"), str(d3), "*.out", "dielectric"]) == 1
    assert ag.main([str(tmp_path / "bad.csv"), str(d), "*.out", "dielectric", "3D", "true"]) == 1   # run token left in: 10 != 9


def test_csv_files_fall_back_to_appending_when_descriptors_run_out(tmp_path):
    """run_sweep batches up to thousands of cases into one ensemble; --csv then needs two files per case.  Handles stay open
    only while they fit RLIMIT_NOFILE with room to spare, otherwise every row is appended -- same bytes either way."""
    import resource
    from polymer_stats_amd.mcmc_eap_chain import CsvFiles
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    texts = {}
    try:
        for tag, lim in (("open", soft), ("append", 80)):
            resource.setrlimit(resource.RLIMIT_NOFILE, (lim, hard))
            f = CsvFiles([str(tmp_path / f"{tag}{k}") for k in range(40)], [f"step,h{k}" for k in range(40)], "step,r")
            assert f.keep_open == (tag == "open") and len(f) == 40
            for step in (500, 1000):
                for k in range(40):
                    f.rows(k, f"{step}.0,{k}.0", f"{step}.0,{k}.5")
            f.close()
            texts[tag] = [(tmp_path / f"{tag}{k}_trajectory.csv").read_text() + (tmp_path / f"{tag}{k}_rolling.csv").read_text()
                          for k in range(40)]
    finally:
        resource.setrlimit(resource.RLIMIT_NOFILE, (soft, hard))
    assert texts["open"] == texts["append"] and texts["open"][7] == "step,h7\n500.0,7.0\n1000.0,7.0\nstep,r\n500.0,7.5\n1000.0,7.5\n"


def test_run_sweep_cli_needs_cases_and_a_gpu(tmp_path):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_sweep.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "no cases" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_sweep.py"), str(tmp_path / "d"), "--dry-run", "--axis", "b=0.5,1,2",
                        "--axis", "n=100,200", "--axis", "E0=0.1,1,10", "--axis", "K1=0,1", "--axis", "K2=0,1", "--skip", "K1==K2",
                        "--num-chains", "4", "--", "--energy-type", "interacting", "--num-steps", "500000", "-v", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "36 cases (0 already there), 2 ensemble(s)" in r.stdout and "n = 200, interacting, dielectric, 500000 steps: 18 cases, 72 chains" in r.stdout
    assert "E0-0000100_K1-0000000_K2-0001000_n-0100000_b-0000500.out" in r.stdout and not (tmp_path / "d").exists()
    # under an external launcher every rank would draw its own fresh base seed: refused before anything runs
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_sweep.py"), str(tmp_path / "x"), "--axis", "n=10", "--gpus", "2"],
                       capture_output=True, text=True, env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "needs --seed" in r.stderr
    import polymer_stats_amd as ps
    if ps._lib.load().pstat_device_count() < 1:           # the product has no CPU path: say so, write nothing
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_sweep.py"), str(tmp_path / "w"), "--axis", "n=10"],
                           capture_output=True, text=True)
        assert r.returncode != 0 and "needs a GPU" in r.stderr
        assert not list((tmp_path).glob("w/*.out"))
