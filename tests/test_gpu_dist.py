"""N > 1 on the GPU box: two processes (gloo, both on device 0) each own a shard of the chains by global
chain id, advance it with libpstat, reduce on the device and merge with ONE all-reduce(SUM) of the 41-double
vector -- the data path of bench.py --gpus N with gloo standing in for RCCL.  The merged summary must equal
the one of a single handle holding all chains (chains are identified by id, not by owner)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCHAINS, NSTEPS = 512, 3000
KW = dict(n=30, E0=1.0, K1=1.0, K2=0.1, Fz=0.6, seed=91)


def worker(rank, world, port, out, moves):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import polymer_stats_amd as ps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = NCHAINS // world
    p = ps.default_params(num_chains=per, chain_id0=rank * per, precision=ps.F64, move_set=moves, **KW)
    with ps.Ensemble(p) as e:
        e.advance(NSTEPS)
        red = torch.from_numpy(e.reduce_host())
    dist.all_reduce(red)
    if rank == 0:
        np.save(out, red.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("moves", [0, 1])
def test_two_ranks_on_one_gpu_merge_to_the_single_handle_result(tmp_path, moves):
    import torch.multiprocessing as mp
    import polymer_stats_amd as ps
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "red.npy")
    mp.spawn(worker, args=(2, port, out, moves), nprocs=2, join=True)
    merged = ps.summary_from_reduction(np.load(out), NSTEPS)
    with ps.Ensemble(ps.default_params(num_chains=NCHAINS, precision=ps.F64, move_set=moves, **KW)) as e:
        e.advance(NSTEPS)
        single = e.summary()
    assert merged.num_chains == single.num_chains == NCHAINS
    np.testing.assert_allclose(np.array(merged.avg), np.array(single.avg), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(np.array(merged.stderr), np.array(single.stderr), rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(np.array(merged.extra_avg), np.array(single.extra_avg), rtol=1e-11, atol=1e-12)
    assert merged.acceptance_ratio == pytest.approx(single.acceptance_ratio, rel=1e-12)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a bare shell (no torchrun, WORLD_SIZE unset): the parent never touches the GPU, starts
    two fresh rank processes (here both on device 0, gloo standing in for RCCL), relays rank 0's single JSON line and
    exits 0; the line pools both ranks' chains.  A failing rank must fail the whole command."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--chains", "4096", "--mc-steps", "3000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["steps"] == 2
    assert d["check"]["chains_pooled"] == 2 * 4096
    assert d["value"] == pytest.approx(2 * 4096 * 3000 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-6)
    assert d["fast_path"]["dtype"].startswith("f32") and d["fast_path"]["check"]["chains_pooled"] == 2 * 4096
    assert "cpu_baseline" not in d                      # rank 0 at N = 1 only
    bad = subprocess.run(cmd + ["--precision", "nonsense"], env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and bad.stdout.strip() == ""
    # RCCL with more ranks than GPUs: the rank without a device exits at once, the launcher stops the other one instead of
    # leaving it in the rendezvous, and nothing is printed on stdout
    import polymer_stats_amd as ps
    if ps._lib.load().pstat_device_count() == 1:
        over = subprocess.run([a if a != "gloo" else "nccl" for a in cmd], env=env, capture_output=True, text=True, timeout=300)
        assert over.returncode != 0 and over.stdout.strip() == "" and "only 1 GPU(s) visible" in over.stderr


def _scan(tmp_path, name, *extra, timeout=900):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = str(tmp_path / name)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "phase_scan.py"), "--points", "8", "--n", "48", "--steps", "1500",
           "--burn-in", "400", "--burn-schedule", "10,1", "--out", out, *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    return r, out


@pytest.mark.parametrize("main", ["clustering", "fixed-force"])
def test_phase_scan_two_ranks_equal_one_rank_with_all_the_chains(tmp_path, main):
    """BASELINE configs[4]'s workflow, multi-rank (the reference: pmap over grid points, run/K1_E0-kT-phase.jl:19-45):
    `python tools/phase_scan.py --gpus 2 --backend gloo` from a bare shell starts two ranks (both on device 0 here),
    each owning 64 of every grid point's 128 chains by global chain id, and merges with ONE all-reduce of the
    [points x 41] reduction tensor; the CSV must equal the single-rank scan holding all 128 chains to 1e-11.  Annealed
    ladder included (every rung a fresh mcmc() call in the clustering main).  A failing rank fails the command."""
    common = ["--main", main, "--energy", "Ising", "--precision", "f64"]
    two, out2 = _scan(tmp_path, "two.csv", "--gpus", "2", "--backend", "gloo", "--chains", "64", *common)
    assert two.returncode == 0, two.stderr[-3000:]
    assert "2 rank(s) [gloo]" in two.stderr
    one, out1 = _scan(tmp_path, "one.csv", "--chains", "128", *common)
    assert one.returncode == 0, one.stderr[-3000:]
    a = np.loadtxt(out1, delimiter=",", skiprows=1)
    b = np.loadtxt(out2, delimiter=",", skiprows=1)
    assert a.shape == b.shape == (8, 12) and np.all(a[:, 2] == 128) and np.all(b[:, 2] == 128)
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-12)
    assert np.all(np.isfinite(a)) and np.all((a[:, 11] > 0) & (a[:, 11] < 1))        # acceptance ratios
    if main == "clustering":
        bad, _ = _scan(tmp_path, "bad.csv", "--gpus", "2", "--backend", "gloo", "--chains", "64", "--precision", "nonsense")
        assert bad.returncode != 0
        # nccl with more ranks than GPUs: every rank beyond the device count exits, the launcher stops the rest
        if __import__("polymer_stats_amd")._lib.load().pstat_device_count() == 1:
            over, _ = _scan(tmp_path, "over.csv", "--gpus", "2", "--backend", "nccl", "--chains", "64", *common, timeout=300)
            assert over.returncode != 0 and "only 1 GPU(s) visible" in over.stderr


SIX = 4      # ranks of the GPU-box rehearsals: the pool allows six processes on one card and the test runner, which has used
             # the GPU in earlier tests, is one of them (six ranks + the runner were killed by the process guard).  The
             # eight-rank rendezvous, launcher and merge are rehearsed on the CPU (tests/test_dist_gloo.py, tests/test_host.py)


def test_bench_four_ranks_pool_the_chains_of_one_rank(tmp_path):
    """`bench.py --gpus 4 --backend gloo` from a bare shell: four children on one port (all on device 0), ONE JSON line; its
    pooled check values equal those of a single rank holding all 4 x 1024 chains (chains are identified by global id)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--backend", "gloo", "--steps", "2", "--warmup", "1", "--mc-steps", "500", "--no-cpu-baseline", "--no-fast-path",
              "--no-rng-named", "--no-configs"]
    six = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(SIX), "--chains", "1024", *common],
                         env=env, capture_output=True, text=True, timeout=900)
    assert six.returncode == 0, six.stderr[-3000:]
    lines = [l for l in six.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, six.stdout
    d6 = json.loads(lines[0])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--chains", str(SIX * 1024), *common],
                         env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads(one.stdout.strip())
    assert d6["n_gpus"] == SIX and d6["check"]["chains_pooled"] == SIX * 1024 == d1["check"]["chains_pooled"]
    for k in ("r3", "p3", "U", "AR", "r3_stderr"):
        assert d6["check"][k] == pytest.approx(d1["check"][k], rel=1e-11, abs=1e-12), k
    assert d6["value"] == pytest.approx(SIX * 1024 * 500 * 2 / (d6["ms_per_step"] * 2e-3), rel=1e-6)


def test_phase_scan_four_ranks_equal_one_rank(tmp_path):
    """BASELINE configs[4]'s workflow with four ranks (one port, all on device 0): the CSV equals the single-rank scan holding
    all the chains to 1e-11."""
    common = ["--main", "fixed-force", "--energy", "Ising", "--precision", "f64"]
    six, out6 = _scan(tmp_path, "six.csv", "--gpus", str(SIX), "--backend", "gloo", "--chains", "16", *common)
    assert six.returncode == 0, six.stderr[-3000:]
    assert f"{SIX} rank(s) [gloo]" in six.stderr
    one, out1 = _scan(tmp_path, "one.csv", "--chains", str(16 * SIX), *common)
    assert one.returncode == 0, one.stderr[-3000:]
    a = np.loadtxt(out1, delimiter=",", skiprows=1)
    b = np.loadtxt(out6, delimiter=",", skiprows=1)
    assert a.shape == b.shape == (8, 12) and np.all(b[:, 2] == 16 * SIX)
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-12)


def test_bench_line_carries_one_measured_entry_per_baseline_config(tmp_path):
    """The default bench line's `configs` array (rank 0 at N = 1): C1..C5 in f64 at BASELINE.json's sizes, each with its
    kernel, rate, roofline fraction and both CPU baselines (oracle faithful and fast) -- the headline keys untouched."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--mc-steps", "3000",
                        "--chains", "4096", "--cpu-seconds", "0.5", "--config-cpu-seconds", "0.2", "--no-fast-path", "--no-rng-named"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["cpu_baseline"]["kind"] == "port"
    cfgs = d["configs"]
    assert [c["id"] for c in cfgs] == ["C1", "C2", "C3", "C4", "C5"]
    kernels = {"C1": "sweep_kernel<double>", "C2": "state in L2", "C3": "state in L2", "C4": "interacting_kernel<double>", "C5": "state in L2"}
    for c in cfgs:
        assert kernels[c["id"]] in c["kernel"] and c["dtype"] == "f64" and c["value"] > 0 and c["kernel_ms"] > 0
        rf = c["roofline"]
        assert 0 < rf["frac"] < 1 and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"]) and rf["unit"] in ("GB/s", "TFLOP/s")
        cb = c["cpu_baseline"]
        assert cb["cores"] >= 1 and cb["faithful"]["value"] > 0 and cb["fast"]["value"] > 0
        assert cb["gpu_over_faithful"] == pytest.approx(c["value"] / cb["faithful"]["value"])
        assert c["check"]["chains_collapsed"] >= 0
    c4 = cfgs[3]
    assert c4["roofline"]["unit"] == "TFLOP/s" and c4["chains"] == 16384 and c4["n"] == 64 and c4["mc_steps"] == 20000
    assert cfgs[4]["chains"] == 546 * 128 and cfgs[4]["n"] == 200
    assert cfgs[1]["cpu_baseline"]["faithful"]["value"] == d["cpu_baseline"]["value"]       # C2 = the headline's own sample
    scan = d["phase_scan"]            # the clustering main on the reference's own phase-scan ensemble: one chain per wavefront
    assert scan["chains"] == 2730 and scan["workgroups"] == 2730 and "cluster_chain_wave_kernel" in scan["kernel"]
    assert 0 < scan["us_per_step"] < 50 and scan["value"] == pytest.approx(2730 / (scan["us_per_step"] * 1e-6), rel=1e-3)
