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
