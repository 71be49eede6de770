"""N > 1 path on CPU: two processes (gloo) each own a shard of the chains (disjoint global chain
ids), build their reduction vectors and merge them with ONE all-reduce(SUM) -- exactly what
bench.py / a multi-GPU job does with RCCL.  The merged summary must equal the single-process one.
The per-chain numbers come from the CPU oracle (this is a test); the merge arithmetic is the
library's own host function pstat_summary_from_reduction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCHAINS, NSTEPS = 24, 3000


def reduction_vector(sums, norm, nacc, steps, extra):
    # 16 observables, acceptance ratio, and the clustering main's two extras (zero in the first main)
    m = np.concatenate([sums / norm[:, None], (nacc / steps)[:, None], extra / norm[:, None]], axis=1)
    # trailing two entries: non-finite-energy rejections and collapsed chains (additive counts; zero here)
    return np.concatenate([[m.shape[0]], m.sum(0), (m ** 2).sum(0), [0.0, 0.0]])


def job(ob, mode):
    if mode == "cluster":
        return ob.make_params(n=10, E0=1.0, K1=1.0, Fz=0.5, num_steps=NSTEPS, seed=5, cluster_prob=0.5, bend_mod=0.4,
                              burn_in=500, burn_sched=[10.0, 1.0])
    return ob.make_params(n=10, E0=1.0, K1=1.0, Fz=0.5, num_steps=NSTEPS, seed=5)


def worker(rank, world, port, out, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = NCHAINS // world
    sums, norm, nacc, extra = ob.run_many(job(ob, mode), rank * per, per, nthreads=1 if world > 2 else 2, mode=mode, extras=True)   # shard by global chain id
    red = torch.from_numpy(reduction_vector(sums, norm, nacc, NSTEPS, extra))
    dist.all_reduce(red)                                                          # the one collective
    if rank == 0:
        np.save(out, red.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("fast", 2), ("cluster", 2), ("fast", 8)])
def test_two_rank_merge_equals_single_process(tmp_path, oracle, mode, world):
    """world = 8: the rank count of the driver's scaling run (the only one that matters; it cannot be rehearsed on one GPU --
    the pool allows six GPU processes per card -- so the eight-rank rendezvous and merge are rehearsed here, on the CPU)."""
    import polymer_stats_amd as ps
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "red.npy")
    mp.spawn(worker, args=(world, port, out, mode), nprocs=world, join=True)
    merged = np.load(out)
    sums, norm, nacc, extra = oracle.run_many(job(oracle, mode), 0, NCHAINS, nthreads=4, mode=mode, extras=True)
    single = reduction_vector(sums, norm, nacc, NSTEPS, extra)
    np.testing.assert_allclose(merged, single, rtol=1e-12)
    assert merged.shape == (ps.NRED,)
    s = ps.summary_from_reduction(merged, NSTEPS)
    m = sums / norm[:, None]
    np.testing.assert_allclose(np.array(s.avg), m.mean(0), rtol=1e-12)
    np.testing.assert_allclose(np.array(s.stderr), m.std(0, ddof=1) / np.sqrt(NCHAINS), rtol=1e-8)
    assert s.num_chains == NCHAINS
    np.testing.assert_allclose(np.array(s.extra_avg), (extra / norm[:, None]).mean(0), rtol=1e-12, atol=1e-300)
    if mode == "cluster":
        assert s.extra_avg[0] > 0 and 0 < s.extra_avg[1] < np.pi       # <sum cos^2 theta>, <psi>
