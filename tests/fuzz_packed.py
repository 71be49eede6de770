#!/usr/bin/env python3
"""Trajectory fuzz of PACKED batches (several few-chain cases per wave, csrc/pstat_device.h run_job_queue<true>) against the CPU
oracle (TEST TOOLING: it runs the oracle, like tests/ do).  Random batches -- 2..40 cases of 1..100 chains, both mains,
non-interacting and Ising energies, per-case physics, seeds and chain ids, either generator, either eps contract, a launch
split and time segments in every run -- forced into packed blocks; three chains per batch must equal the oracle's final
angles, generator state, acceptance count and step sizes bit for bit.

    python tests/fuzz_packed.py [trials=300] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polymer_stats_amd as ps
from helpers import both
from oracle import binding as ob

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
os.environ["PSTAT_PACK"] = "1"
nfail = npacked = 0
for trial in range(trials):
    cluster = bool(rng.integers(0, 2))
    et = int(rng.choice([0, 0, 2]))
    n = int(rng.integers(2, 12)) if rng.random() < 0.3 else int(rng.integers(12, 130))
    ncases = int(rng.integers(2, 41))
    nch = int(rng.choice([1, 2, 3, 5, 7, 16, 25, 33, 100]))
    common = dict(n=n, chain_type=int(rng.integers(0, 2)), energy_type=et, rng=int(rng.integers(0, 2)),
                  steps_per_adjust=int(rng.choice([50, 137, 400, 2500])), adj_scale=float(rng.choice([1.0, 1.1, 1.3])),
                  uniform_bits=int(rng.choice([0, 23])))
    if not cluster:
        common.update(do_flips=int(rng.integers(0, 2)), umbrella=int(rng.integers(0, 2)))
    cases = []
    for i in range(ncases):
        kw = dict(E0=float(rng.uniform(0, 2)), K1=float(rng.uniform(0, 1.2)), K2=float(rng.uniform(0, 0.5)),
                  mu=float(rng.uniform(0.01, 0.6)), kT=float(10 ** rng.uniform(-0.5, 0.7)), Fz=float(rng.uniform(-1, 2)),
                  Fx=float(rng.choice([0.0, rng.uniform(-1, 1)])), b=float(rng.uniform(0.5, 2.0)), seed=int(rng.integers(0, 2 ** 40)))
        if et == 2:
            kw.update(K1=kw["K1"] * 0.3, K2=kw["K2"] * 0.2, mu=kw["mu"] * 0.3)
        if cluster:
            kw.update(cluster_prob=float(rng.uniform(0.1, 0.9)), bend_mod=float(rng.choice([0.0, rng.uniform(0, 0.6)])),
                      bend_angle=float(rng.uniform(0, 0.5)))
        cid = int(rng.integers(0, 2 ** 33))
        if common["rng"] == 0:
            cid %= (1 << 22) - 200
        cases.append((kw, cid))
    nsteps = int(rng.choice([400, 900, 1500]))
    os.environ["PSTAT_SEGMENTS"] = str(int(rng.choice([1, 2, 3])))
    # where the f64 state lives: the sweep's cells fit LDS up to n = 40; the clustering main has both homes at any n here
    os.environ["PSTAT_F64_STATE"] = str(rng.choice(["lds", "global"])) if (cluster or n <= 40) else "global"
    params = []
    for kw, cid in cases:
        _, pp = both(nsteps, num_chains=nch, precision=ps.F64, chain_id0=cid, **common, **kw)
        if cluster:
            pp.move_set = ps.MOVES_CLUSTER
        params.append(pp)
    try:
        with ps.Ensemble(params) as e:
            info = e.launch_info()
            npacked += info.packed_cases
            assert info.packed_cases == 1, "not packed"
            half = nsteps // 3
            e.advance(half); e.advance(nsteps - half)
            e.sync()
            for _ in range(3):
                i = int(rng.integers(0, ncases)); k = int(rng.integers(0, nch))
                kw, cid = cases[i]
                op, _ = both(nsteps, **common, **kw)
                o = ob.run(op, chain_id=cid + k, mode="cluster" if cluster else "fast", trace=True)
                g = e.chain_state(i * nch + k)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), "angles"
                assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total, "stream / count"
                assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step, "steps"
                # (--umbrella-sampling on a cold chain: the reference's weights exp(w - log_gauge) under- or overflow where the
                # device's, whose gauge rises with the chain, stay representable -- DESIGN.md 3.5.  Trial 701 of seed 1:
                # oracle normaliser 1.5e-306 and 0.0, device 2.4e7 and 2.4e3.  Nothing to compare then.)
                if np.all(np.isfinite(o.avg)) and 1e-290 < abs(o.norm) < 1e290:
                    np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-7, atol=1e-7)
    except AssertionError as ex:
        nfail += 1
        print("FAIL trial", trial, "cluster" if cluster else "sweep", "et", et, "n", n, "cases", ncases, "chains", nch, "steps", nsteps,
              common, str(ex)[:160], flush=True)
    if trial % 50 == 49:
        print(f"# {trial + 1} trials, {nfail} failures", flush=True)
print(f"fuzz_packed: {trials} trials ({npacked} packed), {nfail} failures")
sys.exit(1 if nfail else 0)
