#!/usr/bin/env python3
"""Trajectory fuzz of the chain-per-wavefront kernel of the clustering main (csrc/pstat_cluster_cw.hip) against the CPU oracle
(TEST TOOLING: it runs the oracle, like tests/ do).  Random configurations over every chain length the kernel takes (n = 2 ...
256: one, two and four monomers per lane, and the lengths either side of each boundary), both chain types, non-interacting and
Ising energies, bending, umbrella weights, either eps contract, cluster_prob from 0 to 1, hot to cold (aligned starts and low
temperatures: clusters that run to the chain ends and leave one end drawing alone for tens of rounds), adaptation, a launch
split and an annealing rung in every run; three chains per configuration must equal the oracle's final angles, generator
state, acceptance count and step sizes bit for bit, the averages to 1e-8.

    python tests/fuzz_cluster_wave.py [trials=300] [seed=1] [kernel=wave|global|lds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polymer_stats_amd as ps
from helpers import both
from oracle import binding as ob


def run(trials, seed, log=print, home="wave"):
    rng = np.random.default_rng(seed)
    os.environ["PSTAT_F64_STATE"] = home
    nfail = 0
    for trial in range(trials):
        et = int(rng.choice([0, 2]))
        n = int(rng.choice([2, 3, 63, 64, 65, 127, 128, 129, 255, 256])) if rng.random() < 0.35 else int(rng.integers(2, 257))
        cold = rng.random() < 0.4
        kw = dict(n=n, E0=float(rng.uniform(0, 3.0 if cold else 1.5)), K1=float(rng.uniform(0, 1.2)), K2=float(rng.uniform(0, 0.5)),
                  mu=float(rng.uniform(0.01, 0.6)), kT=float(10 ** (rng.uniform(-1.5, -0.5) if cold else rng.uniform(-0.3, 0.7))),
                  Fz=float(rng.uniform(-1, 2)), Fx=float(rng.choice([0.0, rng.uniform(-1, 1)])), b=float(rng.uniform(0.5, 2.0)),
                  chain_type=int(rng.integers(0, 2)), energy_type=et, umbrella=int(rng.integers(0, 2)),
                  bend_mod=float(rng.choice([0.0, 0.0, rng.uniform(0, 1.5)])), bend_angle=float(rng.uniform(0, 1.0)),
                  cluster_prob=float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.8, 1.0])), steps_per_adjust=int(rng.choice([50, 137, 400, 2500])),
                  adj_scale=float(rng.choice([1.0, 1.1, 1.3])), uniform_bits=int(rng.choice([0, 23])), seed=int(rng.integers(0, 2 ** 40)))
        if et == 2:      # keep the Ising coupling weak: collapsed chains amplify rounding into decisions (tests/test_gpu_cluster.py)
            kw.update(K1=kw["K1"] * 0.25, K2=kw["K2"] * 0.2, mu=kw["mu"] * 0.25)
        if cold or rng.random() < 0.3:      # aligned start: every link joins, clusters span the chain
            kw.update(use_x0=1, x0_phi=float(rng.uniform(0, 6)), x0_theta=float(rng.uniform(0.2, 1.2)), dx0_phi=0.05, dx0_theta=0.05)
        nsteps = int(rng.choice([300, 700, 1200])) if n > 128 else int(rng.choice([700, 1500, 2500]))
        burn = int(rng.integers(50, 300))
        sched = (float(rng.choice([10.0, 2.0])),) if rng.random() < 0.5 else ()
        cid = int(rng.integers(0, (1 << 22) - 8))
        nch = int(rng.choice([1, 3, 5]))
        op, pp = both(nsteps, num_chains=nch, precision=ps.F64, chain_id0=cid, **kw)
        op.burn_nsched = len(sched); op.burn_in = burn
        for i, v in enumerate(sched):
            op.burn_sched[i] = v
        pp.move_set = ps.MOVES_CLUSTER
        try:
            with ps.Ensemble(pp) as e:
                assert ("cluster_chain_wave_kernel" in e.launch_info().kernel.decode()) == (home == "wave"), "kernel"
                for f in sched:
                    e.set_kT(pp.kT * f); e.reset_sampler(); e.reset_averages(); e.advance(burn)
                e.set_kT(pp.kT); e.reset_sampler(); e.reset_averages()
                half = nsteps // 3
                e.advance(half); e.advance(nsteps - half)
                e.sync()
                for c in sorted({0, nch - 1, nch // 2}):
                    o = ob.run(op, chain_id=cid + c, mode="cluster", trace=True)
                    g = e.chain_state(c)
                    assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), "angles"
                    assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total, "stream / count"
                    assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step, "steps"
                    # (cold chains under --umbrella-sampling: the reference's weights exp(w - log_gauge) leave the range of a double
                    # where the device's rising gauge keeps them inside -- DESIGN.md 3.5; nothing to compare then)
                    if np.all(np.isfinite(o.avg)) and 1e-290 < abs(o.norm) < 1e290:
                        np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-8, atol=1e-8)
                        x = e.chain_extras(c)
                        np.testing.assert_allclose(x["sums"] / g["normalizer"], o.extra_sums / o.norm, rtol=1e-8, atol=1e-8)
        except AssertionError as ex:
            nfail += 1
            log("FAIL trial", trial, "n", n, "steps", nsteps, "sched", sched, kw, str(ex)[:200])
        if trial % 50 == 49:
            log(f"# {trial + 1} trials, {nfail} failures")
    return nfail


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = run(trials, seed, log=lambda *a: print(*a, flush=True), home=sys.argv[3] if len(sys.argv) > 3 else "wave")
    print(f"fuzz_cluster_wave: {trials} trials, {bad} failures")
    sys.exit(1 if bad else 0)
