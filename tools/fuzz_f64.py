#!/usr/bin/env python3
"""Trajectory fuzz of the f64 kernels against the CPU oracle (TEST TOOLING: it runs the oracle, like tests/ do): random
physics, options, chain lengths, generators and global chain ids; every chain's final angles, generator state,
acceptance count and step sizes must equal the oracle's bit for bit, the running averages to 1e-8.  Beyond what the
test-suite's 80 cases cover, it forces the sweep's cells into memory for SHORT chains (same-monomer hits in consecutive
steps every few steps: the forwarding paths) and varies the number of rows kept in LDS.

    python tools/fuzz_f64.py [trials=400] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polymer_stats_amd as ps
from helpers import both
from oracle import binding as ob

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nfail = 0
for trial in range(trials):
    et = int(rng.choice([0, 0, 0, 2, 2, 1]))
    where = str(rng.choice(["lds", "global", "global", "auto"]))
    n = int(rng.integers(1, 12)) if rng.random() < 0.4 else int(rng.integers(12, 140))
    if et == 1:
        n = int(rng.integers(2, 70))
    kw = dict(n=n, E0=float(rng.uniform(0, 2)), K1=float(rng.uniform(0, 1.2)), K2=float(rng.uniform(0, 0.5)),
              mu=float(rng.uniform(0.01, 0.6)), kT=float(10 ** rng.uniform(-0.5, 0.7)), Fz=float(rng.uniform(-1, 2)),
              Fx=float(rng.choice([0.0, rng.uniform(-1, 1)])), b=float(rng.uniform(0.5, 2.0)),
              chain_type=int(rng.integers(0, 2)), energy_type=et, do_flips=int(rng.integers(0, 2)),
              umbrella=int(rng.integers(0, 2)), steps_per_adjust=int(rng.choice([50, 137, 400, 2500])),
              adj_scale=float(rng.choice([1.0, 1.1, 1.3])), rng=int(rng.integers(0, 2)), seed=int(rng.integers(0, 2 ** 40)))
    if et == 2:      # keep the Ising coupling weak: collapsed chains amplify rounding into decisions
        kw.update(K1=kw["K1"] * 0.3, K2=kw["K2"] * 0.2, mu=kw["mu"] * 0.3)
    nsteps = 300 if et == 1 else int(rng.choice([700, 1500, 3001]))
    inits = int(rng.choice([1, 1, 2]))
    force = int(rng.integers(0, 2))
    cid = int(rng.integers(0, 2 ** 33))
    if kw["rng"] == 0:
        cid %= (1 << 22) - 70
    nch = int(rng.choice([3, 65, 130]))
    if where == "auto":
        os.environ.pop("PSTAT_F64_STATE", None)
    else:
        os.environ["PSTAT_F64_STATE"] = where
    rows = int(rng.choice([0, 1, 5, 39]))
    os.environ["PSTAT_F64_LDS_ROWS"] = str(rows)
    op, pp = both(nsteps, num_chains=nch, precision=ps.F64, num_inits=inits, force_init=force, chain_id0=cid, **kw)
    try:
        with ps.Ensemble(pp) as e:
            kern = e.launch_info().kernel.decode()
            for k in range(inits):
                half = nsteps // 3
                e.advance(half); e.advance(nsteps - half)          # a launch split in every run
                if k + 1 < inits:
                    e.reinit(bool(force))
            e.sync()
            for c in sorted(set([0, nch - 1, nch // 2])):
                o = ob.run(op, chain_id=pp.chain_id0 + c, mode="fast", trace=True)
                g = e.chain_state(c)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), "angles"
                assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total, "rng/nacc"
                assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step, "steps"
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-7, atol=1e-7)
    except AssertionError as ex:
        nfail += 1
        print("FAIL trial", trial, kern, where, rows, nch, nsteps, inits, force, kw, str(ex)[:200], flush=True)
    if trial % 50 == 49:
        print(f"# {trial + 1} trials, {nfail} failures", flush=True)
print(f"fuzz_f64: {trials} trials, {nfail} failures")
sys.exit(1 if nfail else 0)
