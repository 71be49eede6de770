#!/usr/bin/env python3
"""Trajectory fuzz of the f64 kernels against the CPU oracle (TEST TOOLING: it runs the oracle, like tests/ do): random
physics, options, chain lengths, generators and global chain ids; every chain's final angles, generator state,
acceptance count and step sizes must equal the oracle's bit for bit, the running averages to 1e-8.  Beyond what the
test-suite's 80 cases cover, it forces the sweep's cells into memory for SHORT chains (same-monomer hits in consecutive
steps every few steps: the forwarding paths) and varies the number of rows kept in LDS.

One class of mismatch is expected and counted separately: a chain of the all-pairs energy that has COLLAPSED (a 1/r^3
contact, |U| > 1e3 n kT).  There a position rounding of 1e-15 b -- the kernel sums positions by a wave prefix scan, the
oracle sequentially -- moves the contact's term by ~1e-5 kT, enough to flip a decision every ~1e5 steps; the reference
itself would not reproduce such a chain across machines.  (Measured: 3 of ~400 interacting trials.)

    python tests/fuzz_f64.py [trials=400] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (this file lives in tests/: it runs the oracle, which only test code may)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polymer_stats_amd as ps
from helpers import both
from oracle import binding as ob

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nfail = 0
ncollapsed = 0
only = os.environ.get("FUZZ_ONLY")
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
    if et == 1:      # likewise for the all-pairs energy (and no short bonds): see the note on collapsed chains above
        kw.update(K1=kw["K1"] * 0.3, K2=kw["K2"] * 0.2, mu=kw["mu"] * 0.3, b=max(1.0, kw["b"]))
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
    if only is not None and trial != int(only):
        continue
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
                same = (np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi) and
                        np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total)
                if not same and et == 1 and abs(o.U) > 1e3 * n * kw["kT"]:
                    ncollapsed += 1
                    print("diverged after collapse: trial", trial, "chain", c, "U = %.3g" % o.U, flush=True)
                    continue
                assert same, "trajectory"
                assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step, "steps"
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-7, atol=1e-7)
    except AssertionError as ex:
        nfail += 1
        print("FAIL trial", trial, kern, where, rows, "chains", nch, "id0", cid, "steps", nsteps, "inits", inits, "force", force,
              kw, str(ex)[:200], flush=True)
        if os.environ.get("FUZZ_DIAGNOSE"):      # first step at which a chain leaves the oracle's trajectory
            op1, pp1 = both(nsteps, num_chains=nch, precision=ps.F64, num_inits=1, force_init=force, chain_id0=cid, **kw)
            with ps.Ensemble(pp1) as e:
                acc = {c: ob.run(op1, chain_id=cid + c, mode="fast", trace=True).accepted for c in (0, nch - 1, nch // 2)}
                prev = {c: 0 for c in acc}
                for step in range(nsteps):
                    e.advance(1)
                    for c in acc:
                        now = e.chain_state(c)["nacc_total"]
                        if now - prev[c] != int(acc[c][step]):
                            print("  chain", c, "step", step, "kernel accepted", now - prev[c], "oracle", int(acc[c][step]),
                                  "U", e.microstate(c)[6], flush=True)
                            acc[c] = None
                        prev[c] = now
                    acc = {c: a for c, a in acc.items() if a is not None}
                    if not acc:
                        break
    if trial % 50 == 49:
        print(f"# {trial + 1} trials, {nfail} failures", flush=True)
print(f"fuzz_f64: {trials} trials, {nfail} failures, {ncollapsed} chains diverged after collapsing (all-pairs energy, expected)")
sys.exit(1 if nfail else 0)
