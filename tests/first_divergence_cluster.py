#!/usr/bin/env python3
"""Debugging aid (TEST TOOLING: it runs the oracle): the first step at which a kernel of the f64 clustering main leaves the
oracle's trajectory, on five configurations of growing reach (no flips; flips; bending; n = 100; n = 200 Ising).
    python tests/first_divergence_cluster.py [wave|global|lds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polymer_stats_amd as ps
from helpers import both
from oracle import binding as ob

os.environ["PSTAT_F64_STATE"] = sys.argv[1] if len(sys.argv) > 1 else "wave"
CASES = [
    ("no flips n=20", dict(n=20, E0=1.2, K1=1.0, K2=0.2, Fz=0.7, seed=21, cluster_prob=1.0, adj_scale=1.0)),
    ("flips n=20", dict(n=20, E0=1.2, K1=1.0, K2=0.2, Fz=0.7, seed=21, cluster_prob=0.5, adj_scale=1.0)),
    ("flips bend n=20", dict(n=20, E0=1.2, K1=1.0, K2=0.2, Fz=0.7, seed=21, cluster_prob=0.5, bend_mod=0.5, bend_angle=0.3, adj_scale=1.0)),
    ("flips n=100", dict(n=100, E0=1.2, K1=1.0, K2=0.2, Fz=0.7, seed=21, cluster_prob=0.3, adj_scale=1.0)),
    ("flips n=200 ising", dict(n=200, E0=1.2, K1=0.3, K2=0.02, Fz=0.7, seed=21, cluster_prob=0.3, adj_scale=1.0, energy_type=2)),
]
for name, kw in CASES:
    first = None
    for k in [1, 2, 3, 4, 5, 6, 8, 10, 15, 20, 30, 50, 100, 200, 400]:
        op, pp = both(k, num_chains=2, precision=ps.F64, **kw)
        pp.move_set = ps.MOVES_CLUSTER
        with ps.Ensemble(pp) as e:
            kern = e.launch_info().kernel.decode()
            e.advance(k); e.sync()
            g = e.chain_state(1)
        o = ob.run(op, chain_id=1, mode="cluster", trace=True)
        same_ang = np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi)
        same_rng = np.array_equal(g["rng"], o.rng)
        if not (same_ang and same_rng and g["nacc_total"] == o.nacc_total):
            first = k
            bad = np.nonzero((g["theta"] != o.final_theta) | (g["phi"] != o.final_phi))[0]
            print(f"{name}: [{kern}] differs after {k} steps: angles {same_ang} rng {same_rng} nacc {g['nacc_total']} vs {o.nacc_total}; monomers {bad[:12]}",
                  "acc", o.accepted[:k].tolist() if k <= 30 else "", flush=True)
            if len(bad):
                i = bad[0]
                print("   dev theta/phi", g["theta"][i], g["phi"][i], " oracle", o.final_theta[i], o.final_phi[i], "rng", g["rng"], o.rng)
            break
    if first is None:
        print(f"{name}: [{kern}] equal through 400 steps", flush=True)
