"""Ad-hoc soak of the f64 clustering-main kernel against the oracle (TEST TOOLING, not collected by pytest: it runs the oracle, which only code under tests/ may): long runs of a few
chains, default home, bit-exact trajectories expected.  python tests/soak_cluster.py [steps=40000] [chains=12]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polymer_stats_amd as ps
from helpers import both
from oracle import binding as ob

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for name, kw in (("n100 K2-only", dict(n=100, E0=1.0, K1=0.0, K2=1.0, adj_ub=0.40)),
                 ("n100 stiff cold", dict(n=100, E0=0.5, K1=0.5, kT=0.2, bend_mod=1.0, Fz=0.3)),
                 ("n200 Ising weak", dict(n=200, E0=0.6, K1=0.25, kT=2.5, energy_type=2, adj_ub=0.40)),
                 ("n7 polar Fx", dict(n=7, E0=1.0, mu=0.5, chain_type=1, Fz=0.4, Fx=0.3, cluster_prob=0.2))):
    op, pp = both(steps, num_chains=nch, precision=ps.F64, seed=4242, cluster_prob=kw.pop("cluster_prob", 0.5), **kw)
    pp.move_set = ps.MOVES_CLUSTER
    with ps.Ensemble(pp) as e:
        e.advance(steps); e.sync()
        for c in range(nch):
            o = ob.run(op, chain_id=c, mode="cluster", trace=True)
            g = e.chain_state(c)
            same = np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi) and np.array_equal(g["rng"], o.rng) \
                and g["nacc_total"] == o.nacc_total
            bad += 0 if same else 1
            if not same:
                print("MISMATCH", name, "chain", c, g["nacc_total"], o.nacc_total)
        print(name, e.launch_info().kernel.decode(), "AR", e.summary().acceptance_ratio, "ok" if bad == 0 else "FAILED", flush=True)
sys.exit(1 if bad else 0)
