"""One warm-up and `reps` launches of the clustering main on a phase-scan ensemble for rocprofv3 (the counters behind DESIGN.md 3.7.3):
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_cluster_cw.py [whole|hot|cold] [per_case=1] [steps=20000] [reps=2] [n=100]
The (E0, kT) grid of run/K1_E0-kT-phase.jl (26 x 21 points x 5 runs, Ising, K1 = 1, cluster_prob 0.5), `per_case` chains per case;
PSTAT_F64_STATE=wave|global picks the kernel (default: pstat_create's rule)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

which = sys.argv[1] if len(sys.argv) > 1 else "whole"
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n = int(sys.argv[5]) if len(sys.argv) > 5 else 100
kTs = [10 ** (-2 + 0.2 * j) for j in range(21)]
kTs = {"whole": kTs, "hot": [k for k in kTs if k >= 0.99], "cold": [k for k in kTs if k <= 0.11]}[which]
cases = []
for rep in range(5):
    for i in range(26):
        for kT in kTs:
            cases.append(ps.default_params(n=n, E0=0.2 * i, K1=1.0, K2=0.0, kT=kT, num_chains=per, precision=ps.F64, seed=1000 + len(cases),
                                           move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, energy_type=ps.ISING))
with ps.Ensemble(cases) as e:
    for _ in range(1 + reps):
        e.advance(steps)
        e.sync()
    print(e.summary(0).acceptance_ratio, "proposals per launch", len(cases) * per * steps, e.launch_info().kernel.decode())
