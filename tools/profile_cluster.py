"""One warm-up and `reps` launches of the chain-per-lane cluster kernel (n = 100, 65 536 chains, the configuration of
run/Ising_2025-12-18.jl / run/K1_E0-kT-phase.jl) for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_cluster.py [ising|ni] [f64|f32] [steps=5000] [reps=2]
tools/summarize_pmc.py drops the first (warm-up) dispatch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

et = ps.ISING if len(sys.argv) > 1 and sys.argv[1] == "ising" else ps.NONINTERACTING
prec = {"f32": ps.F32, "f64": ps.F64}[sys.argv[2] if len(sys.argv) > 2 else "f64"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
p = ps.default_params(n=100, E0=1.0, K1=0.0, K2=1.0, kT=1.0, energy_type=et, num_chains=65536, precision=prec, seed=6,
                      move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, adj_ub=0.40)
with ps.Ensemble(p) as e:
    for _ in range(1 + reps):
        e.advance(steps)
        e.sync()
    print(e.summary().acceptance_ratio, "proposals per launch", 65536 * steps)
