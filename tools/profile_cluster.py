"""One warm-up and `reps` launches of a clustering-main kernel at n = 100 for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_cluster.py [ising|ni|interacting|cutoff] [f64|f32] [steps] [reps=2] [n=100]
ni / ising: the chain-per-lane kernel, 65 536 chains, K2-only dielectric as run/Ising_2025-12-18.jl / run/K1_E0-kT-phase.jl
launch it; interacting / cutoff: the chain-per-wavefront kernel (cluster_wave_kernel), 16 384 chains, the configuration
of tools/measure_configs.py C7.  tools/summarize_pmc.py drops the first (warm-up) dispatch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

what = sys.argv[1] if len(sys.argv) > 1 else "ising"
prec = {"f32": ps.F32, "f64": ps.F64}[sys.argv[2] if len(sys.argv) > 2 else "f64"]
wave = what in ("interacting", "cutoff")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else (1000 if wave else 5000)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n = int(sys.argv[5]) if len(sys.argv) > 5 else 100
if wave:
    chains = 16384
    p = ps.default_params(n=n, E0=1.0, K1=1.0, Fz=0.5, kT=1.0, cutoff_radius=7.5, num_chains=chains, precision=prec, seed=7,
                          energy_type=ps.INTERACTING if what == "interacting" else ps.CUTOFF,
                          move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, adj_ub=0.40)
else:
    chains = 65536
    p = ps.default_params(n=n, E0=1.0, K1=0.0, K2=1.0, kT=1.0, energy_type=ps.ISING if what == "ising" else ps.NONINTERACTING,
                          num_chains=chains, precision=prec, seed=6, move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, adj_ub=0.40)
with ps.Ensemble(p) as e:
    for _ in range(1 + reps):
        e.advance(steps)
        e.sync()
    print(e.summary().acceptance_ratio, "proposals per launch", chains * steps, e.launch_info().kernel.decode())
