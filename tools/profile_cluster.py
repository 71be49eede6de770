"""One warm-up and one 5000-step launch of the cluster kernel (n = 100, 65 536 chains) for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_cluster.py [ising]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps
et = ps.ISING if len(sys.argv) > 1 and sys.argv[1] == "ising" else ps.NONINTERACTING
p = ps.default_params(n=100, E0=1.0, K1=0.0, K2=1.0, kT=1.0, energy_type=et, num_chains=65536, precision=ps.F32, seed=6,
                      move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, adj_ub=0.40)
with ps.Ensemble(p) as e:
    e.advance(2000); e.sync()
    e.advance(5000); e.sync()
    print(e.summary().acceptance_ratio)
