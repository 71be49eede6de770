"""One warm-up and one 4000-step launch of the interacting kernel (BASELINE configs[3]: n = 64, 16 384 chains)
for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_interacting.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = ps.default_params(n=n, E0=1.0, K1=1.0, K2=0.0, Fz=0.5, kT=1.0, energy_type=ps.INTERACTING, num_chains=16384,
                      precision=ps.F32, seed=4)
with ps.Ensemble(p) as e:
    e.advance(500); e.sync()
    e.advance(4000); e.sync()
    print(e.summary().acceptance_ratio)
