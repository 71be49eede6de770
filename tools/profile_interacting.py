"""One warm-up and `reps` launches of the interacting kernel (BASELINE configs[3]: n = 64, 16 384 chains) for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_interacting.py [n=64] [f64|f32] [steps=4000] [reps=2]
tools/summarize_pmc.py drops the first (warm-up) dispatch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = {"f32": ps.F32, "f64": ps.F64}[sys.argv[2] if len(sys.argv) > 2 else "f64"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
p = ps.default_params(n=n, E0=1.0, K1=1.0, K2=0.0, Fz=0.5, kT=1.0, energy_type=ps.INTERACTING, num_chains=16384,
                      precision=prec, seed=4)
import time
with ps.Ensemble(p) as e:
    best = 1e9
    for _ in range(1 + reps):
        t0 = time.perf_counter()
        e.advance(steps)
        e.sync()
        best = min(best, time.perf_counter() - t0)
    print("AR %.4f; %d updates per launch, best launch %.1f ms = %.3e updates/s (%s)"
          % (e.summary().acceptance_ratio, 16384 * steps, best * 1e3, 16384 * steps / best, e.launch_info().kernel.decode()))
