"""Rate of the all-pairs (interacting) fixed-force kernel (kernel experiments):
    python tools/time_interacting.py [n=64,100,200] [chains=16384] [steps=4000]
BASELINE configs[3]'s physics (E0 = 1, K1 = 1, Fz = 0.5), f64; best of three launches after a warm-up."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 100, 200]
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
for n in ns:
    p = ps.default_params(n=n, E0=1.0, K1=1.0, K2=0.0, Fz=0.5, energy_type=ps.INTERACTING, num_chains=chains, precision=ps.F64, seed=4)
    st = max(200, steps * 64 * 64 // (n * n))
    with ps.Ensemble(p) as e:
        e.advance(st // 4); e.sync()
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter(); e.advance(st); e.sync()
            best = min(best, time.perf_counter() - t0)
        info = e.launch_info()
        print("n = %3d, %d chains x %d steps: %8.1f ms = %.3e updates/s (%s, %d workgroups per CU)" %
              (n, chains, st, best * 1e3, chains * st / best, info.kernel.decode(), info.blocks_per_cu), flush=True)
