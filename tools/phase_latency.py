"""Per-step latency of single grid points of run/K1_E0-kT-phase.jl's (E0, kT) grid in the clustering main (kernel experiments):
one case x 16 chains is one wave, so wall time / steps = the wave's step time at that temperature.
    python tools/phase_latency.py [steps=20000] [chains=16] [n=100]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for E0 in (0.0, 1.0, 3.0, 5.0):
    for kT in (0.01, 0.1, 1.0, 10.0, 100.0):
        p = ps.default_params(n=n, E0=E0, K1=1.0, K2=0.0, kT=kT, energy_type=ps.ISING, num_chains=chains, precision=ps.F64, seed=6,
                              move_set=ps.MOVES_CLUSTER, cluster_prob=0.5)
        with ps.Ensemble(p) as e:
            for mult in (10.0, 1.0):               # a short annealed start like the ladder's
                e.scale_kT(mult); e.advance(steps // 4); e.reset_sampler(); e.reset_averages()
            e.sync()
            t0 = time.perf_counter()
            e.advance(steps); e.sync()
            dt = time.perf_counter() - t0
            s = e.summary()
            print("E0 %.1f kT %6.2f: %7.2f us per step  AR %.4f  <cos2> %.3f  U %.3e collapsed %d" %
                  (E0, kT, dt / steps * 1e6, s.acceptance_ratio, s.extra_avg[0] / n, s.avg[14], s.chains_collapsed), flush=True)
