"""Wall-clock rate of the sweep kernel on the bench workload (kernel experiments; the driver-facing numbers come from bench.py):
    python tools/time_sweep.py [f64|f32|q16] [n=100] [chains=65536] [mc_steps=100000] [reps=5] [energy=0]   (env: UNIFORM_BITS, PSTAT_LIB)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

prec = {"f32": ps.F32, "f64": ps.F64, "q16": ps.Q16}[sys.argv[1] if len(sys.argv) > 1 else "f64"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
chains = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
mc = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
energy = int(sys.argv[6]) if len(sys.argv) > 6 else ps.NONINTERACTING
p = ps.default_params(n=n, E0=1.0, K1=1.0, K2=0.0, kT=1.0, b=1.0, Fz=1.0, num_chains=chains, precision=prec, seed=20260501,
                      energy_type=energy, uniform_bits=int(os.environ.get("UNIFORM_BITS", "0")))
with ps.Ensemble(p) as e:
    e.advance(mc); e.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        e.advance(mc); e.sync()
        ts.append(time.perf_counter() - t0)
    info = e.launch_info()
    s = e.summary()
    print("%s n=%d chains=%d: best %.3e median %.3e updates/s (%.2f ms best of %d; %s; r3 %.5f AR %.4f; bits %s lib %s)"
          % (sys.argv[1] if len(sys.argv) > 1 else "f64", n, chains, chains * mc / min(ts), chains * mc / sorted(ts)[len(ts) // 2],
             min(ts) * 1e3, reps, info.kernel.decode(), s.avg[2], s.acceptance_ratio, os.environ.get("UNIFORM_BITS", "0"),
             os.path.basename(os.path.dirname(os.environ.get("PSTAT_LIB", "default/x")))), flush=True)
