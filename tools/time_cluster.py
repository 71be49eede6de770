"""Wall-clock rate of the clustering main's chain-per-lane kernel on the configuration of tools/profile_cluster.py
(kernel experiments; the driver-facing numbers come from tools/measure_configs.py and bench.py):
    python tools/time_cluster.py [ni|ising] [f64|f32] [steps=5000] [n=100] [chains=65536]      (env: CPROB, E0, K1, K2, KT)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

et = ps.ISING if len(sys.argv) > 1 and sys.argv[1] == "ising" else ps.NONINTERACTING
prec = {"f32": ps.F32, "f64": ps.F64}[sys.argv[2] if len(sys.argv) > 2 else "f64"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
n = int(sys.argv[4]) if len(sys.argv) > 4 else 100
chains = int(sys.argv[5]) if len(sys.argv) > 5 else 65536
p = ps.default_params(n=n, E0=float(os.environ.get("E0", "1.0")), K1=float(os.environ.get("K1", "0.0")), K2=float(os.environ.get("K2", "1.0")), kT=float(os.environ.get("KT", "1.0")), energy_type=et, num_chains=chains, precision=prec, seed=6,
                      move_set=ps.MOVES_CLUSTER, cluster_prob=float(os.environ.get("CPROB", "0.5")), adj_ub=0.40)
with ps.Ensemble(p) as e:
    e.advance(steps); e.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        e.advance(steps); e.sync()
        best = min(best, time.perf_counter() - t0)
    info = e.launch_info()
    s = e.summary()
    print("%s %s n=%d chains=%d: %.3e proposals/s  (%.2f ms per %d steps; %s, %d lanes x %d per CU; AR %.4f p3 %.6f lib %s)"
          % (sys.argv[1] if len(sys.argv) > 1 else "ni", "f64" if prec == ps.F64 else "f32", n, chains, chains * steps / best,
             best * 1e3, steps, info.kernel.decode(), info.lanes_per_block, info.blocks_per_cu, s.acceptance_ratio, s.avg[9],
             os.environ.get("PSTAT_LIB", "default")), flush=True)
