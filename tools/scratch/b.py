import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import polymer_stats_amd._lib as L
if sys.argv[1] != "std": L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"libpstat_{sys.argv[1]}.so")
import polymer_stats_amd as ps
p = ps.default_params(n=100, E0=1.0, K1=1.0, K2=0.0, Fz=0.45, kT=1.0, num_chains=65536, precision=ps.F32, seed=4)
with ps.Ensemble(p) as e:
    e.advance(20000); e.sync()
    best = 1e9
    for _ in range(4):
        t = time.perf_counter(); e.advance(100000); e.sync(); best = min(best, time.perf_counter() - t)
    s = e.summary()
    print(sys.argv[1], f"{65536 * 100000 / best:.4g} upd/s  {best*1e3:.2f} ms", f"r3={s.avg[2]:.4f} U={s.avg[14]:.4f} AR={s.acceptance_ratio:.4f}", flush=True)
