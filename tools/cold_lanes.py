"""`ncases` cold (aligned) cases of `k` chains each, one wave per case (PSTAT_PACK=0), as bare launches for rocprofv3 / timing:
what does a wave's step cost as a function of its active lanes?   python tools/cold_lanes.py [k=16] [ncases=64] [steps=4000] [n=100]"""
import os
import sys
import time

os.environ["PSTAT_PACK"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
n = int(sys.argv[4]) if len(sys.argv) > 4 else 100
cases = [ps.default_params(n=n, E0=3.0, K1=1.0, K2=0.0, kT=0.1, energy_type=ps.ISING, num_chains=k, precision=ps.F64, seed=100 + i,
                           move_set=ps.MOVES_CLUSTER, cluster_prob=0.5) for i in range(ncases)]
with ps.Ensemble(cases) as e:
    e.scale_kT(10.0); e.advance(steps); e.reset_sampler(); e.scale_kT(1.0); e.advance(steps); e.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); e.advance(steps); e.sync(); ts.append(time.perf_counter() - t0)
    info = e.launch_info()
    print("k=%d ncases=%d: %.2f us per step (%s, %d lanes, %d blocks) AR %.3f" % (k, ncases, min(ts) / steps * 1e6, info.kernel.decode(),
          info.lanes_per_block, info.blocks, e.summary().acceptance_ratio), flush=True)
