"""One chain per wavefront against one chain per lane on the clustering main's small ensembles (kernel experiments):
    python tools/time_cluster_cw.py [steps=20000] [n=100] [chains per case=1,2,4,8,16] [homes=global,wave]
times (E0, kT) grids of the reference's phase scan (run/K1_E0-kT-phase.jl: Ising, K1 = 1, cluster_prob 0.5) -- hot points only, the
whole grid, cold points only -- at 1, 2, 4, 8 and 16 chains per case, with PSTAT_F64_STATE=global (pstat_cluster_gm.hip, packed or
not as pstat_create decides) and =wave (pstat_cluster_cw.hip)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pers = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16]
homes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["global", "wave"]
all_kT = [10 ** (-2 + 0.2 * j) for j in range(21)]


def grid(kTs, per, reps=5):
    out = []
    for rep in range(reps):
        for i in range(26):
            for kT in kTs:
                out.append(ps.default_params(n=n, E0=0.2 * i, K1=1.0, K2=0.0, kT=kT, num_chains=per, precision=ps.F64, seed=1000 + len(out),
                                             move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, energy_type=ps.ISING))
    return out


def run(name, cases, per):
    row = []
    for home in homes:
        os.environ["PSTAT_F64_STATE"] = home
        with ps.Ensemble(cases) as e:
            e.advance(max(500, steps // 5)); e.sync()
            t0 = time.perf_counter()
            e.advance(steps); e.sync()
            dt = time.perf_counter() - t0
            info = e.launch_info()
            row.append("%s %8.1f ms = %6.2f us/step (%s%s, %d blocks)" % (home, dt * 1e3, dt / steps * 1e6, info.kernel.decode().split("<")[0],
                                                                        " packed" if info.packed_cases else "", info.blocks))
    print("%-34s %5d chains: %s" % (name, len(cases) * per, " | ".join(row)), flush=True)


for per in pers:
    run("hot (kT >= 1), %d per case" % per, grid([k for k in all_kT if k >= 0.99], per), per)
    run("whole grid, %d per case" % per, grid(all_kT, per), per)
    run("cold (kT <= 0.1), %d per case" % per, grid([k for k in all_kT if k <= 0.11], per), per)
