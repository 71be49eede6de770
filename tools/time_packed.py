"""What packing several few-chain cases into one wave buys (kernel experiments):
    python tools/time_packed.py [chains_per_case=16] [ncases=2730] [steps=20000]
times a fixed-force f64 sweep and three clustering-main grids (hot, the reference's whole (E0, kT) grid, cold) with
PSTAT_PACK=0 (every workgroup inside one case) and PSTAT_PACK=1."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

per = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 2730
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20000


def grid(kTs, **kw):
    out = []
    i = 0
    while len(out) < ncases:
        E0 = 0.2 * (i // len(kTs) % 26)
        out.append(ps.default_params(n=100, E0=E0, K1=1.0, K2=0.0, kT=kTs[i % len(kTs)], num_chains=per, precision=ps.F64, seed=1000 + i, **kw))
        i += 1
    return out


def run(name, cases, nsteps):
    for pack in ("0", "1"):
        os.environ["PSTAT_PACK"] = pack
        with ps.Ensemble(cases) as e:
            e.advance(max(200, nsteps // 10)); e.sync()
            best = 1e30
            for _ in range(2):
                t0 = time.perf_counter()
                e.advance(nsteps); e.sync()
                best = min(best, time.perf_counter() - t0)
            info = e.launch_info()
            print("%-44s pack=%s: %8.1f ms per %d steps = %.3e updates/s  (%s, %d lanes, %d blocks)" %
                  (name, pack, best * 1e3, nsteps, len(cases) * per * nsteps / best, info.kernel.decode(), info.lanes_per_block, info.blocks), flush=True)


cl = dict(move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, energy_type=ps.ISING)
all_kT = [10 ** (-2 + 0.2 * j) for j in range(21)]
run("fixed-force main, non-interacting", grid(all_kT), steps * 4)
run("fixed-force main, Ising", grid(all_kT, energy_type=ps.ISING), steps * 2)
run("clustering main, Ising, kT >= 1 (hot)", grid([k for k in all_kT if k >= 0.99], **cl), steps)
run("clustering main, Ising, whole grid", grid(all_kT, **cl), steps)
run("clustering main, Ising, kT <= 0.1 (cold)", grid([k for k in all_kT if k <= 0.11], **cl), steps // 2)
