"""The bench workload (BASELINE configs[1]; or configs[4]'s n = 200) as bare launches for rocprofv3: one warm-up
launch and `reps` full launches of the sweep kernel, nothing else on the stream.

    rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_sweep.py f64 [n=100] [chains=65536] [mc_steps=100000] [reps=3] [energy=0]

tools/summarize_pmc.py drops the first (warm-up) dispatch of the kernel from every figure."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

prec = {"f32": ps.F32, "f64": ps.F64, "q16": ps.Q16}[sys.argv[1] if len(sys.argv) > 1 else "f64"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
chains = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
mc = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
energy = int(sys.argv[6]) if len(sys.argv) > 6 else ps.NONINTERACTING
p = ps.default_params(n=n, E0=1.0, K1=1.0, K2=0.0, kT=1.0, b=1.0, Fz=1.0, num_chains=chains, precision=prec, seed=20260501,
                      energy_type=energy)
with ps.Ensemble(p) as e:
    for _ in range(1 + reps):
        e.advance(mc)
        e.sync()
    s = e.summary()
    print("r3", s.avg[2], "AR", s.acceptance_ratio, "updates per launch", chains * mc)
