"""One BASELINE configuration (tools/configs.py) as bare launches for rocprofv3: a warm-up launch and `reps` full ones.
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_config.py C4 [reps=2]
tools/summarize_pmc.py drops the first (warm-up) dispatch of the kernel from every figure."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import polymer_stats_amd as ps
from configs import config_list

want = sys.argv[1] if len(sys.argv) > 1 else "C4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = [c for c in config_list(ps) if c["id"] == want][0]
with ps.Ensemble(cfg["cases"]) as e:
    for _ in range(1 + reps):
        e.advance(cfg["mc_steps"])
        e.sync()
    s = e.summary(0)
    print(want, e.launch_info().kernel.decode(), "r3", s.avg[2], "AR", s.acceptance_ratio, "updates per launch",
          e.num_chains * e.ncases * cfg["mc_steps"])
