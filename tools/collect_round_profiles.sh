#!/bin/bash
# Every rocprofv3 record of a round, on the kernel sources as they stand (run on the GPU box):
#   bash tools/collect_round_profiles.sh r04      -> gpurun_out/prof_<tag>_*/ (raw), then
#   bash tools/summarize_round_profiles.sh r04    -> profiles/<tag>/*.csv + profiles/pmc_traffic.json (run anywhere)
# One kernel trace + separate --pmc passes per workload (tools/collect_pmc.sh); every summary is stamped with the sha256
# of the kernel sources (tools/summarize_pmc.py), so a profile can always be attributed to the code it describes.
set -uo pipefail
tag=${1:-r04}
part=${2:-all}        # "A" = the bench command and the BASELINE configurations, "B" = the other kernels, "all" = both (two gpurun calls fit their time limit)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PMC_TIMEOUT=${PMC_TIMEOUT:-150}
run() { echo "== $* ($(date +%T))"; bash tools/collect_pmc.sh "$@" > /dev/null 2>&1 || { echo "collection failed: $*"; exit 1; }; }
if [ "$part" != B ]; then
# the bench command itself (headline f64 kernel; the f32 fast path)
run ${tag}_bench_f64 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-path --no-rng-named --no-configs
run ${tag}_bench_f32 bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision f32 --no-configs
# the other BASELINE configurations exactly as bench.py's `configs` array launches them (tools/configs.py): C1 n = 20 (cells in
# LDS), C3 polar n = 100, C4 all-pairs n = 64, C5 the 546-point n = 200 Ising grid (C2 is the bench command above)
run ${tag}_cfg_C1 tools/profile_config.py C1 2
run ${tag}_cfg_C3 tools/profile_config.py C3 2
run ${tag}_cfg_C4 tools/profile_config.py C4 2
run ${tag}_cfg_C5 tools/profile_config.py C5 2
fi
if [ "$part" != A ]; then
# the clustering main, chain per lane (f64: chains in device memory), n = 100
run ${tag}_cluster_f64_ni tools/profile_cluster.py ni f64 5000 2
run ${tag}_cluster_f64_ising tools/profile_cluster.py ising f64 5000 2
# the same main in f32: cells in LDS (n = 100), and the in-memory home that 65 536 chains of n = 200 get by default
run ${tag}_cluster_f32_ni tools/profile_cluster.py ni f32 5000 2
run ${tag}_cluster_f32_mem_n200 tools/profile_cluster.py ni f32 5000 2 200
# the clustering main one chain per wavefront: the reference's phase scan, 2 730 single-chain cases (run/K1_E0-kT-phase.jl)
run ${tag}_cluster_cw_phase tools/profile_cluster_cw.py whole 1 20000 2
# the all-pairs clustering main, n = 100
run ${tag}_cluster_wave_f64_n100 tools/profile_cluster.py interacting f64 1000 2
# the f64 non-interacting sweep at the phase-scan chain length (the kernel furthest below its roofline)
run ${tag}_sweep_f64_ni_n200 tools/profile_sweep.py f64 200 65536 50000 2 0
fi
echo "collected $part"
