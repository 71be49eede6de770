#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's `roofline` object on the GPU box:
#   tools/collect_profiles.sh <tag>        -> gpurun_out/prof_<tag>/{trace,pmc_*}/...
# One --kernel-trace --stats run and separate --pmc passes (counters never share a pass with the
# HBM ones: MI355X_MICROARCH.md, HBM / rocprofv3 section).  Summarise with tools/summarize_profiles.py.
set -euo pipefail
tag=${1:-now}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cmd="bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 $cmd > "$out/trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmc_$name" -o p -- python3 $cmd > "$out/pmc_$name.log" 2>&1
done
python3 bench.py > "$out/bench_line_f32.json" 2> "$out/bench_f32.err"
python3 bench.py --precision q16 --no-cpu-baseline > "$out/bench_line_q16.json" 2>/dev/null
python3 bench.py --precision f64 --no-cpu-baseline --steps 3 --warmup 1 > "$out/bench_line_f64.json" 2>/dev/null
ls "$out"
