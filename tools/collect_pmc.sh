#!/bin/bash
# rocprofv3 evidence for one kernel workload on the GPU box:
#   tools/collect_pmc.sh <tag> <python script and its arguments ...>   -> gpurun_out/prof_<tag>/{trace,pmc_*}
# One --kernel-trace --stats run, then separate --pmc passes (the HBM counters never share a pass with others:
# MI355X_MICROARCH.md, HBM / rocprofv3 section).  The program itself follows `--` (no env/bash hop).
# Summarise with tools/summarize_pmc.py.
set -euo pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
echo "$@" > "$out/command.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 "$@" > "$out/trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmc_$name" -o p -- python3 "$@" > "$out/pmc_$name.log" 2>&1
done
ls "$out"
