#!/bin/bash
# rocprofv3 evidence for the interacting kernel: tools/collect_interacting.sh <tag> [n]
set -euo pipefail
tag=${1:-now}; n=${2:-64}
out=gpurun_out/prof_inter_$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 tools/profile_interacting.py $n > "$out/trace.log" 2>&1
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmc_$name" -o p -- python3 tools/profile_interacting.py $n > "$out/pmc_$name.log" 2>&1
done
ls "$out"
