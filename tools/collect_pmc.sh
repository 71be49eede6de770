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
# every pass under its own timeout: a counter group the hardware cannot collect makes rocprofv3 abort and then sit there
T=${PMC_TIMEOUT:-240}
timeout -k 10 $T rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 "$@" > "$out/trace.log" 2>&1
# PMC_GROUPS="A B;C D" replaces the default counter groups (one rocprofv3 pass per ';'-separated group); PMC_NOTRACE=1 skips the trace run
if [ -n "${PMC_GROUPS:-}" ]; then IFS=';' read -r -a groups <<< "$PMC_GROUPS"; else
groups=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"
        # the fabric read requests by size: the exact byte count behind FETCH_SIZE (which tallies a 128-byte request at 64)
        "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
        "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
        # the f64 share of the VALU stream (an f64 op issues at 16 lanes per clock, everything else at 32)
        "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"); fi
for grp in "${groups[@]}"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  timeout -k 10 $T rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmc_$name" -o p -- python3 "$@" > "$out/pmc_$name.log" 2>&1
done
ls "$out"
