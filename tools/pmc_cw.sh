# PMC passes of the chain-per-wavefront cluster kernel on the phase-scan grid: instructions and cycles per chain-step
#   bash tools/pmc_cw.sh [whole|hot|cold] [per_case=1] [steps=20000]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
which=${1:-whole}; per=${2:-1}; steps=${3:-20000}
export PMC_TIMEOUT=120 PMC_NOTRACE=1 PSTAT_F64_STATE=${PSTAT_F64_STATE:-wave}
export PMC_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE;SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
bash tools/collect_pmc.sh cw tools/profile_cluster_cw.py $which $per $steps > /dev/null 2>&1
python3 - $which $per $steps <<'PY'
import csv, glob, collections, sys
which, per, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nk = {"whole": 21, "hot": 11, "cold": 6}[which]
units = 5 * 26 * nk * per * steps
per_c = collections.defaultdict(dict)
for fn in glob.glob("gpurun_out/prof_cw/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "cluster" not in r["Kernel_Name"]: continue
        d = per_c[r["Counter_Name"]]; d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
print("%s grid, %d per case, %d steps: per chain-step" % (which, per, steps))
for k in sorted(per_c):
    ids = sorted(per_c[k]); v = [per_c[k][i] for i in ids[1:]]
    m = sum(v) / max(1, len(v))
    print("  %-28s %12.6g   %10.2f per chain-step" % (k, m, m / units))
PY
