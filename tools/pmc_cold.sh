# PMC passes of the clustering main on a cold (aligned) chain: where do a long cluster's cycles go?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export E0=3 KT=0.1 K1=1 K2=0 PMC_TIMEOUT=120 PMC_NOTRACE=1
export PMC_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS;SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"
bash tools/collect_pmc.sh r04_cold tools/time_cluster.py ising f64 2000 100 ${1:-65536} > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
per=collections.defaultdict(dict)
for fn in glob.glob("gpurun_out/prof_r04_cold/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "cluster" not in r["Kernel_Name"]: continue
        d=per[r["Counter_Name"]]; d[int(r["Dispatch_Id"])]=d.get(int(r["Dispatch_Id"]),0.0)+float(r["Counter_Value"])
for k in sorted(per):
    ids=sorted(per[k]); v=[per[k][i] for i in ids[1:]]
    print(k, len(v), "%.6g"%(sum(v)/max(1,len(v))))
PY
