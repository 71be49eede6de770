cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PMC_TIMEOUT=120 PMC_NOTRACE=1
export PMC_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM;TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
for k in 1 16; do
bash tools/collect_pmc.sh r04_coldk$k tools/cold_lanes.py $k 64 4000 > /dev/null 2>&1
python - $k <<'PY'
import csv, glob, collections, sys
k=sys.argv[1]
per=collections.defaultdict(dict)
for fn in glob.glob("gpurun_out/prof_r04_coldk%s/pmc_*/**/*counter_collection.csv"%k, recursive=True):
    for r in csv.DictReader(open(fn)):
        if "cluster" not in r["Kernel_Name"]: continue
        d=per[r["Counter_Name"]]; d[int(r["Dispatch_Id"])]=d.get(int(r["Dispatch_Id"]),0.0)+float(r["Counter_Value"])
print("== k =", k)
for c in sorted(per):
    ids=sorted(per[c]); v=[per[c][i] for i in ids[-3:]]
    print(c, "%.6g"%(sum(v)/len(v)), "per wave-step %.1f"%(sum(v)/len(v)/(64*4000)))
PY
done
