#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) -> profiles/<round>/{kernel_stats.csv, pmc_summary.csv,
bench_line_*.json} and profiles/pmc_traffic.json (what bench.py's `roofline.traffic` and `valu` read).

    python tools/summarize_profiles.py gpurun_out/prof_now profiles/r01_final
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(ks, os.path.join(dst, "kernel_stats.csv"))
    kern_ms = None
    for r in csv.DictReader(open(ks)):
        if "sweep_kernel" in r["Name"]:
            # every launch of bench.py (warm-up included) is one full step of 1e5 MC steps per chain
            kern_ms = float(r["AverageNs"]) / 1e6
            avg_ms = kern_ms
            calls = int(r["Calls"])
    agg = collections.defaultdict(list)
    kinfo = None
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "sweep_kernel" not in r["Kernel_Name"]:
                continue
            agg[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
            kinfo = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count") if k in r}
    per = collections.defaultdict(list)
    for (name, disp), vals in agg.items():
        per[name].append(sum(vals))
    lines = ["# rocprofv3 --pmc passes, one counter group per run, command: python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
             "# kernel: " + json.dumps(kinfo),
             "# rows: dispatches of the sweep kernel whose counter value is within 2x of the largest (the full-length launches)",
             "counter,dispatches,mean_per_dispatch,min,max"]
    full = {}
    for name in sorted(per):
        v = [x for x in per[name] if x > 0.5 * max(per[name])]
        full[name] = sum(v) / len(v)
        lines.append(f"{name},{len(v)},{full[name]:.6g},{min(v):.6g},{max(v):.6g}")
    open(os.path.join(dst, "pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
    for p in ("f32", "q16", "f64"):
        f = os.path.join(src, f"bench_line_{p}.json")
        if os.path.exists(f) and os.path.getsize(f) > 0:
            shutil.copy(f, os.path.join(dst, f"bench_line_{p}.json"))
    line = json.load(open(os.path.join(src, "bench_line_f32.json")))
    updates = line["config"]["chains_per_gpu"] * line["config"]["mc_steps_per_chain"]
    key = f"f32_n{line['config']['n']}_c{line['config']['chains_per_gpu']}_s{line['config']['mc_steps_per_chain']}"
    traffic = {
        key: {
            "hbm_bytes_per_launch": (2 * full["FETCH_SIZE"] + full["WRITE_SIZE"]) * 1024,
            "fetch_size_kib": full["FETCH_SIZE"], "write_size_kib": full["WRITE_SIZE"],
            "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reads half, MI355X_MICROARCH.md HBM)",
            "round": os.path.basename(dst.rstrip("/")),
            "valu_wave_instructions_per_launch": full["SQ_INSTS_VALU"],
            "valu_instructions_per_update_per_lane": full["SQ_INSTS_VALU"] * 64 / updates,
            "kernel_ms_rocprof_trace": kern_ms,
        }
    }
    json.dump(traffic, open(os.path.join(os.path.dirname(dst.rstrip("/")), "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    print("kernel avg ms (all launches incl. warm-up)", avg_ms, "calls", calls)


if __name__ == "__main__":
    main()
