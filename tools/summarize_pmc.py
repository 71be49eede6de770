#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/collect_pmc.sh) -> profiles/<round>/<name>_{kernel_trace,pmc}.csv and, with --record KEY,
an entry in profiles/pmc_traffic.json (what bench.py's `roofline.traffic` and `valu.ops_per_update` read), stamped
with the sha256 of the kernel sources it was collected on.

    python tools/summarize_pmc.py gpurun_out/prof_f64 profiles/r02 sweep_f64 --kernel sweep_kernel \
           --updates 6553600000 --record f64_n100_c65536_s100000

Every figure EXCLUDES the first dispatch of the kernel (the warm-up launch)."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src"); ap.add_argument("dst"); ap.add_argument("name")
    ap.add_argument("--kernel", default="sweep_kernel")
    ap.add_argument("--updates", type=float, default=0.0, help="attempted updates per launch")
    ap.add_argument("--record", default=None)
    a = ap.parse_args()
    os.makedirs(a.dst, exist_ok=True)
    cmd = open(os.path.join(a.src, "command.txt")).read().strip()
    import bench
    stamp = bench.kernel_source_hash()      # every summary names the kernel sources it was taken on
    stamp_line = (f"# kernel_source_sha256_16: {stamp}   (sha256 over polymer_stats_amd/csrc/*.hip, *.h, Makefile and "
                  f"include/pstat.h at collection time; bench.kernel_source_hash())\n")

    # kernel trace: per-dispatch durations of the kernel, warm-up launch dropped
    tr = glob.glob(os.path.join(a.src, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(tr)) if a.kernel in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    full = [d for d in dur[1:] if d > 0.5 * max(dur[1:])]
    kname = rows[0]["Kernel_Name"]
    with open(os.path.join(a.dst, a.name + "_kernel_trace.csv"), "w") as f:
        f.write(stamp_line)
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 {cmd}\n")
        f.write("# per-dispatch durations of the kernel; the first dispatch (warm-up) is excluded from the summary row\n")
        f.write("kernel,dispatches_total,dispatches_summarised,avg_ms,min_ms,max_ms,first_dispatch_ms\n")
        f.write(f"\"{kname}\",{len(dur)},{len(full)},{sum(full) / len(full):.6f},{min(full):.6f},{max(full):.6f},{dur[0]:.6f}\n")
        st = glob.glob(os.path.join(a.src, "trace", "**", "*kernel_stats.csv"), recursive=True)
        if st:
            f.write("# rocprofv3's own --stats table of the same run (all dispatches, warm-up included):\n")
            for line in open(st[0]):
                f.write("# " + line)
    kern_ms = sum(full) / len(full)

    # PMC passes: sum over the XCD/SE instances of a dispatch, then mean over the non-warm-up dispatches
    per = collections.defaultdict(dict)
    kinfo = None
    for fn in glob.glob(os.path.join(a.src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if a.kernel not in r["Kernel_Name"]:
                continue
            d = per[r["Counter_Name"]]
            d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
            kinfo = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count",
                                        "SGPR_Count", "LDS_Block_Size") if k in r}
    mean = {}
    lines = [stamp_line.rstrip(), f"# rocprofv3 --pmc <group> --kernel-trace -- python3 {cmd}   (one counter group per run)",
             "# kernel: " + json.dumps(kinfo),
             "# rows: mean over the kernel's dispatches EXCEPT its first one (the warm-up launch)",
             "counter,dispatches,mean_per_dispatch,min,max"]
    for name in sorted(per):
        ids = sorted(per[name])
        v = [per[name][i] for i in ids[1:]] or [per[name][ids[0]]]
        v = [x for x in v if x >= 0.5 * max(v)] or v
        mean[name] = sum(v) / len(v)
        lines.append(f"{name},{len(v)},{mean[name]:.6g},{min(v):.6g},{max(v):.6g}")
    if a.updates and "SQ_INSTS_VALU" in mean:
        lines.append(f"# VALU wave-instructions per update per lane (x64 lanes / updates): {mean['SQ_INSTS_VALU'] * 64 / a.updates:.4f}")
    raw = corrected = exact = None
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        raw = (mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
        corrected = (2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
        lines.append(f"# fabric bytes per launch, raw = (FETCH_SIZE + WRITE_SIZE)*1024: {raw:.6g}")
        lines.append(f"# fabric bytes per launch, guide's correction for wide streaming reads = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                     f"(MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies a 128-byte request at 64): {corrected:.6g}")
    if all(k in mean for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
        n32, n64, n128, nall = (mean[k] for k in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum",
                                                  "TCC_EA0_RDREQ_sum"))
        rd = 32 * n32 + 64 * n64 + 128 * n128
        lines.append(f"# fabric READ bytes per launch by request size = 32*{n32:.6g} + 64*{n64:.6g} + 128*{n128:.6g} = {rd:.6g} "
                     f"({100 * n128 / max(nall, 1):.1f} % of the {nall:.6g} requests are 128-byte ones"
                     + (f"; FETCH_SIZE*1024 = {mean['FETCH_SIZE'] * 1024:.6g} is {mean['FETCH_SIZE'] * 1024 / rd:.3f} of it)" if "FETCH_SIZE" in mean else ")"))
        if "WRITE_SIZE" in mean:
            exact = rd + mean["WRITE_SIZE"] * 1024
            lines.append(f"# fabric bytes per launch, exact reads + WRITE_SIZE*1024: {exact:.6g}")
    f64n = [mean.get(k) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64")]
    f64_insts = sum(f64n) if all(v is not None for v in f64n) else None
    if f64_insts is not None and "SQ_INSTS_VALU" in mean:
        lines.append(f"# f64 VALU wave-instructions (FMA + ADD + MUL + TRANS): {f64_insts:.6g} = "
                     f"{100 * f64_insts / mean['SQ_INSTS_VALU']:.1f} % of SQ_INSTS_VALU")
    open(os.path.join(a.dst, a.name + "_pmc.csv"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    print("kernel avg ms (warm-up excluded):", kern_ms, "over", len(full), "dispatches; first:", dur[0])

    if a.record:
        path = os.path.join(os.path.dirname(a.dst.rstrip("/")), "pmc_traffic.json")
        try:
            rec = json.load(open(path))
            if rec.get("kernel_source_sha256_16") != stamp:
                rec = {}
        except Exception:
            rec = {}
        rec.setdefault("records", {})
        rec["kernel_source_sha256_16"] = stamp
        rec["note"] = ("collected by tools/collect_pmc.sh + tools/summarize_pmc.py; bench.py uses a record only while the "
                       "sha256 of polymer_stats_amd/csrc/* + include/pstat.h still equals the stamp")
        rec["records"][a.record] = {
            # `hbm_bytes_per_launch` = the figure bench.py prints as roofline.traffic: exact where the request-size
            # counters were collected, else the guide's streaming-read correction
            "hbm_bytes_per_launch": exact if exact is not None else corrected,
            "traffic_method": ("32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B + WRITE_SIZE*1024 (TCC_EA0 request-size counters)"
                               if exact is not None else
                               "(2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md HBM: wide streaming reads; uncalibrated for scattered 16-byte reads)"),
            "raw_bytes_per_launch": raw, "streaming_corrected_bytes_per_launch": corrected,
            "fetch_size_kib": mean.get("FETCH_SIZE"), "write_size_kib": mean.get("WRITE_SIZE"),
            "round": os.path.basename(a.dst.rstrip("/")),
            "valu_wave_instructions_per_launch": mean["SQ_INSTS_VALU"],
            "valu_instructions_per_update_per_lane": mean["SQ_INSTS_VALU"] * 64 / a.updates,
            "valu_f64_instructions_per_update_per_lane": (f64_insts * 64 / a.updates) if f64_insts is not None else None,
            # every counted instruction type (waitcnt, nop and branches have no counter): what a lone wave pays an issue slot for
            "instructions_per_update_per_lane": sum(mean.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                                                                               "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) * 64 / a.updates,
            # share of the waves' cycles in which an instruction of theirs was executing / in which they sat in s_waitcnt
            "wave_cycles_with_an_instruction": (mean["SQ_ACTIVE_INST_ANY"] / mean["SQ_WAVE_CYCLES"])
            if "SQ_ACTIVE_INST_ANY" in mean and mean.get("SQ_WAVE_CYCLES") else None,
            "wave_cycles_waiting": (mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"]) if "SQ_WAIT_ANY" in mean and mean.get("SQ_WAVE_CYCLES") else None,
            "kernel_ms_rocprof_trace": kern_ms,
            "updates_per_launch": a.updates,
        }
        json.dump(rec, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
