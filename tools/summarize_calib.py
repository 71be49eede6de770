#!/usr/bin/env python3
"""tools/fetch_calib.hip under rocprofv3 (gpurun_out/calib/{fetch,req}) -> a text table: per kernel, the known byte counts
beside FETCH_SIZE*1024 and the request-size histogram of TCC_EA0_RDREQ.  `python tools/summarize_calib.py SRC OUT.txt`"""
import collections
import csv
import glob
import os
import sys

src, out = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(dict))
for fn in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"]
        k = "stream" if "stream_kernel" in name else ("scatter" if "scatter_kernel" in name else None)
        if "scatter_aux_kernel<" in name:
            k = "scatter_aux<" + name.split("scatter_aux_kernel<")[1].split(">")[0] + ">"
        if k is None:
            continue
        d = per[k][r["Counter_Name"]]
        d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
lines = [open(os.path.join(src, "stdout.txt")).read().strip(),
         "# counters: second dispatch of each kernel (the first is the warm-up); rocprofv3 --pmc, one group per run"]
for k in ["stream", "scatter"] + sorted((x for x in per if x.startswith("scatter_aux")), key=lambda x: int(x.split("<")[1][:-1])):
    m = {c: v[max(v)] for c, v in per[k].items()}
    lines.append(f"{k}_kernel: " + "  ".join(f"{c}={m[c]:.6g}" for c in sorted(m)))
    if "FETCH_SIZE" in m:
        lines.append(f"  FETCH_SIZE*1024 = {m['FETCH_SIZE'] * 1024:.6g} bytes")
    if "TCC_EA0_RDREQ_128B_sum" in m:
        rd = 32 * m["TCC_EA0_RDREQ_32B_sum"] + 64 * m["TCC_EA0_RDREQ_64B_sum"] + 128 * m["TCC_EA0_RDREQ_128B_sum"]
        lines.append(f"  32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B = {rd:.6g} bytes "
                     f"({100 * m['TCC_EA0_RDREQ_128B_sum'] / max(1.0, m['TCC_EA0_RDREQ_sum']):.1f} % of the requests are 128-byte ones)")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
