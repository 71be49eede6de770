"""The consumer's side of the `.out` files: what scripts/aggregate_mcmc.jl makes of a directory of them.

    python -m polymer_stats_amd.aggregate_mcmc <outfile> <indir> <pattern> <dielectric|polar> [3D] [<kappaflag>] [<runflag>]

Same arguments, same CSV: one row per file matching `pattern` in `indir` (sorted, as Glob.jl's readdir), the input fields
read out of the file NAME (`E0-0001000_K1-...`: the text after the first `-` of every `_`-separated token, times 1e-3,
scripts/aggregate_mcmc.jl:69; with runflag the last token is dropped, :63-65) followed by the values right of the `=` of
every line, vectors flattened (:71).  Numbers are written the way Julia's writedlm prints Float64 (shortest round-trip).
The header names the 22 output columns of the clustering main's twelve lines; a file of the fixed-force main's ten lines
gives 20 values and the header drops Ealign and psi (scripts/aggregate_mcmc_legacy.jl).  The planar (2D) variant of the
reference is outside this implementation."""
from __future__ import annotations

import fnmatch
import os
import sys

from .julia_fmt import jl_row

OUT_3D = ["r1", "r2", "r3", "lambda1", "lambda2", "lambda3", "r1sq", "r2sq", "r3sq", "rsquared", "p1", "p2", "p3", "p1sq",
          "p2sq", "p3sq", "psquared", "U", "Usquared", "Ealign", "psi", "AR"]
USAGE = "usage: julia aggregate.jl <outfile> <indir> <pattern> <dielectric|polar> [<3D|2D>] [<kappaflag>] [<runflag>]"


def julia_value(text: str) -> list[float]:
    """`[a, b, c]` or a scalar, as printed by the mains (NaN, Inf, -Inf included) -> floats."""
    t = text.strip()
    items = t[1:-1].split(",") if t.startswith("[") and t.endswith("]") else [t]
    out = []
    for it in items:
        it = it.strip()
        out.append(float({"NaN": "nan", "Inf": "inf", "-Inf": "-inf"}.get(it, it)))
    return out


def name_fields(path: str, runflag: bool = False) -> list[float]:
    toks = os.path.basename(path).split(".")[0].split("_")
    if runflag:
        toks = toks[:-1]
    return [int(tok.split("-", 1)[1]) * 1e-3 for tok in toks]


def out_values(path: str) -> list[float]:
    vals = []
    with open(path) as f:
        for line in f.read().splitlines():
            if line.strip():
                vals += julia_value(line.split("=")[1])
    return vals


def aggregate(outfile: str, indir: str, pattern: str, chain: str, kappaflag: bool = False, runflag: bool = False) -> int:
    if chain == "dielectric":
        heads = ["E0", "K1", "K2", "kT", "Fz", "Fx", "n", "b"]
    elif chain == "polar":
        heads = ["E0", "mu", "kT", "Fz", "Fx", "n", "b"]
    else:
        print("I don't understand the second to last argument")
        return 1
    if kappaflag:
        heads.append("kappa")
    files = sorted(os.path.join(indir, f) for f in os.listdir(indir) if fnmatch.fnmatchcase(f, pattern))
    rows = [(f, name_fields(f, runflag), out_values(f)) for f in files]
    # The reference hcat's whatever the file name holds under the fixed header (scripts/aggregate_mcmc.jl:54,67-74): a name
    # with fewer or more tokens than the header has input columns shifts every value under the wrong heading, silently.
    # Here that is an error: name the files with all of the header's keys (run_sweep.py --name E0,K1,K2,kT,Fz,Fx,n,b[,kappa][,run:raw]).
    for f, fields, _ in rows:
        if len(fields) != len(heads):
            print(f"{os.path.basename(f)}: the file name holds {len(fields)} input field(s) but the header of a {chain} "
                  f"aggregate has {len(heads)} ({','.join(heads)}{' + a run token' if runflag else ''}): columns would be "
                  "misaligned", file=sys.stderr)
            return 1
    cols = OUT_3D if not rows or len(rows[0][2]) == len(OUT_3D) else [c for c in OUT_3D if c not in ("Ealign", "psi")]
    with open(outfile, "w") as out:
        out.write(",".join(heads + cols) + "\n")
        for f, fields, vals in rows:
            print(f"    processing {f} ... ")
            out.write(jl_row(fields + vals) + "\n")
    return 0


def main(argv=None) -> int:
    a = list(sys.argv[1:] if argv is None else argv)
    if len(a) < 4:
        print(USAGE)
        return 1
    if len(a) >= 5 and a[4] != "3D":
        print("the planar (2D) variant is outside this implementation" if a[4] == "2D" else USAGE)
        return 1
    flag = lambda i: len(a) > i and a[i] == "true"     # parse(Bool, ARGS[i])
    return aggregate(a[0], a[1], a[2], a[3], kappaflag=flag(5), runflag=flag(6))


if __name__ == "__main__":
    sys.exit(main())
