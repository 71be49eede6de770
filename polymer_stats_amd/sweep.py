"""What the reference's sweep drivers do with `pmap` over subprocesses, as batched launches.

A `run/*.jl` script (run/interacting_dielectric_study.jl, run/Ising_2025-12-18.jl, run/K1_E0-kT-phase.jl, ...) builds a
list of cases (dicts of E0, K1, K2, kT, Fz, Fx, n, b, kappa, run ...), and for every case starts
`julia mcmc_[clustering_]eap_chain.jl <fixed flags> <the case's flags> --prefix workdir/<name>` in a worker process,
writes the ten (twelve) stdout lines to `workdir/<name>.out` unless that file already exists, where
`<name>` = `E0-0001000_K1-0000000_..._b-0001000[_run-001]` (every value x 1000, `%07d`).  scripts/aggregate_mcmc.jl
then reads the values back out of the file NAMES and the stdout lines.

Here the cases of a sweep that share everything but their physics scalars become ONE ensemble (`pstat_create` with
ncases > 1) and one launch per stage of the main's protocol -- the hosts' own `mcmc_cases` / `run_cases`, which the command
line calls with one case --; cases are dealt to ranks round-robin (independent work, no exchange: `pmap`'s own partitioning),
and every case's `.out` holds the lines the single-case host prints for the same options, seed and chains.

What "the same" means: a chain's TRAJECTORY depends on its (seed, chain id) and its case's options only -- in f64 bit for bit
whatever else shares the ensemble, however many ranks there are and whatever is already finished.  The printed f64 averages
agree to ~1e-10 relative, not to the last bit: how a launch is cut into time segments (and, in f32, where the chains live)
depends on how many chains are co-batched, and the running sums are folded in 128-step blocks that start at segment
boundaries (the f64 clustering main also re-derives its cached n-hat there).  f32 / q16 cases are statistically equivalent
across partitions, not trajectory-identical, once an ensemble is large enough to change the kernel's home
(tests/test_gpu_sweep.py::test_a_case_does_not_depend_on_what_shares_its_ensemble_even_past_the_resident_slots).

    python tools/run_sweep.py WORKDIR --main mcmc_clustering_eap_chain --num-chains 64 \
        --axis run=1:5 --axis kT='10^(-2:0.2:2)' --axis E0=0:0.2:5 --axis K1=1 --axis K2=0 --axis Fz=0 --axis Fx=0 \
        --axis n=100 --axis b=1 --axis kappa=0 --name E0,K1,K2,kT,Fz,Fx,n,b,kappa,run:raw \
        -- --chain-type dielectric --energy-type Ising --num-steps 2500000 --burn-in 100000 -v 2
"""
from __future__ import annotations

import ast
import itertools
import json
import os
import time

import numpy as np

from . import _lib          # (tools/run_sweep.py counts the devices through it)
from . import mcmc_clustering_eap_chain as cluster_main
from . import mcmc_eap_chain as fixed_main
from .mcmc_eap_chain import ReferenceError_

MAINS = {"mcmc_eap_chain": fixed_main, "mcmc_clustering_eap_chain": cluster_main}

# how the reference's run scripts turn a case's keys into flags of the main (run/K1_E0-kT-phase.jl:45)
KEY_FLAG = {"E0": "--E0", "K1": "--K1", "K2": "--K2", "mu": "--mu", "kT": "--kT", "Fz": "--Fz", "Fx": "--Fx",
            "n": "--num-monomers", "b": "--mlen", "kappa": "--bend-mod", "run": None}
NAME_ORDER = ["E0", "K1", "K2", "mu", "kT", "Fz", "Fx", "n", "b", "kappa", "run"]
# pargs entries that may differ between the cases of one ensemble (pstat_create: "differ only in physics scalars")
PER_CASE = {"E0", "K1", "K2", "mu", "kT", "Fz", "Fx", "mlen", "bend-mod", "bend-angle", "cluster-prob", "cutoff-radius",
            "prefix", "seed"}


# ---------------------------------------------------------------------------------------------- the case list
_BIN = {ast.Add: lambda a, b: a + b, ast.Sub: lambda a, b: a - b, ast.Mult: lambda a, b: a * b, ast.Div: lambda a, b: a / b,
        ast.Pow: lambda a, b: a ** b}


def _arith(node) -> float:
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
        return node.value
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _arith(node.operand)
        return -v if isinstance(node.op, ast.USub) else v
    if isinstance(node, ast.BinOp) and type(node.op) in _BIN:
        return _BIN[type(node.op)](_arith(node.left), _arith(node.right))
    raise ValueError("not a number")


def _number(text: str):
    return _arith(ast.parse(text.strip().replace("^", "**"), mode="eval").body)


def julia_range(text: str) -> list:
    """`a:b` or `a:s:b` the way Julia enumerates it: a, a+s, ... while <= b (count from the rounded quotient, values
    as a + k*s -- Julia's own StepRangeLen is a shade more careful, equal to this to an ulp; names are rounded to 1e-3)."""
    parts = [_number(p) for p in text.split(":")]
    if len(parts) == 2:
        a, s, b = parts[0], 1, parts[1]
    elif len(parts) == 3:
        a, s, b = parts
    else:
        raise ValueError(f"range '{text}' not understood")
    if s == 0:
        raise ValueError(f"range '{text}' has step 0")
    cnt = int(np.floor((b - a) / s + 1e-9)) + 1
    vals = [a + k * s for k in range(max(cnt, 0))]
    if all(isinstance(p, int) for p in (a, s, b)):
        return [int(v) for v in vals]
    return [float(np.round(v, 12)) for v in vals]


def axis_values(text: str) -> list:
    """`1,5,10` | `0:0.2:5` | `10^(-2:0.2:2)` (run/K1_E0-kT-phase.jl:22-27) | `0:0.05:1,1.5:0.5:5` (vcat of ranges,
    run/noninteracting-compare-with-clustering_2021-09-24.jl:21) -> values."""
    t = text.strip()
    if t.startswith("10^(") and t.endswith(")"):
        return [float(10.0 ** x) for x in julia_range(t[4:-1])]
    out = []
    for item in t.split(","):
        if item.strip():
            out += julia_range(item) if ":" in item else [_number(item)]
    return out


def product_cases(axes: list[tuple[str, list]]) -> list[dict]:
    """`for a in As, b in Bs, ...` (first axis outermost, run/interacting_dielectric_study.jl:28)."""
    keys = [k for k, _ in axes]
    return [dict(zip(keys, combo)) for combo in itertools.product(*[v for _, v in axes])]


_CMP = {ast.Eq: lambda a, b: a == b, ast.NotEq: lambda a, b: a != b, ast.Lt: lambda a, b: a < b,
        ast.LtE: lambda a, b: a <= b, ast.Gt: lambda a, b: a > b, ast.GtE: lambda a, b: a >= b}


def _cond(node, case):
    if isinstance(node, ast.BoolOp):
        vals = [_cond(v, case) for v in node.values]
        return all(vals) if isinstance(node.op, ast.And) else any(vals)
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.Not):
        return not _cond(node.operand, case)
    if isinstance(node, ast.Compare):
        left = _cond(node.left, case)
        for op, right in zip(node.ops, node.comparators):
            r = _cond(right, case)
            if not _CMP[type(op)](left, r):
                return False
            left = r
        return True
    if isinstance(node, ast.Name):
        if node.id not in case:
            raise ValueError(f"--skip names '{node.id}', which is not a key of the cases")
        return case[node.id]
    return _arith(node)


def skip_case(expr: str, case: dict) -> bool:
    """`K1==K2` (run/interacting_dielectric_study.jl:29): comparisons of case keys and numbers, and/or/not."""
    return bool(_cond(ast.parse(expr.replace("&&", " and ").replace("||", " or "), mode="eval").body, case))


def fmt(x) -> str:
    """`@sprintf("%07d", round(Int, 1e3*x))` (run/interacting_dielectric_study.jl:12)."""
    return "%07d" % int(round(1e3 * x))


def name_spec(text: str | None, keys) -> list[tuple[str, str]]:
    """`E0,K1,...,run:int` -> [(key, kind)], kind in milli (%07d of 1000 x) | int (%03d) | raw."""
    if not text:
        order = [k for k in NAME_ORDER if k in keys] + [k for k in keys if k not in NAME_ORDER]
        return [(k, "int" if k == "run" else "milli") for k in order]       # run/Ising_2025-12-18.jl:15: `_run-$(fmt_int(run))`
    out = []
    for tok in text.split(","):
        k, _, kind = tok.strip().partition(":")
        kind = kind or "milli"
        if kind not in ("milli", "int", "raw"):
            raise ValueError(f"name kind '{kind}' not understood (milli | int | raw)")
        out.append((k, kind))
    return out


def case_name(case: dict, spec) -> str:
    toks = []
    for k, kind in spec:
        v = case[k]
        toks.append(f"{k}-" + (fmt(v) if kind == "milli" else "%03d" % int(v) if kind == "int" else str(v)))
    return "_".join(toks)


def case_argv(case: dict) -> list[str]:
    argv = []
    for k, v in case.items():
        flag = KEY_FLAG.get(k, f"--{k}")
        if flag is not None:
            whole = isinstance(v, int) or (flag == "--num-monomers" and float(v).is_integer())
            argv += [flag, repr(int(v)) if whole else repr(float(v))]      # (an integer option of the main may be an axis too)
    return argv


# ---------------------------------------------------------------------------------------------- the sweep
def plan(main_name: str, fixed_argv: list[str], cases: list[dict], workdir: str, *, name=None, num_chains: int = 64,
         seed: int | None = None, precision: str | None = None, rng: str | None = None, device: int = 0) -> list[dict]:
    """Every case's parsed options (the dict the main's own parser returns) with its prefix, its `.out` path and its
    seed = base seed + position in the FULL case list (so a case's result depends neither on which other cases still
    have to be run nor on how many ranks share them)."""
    main = MAINS[main_name]
    if not cases:
        return []
    spec = name_spec(name, list(cases[0].keys()))
    base = fixed_main.fresh_seed() & 0x7FFFFFFFFFFF if seed is None else int(seed)
    extra = ["--num-chains", str(int(num_chains)), "--devices", str(int(device))]
    if precision:
        extra += ["--precision", precision]
    if rng:
        extra += ["--rng", rng]
    out, seen = [], set()
    for k, case in enumerate(cases):
        nm = case_name(case, spec)
        if nm in seen:
            raise ValueError(f"two cases share the file name '{nm}': add the key that tells them apart to --name")
        seen.add(nm)
        pargs = main.parse_args(list(fixed_argv) + case_argv(case) + extra +
                                ["--prefix", os.path.join(workdir, nm), "--seed", str(base + k)])
        if pargs["profile"]:
            raise ReferenceError_("not implemented for the HPC env")
        pargs["_case"], pargs["_name"], pargs["_out"], pargs["_index"] = case, nm, os.path.join(workdir, nm + ".out"), k
        out.append(pargs)
    return out


def _signature(pargs: dict):
    return tuple(sorted((k, repr(v)) for k, v in pargs.items() if k not in PER_CASE and not k.startswith("_")))


def run_sweep(main_name: str, fixed_argv: list[str], cases: list[dict], workdir: str, *, name=None, num_chains: int = 64,
              seed: int | None = None, precision: str | None = None, rng: str | None = None, rank: int = 0, world: int = 1,
              device: int = 0, overwrite: bool = False, write_csv: bool = False, max_chains: int = 262144, log=None) -> dict:
    """Runs the cases whose `.out` does not exist yet (run/interacting_dielectric_study.jl:39) and that fall to this rank
    (position in the full case list modulo `world`).  Returns {"ran": [...], "skipped": [...], "launches": k}."""
    main = MAINS[main_name]
    os.makedirs(workdir, exist_ok=True)
    todo_all = plan(main_name, fixed_argv, cases, workdir, name=name, num_chains=num_chains, seed=seed, precision=precision,
                    rng=rng, device=device)
    skipped = [p["_name"] for p in todo_all if os.path.isfile(p["_out"]) and not overwrite]
    todo = [p for p in todo_all if overwrite or not os.path.isfile(p["_out"])]
    mine = [p for p in todo if p["_index"] % world == rank]     # by position in the full list: ranks need not agree on what is done
    groups: dict = {}
    for p in mine:
        groups.setdefault(_signature(p), []).append(p)
    ran, launches = [], 0
    per_launch = max(1, int(max_chains) // max(1, int(num_chains)))
    for plist_all in groups.values():
        for i in range(0, len(plist_all), per_launch):
            plist = plist_all[i:i + per_launch]
            t0 = time.time()
            info = {}
            if main is cluster_main:      # the main's own protocol, every case of the ensemble at once
                res = main.run_cases(plist, write_csv=write_csv, info=info)
            else:
                res = main.mcmc_cases(int(plist[0]["num-steps"]), plist, write_csv=write_csv, info=info)
            for p, (sas, vas, ar) in zip(plist, res):
                tmp = p["_out"] + f".tmp{os.getpid()}"
                with open(tmp, "w") as f:           # println x 10 (12); complete or absent: an interrupted sweep re-runs the case
                    f.write("\n".join(main.summary_lines(sas, vas, ar, p)) + "\n")
                os.replace(tmp, p["_out"])
                ran.append(p["_name"])
            launches += 1
            if log:
                log(f"rank {rank}: {len(plist)} cases x {num_chains} chains, n = {plist[0]['num-monomers']}, "
                    f"{info.get('kernel', '?')}: {time.time() - t0:.2f} s")
    return {"ran": ran, "skipped": skipped, "launches": launches}


def load_cases(path: str) -> list[dict]:
    """An explicit case list (what run/Ising_2025-12-18.jl spells out push! by push!): a JSON array of objects."""
    with open(path) as f:
        cases = json.load(f)
    if not isinstance(cases, list) or not all(isinstance(c, dict) for c in cases):
        raise ValueError(f"{path}: expected a JSON array of objects")
    return cases
