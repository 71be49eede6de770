"""Host side of the drop-in for mcmc_clustering_eap_chain.jl: its command line, its burn-in ladder, its
two CSV files and its twelve stdout lines, with the step loop (single-monomer move + cluster_flip!) on
the GPU through libpstat (move_set = PSTAT_MOVES_CLUSTER).

    python -m polymer_stats_amd.mcmc_clustering_eap_chain -n 100 -e 1 -F 1 -u noninteracting \
           --bend-mod 1 --cluster-prob 0.5 -N 1000000 --num-chains 16384 --prefix out/run1 -v 2

Option names, aliases, types and defaults are the reference's (mcmc_clustering_eap_chain.jl:14-152;
note they differ from mcmc_eap_chain.jl: energy-type defaults to Ising, step-adjust-ub to 0.40,
num-steps to 1e6, and there is a 5-rung burn-in ladder by default).  Added: --num-chains, --seed,
--devices, --precision, --rng.

All four energy types run on the device (interacting and cutoff: one chain per wavefront, n <= 512),
and both forms of --x0 ([phi; theta] for every monomer, or 2 n interleaved per-monomer angles).
"""
from __future__ import annotations

import ast
import math
import operator
import sys
import time

import argparse
import numpy as np

from . import _lib
from .julia_fmt import jl_float, jl_row, jl_vector
from .mcmc_eap_chain import Averager, CsvFiles, ReferenceError_, _Pool, _averagers, _log, get_avg, resolve_seed

ROLL_HEADER = "step,r1,r2,r3,r1sq,r2sq,r3sq,rsq,p1,p2,p3,p1sq,p2sq,p3sq,psq,U,Usq,Ealign,psi"   # :259


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="mcmc_clustering_eap_chain", add_help=True, allow_abbrev=False)
    a = p.add_argument
    # --- the reference's table, mcmc_clustering_eap_chain.jl:14-152
    a("--E0", "-e", dest="E0", type=float, default=0.0, help="magnitude of electric field")
    a("--chain-type", "-T", dest="chain-type", type=str, default="dielectric", help="chain type (dielectric|polar)")
    a("--K1", "-J", dest="K1", type=float, default=1.0, help="dipole susceptibility along the monomer axis (dielectric chain)")
    a("--K2", "-K", dest="K2", type=float, default=0.0, help="dipole susceptibility orthogonal to the monomer axis (dielectric chain)")
    a("--mu", "-m", dest="mu", type=float, default=1e-2, help="dipole magnitude (electret chain)")
    a("--bend-mod", "-a", dest="bend-mod", type=float, default=0.0, help="bending modulus of chain")
    a("--bend-angle", "-g", dest="bend-angle", type=float, default=0.0, help="zero energy bond angle")
    a("--energy-type", "-u", dest="energy-type", type=str, default="Ising", help="energy type (interacting|cutoff|Ising|noninteracting)")
    a("--cutoff-radius", dest="cutoff-radius", type=float, default=7.5, help="cut off radius (units of monomer lengths)")
    a("--kT", "-k", dest="kT", type=float, default=1.0, help="dimensionless temperature")
    a("--Fz", "-F", dest="Fz", type=float, default=0.0, help="force in the z-direction (direction of E-field; force ensemble)")
    a("--Fx", "-G", dest="Fx", type=float, default=0.0, help="force in the x-direction (force ensemble)")
    a("--mlen", "-b", dest="mlen", type=float, default=1.0, help="monomer length")
    a("--num-monomers", "-n", dest="num-monomers", type=int, default=100, help="number of monomers")
    a("--num-steps", "-N", dest="num-steps", type=int, default=int(1e6), help="number of steps")
    a("--phi-step", "-p", dest="phi-step", type=float, default=3 * math.pi / 8, help="maximum phi step length")
    a("--theta-step", "-q", dest="theta-step", type=float, default=3 * math.pi / 16, help="maximum theta step length")
    a("--cluster-prob", dest="cluster-prob", type=float, default=0.5, help="probability of flipping a cluster")
    a("--step-adjust-lb", "-L", dest="step-adjust-lb", type=float, default=0.15, help="adjust step sizes if acc. ratio below this threshold")
    a("--step-adjust-ub", "-U", dest="step-adjust-ub", type=float, default=0.40, help="adjust step sizes if acc. ratio above this threshold")
    a("--step-adjust-scale", "-A", dest="step-adjust-scale", type=float, default=1.1, help="scale factor for adjusting step sizes (> 1.0)")
    a("--steps-per-adjust", "-S", dest="steps-per-adjust", type=int, default=2500, help="steps between step size adjustments")
    a("--umbrella-sampling", "-B", dest="umbrella-sampling", action="store_true", help="use umbrella sampling (w/ electrostatic weight function)")
    a("--update-freq", dest="update-freq", type=float, default=15.0, help="update frequency (seconds)")
    a("--verbose", "-v", dest="verbose", type=int, default=3, help="verbosity level: 0-nothing, 1-errors, 2-warnings, 3-info")
    a("--prefix", "-P", dest="prefix", type=str, default="eap-mcmc", help="prefix for output files")
    a("--postfix", "-Q", dest="postfix", type=str, default="", help="postfix for output files")
    a("--stepout", "-s", dest="stepout", type=int, default=500, help="steps between storing microstates")
    a("--numeric-type", dest="numeric-type", type=str, default="float64", help="numerical data type for averaging (float64|float128|dec128|big)")
    a("--burn-in", dest="burn-in", type=int, default=50000, help="steps for burn-in; i.e. steps before averaging")
    a("--burn-schedule", dest="burn-schedule", type=str, default="[1000; 100; 10; 2; 1]", help="temperature schedule for burn-in")
    a("--x0", dest="x0", type=str, default=None, help="initial configuration")
    a("--dx0", dest="dx0", type=str, default="[2*pi, 1e-1]", help="random perturbation of x0")
    a("--profile", "-Z", dest="profile", action="store_true", help="profile the program")
    # --- ours
    a("--num-chains", dest="num-chains", type=int, default=4096, help="independent chains run at once on the GPU(s) and pooled")
    a("--seed", dest="seed", type=int, default=None,
      help="seed of the per-chain generators; default: fresh OS entropy per run, like the reference's unseeded RNG "
           "(the seed drawn is echoed on stderr at --verbose >= 2)")
    a("--devices", dest="devices", type=str, default="0", help="comma-separated HIP device ordinals; chains are sharded over them")
    a("--rng", dest="rng", type=str, default="mwc64x", help="per-chain generator: mwc64x | xoshiro128++")
    a("--precision", dest="precision", type=str, default="f64",
      help="device arithmetic: f64 (the reference's Float64; default) | f32 (fast path: f32 state, f64 running sums; not for collapsed "
           "chains of the pair energies) | q16 (lattice angles, f32 arithmetic)")
    a("--uniform-bits", dest="uniform-bits", type=int, default=0,
      help="random bits of the Metropolis draw rand(): 0 = the precision's default (53 for f64, 23 for f32) | 23 | 53 (f64 only)")
    return p


def parse_args(argv=None) -> dict:
    return vars(build_parser().parse_args(argv))


def default_pargs(**overrides) -> dict:
    d = parse_args([])
    for k, v in overrides.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return d


_BIN = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
        ast.Pow: operator.pow}


def _num(node) -> float:
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return float(node.value)
    if isinstance(node, ast.Name) and node.id in ("pi", "π"):
        return math.pi
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _num(node.operand)
        return -v if isinstance(node.op, ast.USub) else v
    if isinstance(node, ast.BinOp) and type(node.op) in _BIN:
        return _BIN[type(node.op)](_num(node.left), _num(node.right))
    raise ValueError("unsupported expression")


def julia_vector(text: str) -> list[float]:
    """A Julia vector literal of arithmetic constants, '[2*pi, 1e-1]' or '[1000; 100; 10]' -> floats.
    (The reference eval()s the string; only literals and pi are understood here.)"""
    t = text.strip()
    if not (t.startswith("[") and t.endswith("]")):
        raise ValueError(text)
    body = t[1:-1].replace(";", ",").replace("^", "**").replace("π", "pi")
    return [_num(ast.parse(x.strip(), mode="eval").body) for x in body.split(",") if x.strip()]


def params_from_pargs(pargs: dict, num_chains: int, chain_id0: int, device: int) -> _lib.Params:
    resolve_seed(pargs)
    ct = {"dielectric": _lib.DIELECTRIC, "polar": _lib.POLAR}.get(pargs["chain-type"])
    if ct is None:
        raise ReferenceError_("chain-type is not understood.")                       # inc/eap_chain.jl:86
    et = {"noninteracting": _lib.NONINTERACTING, "Ising": _lib.ISING, "interacting": _lib.INTERACTING,
          "cutoff": _lib.CUTOFF}.get(pargs["energy-type"])
    if et is None:
        raise ReferenceError_("energy-type is not understood.")                      # inc/eap_chain.jl:104
    prec = {"f32": _lib.F32, "f64": _lib.F64}.get(pargs["precision"])
    if prec is None:
        raise ReferenceError_(f"precision '{pargs['precision']}' not understood")
    rng = {"mwc64x": _lib.RNG_MWC64X, "xoshiro128++": _lib.RNG_XOSHIRO128PP}.get(pargs["rng"])
    if rng is None:
        raise ReferenceError_(f"rng '{pargs['rng']}' not understood")
    x0kw = {}
    if pargs.get("x0") is not None:                                                   # inc/eap_chain.jl:61-80
        try:
            x0, dx0 = julia_vector(pargs["x0"]), julia_vector(pargs["dx0"])
        except (ValueError, SyntaxError):
            raise ReferenceError_(f"Invalid input for 'x0' and/or 'dx0', {pargs['x0']}; {pargs['dx0']}")
        if len(x0) == 2 and len(dx0) >= 2:
            x0kw = dict(use_x0=1, x0_phi=x0[0], x0_theta=x0[1], dx0_phi=dx0[0], dx0_theta=dx0[1])
        elif len(x0) == 2 * pargs["num-monomers"] and len(dx0) >= 2:
            pass        # per-monomer start: applied after creation (run(): Ensemble.restart_from_x0)
        else:
            raise ReferenceError_(f"Invalid input for 'x0' and/or 'dx0', {pargs['x0']}; {pargs['dx0']}")
    return _lib.default_params(
        E0=pargs["E0"], K1=pargs["K1"], K2=pargs["K2"], mu=pargs["mu"], kT=pargs["kT"],
        Fz=pargs["Fz"], Fx=pargs["Fx"], b=pargs["mlen"],
        phi_step=pargs["phi-step"], theta_step=pargs["theta-step"],
        adj_lb=pargs["step-adjust-lb"], adj_ub=pargs["step-adjust-ub"], adj_scale=pargs["step-adjust-scale"],
        steps_per_adjust=pargs["steps-per-adjust"], n=pargs["num-monomers"], num_chains=num_chains,
        seed=pargs["seed"], chain_id0=chain_id0, chain_type=ct, energy_type=et,
        umbrella=1 if pargs["umbrella-sampling"] else 0, precision=prec, device=device, rng=rng,
        uniform_bits=int(pargs.get("uniform-bits", 0)),
        move_set=_lib.MOVES_CLUSTER, bend_mod=pargs["bend-mod"], bend_angle=pargs["bend-angle"],
        cluster_prob=pargs["cluster-prob"], cutoff_radius=pargs["cutoff-radius"], **x0kw)


def traj_header(n: int) -> str:
    """mcmc_clustering_eap_chain.jl:253-258: phi/theta interleaved per monomer, then mux/muy/muz."""
    cols = ["step", "r1", "r2", "r3", "p1", "p2", "p3", "U"]
    for i in range(1, n + 1):
        cols += [f"phi{i}", f"theta{i}"]
    for i in range(1, n + 1):
        cols += [f"mux{i}", f"muy{i}", f"muz{i}"]
    return ",".join(cols)


def _dipoles(pargs, phi, theta) -> np.ndarray:
    """mu_i of the printed microstate (inc/dipole_response.jl:7-29), for the trajectory file only."""
    nx, ny, nz = np.cos(phi) * np.sin(theta), np.sin(phi) * np.sin(theta), np.cos(theta)
    if pargs["chain-type"] == "dielectric":
        a = (pargs["K1"] - pargs["K2"]) * pargs["E0"] * nz
        return np.stack([a * nx, a * ny, a * nz + pargs["K2"] * pargs["E0"]], axis=1)
    return pargs["mu"] * np.stack([nx, ny, nz], axis=1)


def _stage(pool, nsteps, mult, write: bool):
    """One call of the reference's mcmc(nsteps, pargs, chain) (:172-352) at kT x mult, for every case of the pool.  Every
    call rewrites the two CSV files, so only the last one's survive; earlier rungs skip the writing."""
    pool.stage(mult)
    plist = pool.plist
    pargs = plist[0]
    stepout = int(pargs["stepout"]) if write else 0
    files = None
    try:
        if write:
            files = CsvFiles([p["prefix"] for p in plist], [traj_header(p["num-monomers"]) for p in plist], ROLL_HEADER)
        start = last_update = time.time()
        step = 0
        while step < nsteps:
            seg = nsteps - step
            if stepout > 0:
                seg = min(seg, stepout - step % stepout)
            pool.advance(seg)
            step += seg
            if time.time() - last_update > pargs["update-freq"]:        # :280-284
                _log(pargs, 3, "Info", f"elapsed: {time.time() - start}")
                _log(pargs, 3, "Info", f"step:    {step} / {nsteps}")
                last_update = time.time()
            if stepout > 0 and step % stepout == 0:                     # :312-335
                for k in range(len(files) if files else 0):
                    micro = pool.microstate(k)
                    st = pool.chain0(k)
                    mus = _dipoles(plist[k], st["phi"], st["theta"])
                    angles = np.stack([st["phi"], st["theta"]], axis=1).reshape(-1)
                    s = pool.summary(k)
                    files.rows(k, jl_row([step, *micro, *angles, *mus.reshape(-1)]), jl_row([step, *s.avg, *s.extra_avg]))
        out = [pool.summary(k) for k in range(len(plist))]
        _log(pargs, 3, "Info", f"total time elapsed: {time.time() - start}")
        for k, s in enumerate(out):
            _log(plist[k], 3, "Info", f"acceptance rate: {s.acceptance_ratio}")
        return out
    finally:
        if files:
            files.close()


def run(pargs: dict):
    """The top level of mcmc_clustering_eap_chain.jl:354-387 -> (scalar_averagers, vector_averagers, ar)."""
    return run_cases([pargs])[0]


def run_cases(plist: list, write_csv: bool = True, info: dict | None = None) -> list:
    """The top level of the clustering main for every case of `plist` at once -- parsed options that differ only in their
    physics scalars, prefix and seed (one case: the command line; many: a sweep, polymer_stats_amd/sweep.py) -- as ONE
    ensemble: every rung of the ladder and the recorded run are one launch (per segment) for all of them."""
    pargs = plist[0]
    if pargs["numeric-type"] not in ("float64", "float128", "dec128", "big"):
        raise ReferenceError_(f"numeric-type '{pargs['numeric-type']}' not understood")    # :191
    try:
        ladder = julia_vector(pargs["burn-schedule"])
    except (ValueError, SyntaxError):
        raise ReferenceError_(f"burn-schedule '{pargs['burn-schedule']}' not understood")
    pool = _Pool(plist, factory=params_from_pargs)
    try:
        if pargs.get("x0") is not None:
            x0 = julia_vector(pargs["x0"])
            if len(x0) == 2 * pargs["num-monomers"] and len(x0) != 2:      # inc/eap_chain.jl:73-75
                dx0 = julia_vector(pargs["dx0"])
                for e in pool.parts:
                    e.restart_from_x0(x0, dx0[0], dx0[1])
        for mult in ladder:                                             # :366-383
            _stage(pool, int(pargs["burn-in"]), mult, write=False)
        out = _stage(pool, int(pargs["num-steps"]), 1.0, write=write_csv)   # :385-386
        for k, s in enumerate(out):
            pool.report_failures(k, s)
        if info is not None:
            info["kernel"] = pool.kernel()
    finally:
        pool.close()
    res = []
    for s in out:
        sas, vas, ar = _averagers(s)
        ex, exse = np.array(s.extra_avg), np.array(s.extra_stderr)
        res.append((sas + [Averager(ex[0], exse[0]), Averager(ex[1], exse[1])], vas, ar))
    return res


def summary_lines(sas, vas, ar, pargs) -> list[str]:
    """The twelve println lines, mcmc_clustering_eap_chain.jl:389-400."""
    nb = pargs["mlen"] * pargs["num-monomers"]
    return [
        f"<r>    =   {jl_vector(get_avg(vas[0]))}",
        f"<r/nb> =   {jl_vector(np.asarray(get_avg(vas[0])) / nb)}",
        f"<rj2>  =   {jl_vector(get_avg(vas[1]))}",
        f"<r2>   =   {jl_float(get_avg(sas[0]))}",
        f"<p>    =   {jl_vector(get_avg(vas[2]))}",
        f"<pj2>  =   {jl_vector(get_avg(vas[3]))}",
        f"<p2>   =   {jl_float(get_avg(sas[1]))}",
        f"<U>    =   {jl_float(get_avg(sas[2]))}",
        f"<U2>   =   {jl_float(get_avg(sas[3]))}",
        f"<cos2(θ)>   =   {jl_float(get_avg(sas[4]))}",
        f"<ψ>    =   {jl_float(get_avg(sas[5]))}",
        f"AR     =   {jl_float(ar)}",
    ]


def main(argv=None) -> int:
    pargs = parse_args(argv)
    if pargs["profile"]:
        raise ReferenceError_("Not currently implemented...")           # :358
    sas, vas, ar = run(pargs)
    for line in summary_lines(sas, vas, ar, pargs):
        print(line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
