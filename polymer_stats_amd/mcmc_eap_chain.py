"""Host side of the drop-in: the command line, `mcmc(nsteps, pargs)` and the three outputs of the
reference's mcmc_eap_chain.jl, with the step loop running on the GPU through libpstat (C ABI).

    python -m polymer_stats_amd.mcmc_eap_chain --chain-type dielectric -n 100 -e 1 -F 1 -N 100000 \
           --num-chains 65536 --prefix out/run1 -v 2

Same option names, short aliases, types and defaults as mcmc_eap_chain.jl:19-153; same files
(<prefix>_trajectory.csv, <prefix>_rolling.csv: :256-259,329-348) and the same ten stdout lines
(:386-395).  Options added by this implementation: --num-chains, --seed, --devices, --precision, --rng, --uniform-bits,
--burn-in, --burn-schedule.
`--num-chains C` runs C independent chains, each statistically one reference run with the given
options, and pools them; everything the reference prints is then the pooled estimate.

This is the Python twin of julia/mcmc_eap_chain.jl (no Julia toolchain exists in the build image);
both are thin: every number they print comes out of the library.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
from dataclasses import dataclass

import numpy as np

from . import _lib
from .ensemble import Ensemble, summary_from_reduction
from .julia_fmt import jl_float, jl_row, jl_vector

TRAJ_HEADER = "step,r1,r2,r3,p1,p2,p3,U"
ROLL_HEADER = "step,r1,r2,r3,r1sq,r2sq,r3sq,rsq,p1,p2,p3,p1sq,p2sq,p3sq,psq,U,Usq"


class ReferenceError_(RuntimeError):
    """Raised where the reference calls error(...) -- same message text."""


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="mcmc_eap_chain", add_help=True, allow_abbrev=False)
    a = p.add_argument
    # --- the reference's table, mcmc_eap_chain.jl:19-153 (dest = ArgParse.jl's dict key)
    a("--E0", "-e", dest="E0", type=float, default=0.0, help="magnitude of electric field")
    a("--chain-type", "-T", dest="chain-type", type=str, default="dielectric", help="chain type (dielectric|polar)")
    a("--K1", "-J", dest="K1", type=float, default=1.0, help="dipole susceptibility along the monomer axis (dielectric chain)")
    a("--K2", "-K", dest="K2", type=float, default=0.0, help="dipole susceptibility orthogonal to the monomer axis (dielectric chain)")
    a("--mu", "-m", dest="mu", type=float, default=1e-2, help="dipole magnitude (electret chain)")
    a("--energy-type", "-u", dest="energy-type", type=str, default="noninteracting", help="energy type (noninteracting|interacting)")
    a("--kT", "-k", dest="kT", type=float, default=1.0, help="dimensionless temperature")
    a("--ensemble-type", "-E", dest="ensemble-type", type=str, default="force", help="ensemble type (force|end-to-end)")
    a("--Fz", "-F", dest="Fz", type=float, default=0.0, help="force in the z-direction (direction of E-field; force ensemble)")
    a("--Fx", "-G", dest="Fx", type=float, default=0.0, help="force in the x-direction (force ensemble)")
    a("--rz", "-z", dest="rz", type=float, default=0.0, help="end-to-end vector in the z-direction (etoe ensemble)")
    a("--rx", "-x", dest="rx", type=float, default=0.0, help="end-to-end vector in the x-direction (etoe ensemble)")
    a("--mlen", "-b", dest="mlen", type=float, default=1.0, help="monomer length")
    a("--num-monomers", "-n", dest="num-monomers", type=int, default=100, help="number of monomers")
    a("--num-steps", "-N", dest="num-steps", type=int, default=int(1e5), help="number of steps")
    a("--num-inits", "-M", dest="num-inits", type=int, default=1, help="number of random initializations")
    a("--force-init", "-I", dest="force-init", action="store_true", help="force each random initialization (false to use metro.)")
    a("--phi-step", "-p", dest="phi-step", type=float, default=3 * math.pi / 8, help="maximum phi step length")
    a("--do-flips", dest="do-flips", action="store_true", help="trial moves with flipping monomers")
    a("--theta-step", "-q", dest="theta-step", type=float, default=3 * math.pi / 16, help="maximum theta step length")
    a("--chain-frac-step", "-f", dest="chain-frac-step", type=float, default=0.15, help="fraction of monomers to step (end-to-end ensemble)")
    a("--step-adjust-lb", "-L", dest="step-adjust-lb", type=float, default=0.15, help="adjust step sizes if acc. ratio below this threshold")
    a("--step-adjust-ub", "-U", dest="step-adjust-ub", type=float, default=0.55, help="adjust step sizes if acc. ratio above this threshold")
    a("--step-adjust-scale", "-A", dest="step-adjust-scale", type=float, default=1.1, help="scale factor for adjusting step sizes (> 1.0)")
    a("--steps-per-adjust", "-S", dest="steps-per-adjust", type=int, default=2500, help="steps between step size adjustments")
    a("--acc", "-a", dest="acc", type=str, default="metropolis", help="acceptance function (metropolis|kawasaki)")
    a("--umbrella-sampling", "-B", dest="umbrella-sampling", action="store_true", help="use umbrella sampling (w/ electrostatic weight function)")
    a("--update-freq", dest="update-freq", type=float, default=15.0, help="update frequency (seconds)")
    a("--verbose", "-v", dest="verbose", type=int, default=3, help="verbosity level: 0-nothing, 1-errors, 2-warnings, 3-info")
    a("--prefix", "-P", dest="prefix", type=str, default="eap-mcmc", help="prefix for output files")
    a("--postfix", "-Q", dest="postfix", type=str, default="", help="postfix for output files")
    a("--stepout", "-s", dest="stepout", type=int, default=500, help="steps between storing microstates")
    a("--numeric-type", dest="numeric-type", type=str, default="float64", help="numerical data type for averaging (float64|float128|dec128|big)")
    a("--profile", "-Z", dest="profile", action="store_true", help="profile the program")
    # --- ours
    a("--num-chains", dest="num-chains", type=int, default=4096, help="independent chains run at once on the GPU(s) and pooled")
    a("--seed", dest="seed", type=int, default=None,
      help="seed of the per-chain generators; default: fresh OS entropy per run, like the reference's unseeded RNG "
           "(the seed drawn is echoed on stderr at --verbose >= 2)")
    a("--devices", dest="devices", type=str, default="0", help="comma-separated HIP device ordinals; chains are sharded over them")
    a("--burn-in", dest="burn-in", type=int, default=0,
      help="steps discarded before averaging, per rung of --burn-schedule (0 = the reference's behaviour: record from step 1)")
    a("--burn-schedule", dest="burn-schedule", type=str, default="[1]",
      help="kT multipliers of the burn-in ladder, e.g. '[1000; 100; 10; 2; 1]' (mcmc_clustering_eap_chain.jl:138-141)")
    a("--rng", dest="rng", type=str, default="mwc64x", help="per-chain generator: mwc64x | xoshiro128++")
    a("--precision", dest="precision", type=str, default="f64",
      help="device arithmetic: f64 (the reference's Float64; default) | f32 (fast path: f32 state, f64 running sums; not for collapsed "
           "chains of the pair energies) | q16 (lattice angles, f32 arithmetic)")
    a("--uniform-bits", dest="uniform-bits", type=int, default=0,
      help="random bits of the Metropolis draw rand() (mcmc_eap_chain.jl:287): 0 = the precision's default (53 for f64, like "
           "Julia's Float64 rand(); 23 for f32 / q16) | 23 | 53 (f64 only)")
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


def fresh_seed() -> int:
    """A new 63-bit seed per call: OS entropy, mixed with the clock and the pid in case the pool is a stub.
    The reference never seeds Julia's RNG, so repeated identical command lines give independent samples
    (run/interacting-compare-with-clustering_2021-09-28.jl:26-27 launches each case 25 times and takes the
    scatter as its error bar); the drop-in must do the same unless --seed is given."""
    v = int.from_bytes(os.urandom(8), "little") ^ time.time_ns() ^ (os.getpid() << 40)
    return v & 0x7FFFFFFFFFFFFFFF


WIDE_TYPES = {"float128": "80-bit extended (numpy.longdouble)", "dec128": "80-bit extended (numpy.longdouble)",
              "big": "80-bit extended (numpy.longdouble)"}


def _log(pargs, level: int, tag: str, msg: str):
    # Logging to stderr gated by --verbose (mcmc_eap_chain.jl:157-165): 3 info, 2 warn, 1 error
    if pargs["verbose"] >= level:
        print(f"[ {tag}: {msg}", file=sys.stderr)


def resolve_seed(pargs: dict) -> int:
    """--seed as given, or (once per pargs) a fresh one, echoed on stderr at --verbose >= 2 so the run can be repeated."""
    if pargs.get("seed") is None:
        pargs["seed"] = fresh_seed()
        _log(pargs, 2, "Info", f"seed: {pargs['seed']} (fresh entropy; pass --seed {pargs['seed']} to reproduce this run)")
    return pargs["seed"]


def params_from_pargs(pargs: dict, num_chains: int, chain_id0: int, device: int) -> _lib.Params:
    """pargs -> pstat_params, with the reference's error() branches (inc/eap_chain.jl:81-105)."""
    resolve_seed(pargs)
    ct = {"dielectric": _lib.DIELECTRIC, "polar": _lib.POLAR}.get(pargs["chain-type"])
    if ct is None:
        raise ReferenceError_("chain-type is not understood.")
    et = {"noninteracting": _lib.NONINTERACTING, "interacting": _lib.INTERACTING,
          "Ising": _lib.ISING}.get(pargs["energy-type"])
    if et is None:
        raise ReferenceError_("energy-type is not understood.")
    prec = {"f32": _lib.F32, "f64": _lib.F64, "q16": _lib.Q16}.get(pargs["precision"])
    if prec is None:
        raise ReferenceError_(f"precision '{pargs['precision']}' not understood")
    rng = {"mwc64x": _lib.RNG_MWC64X, "xoshiro128++": _lib.RNG_XOSHIRO128PP}.get(pargs["rng"])
    if rng is None:
        raise ReferenceError_(f"rng '{pargs['rng']}' not understood")
    return _lib.default_params(
        E0=pargs["E0"], K1=pargs["K1"], K2=pargs["K2"], mu=pargs["mu"], kT=pargs["kT"],
        Fz=pargs["Fz"], Fx=pargs["Fx"], b=pargs["mlen"],
        phi_step=pargs["phi-step"], theta_step=pargs["theta-step"],
        adj_lb=pargs["step-adjust-lb"], adj_ub=pargs["step-adjust-ub"], adj_scale=pargs["step-adjust-scale"],
        steps_per_adjust=pargs["steps-per-adjust"], n=pargs["num-monomers"], num_chains=num_chains,
        seed=pargs["seed"], chain_id0=chain_id0, chain_type=ct, energy_type=et,
        do_flips=1 if pargs["do-flips"] else 0, umbrella=1 if pargs["umbrella-sampling"] else 0,
        precision=prec, device=device, rng=rng, uniform_bits=int(pargs.get("uniform-bits", 0)))


@dataclass
class Averager:
    """What the caller of mcmc() gets back in place of a StandardAverager (inc/average.jl:8-48)."""
    value: object
    stderr: object

    def get_avg(self):
        return self.value


def get_avg(a: Averager):
    return a.get_avg()


class _Pool:
    """The cases of one ensemble -- one for the command line, many for a sweep (polymer_stats_amd/sweep.py; they differ only
    in their physics scalars, pstat_create) --, every case's chains sharded over one or more devices in this process;
    reductions merged on the host (every entry of the reduction vector is additive)."""

    def __init__(self, pargs, factory=None):
        self.plist = plist = pargs if isinstance(pargs, list) else [pargs]
        factory = factory or params_from_pargs
        for p in plist:
            resolve_seed(p)      # before the shards are made: every device gets the same seed, disjoint chain ids
        p0 = plist[0]
        self.numeric_type = p0.get("numeric-type", "float64")
        if self.numeric_type != "float64":
            # mcmc_eap_chain.jl:186-197 switches the averagers' accumulation type.  Here the per-chain sums are
            # Float64 on the device (the reference's default); what the option changes is the merge over chains.
            _log(p0, 2, "Warning", f"--numeric-type {self.numeric_type}: per-chain sums are Float64 on the device; the "
                                   f"merge over chains is carried out in {WIDE_TYPES[self.numeric_type]}")
        devices = [int(d) for d in str(p0["devices"]).split(",") if d != ""]
        total = int(p0["num-chains"])
        if total < 1:
            raise ReferenceError_("num-chains must be >= 1")
        devices = devices[:total] or [0]
        base, extra = divmod(total, len(devices))
        self.parts, self.counts = [], []
        first = 0
        for i, dev in enumerate(devices):
            cnt = base + (1 if i < extra else 0)
            self.parts.append(Ensemble([factory(p, cnt, first, dev) for p in plist]))
            self.counts.append(cnt)
            first += cnt
        self.steps = 0

    def advance(self, n):
        for e in self.parts:
            e.advance(n)            # asynchronous: the devices run concurrently
        self.steps += n

    def reinit(self, force):
        for e in self.parts:
            e.reinit(force)

    def burn_in(self, nsteps, multipliers):
        """Run the temperature ladder (every case's own kT times the rung's multiplier) without keeping anything it records."""
        for mult in multipliers:
            for e in self.parts:
                e.scale_kT(mult)
            for e in self.parts:
                e.advance(nsteps)
        for e in self.parts:
            e.scale_kT(1.0)
            e.reset_averages()
        self.steps = 0

    def stage(self, mult):
        """Start of a fresh mcmc(nsteps, pargs, chain) call of the clustering main: new temperature (kT x mult),
        default step sizes, empty acceptor cache and averagers (mcmc_clustering_eap_chain.jl:172-181)."""
        for e in self.parts:
            e.scale_kT(mult)
            e.reset_sampler()
            e.reset_averages()
        self.steps = 0

    def chain0(self, k=0):
        return self.parts[0].chain_state(k * self.counts[0])      # the first chain of case k

    def microstate(self, k=0):
        return self.parts[0].microstate(k * self.counts[0])

    def summary(self, k=0):
        red = np.zeros(_lib.NRED)
        for e in self.parts:
            red += e.reduce_host(k)
        s = summary_from_reduction(red, self.steps)
        if self.numeric_type != "float64":
            # --numeric-type: pooled mean and across-chain standard error re-done in the wide type from the per-chain
            # means (the same quantities the device reduction folds in Float64)
            m = np.concatenate([e.chain_means(k) for e in self.parts], axis=1).astype(np.longdouble)
            C = m.shape[1]
            mean = m.sum(axis=1) / C
            se = np.sqrt(((m - mean[:, None]) ** 2).sum(axis=1) / (C - 1) / C) if C > 1 else np.zeros_like(mean)
            for q in range(_lib.NOBS):
                s.avg[q], s.stderr[q] = float(mean[q]), float(se[q])
            s.acceptance_ratio, s.ar_stderr = float(mean[16]), float(se[16])
            for q in range(2):
                s.extra_avg[q], s.extra_stderr[q] = float(mean[17 + q]), float(se[17 + q])
        return s

    def report_failures(self, k, s):
        """stderr only (stdout stays the reference's lines): what the reference hides -- proposals it rejected because
        their energy was NaN/Inf, and chains sitting in a 1/r^3 singularity (no excluded volume, inc/eap_chain.jl:200-207)."""
        pargs = self.plist[k]
        who = f"{os.path.basename(pargs['prefix'])}: " if len(self.plist) > 1 else ""
        if s.nan_rejects:
            _log(pargs, 2, "Warning", f"{who}{s.nan_rejects} proposals had a non-finite energy and were rejected "
                                      f"({s.nan_rejects / max(1.0, s.attempted_updates):.3g} of all attempts)")
        if s.chains_collapsed:
            _log(pargs, 2, "Warning", f"{who}{s.chains_collapsed} of {s.num_chains} chains have collapsed "
                                      f"(|U| a thousand times beyond field + force + thermal energy: monomers on top of each other)")

    def kernel(self) -> str:
        return self.parts[0].launch_info().kernel.decode()

    def close(self):
        for e in self.parts:
            e.close()


def _averagers(s):
    avg, se = np.array(s.avg), np.array(s.stderr)
    sas = [Averager(avg[6], se[6]), Averager(avg[13], se[13]), Averager(avg[14], se[14]), Averager(avg[15], se[15])]
    vas = [Averager(avg[0:3], se[0:3]), Averager(avg[3:6], se[3:6]), Averager(avg[7:10], se[7:10]),
           Averager(avg[10:13], se[10:13])]
    return sas, vas, s.acceptance_ratio


class CsvFiles:
    """The `<prefix>_trajectory.csv` / `<prefix>_rolling.csv` pair of every case of an ensemble.  The reference holds its two
    files open for the whole run (mcmc_eap_chain.jl:256-258,372-373); a batched sweep has thousands of cases, so the handles
    stay open only while two per case fit the process's descriptor limit with room to spare -- beyond that every row is
    appended by open/write/close (same bytes on disk)."""

    def __init__(self, prefixes, traj_headers, roll_header):
        try:
            import resource
            limit = resource.getrlimit(resource.RLIMIT_NOFILE)[0]
        except Exception:
            limit = 256
        self.paths = [(f"{p}_trajectory.csv", f"{p}_rolling.csv") for p in prefixes]
        self.keep_open = 2 * len(self.paths) <= max(0, limit - 64) // 2
        self.handles = []
        for (tp, rp), th in zip(self.paths, traj_headers):
            ft, fr = open(tp, "w"), open(rp, "w")
            ft.write(th + "\n")
            fr.write(roll_header + "\n")
            if self.keep_open:
                self.handles.append((ft, fr))
            else:
                ft.close()
                fr.close()

    def __len__(self):
        return len(self.paths)

    def rows(self, k, traj_row, roll_row):
        if self.keep_open:
            ft, fr = self.handles[k]
            ft.write(traj_row + "\n")
            fr.write(roll_row + "\n")
        else:
            for path, row in zip(self.paths[k], (traj_row, roll_row)):
                with open(path, "a") as f:
                    f.write(row + "\n")

    def close(self):
        for ft, fr in self.handles:
            ft.close()
            fr.close()
        self.handles = []


def mcmc(nsteps: int, pargs: dict):
    """mcmc(nsteps, pargs) of mcmc_eap_chain.jl:171-376 -> (scalar_averagers, vector_averagers, ar)."""
    return mcmc_cases(nsteps, [pargs])[0]


def mcmc_cases(nsteps: int, plist: list, write_csv: bool = True, info: dict | None = None) -> list:
    """mcmc(nsteps, pargs) for every case of `plist` at once -- parsed options that differ only in their physics scalars,
    prefix and seed (one case: the command line; many: a sweep, polymer_stats_amd/sweep.py) -- as ONE ensemble: one launch
    per segment for all of them.  `write_csv=False` skips the two CSV files of every case (then one launch per init)."""
    pargs = plist[0]
    if pargs["acc"] != "metropolis":
        raise ReferenceError_(f"'{pargs['acc']}' acceptance criteria has not yet been implemented.")  # :184
    if pargs["numeric-type"] not in ("float64", "float128", "dec128", "big"):
        raise ReferenceError_(f"numeric-type '{pargs['numeric-type']}' not understood")                # :195
    if pargs["ensemble-type"] != "force":
        raise ReferenceError_("'end-to-end' ensemble is an experimental option of the reference; "
                              "it has no device implementation")
    pool = _Pool(plist)
    stepout = int(pargs["stepout"]) if write_csv else 0
    files = None
    try:
        if pargs["burn-in"] > 0:
            ladder = [float(x) for x in pargs["burn-schedule"].strip("[] ").replace(",", ";").split(";") if x.strip()]
            pool.burn_in(int(pargs["burn-in"]), ladder or [1.0])
        if write_csv:                                # :256-259
            files = CsvFiles([p["prefix"] for p in plist], [TRAJ_HEADER] * len(plist), ROLL_HEADER)
        start = last_update = time.time()
        for init in range(1, pargs["num-inits"] + 1):           # :266
            step = 0
            while step < nsteps:                                # :276 (in segments)
                seg = nsteps - step
                if stepout > 0:
                    seg = min(seg, stepout - step % stepout)
                pool.advance(seg)
                step += seg
                if time.time() - last_update > pargs["update-freq"]:   # :294-299
                    _log(pargs, 3, "Info", f"elapsed: {time.time() - start}")
                    _log(pargs, 3, "Info", f"init:    {init} / {pargs['num-inits']}")
                    _log(pargs, 3, "Info", f"step:    {step} / {nsteps}")
                    last_update = time.time()
                if stepout > 0 and step % stepout == 0:          # :329-348
                    for k in range(len(files) if files else 0):
                        micro = pool.microstate(k)
                        s = pool.summary(k)
                        files.rows(k, jl_row([step, *micro]), jl_row([step, *s.avg]))
            if init < pargs["num-inits"]:                        # :352-361
                pool.reinit(bool(pargs["force-init"]))
        out = [pool.summary(k) for k in range(len(plist))]
        _log(pargs, 3, "Info", f"total time elapsed: {time.time() - start}")
        for k, s in enumerate(out):
            _log(plist[k], 3, "Info", f"acceptance rate: {s.acceptance_ratio}")
            pool.report_failures(k, s)
        if info is not None:
            info["kernel"] = pool.kernel()
    finally:
        if files:
            files.close()
        pool.close()
    return [_averagers(s) for s in out]


def summary_lines(sas, vas, ar, pargs) -> list[str]:
    """The ten println lines, mcmc_eap_chain.jl:386-395."""
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
        f"AR     =   {jl_float(ar)}",
    ]


def main(argv=None) -> int:
    pargs = parse_args(argv)
    if pargs["ensemble-type"] == "end-to-end":
        _log(pargs, 2, "Warning", "'end-to-end' ensemble is an experimental option; it has not been validated.")
    if pargs["profile"]:
        raise ReferenceError_("not implemented for the HPC env")     # :379
    sas, vas, ar = mcmc(pargs["num-steps"], pargs)
    for line in summary_lines(sas, vas, ar, pargs):
        print(line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
