#!/usr/bin/env python3
"""bench.py -- throughput of the fixed-force MCMC hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (starts its own N rank processes when N > 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (also fine)

Workload (BASELINE.json configs[1]): non-interacting dielectric chain, n = 100, E0 = 1, K1 = 1, K2 = 0, kT = 1,
b = 1, one point of the Fz sweep per bench step, 65 536 chains per GPU, 1e5 MC steps per chain (6.55e9 attempted
monomer updates per GPU per step).  Weak scaling: every rank runs its own 65 536 chains (global chain ids
rank*65536 ...), no data-path collective; the only exchange is one RCCL all-reduce of the 41-double reduction
vector per step.

A bench "step" = advance every chain of one Fz point by --mc-steps Monte-Carlo steps (ONE launch of the sweep
kernel) + the on-device reduction + the all-reduce.  Chain states are created (on the device) before the timed
region, so inputs are resident in HBM when it starts.

The headline (`value`, `dtype`, `roofline`, `valu`, `lds`) is measured on the f64 kernel -- Float64 is the
reference's arithmetic (inc/types.jl, inc/eap_chain.jl:12-36) and that kernel reproduces the CPU oracle bit for
bit.  The f32 kernel (f32 state and transcendentals, f64 running sums; the survey-sanctioned fast path) is timed
in the same run with the same K/W and reported in the sibling object `fast_path`.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes (32 B per attempted update in f64: one
(theta, phi) pair read and written; 16 B in f32 -- SURVEY.md 8(d)) x updates per launch / mean launch time
measured with HIP events on the launch stream.  The f64 kernel at n = 100 keeps 39 of the 100 monomers' cells in LDS
and the rest in memory (L2 / Infinity Cache), so its `traffic` is real fabric traffic -- about 1.7x
the algorithmic bytes, the price of 16-byte random accesses at 64-byte granularity; the f32 kernel keeps all state in
LDS, its figure is an EQUIVALENT rate (`equivalent: true`) and its traffic is ~1 % of it.  Both kernels are bound by
VALU issue (`valu`); `lds` is the third fraction SURVEY 8(d) asks for.  `traffic` / `valu.ops_per_update` come from
the PMC passes in profiles/pmc_traffic.json and are used only if that file was collected on the kernel sources being
benchmarked (sha256 stamp), else null / estimate.

Beside the headline: `configs` (one measured entry per BASELINE configuration) and `phase_scan` (the clustering main on the
ensemble shape the reference launches its phase scans with: 2 730 single-chain cases; microseconds per step).
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FZ_SWEEP = [round(0.05 * i, 2) for i in range(21)] + [1.5 + 0.5 * i for i in range(8)]  # run/noninteracting-compare-*.jl:21
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s spec
VALU_LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes/clk x Hz = 7.86e13 f32 lane-ops/s (f64: half of it)
NUM_SIMDS, CLOCK_HZ = 256 * 4, 2.4e9
LDS_PEAK_GBS = 256 * 256 * 2.4              # CUs x 256 B/clk (MI355X_MICROARCH.md, LDS) x GHz = 157 TB/s
VALU_OPS_ESTIMATE = {"f32": 85, "f64": 257, "q16": 85}   # last PMC counts (profiles/r02); replaced by the stamped record when valid
VALU_F64_OPS_ESTIMATE = {"f64": 128.0}                   # of which f64 ops; likewise replaced
STATE_BYTES = {"f32": 16, "f64": 32, "q16": 8}              # one state cell read + written per attempted update
DTYPE = {"f32": "f32 state+transcendentals, f64 running sums", "f64": "f64",
         "q16": "u16 lattice angles, f32 transcendentals, f64 running sums"}


def kernel_source_hash() -> str:
    """sha256 over the kernel sources and their build recipe: the stamp of profiles/pmc_traffic.json."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "polymer_stats_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "polymer_stats_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "polymer_stats_amd", "csrc", "Makefile"), os.path.join(ROOT, "include", "pstat.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            if parse:
                q, per = parse(open(path).read())
            else:
                q = open(path).read().strip()
                per = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if q not in ("max", "-1"):
                n = min(n, max(1, int(float(q) / float(per))))
            break
        except Exception:
            continue
    return max(1, n)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline_oracle():
    """The CPU restatement of the reference (oracle/): loaded HERE and nowhere else in this file -- for the `cpu_baseline`
    objects of the bench line (the headline's and the configs'), timed on the host cores beside the GPU numbers.  The
    product path (polymer_stats_amd/, libpstat.so) never touches it (tests/test_abi.py)."""
    from oracle import binding as ob
    return ob


def cpu_baseline(n, mc_steps, target_seconds=12.0):
    """Times the CPU restatement of the reference algorithm (oracle, faithful mode: deep copy +
    full recompute per step, fp64) on this host: one single-threaded chain per worker thread over all
    cores, like the reference's pmap farm.  Bounded sample of the same workload."""
    ob = cpu_baseline_oracle()
    cores = host_cores()
    P = ob.make_params(n=n, E0=1.0, K1=1.0, K2=0.0, kT=1.0, Fz=1.0, b=1.0, num_steps=mc_steps, seed=1, stepout=0)
    t0 = time.perf_counter()
    ob.run(P, chain_id=0, mode="faithful")
    one = max(time.perf_counter() - t0, 1e-4)
    per_thread = max(1, min(1024, int(target_seconds / one)))
    nchains = cores * per_thread
    log(f"cpu baseline: {cores} threads x {per_thread} chains, one chain takes {one:.3f} s")
    t0 = time.perf_counter()
    sums, norm, _ = ob.run_many(P, id0=1, nchains=nchains, nthreads=cores, mode="faithful")
    wall = time.perf_counter() - t0
    m = sums / norm[:, None]
    mean, se = m.mean(axis=0), m.std(axis=0, ddof=1) / max(1.0, float(nchains)) ** 0.5
    return {"value": nchains * mc_steps / wall, "unit": "MC monomer-updates/s", "cores": cores,
            "kind": "port",
            "sample": f"{nchains} chains x {mc_steps} steps, n={n}, Fz=1, oracle faithful mode (deep copy + "
                      f"full recompute per step, fp64, one thread per chain), {wall:.1f} s wall"}, mean, se


def cpu_sample(ob, okw, mode, cores, budget_s):
    """A bounded CPU sample of one configuration in one oracle mode: one single-threaded chain per host thread (the reference's
    pmap farm), the number of steps sized from a calibration run so that the sample takes about `budget_s` seconds."""
    cal = 400 if okw.get("energy_type") == 1 else 20000
    P = ob.make_params(num_steps=cal, seed=11, stepout=0, **okw)
    t0 = time.perf_counter()
    ob.run(P, chain_id=0, mode=mode)
    per_step = max(time.perf_counter() - t0, 1e-5) / cal
    steps = int(max(cal, min(100_000_000, budget_s / per_step)))
    P = ob.make_params(num_steps=steps, seed=11, stepout=0, **okw)
    t0 = time.perf_counter()
    ob.run_many(P, id0=1, nchains=cores, nthreads=cores, mode=mode)
    wall = time.perf_counter() - t0
    return {"value": cores * steps / wall, "sample": f"{cores} chains x {steps} steps in {wall:.2f} s"}


def measure_configs(ps, torch, stream, pmc, headline, head_cpu, skip_cpu, budget_s):
    """One measured line per BASELINE configuration (tools/configs.py), each in f64 -- the reference's arithmetic -- with its
    roofline fraction (SURVEY 8(d)'s algorithmic work per update), the VALU figures of the stamped PMC record of ITS kernel
    (profiles/pmc_traffic.json; null when the kernel sources have changed since), and both CPU baselines beside it: the
    oracle's faithful mode (the reference's literal algorithm: deep copy + full recompute per step) and its fast mode
    (O(1) energy differences; the best-effort CPU of BASELINE.md section 2).  Not part of the headline's timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from configs import config_list, F64_VECTOR_PEAK_TFLOPS
    out = []
    cores = host_cores()
    ob = None if skip_cpu else cpu_baseline_oracle()
    for cfg in config_list(ps):
        t_cfg = time.perf_counter()
        entry = {"id": cfg["id"], "workload": cfg["workload"], "dtype": "f64"}
        if cfg.get("headline"):
            rate, kernel_ms, kernel = headline["rate"], headline["kernel_ms"], headline["kernel"]
            entry["check"] = headline["check"]
        else:
            with torch.cuda.stream(stream):
                with ps.Ensemble(cfg["cases"], stream=stream.cuda_stream) as e:
                    e.advance(min(cfg["mc_steps"], 4000))          # warm-up: first touch, clocks, the adaptation's regime
                    torch.cuda.synchronize()
                    ms = []
                    for _ in range(2):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(stream)
                        e.advance(cfg["mc_steps"])
                        b.record(stream)
                        torch.cuda.synchronize()
                        ms.append(a.elapsed_time(b))
                    e.sync()
                    info, s = e.launch_info(), e.summary()
                    chains = e.num_chains * e.ncases
            kernel_ms, kernel = sum(ms) / len(ms), info.kernel.decode()
            rate = chains * cfg["mc_steps"] / (kernel_ms * 1e-3)
            entry.update({"chains": chains, "n": int(cfg["cases"][0].n), "mc_steps": cfg["mc_steps"],
                          "check": {"r3": s.avg[2], "p3": s.avg[9], "U": s.avg[14], "AR": s.acceptance_ratio,
                                    "nan_rejects": int(s.nan_rejects), "chains_collapsed": int(s.chains_collapsed)}})
        entry.update({"kernel": kernel, "value": rate, "unit": "MC monomer-updates/s", "kernel_ms": kernel_ms})
        if "flop_per_update" in cfg:
            ach = cfg["flop_per_update"] * rate / 1e12
            entry["roofline"] = {"bound": "valu-f64", "achieved": ach, "peak": F64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": ach / F64_VECTOR_PEAK_TFLOPS,
                                 "algorithmic": "%d flop per update = n(n-1)/2 pair terms x 36 flop (SURVEY 8(d); the reference's full "
                                                "recomputation, inc/eap_chain.jl:196-211) against the f64 vector peak" % cfg["flop_per_update"]}
        else:
            ach = cfg["bytes_per_update"] * rate / 1e9
            entry["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                 "algorithmic": "%d B per update: one f64 (theta, phi) cell read and written (SURVEY 8(d))" % cfg["bytes_per_update"]}
        rec = (pmc or {}).get("records", {}).get(cfg["pmc_record"])
        if rec:
            v, f = rec["valu_instructions_per_update_per_lane"], rec.get("valu_f64_instructions_per_update_per_lane") or 0.0
            entry["valu"] = {"frac": rate * (f / (0.5 * VALU_LANE_OPS_PEAK) + max(0.0, v - f) / VALU_LANE_OPS_PEAK),
                             "ops_per_update": v, "f64_ops_per_update": f,
                             "wave_cycles_with_an_instruction": rec.get("wave_cycles_with_an_instruction"),
                             "wave_cycles_waiting": rec.get("wave_cycles_waiting"),
                             "traffic_bytes_per_update": (rec["hbm_bytes_per_launch"] / rec["updates_per_launch"])
                             if rec.get("hbm_bytes_per_launch") and rec.get("updates_per_launch") else None,
                             "source": "profiles/pmc_traffic.json record '%s' (round %s, stamp = sha256 of the kernel sources)"
                                       % (cfg["pmc_record"], rec.get("round"))}
        else:
            entry["valu"] = None
        if ob is not None:
            # (`options`: what the CPU sample ran -- for the C5 grid one representative point, the cost per update does not
            # depend on the physics scalars)
            cb = {"cores": cores, "kind": "port", "unit": "MC monomer-updates/s", "options": cfg["oracle"]}
            if cfg.get("headline") and head_cpu is not None:
                cb["faithful"] = {"value": head_cpu["value"], "sample": head_cpu["sample"]}
            else:
                cb["faithful"] = cpu_sample(ob, cfg["oracle"], "faithful", cores, budget_s)
            cb["fast"] = cpu_sample(ob, cfg["oracle"], "fast", cores, budget_s)
            cb["gpu_over_faithful"] = rate / cb["faithful"]["value"]
            cb["gpu_over_fast"] = rate / cb["fast"]["value"]
            entry["cpu_baseline"] = cb
        log(f"config {cfg['id']}: {rate:.3e} updates/s ({kernel}); {time.perf_counter() - t_cfg:.1f} s")
        out.append(entry)
    return out


def measure_phase_scan(ps, steps=20000):
    """The reference's own phase scan as an ensemble (run/K1_E0-kT-phase.jl:17-45: 26 x 21 (E0, kT) points x 5 runs = 2 730
    single-chain cases, clustering main, Ising, n = 100): microseconds per MC step of the whole ensemble, best of three timed
    launches of `steps` steps after a warm-up.  Not the headline metric: the "next" row f3 + f4 of SURVEY.md section 8 on the
    ensemble shape the reference launches it with."""
    cases = [ps.default_params(n=100, E0=0.2 * (i // 21 % 26), K1=1.0, K2=0.0, kT=10 ** (-2 + 0.2 * (i % 21)), num_chains=1,
                               precision=ps.F64, seed=1000 + i, move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, energy_type=ps.ISING)
             for i in range(2730)]
    with ps.Ensemble(cases) as e:
        e.advance(max(500, steps // 10))
        e.sync()
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            e.advance(steps)
            e.sync()
            best = min(best, time.perf_counter() - t0)
        info = e.launch_info()
        return {"workload": "run/K1_E0-kT-phase.jl as the reference launches it: 546 (E0, kT) points x 5 runs = 2 730 single-chain cases, "
                            "mcmc_clustering_eap_chain.jl step (single move + cluster_flip!), Ising, n = 100, f64",
                "chains": len(cases), "mc_steps": steps, "us_per_step": round(best / steps * 1e6, 3),
                "value": len(cases) * steps / best, "unit": "proposals/s", "kernel": info.kernel.decode(),
                "workgroups": int(info.blocks), "workgroups_per_cu": int(info.blocks_per_cu), "timing": "host clock around advance + sync"}


def parity_vs_cpu(ps, prec, n, chains, mc_steps, device, cpu_mean, cpu_se):
    """The north star's acceptance line: <r_z>, <p_z>, <U> of the device path against the CPU restatement run
    under the same options and protocol (same Fz = 1 point, same number of steps, no burn-in on either
    side), as z = (gpu - cpu) / sqrt(se_gpu^2 + se_cpu^2).  One extra untimed launch."""
    p = ps.default_params(n=n, E0=1.0, K1=1.0, K2=0.0, kT=1.0, b=1.0, Fz=1.0, num_chains=chains, seed=20260499,
                          precision=prec, device=device)
    with ps.Ensemble(p) as e:
        e.advance(mc_steps)
        s = e.summary()
    out = {}
    for name, k in (("r3", 2), ("p3", 9), ("U", 14)):
        z = (s.avg[k] - cpu_mean[k]) / ((s.stderr[k] ** 2 + cpu_se[k] ** 2) ** 0.5 + 1e-300)
        out[name] = {"gpu": s.avg[k], "cpu": float(cpu_mean[k]), "z": float(z)}
    return out


def spawn_ranks(args) -> int:
    """--gpus N > 1 from a bare shell (WORLD_SIZE unset): start N fresh rank processes, one per GPU, BEFORE this
    process makes any GPU call (it never does: torch is not even imported here), relay rank 0's JSON line, and
    fail -- stopping the surviving ranks -- as soon as any rank fails (tools/rank_spawn.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rank_spawn import spawn_ranks as spawn
    rc, out0 = spawn(os.path.abspath(__file__), sys.argv[1:], args.gpus)
    if rc:
        return rc
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU")
    ap.add_argument("--mc-steps", type=int, default=100000, help="MC steps per chain per bench step")
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--precision", choices=["f64", "f32", "q16"], default="f64",
                    help="arithmetic of the headline line (default f64 = the reference's Float64)")
    ap.add_argument("--no-fast-path", action="store_true", help="skip the sibling f32 measurement")
    ap.add_argument("--rng", choices=["mwc64x", "xoshiro128++"], default="mwc64x",
                    help="per-chain generator (default MWC64X; xoshiro128++ is the north star's named one, measured slower here)")
    ap.add_argument("--no-rng-named", action="store_true", help="skip the sibling f64 measurement under xoshiro128++")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` array (one measured line per BASELINE configuration)")
    ap.add_argument("--config-cpu-seconds", type=float, default=2.5, help="CPU seconds per configuration and oracle mode")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU seconds of the headline's cpu_baseline sample")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even for one rank")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend; gloo (via host memory) only to rehearse N>1 on a box with fewer GPUs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    # Libraries (RCCL's version banner, HIP warnings) may write to stdout; the contract is ONE JSON
    # line there.  Park the real stdout and send everything else to stderr until the line is printed.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import polymer_stats_amd as ps

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libpstat has no CPU path")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    local_rank = local_rank % ndev          # (gloo rehearsal: ranks may share a GPU)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def all_reduce(tensor, op=None):
        """Sum (or op) over ranks: RCCL on the device tensor, or gloo through host memory."""
        kw = {} if op is None else {"op": op}
        if args.backend == "nccl":
            dist.all_reduce(tensor, **kw)
        else:
            host = tensor.cpu()
            dist.all_reduce(host, **kw)
            tensor.copy_(host)

    PREC = {"f32": ps.F32, "f64": ps.F64, "q16": ps.Q16}
    stream = torch.cuda.Stream()
    nstep_total = args.steps + args.warmup
    red = torch.zeros(ps.NRED, dtype=torch.float64, device="cuda")

    def measure(precision: str, rng_name: str = None):
        """W warm-up + K timed bench steps of the sweep kernel in `precision`; returns the timing record."""
        prec = PREC[precision]
        rng_name = rng_name or args.rng
        with torch.cuda.stream(stream):
            # one ensemble per bench step (a point of the Fz sweep), initialised on the device up front
            ens = []
            for i in range(nstep_total):
                p = ps.default_params(n=args.n, E0=1.0, K1=1.0, K2=0.0, kT=1.0, b=1.0, Fz=FZ_SWEEP[i % len(FZ_SWEEP)],
                                      num_chains=args.chains, chain_id0=rank * args.chains,
                                      seed=20260501 + i, precision=prec, device=local_rank,
                                      rng=ps.RNG_XOSHIRO128PP if rng_name == "xoshiro128++" else ps.RNG_MWC64X)
                ens.append(ps.Ensemble(p, stream=stream.cuda_stream))
            info = ens[0].launch_info()
            log(f"rank {rank}: {precision}: {nstep_total} ensembles ready; kernel {info.kernel.decode()} "
                f"lds={info.lds_bytes} lanes={info.lanes_per_block} wgs={info.blocks} wg/cu={info.blocks_per_cu}")

            def one_step(e, ev=None):
                if ev:
                    ev[0].record(stream)
                e.advance(args.mc_steps)
                if ev:
                    ev[1].record(stream)
                e.reduce_into(red.data_ptr())
                if use_dist:
                    all_reduce(red)

            for i in range(args.warmup):
                tw = time.perf_counter()
                one_step(ens[i])
                torch.cuda.synchronize()
                log(f"rank {rank}: {precision}: warmup step {i} took {time.perf_counter() - tw:.3f} s")
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                      for _ in range(args.steps)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                one_step(ens[args.warmup + i], events[i])
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            log(f"rank {rank}: {precision}: {args.steps} timed steps in {elapsed:.3f} s")
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        if use_dist:
            all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kernel_ms = [a.elapsed_time(b) for a, b in events]
        ens[-1].sync()           # surfaces a timed-out persistent launch (PSTAT_ERR_HIP) instead of timing garbage
        last = ps.summary_from_reduction(red.cpu().tolist(), args.mc_steps)
        for e in ens:
            e.close()
        return dict(precision=precision, elapsed=elapsed, kernel_ms=kernel_ms, last=last, info=info)

    def price(rec, pmc):
        """value / roofline / valu / lds of one timing record."""
        precision = rec["precision"]
        upd_per_launch = args.chains * args.mc_steps
        mean_ms = sum(rec["kernel_ms"]) / len(rec["kernel_ms"])
        rate = upd_per_launch / (mean_ms * 1e-3)             # updates/s of the kernel alone
        achieved = STATE_BYTES[precision] * rate / 1e9
        key = f"{precision}_n{args.n}_c{args.chains}_s{args.mc_steps}"
        traffic, traffic_raw, traffic_how = None, None, None
        all_ops = with_inst = waiting = None
        valu_ops, src = VALU_OPS_ESTIMATE[precision], "estimate (no PMC record for these kernel sources)"
        f64_ops = VALU_F64_OPS_ESTIMATE.get(precision, 0.0)
        if pmc and key in pmc.get("records", {}):
            r = pmc["records"][key]
            traffic, traffic_raw, traffic_how = r.get("hbm_bytes_per_launch"), r.get("raw_bytes_per_launch"), r.get("traffic_method")
            if r.get("valu_instructions_per_update_per_lane"):
                valu_ops = r["valu_instructions_per_update_per_lane"]
                src = "SQ_INSTS_VALU x 64 / updates, profiles/pmc_traffic.json (stamp = sha256 of the kernel sources)"
            if r.get("valu_f64_instructions_per_update_per_lane") is not None:
                f64_ops = r["valu_f64_instructions_per_update_per_lane"]
            all_ops, with_inst, waiting = r.get("instructions_per_update_per_lane"), r.get("wave_cycles_with_an_instruction"), r.get("wave_cycles_waiting")
        # an f64 op (v_fma/add/mul_f64 and the f64 transcendentals) issues at 16 lanes per clock and SIMD, everything else
        # (integer, select, f32, conversions: the generator, the address steering, the f32 Metropolis filter) at 32
        other_ops = max(0.0, valu_ops - f64_ops)
        issue_frac = rate * (f64_ops / (0.5 * VALU_LANE_OPS_PEAK) + other_ops / VALU_LANE_OPS_PEAK)
        last, info = rec["last"], rec["info"]
        # what ONE wave per SIMD can issue: a vector instruction every 4 cycles whatever its type (MI355X_MICROARCH.md, row
        # 'vector-instruction ISSUE cost'); the 2-cycle rate of 32-bit ops needs a second wave on the SIMD to fill the other slot
        waves_per_simd = info.blocks * ((info.lanes_per_block + 63) // 64) / float(NUM_SIMDS)
        one_wave_frac = rate * valu_ops * 4.0 / 64.0 / (NUM_SIMDS * CLOCK_HZ) if waves_per_simd <= 1.0 else None
        in_memory = "state in L2" in info.kernel.decode()
        # state in memory: (LDS bytes / 1 KiB per row of 64 lanes x 16 B) - 1 trash row = monomers whose cells are in LDS
        lds_share = min(1.0, (info.lds_bytes // 1024 - 1) / args.n) if in_memory else 1.0
        return {
            "value": world * upd_per_launch * args.steps / rec["elapsed"],
            "ms_per_step": rec["elapsed"] / args.steps * 1e3,
            "dtype": DTYPE[precision],
            "kernel": info.kernel.decode(), "lds_bytes_per_wg": info.lds_bytes, "lanes_per_wg": info.lanes_per_block,
            "workgroups": int(info.blocks), "wg_per_cu_resident": info.blocks_per_cu,
            "roofline": {"bound": "hbm", "bound_measured": "valu-issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw": traffic_raw,
                         "traffic_note": ("`traffic` = fabric bytes per launch (L2 misses + write-backs; for this 105 MB working set they "
                                          "are served by the Infinity Cache, not HBM): " + (traffic_how or "no PMC record for these kernel sources") +
                                          "; `traffic_raw` = (FETCH_SIZE + WRITE_SIZE)*1024 as rocprofv3 prints them.  FETCH_SIZE tallies every "
                                          "128-byte fabric request at 64 bytes, for scattered 16-byte reads as for streaming ones "
                                          "(profiles/r03/fetch_calib.txt: 1.07e9 scattered 16-byte reads = 1.07e9 requests, all 128-byte)"),
                         "kernel_ms": mean_ms, "equivalent": not in_memory,
                         "note": ("algorithmic bytes = %d B/update x %d updates per launch / HIP-event kernel time; " % (STATE_BYTES[precision], upd_per_launch)) +
                                 ("part of the state lives in memory (L2 / Infinity Cache): `traffic` is what the fabric carried; "
                                  "the kernel's bound is `valu`" if in_memory else
                                  "state is on-chip resident, so this is an equivalent rate, not HBM traffic (see `traffic`); "
                                  "the kernel's real bound is `valu`")},
            "valu": {"bound": "valu-issue", "achieved": rate * valu_ops / 1e12, "peak": VALU_LANE_OPS_PEAK / 1e12,
                     "unit": "T lane-ops/s", "frac": issue_frac,
                     "ops_per_update": valu_ops, "f64_ops_per_update": f64_ops, "other_ops_per_update": other_ops,
                     "frac_note": "issue time: f64 ops priced at 16 lanes/clk/SIMD (3.93e13 lane-ops/s), all others at 32 (7.86e13)",
                     "waves_per_simd": waves_per_simd, "frac_one_wave_issue": one_wave_frac,
                     "instructions_per_update": all_ops, "wave_cycles_with_an_instruction": with_inst, "wave_cycles_waiting": waiting,
                     "wave_cycles_note": "PMC record (profiled passes of this command): VALU + SALU + LDS + VMEM instructions per update and "
                                         "lane; the share of the waves' cycles in which one of their instructions was executing "
                                         "(SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES) and in which they waited (SQ_WAIT_ANY / SQ_WAVE_CYCLES)",
                     "frac_one_wave_issue_note": "this launch seats at most one wave per SIMD, and one wave issues one vector "
                                                 "instruction per 4 cycles whatever its type: ops_per_update x 4 cycles x wave-steps/s "
                                                 "/ (1024 SIMDs x 2.4 GHz) -- the ceiling that applies to this ensemble size "
                                                 "(the chip holds ~1.9 GHz under this load, DESIGN 3.9)",
                     "ops_source": src},
            "lds": {"bound": "lds-bandwidth", "achieved": lds_share * STATE_BYTES[precision] * rate / 1e9, "peak": LDS_PEAK_GBS,
                    "unit": "GB/s", "frac": lds_share * STATE_BYTES[precision] * rate / 1e9 / LDS_PEAK_GBS,
                    "note": "one state cell read + one written per update (ds_read/ds_write of %d B)%s" %
                            (STATE_BYTES[precision] // 2, "; %.0f %% of the cells live in LDS" % (100 * lds_share) if in_memory else "")},
            "check": {"Fz": FZ_SWEEP[(nstep_total - 1) % len(FZ_SWEEP)], "r3": last.avg[2], "r3_stderr": last.stderr[2],
                      "p3": last.avg[9], "U": last.avg[14], "AR": last.acceptance_ratio,
                      "chains_pooled": int(last.num_chains), "nan_rejects": int(last.nan_rejects),
                      "chains_collapsed": int(last.chains_collapsed)},
        }

    head = measure(args.precision)
    fast = named = None
    if args.precision == "f64" and not args.no_fast_path:
        fast = measure("f32")
    if args.precision == "f64" and args.rng == "mwc64x" and not args.no_rng_named:
        named = measure("f64", "xoshiro128++")    # the generator the north star names, on the headline kernel

    if rank == 0:
        pmc = None
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if rec.get("kernel_source_sha256_16") == kernel_source_hash():
                pmc = rec
            else:
                log("profiles/pmc_traffic.json was collected on other kernel sources: traffic = null, VALU count = estimate")
        except Exception:
            pass
        h = price(head, pmc)
        out = {
            "metric": "MC monomer-updates/sec (whole node) at n=100",
            "value": h["value"], "unit": "MC monomer-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": h["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": h["dtype"], "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: non-interacting dielectric chain, n=%d, E0=1, K1=1, K2=0, "
                                   "kT=1, b=1, Fz sweep (one point per step), %d chains/GPU x %d MC steps"
                                   % (args.n, args.chains, args.mc_steps),
                       "chains_per_gpu": args.chains, "mc_steps_per_chain": args.mc_steps, "n": args.n,
                       "parallelism": f"chains sharded over {world} GPU(s), one RCCL all-reduce of {ps.NRED} doubles per step",
                       "kernel": h["kernel"], "rng": args.rng, "lds_bytes_per_wg": h["lds_bytes_per_wg"],
                       "lanes_per_wg": h["lanes_per_wg"], "workgroups": h["workgroups"],
                       "wg_per_cu_resident": h["wg_per_cu_resident"]},
            "roofline": h["roofline"], "valu": h["valu"], "lds": h["lds"], "check": h["check"],
        }
        if fast is not None:
            f = price(fast, pmc)
            out["fast_path"] = {k: f[k] for k in ("value", "ms_per_step", "dtype", "kernel", "lanes_per_wg",
                                                  "wg_per_cu_resident", "roofline", "valu", "lds", "check")}
            out["fast_path"]["bias_bound"] = ("<= 5e-6 relative on every pooled average against the closed form (round-1 evidence: "
                                              "profiles/r01_final/bias_f32_long.json, 6.6e12 updates; the f32 kernel is unchanged "
                                              "since; DESIGN.md section 5)")
        if named is not None:
            g = price(named, None)
            out["rng_named"] = {"rng": "xoshiro128++ (Philox4x32-10-seeded per chain; BASELINE.json north_star: 'Philox/xoshiro counter-based RNG')",
                                "value": g["value"], "ms_per_step": g["ms_per_step"], "dtype": g["dtype"], "kernel": g["kernel"],
                                "roofline_frac": g["roofline"]["frac"], "kernel_ms": g["roofline"]["kernel_ms"], "check": g["check"],
                                "note": "same kernel, workload, K and W as the headline, which runs the default MWC64X "
                                        "(one v_mad_u64_u32 + one v_xor per draw against ten 32-bit ops; DESIGN.md section 4)"}
        base = None
        if world == 1 and not args.no_cpu_baseline:
            base, cpu_mean, cpu_se = cpu_baseline(args.n, args.mc_steps, args.cpu_seconds)
            base["parity"] = parity_vs_cpu(ps, PREC[args.precision], args.n, args.chains, args.mc_steps, local_rank, cpu_mean, cpu_se)
            if fast is not None:
                base["parity_fast_path"] = parity_vs_cpu(ps, ps.F32, args.n, args.chains, args.mc_steps, local_rank, cpu_mean, cpu_se)
            out["cpu_baseline"] = base
        if world == 1 and not args.no_configs and args.precision == "f64" and args.n == 100:
            kern_ms = h["roofline"]["kernel_ms"]
            headline = {"rate": args.chains * args.mc_steps / (kern_ms * 1e-3), "kernel_ms": kern_ms, "kernel": h["kernel"],
                        "check": h["check"]}
            out["configs"] = measure_configs(ps, torch, stream, pmc, headline, base, args.no_cpu_baseline, args.config_cpu_seconds)
            out["phase_scan"] = measure_phase_scan(ps)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
