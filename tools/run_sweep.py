#!/usr/bin/env python3
"""A whole parameter sweep of the reference's run/*.jl kind in batched launches: the cases that `pmap` hands to worker
processes one `julia mcmc_*.jl` at a time (run/interacting_dielectric_study.jl:37-47) become the cases of one ensemble
per chain length, dealt round-robin to the ranks; every case gets its `<name>.out` exactly as the run script would have
written it (polymer_stats_amd/sweep.py), ready for scripts/aggregate_mcmc.jl (or `--aggregate`, its twin).

    # run/interacting_dielectric_study.jl
    python tools/run_sweep.py out/ --main mcmc_eap_chain --axis b=0.5,1,2 --axis n=100,200 --axis Fx=0,0.5,1,2 \
        --axis Fz=0,0.5,1,2 --axis kT=1 --axis E0=0.1,1,10 --axis K1=0,0.1,0.5,1,2 --axis K2=0,0.1,0.5,1,2 --skip 'K1==K2' \
        --num-chains 16 -- --chain-type dielectric --energy-type interacting --num-steps 500000 -v 2
    # run/K1_E0-kT-phase.jl: 546 grid points x 5 runs of the clustering main, on 8 GPUs
    python tools/run_sweep.py out/ --gpus 8 --main mcmc_clustering_eap_chain --axis b=1 --axis n=100 --axis Fx=0 --axis Fz=0 \
        --axis kT='10^(-2:0.2:2)' --axis E0=0:0.2:5 --axis K1=1 --axis K2=0 --axis kappa=0 --axis run=1:5 \
        --name E0,K1,K2,kT,Fz,Fx,n,b,kappa,run:raw \
        -- --chain-type dielectric --energy-type Ising --num-steps 2500000 --burn-in 100000 -v 2 --stepout 250

The first --axis is the outermost loop (`for b in bs, n in ns, ...`).  Everything after `--` goes to the main unchanged.
A case whose .out exists is not run again (the run scripts' `isfile(outfile)`); --overwrite runs it anyway.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    argv = sys.argv[1:]
    fixed = []
    if "--" in argv:
        k = argv.index("--")
        argv, fixed = argv[:k], argv[k + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("workdir")
    ap.add_argument("--main", choices=["mcmc_eap_chain", "mcmc_clustering_eap_chain"], default="mcmc_eap_chain")
    ap.add_argument("--axis", action="append", default=[], metavar="KEY=VALUES",
                    help="values: a,b,c | start:step:stop (also several, comma-separated) | 10^(start:step:stop); first axis = outermost loop")
    ap.add_argument("--cases", default="", help="JSON array of case objects instead of (or appended to) the axes' product")
    ap.add_argument("--skip", default="", help="leave out the cases for which this holds, e.g. 'K1==K2'")
    ap.add_argument("--name", default="", help="file-name tokens, e.g. E0,K1,K2,kT,Fz,Fx,n,b,run:int (kinds: milli [default] | int | raw)")
    ap.add_argument("--num-chains", type=int, default=64, help="independent chains per case, pooled (1 = literally one reference run)")
    ap.add_argument("--seed", type=int, default=None, help="base seed; case k of the full list runs on seed + k (default: fresh entropy)")
    ap.add_argument("--precision", choices=["f64", "f32", "q16"], default=None)
    ap.add_argument("--rng", choices=["mwc64x", "xoshiro128++"], default=None)
    ap.add_argument("--gpus", type=int, default=1, help="ranks; > 1 from a bare shell starts them (one per GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: ranks beyond the visible GPUs share them")
    ap.add_argument("--overwrite", action="store_true")
    ap.add_argument("--csv", action="store_true", help="also write every case's _trajectory.csv and _rolling.csv (launches split at --stepout)")
    ap.add_argument("--max-chains", type=int, default=262144, help="chains per launch (cases per ensemble = this / num-chains)")
    ap.add_argument("--dry-run", action="store_true", help="print the plan (cases, file names, ensembles) and stop: no GPU needed")
    ap.add_argument("--aggregate", default="", help="afterwards write scripts/aggregate_mcmc.jl's CSV of the whole directory here")
    ap.add_argument("--aggregate-args", default="*.out,dielectric", help="pattern,dielectric|polar[,kappaflag[,runflag]]")
    args = ap.parse_args(argv)

    from polymer_stats_amd import sweep as sw          # (no GPU call on import)
    axes = []
    for a in args.axis:
        key, _, vals = a.partition("=")
        axes.append((key.strip(), sw.axis_values(vals)))
    cases = sw.product_cases(axes) if axes else []
    if args.cases:
        cases += sw.load_cases(args.cases)
    if args.skip:
        cases = [c for c in cases if not sw.skip_case(args.skip, c)]
    if not cases:
        raise SystemExit("no cases: give --axis and/or --cases")

    if args.dry_run:
        pl = sw.plan(args.main, fixed, cases, args.workdir, name=args.name or None, num_chains=args.num_chains,
                     seed=0 if args.seed is None else args.seed, precision=args.precision, rng=args.rng)
        groups = {}
        for p in pl:
            groups.setdefault(sw._signature(p), []).append(p)
        done = sum(os.path.isfile(p["_out"]) for p in pl)
        print(f"{len(pl)} cases ({done} already there), {len(groups)} ensemble(s) per rank at most, {args.num_chains} chain(s) per case, "
              f"{args.gpus} rank(s); main {args.main}; seed {'fresh entropy' if args.seed is None else args.seed} + position")
        for g in groups.values():
            p0 = g[0]
            print(f"  n = {p0['num-monomers']}, {p0['energy-type']}, {p0['chain-type']}, {p0['num-steps']} steps: {len(g)} cases, "
                  f"{len(g) * args.num_chains} chains")
        for p in pl[:3] + (pl[-1:] if len(pl) > 3 else []):
            print("  " + os.path.basename(p["_out"]))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the ranks are started before anything touches the GPU; this process never does.  A fresh default seed has to be
        # drawn HERE: every rank must name the same seed for case k
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from rank_spawn import spawn_ranks
        extra = [] if args.seed is not None else ["--seed", str(sw.fixed_main.fresh_seed() & 0x7FFFFFFFFFFF)]
        pre = sys.argv[1:sys.argv.index("--")] if "--" in sys.argv else sys.argv[1:]
        agg = bool(args.aggregate)              # the ranks only run; the parent aggregates when all of them are done
        kept, skip = [], False
        for a in pre:                           # drop `--aggregate X` and `--aggregate=X` from what the ranks are given
            if skip:
                skip = False
            elif a == "--aggregate":
                skip = True
            elif not a.startswith("--aggregate="):
                kept.append(a)
        pre = kept
        rc, out0 = spawn_ranks(os.path.abspath(__file__), pre + extra + (["--"] + fixed if fixed else []), args.gpus)
        sys.stdout.write(out0)
        if rc == 0 and agg:
            rc = aggregate(args)
        sys.exit(rc)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1 and args.seed is None:
        # started by an external launcher: every rank would draw its own fresh base seed, and case k's seed = base + k must
        # be the same whichever rank runs it
        raise SystemExit("run_sweep.py under an external launcher (WORLD_SIZE preset) needs --seed: all ranks must name the same one")
    lib = sw._lib.load()
    ndev = lib.pstat_device_count()
    if ndev < 1:
        raise SystemExit("run_sweep.py needs a GPU: libpstat has no CPU path")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if local >= ndev and not args.share_gpu:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible (--share-gpu to rehearse)")
    t0 = time.time()
    res = sw.run_sweep(args.main, fixed, cases, args.workdir, name=args.name or None, num_chains=args.num_chains, seed=args.seed,
                       precision=args.precision, rng=args.rng, rank=rank, world=world, device=local % ndev,
                       overwrite=args.overwrite, write_csv=args.csv, max_chains=args.max_chains,
                       log=lambda m: print("# " + m, file=sys.stderr, flush=True))
    print(f"# rank {rank} of {world}: {len(res['ran'])} cases run in {res['launches']} ensembles, {len(res['skipped'])} already "
          f"there; {time.time() - t0:.2f} s", file=sys.stderr, flush=True)
    if world == 1 and args.aggregate:
        sys.exit(aggregate(args))


def aggregate(args) -> int:
    from polymer_stats_amd import aggregate_mcmc as ag
    parts = args.aggregate_args.split(",")
    flag = lambda i: len(parts) > i and parts[i].strip() == "true"
    return ag.aggregate(args.aggregate, args.workdir, parts[0], parts[1] if len(parts) > 1 else "dielectric", flag(2), flag(3))


if __name__ == "__main__":
    main()
