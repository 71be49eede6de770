#!/usr/bin/env python3
"""BASELINE configs[4]: (E0, kT) phase-diagram scan at n = 200 (grid of run/K1_E0-kT-phase.jl:21-24:
E0 in 0:0.2:5, kT in 10^(-2:0.2:2), K1 = 1, K2 = 0, b = 1, F = 0), all 546 grid points in ONE
batched launch per GPU.  The reference farms the grid points out to worker processes with `pmap`
(run/K1_E0-kT-phase.jl:19-45); here the chains of every grid point are sharded over the ranks by global chain id
and the only exchange is one all-reduce(SUM) of the [points x 41] reduction tensor (RCCL; nothing at all for one GPU).

    python tools/phase_scan.py --chains 128 --steps 50000 --burn-in 20000 --energy Ising --out scan.csv
    python tools/phase_scan.py --main clustering --burn-schedule "1000,100,10,2,1" ...   # what the reference's
        # run/K1_E0-kT-phase.jl:45 launches per grid point: mcmc_clustering_eap_chain.jl with its annealed ladder
    python tools/phase_scan.py --gpus 8 ...                      # starts its own 8 ranks, one per GPU (RCCL)
    python tools/phase_scan.py --gpus 2 --backend gloo ...       # rehearsal: ranks may share a GPU, all-reduce via host
    python -m torch.distributed.run --nproc-per-node 8 tools/phase_scan.py --gpus 8 ...   (also fine)

`--chains` is chains per grid point PER RANK: rank r owns global chain ids [r * chains, (r + 1) * chains) of every
point, so N ranks x C chains pool exactly the chains of one rank x N*C (tests/test_gpu_dist.py).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def grid_points(points: int):
    g = [(0.2 * i, 10.0 ** (-2 + 0.2 * j)) for i in range(26) for j in range(21)]   # run/K1_E0-kT-phase.jl:21-24
    if points and points < len(g):      # an evenly spread subset (tests, rehearsals)
        g = [g[(k * len(g)) // points] for k in range(points)]
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--chains", type=int, default=128, help="chains per grid point per rank")
    ap.add_argument("--steps", type=int, default=50000)
    ap.add_argument("--burn-in", type=int, default=20000)
    ap.add_argument("--energy", choices=["noninteracting", "Ising"], default="Ising")
    ap.add_argument("--precision", choices=["f32", "f64", "q16"], default="f64", help="f64 = the reference's arithmetic (default); f32 / q16 = fast paths")
    ap.add_argument("--main", choices=["fixed-force", "clustering"], default="fixed-force",
                    help="which main's step: mcmc_eap_chain.jl or mcmc_clustering_eap_chain.jl (cluster flips)")
    ap.add_argument("--burn-schedule", default="1", help="kT multipliers of the burn-in ladder, comma-separated")
    ap.add_argument("--bend-mod", type=float, default=0.0)
    ap.add_argument("--cluster-prob", type=float, default=0.5)
    ap.add_argument("--points", type=int, default=0, help="use only this many grid points, evenly spread (0 = all 546)")
    ap.add_argument("--seed", type=int, default=20260501)
    ap.add_argument("--gpus", type=int, default=1, help="ranks; > 1 from a bare shell starts them (one per GPU)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend; gloo (through host memory, ranks may share a GPU) to rehearse N > 1 on a box with fewer GPUs")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks BEFORE anything touches the GPU; this process never imports torch
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from rank_spawn import spawn_ranks
        rc, out0 = spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus)
        sys.stdout.write(out0)
        sys.exit(rc)

    import torch
    import torch.distributed as dist
    import polymer_stats_amd as ps

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("phase_scan.py needs a GPU: libpstat has no CPU path")
    ndev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "nccl" and local >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible (use --backend gloo to rehearse)")
    local %= ndev
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    prec = {"f32": ps.F32, "f64": ps.F64, "q16": ps.Q16}[args.precision]
    en = ps.ISING if args.energy == "Ising" else ps.NONINTERACTING
    grid = grid_points(args.points)
    cases = [ps.default_params(n=args.n, E0=E0, kT=kT, K1=1.0, K2=0.0, b=1.0, num_chains=args.chains,
                               chain_id0=rank * args.chains, seed=args.seed + k, precision=prec,
                               energy_type=en, device=local,
                               **(dict(move_set=ps.MOVES_CLUSTER, bend_mod=args.bend_mod, cluster_prob=args.cluster_prob,
                                       adj_ub=0.40) if args.main == "clustering" else {}))
             for k, (E0, kT) in enumerate(grid)]
    stream = torch.cuda.Stream()
    red = torch.zeros(len(grid), ps.NRED, dtype=torch.float64, device="cuda")
    with torch.cuda.stream(stream):
        e = ps.Ensemble(cases, stream=stream.cuda_stream)
        t0 = time.perf_counter()
        ladder = [float(x) for x in args.burn_schedule.split(",") if x.strip()]
        rungs = 0
        if args.burn_in > 0:
            for mult in ladder:             # every rung is a fresh mcmc() call in the clustering main
                e.scale_kT(mult)
                if args.main == "clustering":
                    e.reset_sampler()
                e.advance(args.burn_in)
                rungs += 1
            e.scale_kT(1.0)
            if args.main == "clustering":
                e.reset_sampler()
            e.reset_averages()
        e.advance(args.steps)
        for k in range(len(grid)):
            e.reduce_into(red[k].data_ptr(), icase=k)
        if world > 1:
            if args.backend == "nccl":
                dist.all_reduce(red)
            else:
                host = red.cpu()
                dist.all_reduce(host)
                red.copy_(host)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        e.sync()
        info = e.launch_info()
    if rank == 0:
        host = red.cpu().numpy()
        rows = ["E0,kT,chains,r3,r3_stderr,rsq,p3,p3_stderr,psq,U,U_stderr,AR"]
        for (E0, kT), v in zip(grid, host):
            s = ps.summary_from_reduction(v, args.steps)
            rows.append(",".join(f"{x:.12g}" for x in (E0, kT, s.num_chains, s.avg[2], s.stderr[2], s.avg[6], s.avg[9],
                                                        s.stderr[9], s.avg[13], s.avg[14], s.stderr[14], s.acceptance_ratio)))
        text = "\n".join(rows) + "\n"
        if args.out:
            open(args.out, "w").write(text)
        else:
            sys.stdout.write(text[:2000] + ("...\n" if len(text) > 2000 else ""))
        upd = world * len(grid) * args.chains * (args.steps + rungs * args.burn_in)
        print(f"# {len(grid)} grid points x {world * args.chains} chains, n={args.n}, {args.main} main, {args.energy}, {args.precision}, "
              f"{world} rank(s) [{args.backend if world > 1 else 'no collective'}]: {wall:.3f} s wall, {upd / wall:.3e} attempted updates/s; "
              f"{info.kernel.decode()}, {info.lanes_per_block} lanes x {info.blocks_per_cu} per CU", file=sys.stderr)
    e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
