#!/usr/bin/env python3
"""Throughput of the BASELINE.json parity configurations (not the bench line): one launch each,
timed with HIP events through torch on the stream the library launches on.
    python tools/measure_configs.py [--quick]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


ONLY = ""


def run(ps, torch, name, cases, nsteps, repeats=3):
    if ONLY and ONLY not in name:
        return None
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        e = ps.Ensemble(cases, stream=stream.cuda_stream)
        e.advance(min(nsteps, 2000))          # warm-up (also leaves the adaptation in its regime)
        torch.cuda.synchronize()
        times = []
        for _ in range(repeats):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            e.advance(nsteps)
            b.record(stream)
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        info = e.launch_info()
        s = e.summary(0)
        chains = e.num_chains * e.ncases
        ms = min(times)
        out = dict(config=name, chains=chains, n=e.n, steps=nsteps, ms=round(ms, 3),
                   updates_per_s=chains * nsteps / (ms * 1e-3), kernel=info.kernel.decode(),
                   lds_bytes=info.lds_bytes, wg_per_cu=info.blocks_per_cu,
                   r3=s.avg[2], p3=s.avg[9], U=s.avg[14], AR=s.acceptance_ratio)
        e.close()
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="substring filter on the configuration name, e.g. 'C4'")
    ap.add_argument("--precisions", default="f32,q16,f64")
    args = ap.parse_args()
    import torch
    import polymer_stats_amd as ps
    global ONLY
    ONLY = args.only
    q = 10 if args.quick else 1
    P = ps.default_params
    res = []
    for prec, tag in ((ps.F32, "f32"), (ps.Q16, "q16"), (ps.F64, "f64")):
        if tag not in args.precisions.split(","):
            continue
        res.append(run(ps, torch, f"C1 n=20 E0=0 Fz=1 [{tag}]",
                       P(n=20, E0=0.0, Fz=1.0, num_chains=65536, precision=prec, seed=1), 100000 // q))
        res.append(run(ps, torch, f"C2 n=100 dielectric E0=1 K1=1 Fz=1 [{tag}]",
                       P(n=100, E0=1.0, K1=1.0, Fz=1.0, num_chains=65536, precision=prec, seed=2), 100000 // q))
        res.append(run(ps, torch, f"C3 n=100 polar mu=1 E0=1 Fz=1 [{tag}]",
                       P(n=100, E0=1.0, mu=1.0, Fz=1.0, chain_type=ps.POLAR, num_chains=65536, precision=prec, seed=3),
                       100000 // q))
        if prec != ps.Q16:
            res.append(run(ps, torch, f"C4 n=64 interacting dielectric E0=1 Fz=0.5 [{tag}]",
                           P(n=64, E0=1.0, K1=1.0, Fz=0.5, energy_type=ps.INTERACTING, num_chains=16384, precision=prec,
                             seed=4), 20000 // q))
            for nn, st in ((100, 6000), (200, 2000)):       # the reference's interacting sweep sizes (run/interacting_*_study.jl)
                res.append(run(ps, torch, f"C4b n={nn} interacting dielectric E0=1 Fz=0.5 [{tag}]",
                               P(n=nn, E0=1.0, K1=1.0, Fz=0.5, energy_type=ps.INTERACTING, num_chains=16384, precision=prec,
                                 seed=4), st // q))
        # C5: (E0, kT) phase grid of run/K1_E0-kT-phase.jl:21-24 (26 x 21 = 546 points), n = 200
        grid = [P(n=200, E0=0.2 * i, kT=10 ** (-2 + 0.2 * j), K1=1.0, num_chains=128, precision=prec, seed=1000 + 21 * i + j,
                  energy_type=et) for i in range(26) for j in range(21) for et in (ps.NONINTERACTING,)]
        res.append(run(ps, torch, f"C5 n=200 (E0,kT) grid 546 points x 128 chains, non-interacting [{tag}]", grid, 50000 // q))
        grid = [P(n=200, E0=0.2 * i, kT=10 ** (-2 + 0.2 * j), K1=1.0, num_chains=128, precision=prec, seed=1000 + 21 * i + j,
                  energy_type=ps.ISING) for i in range(26) for j in range(21)]
        res.append(run(ps, torch, f"C5 n=200 (E0,kT) grid 546 points x 128 chains, Ising [{tag}]", grid, 50000 // q))
        # C6: the clustering main as run/Ising_2025-12-18.jl:25-27 and run/K1_E0-kT-phase.jl:45 launch it
        # (Ising energy, K2-only dielectric, default cluster-prob 0.5, step-adjust-ub 0.40)
        for et, en in ((ps.ISING, "Ising"), (ps.NONINTERACTING, "non-interacting")):
            res.append(run(ps, torch, f"C6 clustering main n=100 {en} E0=1 K1=0 K2=1 kT=1 [{tag}]",
                           P(n=100, E0=1.0, K1=0.0, K2=1.0, kT=1.0, energy_type=et, num_chains=65536, precision=prec,
                             seed=6, move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, adj_ub=0.40), 20000 // q))
        if prec != ps.Q16:
            # C7: clustering main with the all-pairs energies (one chain per wavefront)
            res.append(run(ps, torch, f"C7 clustering main n=100 interacting E0=1 K1=1 Fz=0.5 [{tag}]",
                           P(n=100, E0=1.0, K1=1.0, Fz=0.5, energy_type=ps.INTERACTING, num_chains=16384, precision=prec,
                             seed=7, move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, adj_ub=0.40), 4000 // q))
            res.append(run(ps, torch, f"C7 clustering main n=100 cutoff 7.5 E0=1 K1=1 Fz=0.5 [{tag}]",
                           P(n=100, E0=1.0, K1=1.0, Fz=0.5, energy_type=ps.CUTOFF, cutoff_radius=7.5, num_chains=16384,
                             precision=prec, seed=7, move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, adj_ub=0.40), 4000 // q))
    return 0


if __name__ == "__main__":
    sys.exit(main())
