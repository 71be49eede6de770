#!/usr/bin/env python3
"""How small is the bias of the reduced-precision kernels?  Equilibrium averages of BASELINE configs[1]
(n = 100, E0 = 1, K1 = 1, Fz = 1; closed form in tests/golden/ni_closed_form.json) from 65 536 chains x
`--steps` recorded steps after a burn-in, against the closed form, in units of the pooled standard error and
as a relative deviation.
    python tools/bias_probe.py --precision f32 --steps 20000000
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--steps", type=int, default=20_000_000)
    ap.add_argument("--chains", type=int, default=65536)
    ap.add_argument("--rng", default="mwc64x")
    args = ap.parse_args()
    import numpy as np
    import polymer_stats_amd as ps
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ni_closed_form.json")))["cases"]
    eq = gold["cfg2_n100_E0_1_K1_1_Fz1"]["avg"]
    prec = {"f32": ps.F32, "f64": ps.F64, "q16": ps.Q16}[args.precision]
    p = ps.default_params(n=100, E0=1.0, K1=1.0, K2=0.0, Fz=1.0, kT=1.0, num_chains=args.chains, precision=prec, seed=424242,
                          rng=ps.RNG_XOSHIRO128PP if args.rng != "mwc64x" else ps.RNG_MWC64X)
    t0 = time.time()
    with ps.Ensemble(p) as e:
        e.advance(200_000)
        e.reset_averages()
        left = args.steps
        while left > 0:                       # progress lines for long runs
            k = min(left, 5_000_000)
            e.advance(k); e.sync()
            left -= k
            print(f"# {args.steps - left} / {args.steps} steps, {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
        s = e.summary()
    out = {"precision": args.precision, "rng": args.rng, "chains": args.chains, "steps": args.steps, "obs": {}}
    for k, name in enumerate(ps.OBS_NAMES):
        want = eq[name]
        z = (s.avg[k] - want) / (s.stderr[k] + 1e-300) if s.stderr[k] > 0 else 0.0
        rel = (s.avg[k] - want) / want if abs(want) > 1e-9 else None
        out["obs"][name] = {"gpu": s.avg[k], "closed_form": want, "stderr": s.stderr[k], "z": z, "rel": rel}
    out["max_abs_z"] = max(abs(v["z"]) for v in out["obs"].values())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
