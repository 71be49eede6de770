"""BASELINE configs[3] exactly as stated (interacting dielectric chain, n = 64, E0 = 1, K1 = 1, K2 = 0, Fz = 0.5, kT = b = 1,
16 384 chains x 2e4 steps, random starts): the f32 all-pairs kernel against the f64 one, as two independent samples.
Writes the pooled averages with their across-chain standard errors and the two-sample z of every observable.

    python tools/config4_f32_vs_f64.py > gpurun_out/config4_f32_vs_f64.json      (then copy to profiles/<round>/)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polymer_stats_amd as ps

kw = dict(n=64, E0=1.0, K1=1.0, K2=0.0, Fz=0.5, energy_type=ps.INTERACTING, num_chains=16384)
out = {"workload": "BASELINE configs[3]: interacting dielectric, n=64, E0=1, K1=1, K2=0, Fz=0.5, kT=1, b=1, 16384 chains x 20000 steps, "
                   "random starts, no burn-in (mcmc_eap_chain.jl's protocol)"}
res = {}
for name, prec, seed in (("f32", ps.F32, 51), ("f64", ps.F64, 52), ("f64_second_seed", ps.F64, 53)):
    with ps.Ensemble(ps.default_params(precision=prec, seed=seed, **kw)) as e:
        e.advance(20000)
        s = e.summary()
        U = np.array([e.microstate(c)[6] for c in range(0, 16384, 16)])
    res[name] = s
    out[name] = {"seed": seed, "avg": dict(zip(ps.OBS_NAMES, s.avg)), "stderr": dict(zip(ps.OBS_NAMES, s.stderr)),
                 "AR": s.acceptance_ratio, "AR_stderr": s.ar_stderr, "nan_rejects": s.nan_rejects,
                 "chains_collapsed": s.chains_collapsed,
                 "final_U_quantiles_of_1024_chains": dict(zip(("q05", "q25", "q50", "q75", "q95"),
                                                              np.quantile(U, [0.05, 0.25, 0.5, 0.75, 0.95]).tolist()))}
for a, b in (("f32", "f64"), ("f64_second_seed", "f64")):
    sa, sb = res[a], res[b]
    z = {nm: (sa.avg[k] - sb.avg[k]) / float(np.hypot(sa.stderr[k], sb.stderr[k]) + 1e-300) for k, nm in enumerate(ps.OBS_NAMES)}
    z["AR"] = (sa.acceptance_ratio - sb.acceptance_ratio) / float(np.hypot(sa.ar_stderr, sb.ar_stderr))
    out[f"z_{a}_minus_{b}"] = z
print(json.dumps(out, indent=1))
