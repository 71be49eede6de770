"""The five BASELINE.json configurations as launchable workloads (SURVEY.md 8(d) "Synthetic inputs"): one definition shared by
bench.py (its `configs` array), tools/profile_config.py (the rocprofv3 passes behind that array's counters) and
tools/measure_configs.py.  Reference work each one times: mcmc_eap_chain.jl:276-328 (the step loop) with
inc/energy.jl:7-9 (C1-C3), inc/eap_chain.jl:196-211 (C4: the O(n^2) pair loop, "probably by far the slowest calculation")
and :215-228 (C5: nearest-neighbour pairs)."""
from __future__ import annotations

F64_VECTOR_PEAK_TFLOPS = 256 * 4 * 16 * 2 * 2.4e9 / 1e12      # CUs x SIMDs x DP lanes/clk x (mul + add) x Hz = 78.6
PAIR_FLOP = 36                                                # ~34 flop + rsqrt + rcp per dipole-dipole term


def config_list(ps):
    """-> [dict(id, workload, cases (list of pstat_params), oracle (kwargs of oracle.make_params for one representative case),
    chains, mc_steps, bytes_per_update | flop_per_update, pmc_record)]"""
    P = ps.default_params
    F = ps.F64
    grid = [P(n=200, E0=0.2 * i, kT=10 ** (-2 + 0.2 * j), K1=1.0, K2=0.0, num_chains=128, precision=F, seed=1000 + 21 * i + j,
              energy_type=ps.ISING) for i in range(26) for j in range(21)]       # run/K1_E0-kT-phase.jl:21-24
    return [
        dict(id="C1", workload="BASELINE configs[0]: non-interacting dielectric, n=20, E0=0, K1=1, Fz=1, kT=1 (the reference's own "
                               "CPU-runnable case), 65 536 chains x 1e5 steps on the device",
             cases=[P(n=20, E0=0.0, K1=1.0, K2=0.0, Fz=1.0, num_chains=65536, precision=F, seed=1)], mc_steps=100000,
             oracle=dict(n=20, E0=0.0, K1=1.0, K2=0.0, Fz=1.0), bytes_per_update=32, pmc_record="cfg_C1"),
        dict(id="C2", workload="BASELINE configs[1]: the headline workload (see the top-level keys)", headline=True,
             cases=[P(n=100, E0=1.0, K1=1.0, K2=0.0, Fz=1.0, num_chains=65536, precision=F, seed=2)], mc_steps=100000,
             oracle=dict(n=100, E0=1.0, K1=1.0, K2=0.0, Fz=1.0), bytes_per_update=32, pmc_record="f64_n100_c65536_s100000"),
        dict(id="C3", workload="BASELINE configs[2]: polar chain (|mu| = 1), non-interacting, n=100, E0=1, Fz=1 (one point of the "
                               "E0 x Fz grid), 65 536 chains x 1e5 steps",
             cases=[P(n=100, E0=1.0, mu=1.0, Fz=1.0, chain_type=ps.POLAR, num_chains=65536, precision=F, seed=3)], mc_steps=100000,
             oracle=dict(n=100, E0=1.0, mu=1.0, Fz=1.0, chain_type=1), bytes_per_update=32, pmc_record="cfg_C3"),
        dict(id="C4", workload="BASELINE configs[3]: interacting dipole-dipole dielectric chain, n=64, E0=1, K1=1, Fz=0.5, all "
                               "n(n-1)/2 = 2016 pair terms per update, 16 384 chains x 2e4 steps",
             cases=[P(n=64, E0=1.0, K1=1.0, K2=0.0, Fz=0.5, energy_type=ps.INTERACTING, num_chains=16384, precision=F, seed=4)],
             mc_steps=20000, oracle=dict(n=64, E0=1.0, K1=1.0, K2=0.0, Fz=0.5, energy_type=1),
             flop_per_update=64 * 63 // 2 * PAIR_FLOP, pmc_record="cfg_C4"),
        dict(id="C5", workload="BASELINE configs[4]: (E0, kT) phase grid of run/K1_E0-kT-phase.jl, 26 x 21 = 546 points, n=200, Ising "
                               "(nearest-neighbour) dielectric, 128 chains per point in ONE launch x 2e4 steps",
             cases=grid, mc_steps=20000, oracle=dict(n=200, E0=2.6, K1=1.0, K2=0.0, kT=1.0, energy_type=2),
             bytes_per_update=32, pmc_record="cfg_C5"),
    ]
