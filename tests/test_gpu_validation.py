"""Round-2 parity cases on the GPU (all through the C ABI):
  * every committed closed-form fixture (tests/golden/ni_closed_form.json) against the f32, q16 and f64 kernels,
    sharply (burn-in, then the pooled averages must sit on the closed form within their own standard error):
    BASELINE configs[1] at Fz = 0, 1, 5, configs[2] at E0 = 1 and 10, Fx != 0, K2 != 0;
  * the reference's own validation workflow: E0 = 0 against the Langevin law over the force range of
    run/prelim/test_noE.jl:18-21 (Fz up to 100), with that script's adaptation band (:38);
  * the f32 all-pairs kernels at the reference's sweep sizes -- n = 64 (one monomer per lane), n = 100 (two,
    packed ring sum), n = 200 (four) -- against the oracle's literal O(n^2) mode and against the f64 kernel;
  * BASELINE configs[3] exactly as stated, f32 against f64;
  * the failure counters of ABI v5.
"""
import numpy as np
import pytest

from helpers import both, pooled

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ps():
    import polymer_stats_amd as ps
    assert ps._lib.load().pstat_device_count() >= 1, "no HIP device visible"
    return ps


def _golden_params(ps, case):
    P = case["params"]
    return dict(n=int(P["n"]), E0=P["E0"], K1=P["K1"], K2=P["K2"], mu=P["mu"], kT=P["kT"], Fz=P["Fz"], Fx=P["Fx"], b=P["b"],
                chain_type=ps.POLAR if P["chain"] == "polar" else ps.DIELECTRIC)


GOLDEN_N100 = ["cfg2_n100_E0_1_K1_1_Fz0", "cfg2_n100_E0_1_K1_1_Fz1", "cfg2_n100_E0_1_K1_1_Fz5",
               "cfg3_polar_n100_E0_10_mu1_Fz025", "cfg3_polar_n100_E0_1_mu1_Fz1", "diel_n100_E0_1_K1_1_Fx1",
               "diel_n100_E0_2_K2_1_Fz05"]


@pytest.mark.parametrize("prec", [0, 2, 1], ids=["f32", "q16", "f64"])
@pytest.mark.parametrize("name", GOLDEN_N100 + ["cfg1_n20_E0_0_Fz1", "diel_n8_E0_15_K1_07_K2_03_Fz04_Fx03_kT07_b13"])
def test_every_closed_form_fixture_sharply(ps, golden, name, prec):
    """All 16 pooled averages within 5 of THEIR OWN standard errors of the closed form (no transient: hot rung, target
    temperature, reset, record).  65 536 chains x 5e4 recorded steps resolve <r_z> to ~1e-4 relative."""
    kw = _golden_params(ps, golden[name])
    nch = 8192 if prec == 1 else 65536
    pp = ps.default_params(num_chains=nch, precision=prec, seed=20260502, **kw)
    with ps.Ensemble(pp) as e:
        e.set_kT(3.0 * kw["kT"])
        e.advance(10000)
        e.set_kT(kw["kT"])
        e.advance(30000)
        e.reset_averages()
        e.advance(50000)
        s = e.summary()
    avg, se = np.array(s.avg), np.array(s.stderr)
    eq = golden[name]["avg"]
    z = np.array([(avg[k] - eq[nm]) / (se[k] + 1e-12 * (1 + abs(eq[nm]))) for k, nm in enumerate(ps.OBS_NAMES)])
    assert np.all(np.abs(z) < 5.0), dict(zip(ps.OBS_NAMES, np.round(z, 2)))
    assert s.nan_rejects == 0 and s.chains_collapsed == 0


def _langevin(x):
    return 1.0 / np.tanh(x) - 1.0 / x


@pytest.mark.parametrize("prec", [0, 2, 1], ids=["f32", "q16", "f64"])
def test_reference_validation_sweep_against_langevin(ps, prec):
    """run/prelim/test_noE.jl:18-21,38: E0 = 0, n = 100, kT = b = 1, Fz from 0.05 to 100, adaptation band
    [0.19, 0.39], checked against <r_z> = n b L(F b / kT), <r_z^2> and <U> = -F <r_z> (inc/langevin.jl is the
    inverse of the same law).  The whole sweep is ONE batched launch (cases differ in Fz only).  Large forces
    are where f32 differs from f64 most: exp2 range, step sizes adapted far down, theta pinned near 0."""
    Fzs = [0.05, 0.5, 1.0, 2.0, 10.0, 40.0, 100.0]
    nch = 2048 if prec == 1 else 16384
    cases = [ps.default_params(num_chains=nch, precision=prec, n=100, E0=0.0, K1=1.0, K2=0.0, kT=1.0, b=1.0, Fz=F,
                               adj_lb=0.19, adj_ub=0.39, adj_scale=1.1, seed=20260503) for F in Fzs]
    with ps.Ensemble(cases) as e:
        e.advance(60000)          # adaptation needs ~20 windows of 2500 steps to bring the steps down at F = 100
        e.reset_averages()
        e.advance(60000)
        for i, F in enumerate(Fzs):
            s = e.summary(i)
            L = _langevin(F)
            rz = 100.0 * L
            var1 = 1.0 - 2.0 * L / F - L * L                 # Var(cos theta) of one monomer
            rz2 = 100.0 * var1 + rz * rz
            z = [(s.avg[2] - rz) / s.stderr[2], (s.avg[5] - rz2) / s.stderr[5], (s.avg[14] + F * rz) / s.stderr[14]]
            assert np.all(np.abs(z) < 5.0), (F, np.round(z, 2), s.avg[2], rz)
            assert abs(s.avg[0]) < 5 * s.stderr[0] and abs(s.avg[1]) < 5 * s.stderr[1]
            assert 0.15 < s.acceptance_ratio < 0.75, (F, s.acceptance_ratio)   # the band holds it (0.6 at small F: steps saturate)
            assert s.avg[7] == 0.0 and s.avg[9] == 0.0                           # E0 = 0: no dipoles


@pytest.mark.parametrize("n,nsteps", [(64, 1500), (100, 1000), (200, 500), (300, 250)], ids=["n64-M1", "n100-M2", "n200-M4", "n300-M8"])
def test_f32_all_pairs_kernels_against_oracle_and_f64(ps, oracle, n, nsteps):
    """interacting_kernel<float, M = 1 | 2 | 4 | 8>: M >= 2 runs the packed ring sum (ring_pair_sum_pk), a different
    LDS layout and code path from M = 1 and from the f64 template.  (i) a zero-step launch: the initial pair
    energy of the same angles agrees with the f64 kernel's to f32 rounding; (ii) a short pre-collapse run:
    pooled means within 4.5 sigma of the oracle's literal O(n^2) mode; (iii) same seeds, f32 vs f64."""
    kw = dict(n=n, E0=1.0, K1=1.0, K2=0.0, Fz=0.5, energy_type=ps.INTERACTING, seed=31)
    # (i)
    U0 = {}
    for prec in (ps.F32, ps.F64):
        with ps.Ensemble(ps.default_params(num_chains=64, precision=prec, **kw)) as e:
            U0[prec] = np.array([e.microstate(c) for c in range(64)])
    # f32 positions are prefix sums of unit vectors (absolute error ~1e-6 at n = 100), so a pair that a random start
    # happens to put at r ~ 1e-2 carries a relative error ~3 dr / r ~ 1e-3 on a term that dominates |U|: the typical chain
    # agrees to f32 rounding, the few with a near contact to a fraction of that one term
    rel = np.abs(U0[ps.F32][:, 6] - U0[ps.F64][:, 6]) / (np.abs(U0[ps.F64][:, 6]) + n * 0.5)
    assert np.median(rel) < 2e-5 and rel.max() < 5e-2, (np.median(rel), rel.max())
    np.testing.assert_allclose(U0[ps.F32][:, :6], U0[ps.F64][:, :6], rtol=0, atol=2e-4)
    # (ii) + (iii)
    nch = 1024
    res = {}
    for prec in (ps.F32, ps.F64):
        op, pp = both(nsteps, num_chains=nch if prec == ps.F32 else 256, precision=prec, **kw)
        with ps.Ensemble(pp) as e:
            e.advance(nsteps)
            res[prec] = e.rolling() + (e.summary(),)
    osums, onorm, _ = oracle.run_many(op, 7_000_000, 96, nthreads=8, mode="faithful")
    o_avg, o_se = pooled(osums, onorm)
    g_avg, g_se, s32 = res[ps.F32]
    z = (g_avg - o_avg) / np.sqrt(g_se ** 2 + o_se ** 2 + 1e-300)
    assert np.all(np.abs(z) < 4.5), dict(zip(ps.OBS_NAMES, np.round(z, 2)))
    d_avg, d_se, s64 = res[ps.F64]
    z2 = (g_avg - d_avg) / np.sqrt(g_se ** 2 + d_se ** 2 + 1e-300)
    assert np.all(np.abs(z2) < 4.5), dict(zip(ps.OBS_NAMES, np.round(z2, 2)))
    assert abs(s32.acceptance_ratio - s64.acceptance_ratio) < 5 * np.hypot(s32.ar_stderr, s64.ar_stderr) + 2e-3


def test_config4_as_stated_f32_against_f64(ps):
    """BASELINE configs[3] exactly as stated: interacting dielectric chain, n = 64, E0 = 1, K1 = 1, K2 = 0, Fz = 0.5,
    kT = b = 1, 16 384 chains x 2e4 steps from random starts.  Without excluded volume (inc/eap_chain.jl:200-207)
    chains fold back onto themselves and fall into the 1/r^3 singularity -- the reference's behaviour: after 2e4
    steps the typical chain holds a few contacts at r ~ 4e-3 .. 1e-2 b and U ~ -1e5 .. -1e6 kT.  There the f32
    kernel is NOT equivalent to Float64 at the kT level: a contact's term changes by ~3 U dr / r per displacement dr,
    so kT-level fidelity needs position differences good to ~1e-10 b, and f32 positions (and the f32 ulp of U itself,
    0.01 - 0.1 kT at |U| ~ 1e5 - 1e6) are orders of magnitude coarser.  Measured (profiles/r02/config4_f32_vs_f64.json):
    <r>, <p> agree within errors, the acceptance ratio does not (0.051 vs 0.057).  The hosts therefore run the
    all-pairs energies in f64 unless told otherwise; this test pins what does agree and bounds what does not.
    Two independent samples (different seeds: same-seed trajectories diverge at the first near-singular step anyway)."""
    kw = dict(n=64, E0=1.0, K1=1.0, K2=0.0, Fz=0.5, energy_type=ps.INTERACTING, num_chains=16384)
    out = {}
    for prec, seed in ((ps.F32, 51), (ps.F64, 52)):
        with ps.Ensemble(ps.default_params(precision=prec, seed=seed, **kw)) as e:
            e.advance(20000)
            out[prec] = e.summary()
    a, b = out[ps.F32], out[ps.F64]
    for k in (0, 1, 2, 7, 8, 9):
        z = (a.avg[k] - b.avg[k]) / np.hypot(a.stderr[k], b.stderr[k])
        assert abs(z) < 5.0, (ps.OBS_NAMES[k], a.avg[k], b.avg[k], z)
    assert abs(a.acceptance_ratio / b.acceptance_ratio - 1) < 0.2, (a.acceptance_ratio, b.acceptance_ratio)
    assert a.chains_collapsed > 0.2 * 16384 and b.chains_collapsed > 0.2 * 16384


def test_config4_as_stated_f64_against_the_oracle_in_distribution(ps, oracle):
    """BASELINE configs[3] as stated, f64 kernel against the oracle's literal O(n^2) mode as two independent samples.
    On collapsed chains a 1e-15 rounding of a position can flip a decision (the trajectory fuzz), so bit
    parity is not the right question there; equality in distribution is: <r>, <r_j^2>, <p>, AR within 5 sigma.

    The ENERGY observable (the pair sum of inc/eap_chain.jl:200-207 is what this configuration exists for): the pooled mean
    of U is dominated by the 1/r^3 tail of a few collapsed chains (its standard error is as large as the mean), so the
    comparison is made on a robust statistic -- the quartiles of the per-chain running mean <U> over the 16 384 device
    chains (pstat_chain_means) against the 192 oracle chains, each within 5 standard errors of a sample quantile,
    sqrt(p (1 - p) / N) / f(q_p), with the density f read off the large device sample."""
    nsteps = 20000
    op, pp = both(nsteps, num_chains=16384, precision=ps.F64, n=64, E0=1.0, K1=1.0, K2=0.0, Fz=0.5,
                  energy_type=ps.INTERACTING, seed=61)
    with ps.Ensemble(pp) as e:
        e.advance(nsteps)
        s = e.summary()
        u_dev = e.chain_means()[14]                 # per-chain <U> (row 14 = U of the reduction vector's observables)
    osums, onorm, onacc = oracle.run_many(op, 9_000_000, 192, nthreads=16, mode="faithful")
    o_avg, o_se = pooled(osums, onorm)
    for k in (0, 1, 2, 3, 4, 5, 7, 8, 9):
        z = (s.avg[k] - o_avg[k]) / np.hypot(s.stderr[k], o_se[k])
        assert abs(z) < 5.0, (ps.OBS_NAMES[k], s.avg[k], o_avg[k], z)
    o_ar = onacc / nsteps
    assert abs(s.acceptance_ratio - o_ar.mean()) < 5 * np.hypot(s.ar_stderr, o_ar.std(ddof=1) / np.sqrt(len(o_ar)))
    u_orc = osums[:, 14] / onorm
    assert np.all(np.isfinite(u_dev)) and np.all(np.isfinite(u_orc))
    assert np.median(u_dev) < -100.0              # the pair energy dominates: non-interacting chains hold U ~ -30 here
    for p in (0.25, 0.5, 0.75):
        h = 0.05
        q_dev, q_orc = np.quantile(u_dev, p), np.quantile(u_orc, p)
        f = 2 * h / (np.quantile(u_dev, p + h) - np.quantile(u_dev, p - h))          # density of per-chain <U> at the quantile
        se = np.sqrt(p * (1 - p)) / f * np.sqrt(1.0 / len(u_dev) + 1.0 / len(u_orc))
        assert abs(q_dev - q_orc) < 5.0 * se, (p, q_dev, q_orc, se)
        assert se < 0.25 * abs(q_dev), (p, q_dev, se)   # the comparison has teeth: a quartile is known to better than 25 %


def test_f64_state_in_memory_matches_state_in_lds(ps, monkeypatch):
    """The f64 non-interacting sweep keeps its (theta, phi) cells in global memory (L2 / Infinity Cache) once LDS would
    seat fewer than four full waves per CU (n > 40), else in LDS.  Same kernel template, two homes for the state: forced
    either way, every chain's trajectory, generator, counters and sums are bit-identical -- also across launch splits
    (fill/spill of the working copy) and with the rare options on (flips, umbrella, re-init offset)."""
    for kw in (dict(n=100, E0=1.0, K1=1.0, Fz=1.0), dict(n=17, E0=0.5, K1=0.7, K2=0.2, Fz=0.3, Fx=0.4, chain_type=ps.POLAR, mu=0.7),
               dict(n=200, E0=2.0, K1=0.3, K2=0.9, Fz=0.1, do_flips=1, umbrella=1),
               # Ising: the two neighbour rows travel with the prefetched row (weak coupling: no collapse in 1500 steps)
               dict(n=70, E0=1.0, K1=0.3, K2=0.05, Fz=0.3, Fx=0.2, energy_type=ps.ISING),
               dict(n=41, E0=0.6, mu=0.25, Fz=0.4, chain_type=ps.POLAR, energy_type=ps.ISING, do_flips=1, umbrella=1)):
        res = {}
        for where in ("lds", "global"):
            monkeypatch.setenv("PSTAT_F64_STATE", where)
            with ps.Ensemble(ps.default_params(num_chains=200, precision=ps.F64, seed=77, steps_per_adjust=300, **kw)) as e:
                assert ("state in L2" in e.launch_info().kernel.decode()) == (where == "global")
                e.advance(700)
                e.reinit(False)
                e.advance(501)
                e.advance(299)
                res[where] = [e.chain_state(c) for c in (0, 63, 64, 199)] + [e.reduce_host()]
        ising = kw.get("energy_type") == ps.ISING    # (its two variants order the neighbour terms' roundings differently)
        for x, y in zip(res["lds"][:-1], res["global"][:-1]):
            for key in ("theta", "phi", "rng"):
                assert np.array_equal(x[key], y[key]), (kw, key)
            if ising:
                np.testing.assert_allclose(x["sums"], y["sums"], rtol=1e-10, atol=1e-8)
            else:
                assert np.array_equal(x["sums"], y["sums"]), kw
            assert (x["nacc_total"], x["phi_step"], x["theta_step"], x["normalizer"]) == \
                   (y["nacc_total"], y["phi_step"], y["theta_step"], y["normalizer"])
        if ising:
            np.testing.assert_allclose(res["lds"][-1], res["global"][-1], rtol=1e-10, atol=1e-8)
        else:
            assert np.array_equal(res["lds"][-1], res["global"][-1])
    monkeypatch.delenv("PSTAT_F64_STATE")
    with ps.Ensemble(ps.default_params(num_chains=64, precision=ps.F64, n=40)) as e:
        assert "state in L2" not in e.launch_info().kernel.decode()       # LDS seats 4 full waves: stays there
    with ps.Ensemble(ps.default_params(num_chains=64, precision=ps.F64, n=41)) as e:
        assert "state in L2" in e.launch_info().kernel.decode()


def test_failure_counters(ps, oracle):
    """nan_rejects / chains_collapsed (ABI v5).  Non-interacting: identically zero.  A strongly coupled polar Ising
    chain runs into r -> 0 between neighbours (the reference's behaviour): every chain is reported collapsed,
    the reduction carries both counters and pstat_reset_averages clears the first.  (The counter's positive path:
    test_nonfinite_energy_counter_counts_every_such_proposal below.)"""
    with ps.Ensemble(ps.default_params(num_chains=256, n=30, E0=1.0, Fz=1.0, seed=2)) as e:
        e.advance(3000)
        s = e.summary()
        assert s.nan_rejects == 0 and s.chains_collapsed == 0
    kw = dict(n=12, E0=1.0, mu=2.0, Fz=0.2, chain_type=ps.POLAR, energy_type=ps.ISING, seed=8)
    for prec in (ps.F32, ps.F64):
        with ps.Ensemble(ps.default_params(num_chains=512, precision=prec, **kw)) as e:
            e.advance(20000)
            s = e.summary()
            assert s.chains_collapsed > 256, (prec, s.chains_collapsed)
            assert abs(s.avg[14]) > 1e5 * 12
            red = e.reduce_host()
            assert red[ps.NRED - 2] == s.nan_rejects and red[ps.NRED - 1] == s.chains_collapsed
            e.reset_averages()
            assert e.summary().nan_rejects == 0


def test_a_timed_out_launch_poisons_the_handle(ps, monkeypatch):
    """a segment job that gives up waiting for its predecessor used to be visible in pstat_sync only, and the
    next launch wiped the flag.  Forced here (a 1-spin bound with every (block, segment) job co-resident: the second
    segments start while the first are still running and give up at once; the kernel then stops taking jobs and ends
    normally).  Every accessor that would hand out results, and every further advance, must fail with PSTAT_ERR_HIP --
    the averages would be over steps that never ran -- and the failure must survive later calls.  Other handles are
    unaffected."""
    monkeypatch.setenv("PSTAT_SEGMENTS", "7")
    monkeypatch.setenv("PSTAT_MAX_SPINS", "1")
    e = ps.Ensemble(ps.default_params(num_chains=4096, precision=ps.F32, n=40, E0=1.0, Fz=0.5, seed=6))
    try:
        e.advance(70000)
        calls = [e.sync, e.summary, e.rolling, lambda: e.microstate(0), lambda: e.chain_state(0), e.reduce_host,
                 e.checkpoint, lambda: e.chain_means(), lambda: e.advance(10), e.summary]
        for f in calls:
            with pytest.raises(ps.PstatError) as ei:
                f()
            assert ei.value.code == -3 and "did not complete" in str(ei.value), str(ei.value)
    finally:
        e.close()
    monkeypatch.delenv("PSTAT_SEGMENTS")
    monkeypatch.delenv("PSTAT_MAX_SPINS")
    with ps.Ensemble(ps.default_params(num_chains=4096, precision=ps.F32, n=40, E0=1.0, Fz=0.5, seed=6)) as ok:
        ok.advance(7000)
        assert ok.summary().steps_per_chain == 7000


@pytest.mark.parametrize("moves,et,prec", [
    (0, 2, 1), (0, 2, 0), (0, 2, 2), (0, 1, 1), (0, 1, 0),            # fixed-force main: Ising (f64, f32, q16), all pairs
    (1, 2, 1), (1, 2, 0), (1, 1, 1), (1, 1, 0), (1, 3, 1), (1, 3, 0),  # clustering main: Ising, interacting, cutoff
])
def test_nonfinite_energy_counter_counts_every_such_proposal(ps, oracle, monkeypatch, moves, et, prec):
    """The positive path of pstat_summary.nan_rejects, where the non-finite value does not hinge on rounding: with
    --mlen 0 every monomer sits on one point, every pair term of inc/eap_chain.jl:200-207 is 0 * inf = NaN in any
    arithmetic, so EVERY proposal has a non-finite trial energy and is rejected (inc/acceptance.jl:29-39).  Every kernel
    that evaluates a pair energy must count all of them -- equal to the oracle's count on the same chains -- leave the
    angles where they were, carry the count in the reduction vector and through a checkpoint, and clear it with the
    averagers.  The non-interacting energy never sees r: its count stays zero."""
    nsteps, nchains, n = 700, 96, 43 if et == 2 else 21
    kw = dict(n=n, E0=1.0, K1=0.8, K2=0.1, Fz=0.3, b=0.0, seed=5, energy_type=et, cutoff_radius=7.5)
    if moves:
        kw.update(cluster_prob=0.5, bend_mod=0.2)
    homes = (("lds", "global", "wave") if moves else ("lds", "global")) if (prec == 1 and et == 2) else ("default",)
    for home in homes:
        if home != "default":
            monkeypatch.setenv("PSTAT_F64_STATE", home)
        op, pp = both(nsteps, num_chains=nchains, precision=prec, **kw)
        pp.move_set = moves
        with ps.Ensemble(pp) as e:
            first = e.chain_state(7)
            e.advance(nsteps)
            s = e.summary()
            assert s.nan_rejects == nsteps * nchains, (home, s.nan_rejects)
            assert s.acceptance_ratio == 0.0 and s.chains_collapsed == nchains
            last = e.chain_state(7)
            assert np.array_equal(first["theta"], last["theta"]) and np.array_equal(first["phi"], last["phi"])
            red = e.reduce_host()
            assert red[ps.NRED - 2] == nsteps * nchains and red[ps.NRED - 1] == nchains
            if prec == 1:     # the oracle's count on the same chains (f64: same streams, same proposals)
                for c in (0, 7, nchains - 1):
                    o = oracle.run(op, chain_id=c, mode="cluster" if moves else "fast")
                    assert o.nan_rejects == nsteps and o.nacc_total == 0
            blob = e.checkpoint()
            e.advance(100)
            assert e.summary().nan_rejects == (nsteps + 100) * nchains
            e.restore(blob)
            assert e.summary().nan_rejects == nsteps * nchains
            e.reset_averages()
            assert e.summary().nan_rejects == 0
            e.advance(50)
            assert e.summary().nan_rejects == 50 * nchains
    # the non-interacting energy under the same --mlen 0: never a non-finite proposal
    op, pp = both(nsteps, num_chains=nchains, precision=prec, **dict(kw, energy_type=0))
    pp.move_set = moves
    with ps.Ensemble(pp) as e:
        e.advance(nsteps)
        s = e.summary()
        assert s.nan_rejects == 0 and s.chains_collapsed == 0 and s.acceptance_ratio > 0.1


@pytest.mark.parametrize("home", ["lds", "global", "wave"])
def test_umbrella_weights_are_regauged_on_a_cold_drifting_chain(ps, oracle, monkeypatch, home):
    """--umbrella-sampling weights records by 1 / e^w; value / normaliser does not depend on the gauge of w, but a double does:
    on a cold chain relaxing from an aligned start the first-configuration gauge overflows (seen in tests/fuzz_cluster_wave.py:
    NaN averages on trajectories that equal the oracle's).  The kernels re-gauge at block boundaries (umbrella_regauge,
    csrc/pstat_math.h): averages stay finite and equal the oracle's, whose a-priori gauge happens to hold here."""
    monkeypatch.setenv("PSTAT_F64_STATE", home)
    kw = dict(n=128, E0=2.9683420157514435, K1=0.2620076967924292, K2=0.22902565388456103, mu=0.583542905988572,
              kT=0.032115462164346804, Fz=-0.1148105231949863, Fx=0.0, b=0.6435975332131777, chain_type=1, energy_type=0, umbrella=1,
              cluster_prob=0.0, steps_per_adjust=137, adj_scale=1.1, seed=337374529627, use_x0=1, x0_phi=2.30492766419305,
              x0_theta=0.4408010319490809, dx0_phi=0.05, dx0_theta=0.05)
    nsteps = 2500
    op, pp = both(nsteps, num_chains=3, precision=ps.F64, **kw)
    pp.move_set = ps.MOVES_CLUSTER
    with ps.Ensemble(pp) as e:
        e.advance(900); e.advance(nsteps - 900)
        for c in range(3):
            o = oracle.run(op, chain_id=c, mode="cluster", trace=True)
            g = e.chain_state(c)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["rng"], o.rng)
            avg = g["sums"] / g["normalizer"]
            assert np.all(np.isfinite(avg)) and np.isfinite(g["normalizer"]) and g["normalizer"] > 0, (home, c, g["normalizer"])
            assert np.all(np.isfinite(o.avg)) and 1e-290 < o.norm < 1e290      # (the reference's gauge holds on this one)
            np.testing.assert_allclose(avg, o.avg, rtol=1e-8, atol=1e-8)
        s = e.summary()
        assert np.all(np.isfinite(np.array(s.avg))) and np.all(np.isfinite(np.array(s.stderr)))


def test_umbrella_weights_survive_where_the_references_underflow(ps, oracle):
    """The other direction (found by tests/fuzz_packed.py): on this cold n = 114 chain the reference's a-priori gauge sends its
    normaliser to 1.5e-306 and then 0 -- NaN averages -- within 900 steps; the device's averages stay finite (same trajectories)."""
    kw = dict(n=114, E0=1.983535541035057, K1=1.0913260760266574, K2=0.42644128996853004, kT=0.3939373987076173, Fz=1.4196626815208542,
              Fx=-0.5148310727063605, b=0.5014461273162807, seed=565127474194, umbrella=1, steps_per_adjust=400, adj_scale=1.1, uniform_bits=23)
    op, pp = both(900, num_chains=7, precision=ps.F64, **kw)
    with ps.Ensemble(pp) as e:
        e.advance(900)
        lost = 0
        for c in range(7):
            o = oracle.run(op, chain_id=c, mode="fast", trace=True)
            g = e.chain_state(c)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["rng"], o.rng)
            assert np.all(np.isfinite(g["sums"] / g["normalizer"])) and g["normalizer"] > 1e-200, (c, g["normalizer"])
            lost += int(not (np.all(np.isfinite(o.avg)) and abs(o.norm) > 1e-290))
        assert lost >= 1          # (otherwise this test no longer shows what it is meant to)
