"""GPU parity tests (run on the MI355X box with -m gpu).  Everything goes through the C ABI
(libpstat.so via ctypes); the CPU oracle is only the checker.

Tolerances:
 * f64 kernel vs oracle, same stream: (theta, phi), generator state, acceptance counts and step
   sizes BIT-EXACT (they depend only on additions, clamps and accept decisions); running sums to
   1e-9 relative (device sin/cos/exp/log differ from glibc's in the last ulp).
 * f32 kernel: statistical -- pooled ensemble means within 4 standard errors of the oracle's pooled
   means (two-sample z), and of the f64 kernel's.
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


def _bit_parity(ps, oracle, nsteps, nchains, obs_tol=(1e-9, 1e-9), **kw):
    """Trajectory (angles, generator, counters, step sizes): bit for bit.  Observables: to `obs_tol` = (rtol, atol) -- the
    kernels' energies are rounded differently from the oracle's literal ones in the last bits (fused multiply-adds, the
    rsq-Newton pair term), and a running U that has passed through large Ising pair terms keeps their absolute rounding."""
    op, pp = both(nsteps, num_chains=nchains, precision=ps.F64, **kw)
    with ps.Ensemble(pp) as e:
        e.advance(nsteps)
        e.sync()
        nan_oracle = nacc_all = 0
        for c in range(nchains):
            o = oracle.run(op, chain_id=c, mode="fast", trace=True)
            g = e.chain_state(c)
            nacc_all += g["nacc_total"]
            assert np.array_equal(g["theta"], o.final_theta), f"theta differs, chain {c}"
            assert np.array_equal(g["phi"], o.final_phi), f"phi differs, chain {c}"
            assert np.array_equal(g["rng"], o.rng), f"rng differs, chain {c}"
            assert g["nacc_total"] == o.nacc_total
            assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step
            np.testing.assert_allclose(g["sums"], o.sums, rtol=1e-9, atol=1e-7)
            np.testing.assert_allclose(e.microstate(c), np.r_[o.r, o.p, o.U], rtol=obs_tol[0], atol=obs_tol[1])
            nan_oracle += o.nan_rejects
        # Proposals whose trial energy was not finite (include/pstat.h, pstat_summary.nan_rejects): the DEVICE's arithmetic,
        # so not equal to the oracle's count in general -- two pole-clamped monomers land on exactly one point under the
        # reference's cumsum (0 * inf = NaN) and 1e-17 apart on the device (a finite 1e50); either way sin(theta') = 0
        # rejects the proposal.  What must hold: a non-finite proposal is never accepted, and the device, whose positions
        # never cancel exactly where the cumsum's do, sees no more of them than the oracle on these seeded cases.
        dev = e.summary().nan_rejects
        assert dev <= nsteps * nchains - nacc_all, (dev, nsteps * nchains - nacc_all)
        assert dev <= nan_oracle, (dev, nan_oracle)
        if kw.get("energy_type", 0) == 0:
            assert dev == 0 and nan_oracle == 0


@pytest.mark.parametrize("rng", [0, 1])
def test_f64_bit_parity_config1(ps, oracle, rng):
    # BASELINE configs[0]: n=20, E0=0, Fz=1, kT=1 (adaptation active: 2500-step windows); both generators
    _bit_parity(ps, oracle, 12000, 70, n=20, E0=0.0, Fz=1.0, kT=1.0, seed=11, rng=rng)


def test_f64_bit_parity_dielectric_fx_flips(ps, oracle):
    _bit_parity(ps, oracle, 6000, 64, n=33, E0=1.5, K1=0.7, K2=0.3, Fz=0.4, Fx=0.3, kT=0.7, b=1.3,
                do_flips=1, seed=5, steps_per_adjust=500, rng=1)
    # sharded ids far apart: the seeding must agree for large chain ids too -- MWC64X (skip-ahead) up to its
    # disjointness bound of 2^22 ids, xoshiro128++ (Philox-seeded) for any 64-bit id
    for rng, id0 in ((0, (1 << 22) - 64), (1, (1 << 33) + 5)):
        op, pp = both(800, num_chains=64, precision=ps.F64, chain_id0=id0, n=10, E0=1.0, Fz=0.5, seed=5, rng=rng)
        with ps.Ensemble(pp) as e:
            e.advance(800)
            o = oracle.run(op, chain_id=id0 + 63, mode="fast", trace=True)
            g = e.chain_state(63)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["rng"], o.rng)
    # beyond the bound the MWC64X streams would wrap onto each other: refused, not silently accepted
    with pytest.raises(ps.PstatError, match="MWC64X streams are disjoint only"):
        ps.Ensemble(ps.default_params(num_chains=64, chain_id0=(1 << 22) - 63, n=10))


def test_f64_bit_parity_polar(ps, oracle):
    _bit_parity(ps, oracle, 6000, 64, n=100, E0=1.0, mu=1.0, Fz=1.0, chain_type=1, seed=3)


def test_f64_bit_parity_ising(ps, oracle):
    _bit_parity(ps, oracle, 4000, 64, n=24, E0=1.0, K1=1.0, Fz=0.25, energy_type=2, seed=9,
                steps_per_adjust=400)
    # n > 40: the cells live in LDS + memory and the neighbours are fetched with the row (run_segment, ST = 2)
    _bit_parity(ps, oracle, 3000, 70, obs_tol=(1e-7, 1e-5), n=90, E0=1.0, K1=0.3, K2=0.05, Fz=0.25, Fx=0.1, energy_type=2,
                seed=10, steps_per_adjust=400)


@pytest.mark.parametrize("rng", [0, 1])
def test_f64_bit_parity_interacting(ps, oracle, rng):
    # BASELINE configs[3] family: dipole-dipole interacting dielectric chain, one chain per wavefront
    _bit_parity(ps, oracle, 1500, 6, n=64, E0=1.0, K1=1.0, K2=0.0, Fz=0.5, energy_type=1, seed=13,
                steps_per_adjust=300, rng=rng)


def test_f64_bit_parity_interacting_short_polar_flips(ps, oracle):
    _bit_parity(ps, oracle, 1500, 5, n=23, E0=1.2, mu=0.8, Fz=0.3, Fx=0.2, chain_type=1, energy_type=1,
                do_flips=1, seed=14, steps_per_adjust=250)


@pytest.mark.parametrize("n", [100, 130, 200])
def test_f64_bit_parity_interacting_long_chains(ps, oracle, n):
    """n = 100 and 200 are what the reference's interacting sweeps run (run/interacting_*_study.jl):
    2 or 4 monomers per lane."""
    _bit_parity(ps, oracle, 400, 3, n=n, E0=1.0, K1=1.0, K2=0.1, Fz=0.5, Fx=0.1, energy_type=1, seed=23,
                steps_per_adjust=100)


@pytest.mark.parametrize("n,kw", [(300, dict(K2=0.1, Fx=0.1)), (512, dict(chain_type=1, mu=0.6, do_flips=1)), (257, dict(rng=1))],
                         ids=["n300", "n512-polar-flips", "n257-xoshiro"])
def test_f64_bit_parity_interacting_eight_monomers_per_lane(ps, oracle, n, kw):
    """Beyond the reference's own sweeps (n <= 200): 257 <= n <= 512 runs eight monomers per lane in the fixed-force main
    too, bit-identical to the oracle; n = 513 is refused by both mains."""
    _bit_parity(ps, oracle, 160, 2, n=n, E0=1.0, K1=1.0, Fz=0.5, energy_type=1, seed=29, steps_per_adjust=50, **kw)
    for move_set in (ps.MOVES_SINGLE, ps.MOVES_CLUSTER):
        with pytest.raises(ps.PstatError) as ei:
            ps.Ensemble(ps.default_params(n=513, energy_type=1, move_set=move_set))
        assert ei.value.code == -4 and "512" in str(ei.value)


def test_f64_interacting_umbrella_and_reinit(ps, oracle):
    """The remaining options of the in-scope main on the interacting kernel: --umbrella-sampling and
    --num-inits (forced and Metropolis re-initialisation, with the acceptor's stale cache)."""
    for force, umb, n in ((1, 0, 20), (0, 0, 20), (0, 1, 70)):
        nsteps, inits = 600, 3
        op, pp = both(nsteps, num_chains=5, precision=ps.F64, num_inits=inits, force_init=force, umbrella=umb,
                      n=n, E0=1.0, K1=1.0, K2=0.1, Fz=0.3, energy_type=1, seed=37, steps_per_adjust=200)
        with ps.Ensemble(pp) as e:
            for k in range(inits):
                e.advance(nsteps)
                if k + 1 < inits:
                    e.reinit(bool(force))
            for c in (0, 4):
                o = oracle.run(op, chain_id=c, mode="fast", trace=True)
                g = e.chain_state(c)
                assert np.array_equal(g["theta"], o.final_theta), (force, umb, c)
                assert np.array_equal(g["rng"], o.rng)
                assert g["nacc_total"] == o.nacc_total
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-8, atol=1e-8)


def test_f32_interacting_statistical_parity(ps, oracle):
    """f32 interacting kernel vs CPU oracle (faithful = full O(n^2) recompute) under the same
    protocol: pooled means within 4.5 sigma."""
    nsteps, nch = 4000, 1024
    op, pp = both(nsteps, num_chains=nch, precision=ps.F32, n=32, E0=1.0, K1=1.0, Fz=0.5, energy_type=1, seed=15)
    with ps.Ensemble(pp) as e:
        e.advance(nsteps)
        g_avg, g_se = e.rolling()
    osums, onorm, _ = oracle.run_many(op, 5_000_000, 128, nthreads=8, mode="faithful")
    o_avg, o_se = pooled(osums, onorm)
    z = (g_avg - o_avg) / np.sqrt(g_se ** 2 + o_se ** 2 + 1e-300)
    assert np.all(np.abs(z) < 4.5), dict(zip(ps.OBS_NAMES, np.round(z, 2)))


def test_f64_umbrella_parity(ps, oracle):
    """--umbrella-sampling: the weight function enters the acceptance and every record is weighted
    by 1/e^w (inc/average.jl:52-124).  Trajectory bit-exact; per-chain averages (value/normalizer,
    in which the gauge constant cancels) to 1e-9."""
    for kw in (dict(n=14, E0=1.0, K1=1.0, K2=0.2, Fz=0.5), dict(n=9, E0=1.0, mu=0.5, Fz=0.2, chain_type=1, energy_type=2)):
        nsteps = 4000
        op, pp = both(nsteps, num_chains=64, precision=ps.F64, umbrella=1, seed=19, steps_per_adjust=500, **kw)
        with ps.Ensemble(pp) as e:
            e.advance(nsteps)
            for c in (0, 31, 63):
                o = oracle.run(op, chain_id=c, mode="fast", trace=True)
                g = e.chain_state(c)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi)
                assert g["nacc_total"] == o.nacc_total
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-9, atol=1e-9)


def test_f32_umbrella_matches_plain_sampling(ps):
    """The reference's own cross-check (run/noninteracting-compare-with-clustering_2021-09-24.jl):
    umbrella-weighted and plain sampling estimate the same averages."""
    out = {}
    for umb in (0, 1):
        pp = ps.default_params(num_chains=2048, precision=ps.F32, n=10, E0=0.8, K1=1.0, Fz=0.3, seed=8 + umb,
                               umbrella=umb)
        with ps.Ensemble(pp) as e:
            e.advance(40000)
            out[umb] = e.rolling()
    z = (out[0][0] - out[1][0]) / np.sqrt(out[0][1] ** 2 + out[1][1] ** 2 + 1e-300)
    assert np.all(np.abs(z) < 4.5), z


def test_f64_bit_parity_no_adaptation_single_monomer(ps, oracle):
    _bit_parity(ps, oracle, 3000, 3, n=1, E0=2.0, K1=1.0, Fz=0.5, adj_scale=1.0, seed=2)


def test_segments_checkpoint_and_sharding_invariance(ps, oracle):
    """advance(a)+advance(b) == advance(a+b); checkpoint/restore resumes exactly; chains are
    identified by global id, so a shard [32,64) of a 64-chain job reproduces those chains."""
    op, pp = both(5000, num_chains=64, precision=ps.F32, n=40, E0=1.0, Fz=0.5, seed=21)
    with ps.Ensemble(pp) as whole, ps.Ensemble(pp) as parts:
        whole.advance(5000)
        parts.advance(1234)
        blob = parts.checkpoint()
        parts.advance(777)           # diverge, then rewind
        parts.restore(blob)
        parts.advance(5000 - 1234)
        a, b = whole.chain_state(17), parts.chain_state(17)
        for k in ("theta", "phi", "rng"):
            assert np.array_equal(a[k], b[k]), k
        assert a["nacc_total"] == b["nacc_total"]
        np.testing.assert_allclose(a["sums"], b["sums"], rtol=1e-5, atol=1e-6 * 5000 * 40)
    _, shard = both(5000, num_chains=32, chain_id0=32, precision=ps.F32, n=40, E0=1.0, Fz=0.5, seed=21)
    with ps.Ensemble(pp) as whole, ps.Ensemble(shard) as half:
        whole.advance(5000)
        half.advance(5000)
        a, b = whole.chain_state(32 + 5), half.chain_state(5)
        assert np.array_equal(a["theta"], b["theta"]) and np.array_equal(a["phi"], b["phi"])


@pytest.mark.parametrize("prec,n,et", [(0, 40, 0), (2, 40, 0), (1, 40, 0), (1, 60, 0), (1, 60, 2)],
                         ids=["f32", "q16", "f64-lds", "f64-memory", "f64-memory-ising"])
def test_time_segments_match_single_launch(ps, monkeypatch, prec, n, et):
    """The persistent sweep kernel splits a launch into (chain block, time segment) jobs.  With every
    job co-resident (64 blocks x 3 segments) later segments really wait on their predecessors; the
    result must be bit-identical to the unsegmented launch.  For the f64 kernel with its cells in memory a segment
    boundary is also a spill of the working buffer to the checkpoint layout and a refill, possibly on another CU."""
    pp = ps.default_params(num_chains=4096, precision=prec, n=n, E0=1.0, Fz=0.5, seed=6, energy_type=et, K1=0.3 if et else 1.0)
    monkeypatch.setenv("PSTAT_MAX_SPINS", str(1 << 19))     # fail within ~1 s instead of hanging
    states = {}
    for nseg in ("1", "3", "7"):
        monkeypatch.setenv("PSTAT_SEGMENTS", nseg)
        with ps.Ensemble(pp) as e:
            e.advance(9000)
            e.sync()
            states[nseg] = [e.chain_state(c) for c in (0, 63, 64, 1000, 4095)]
            avg, _ = e.rolling()
            states[nseg + "avg"] = avg
    for nseg in ("3", "7"):
        for a, b in zip(states["1"], states[nseg]):
            for k in ("theta", "phi", "rng"):
                assert np.array_equal(a[k], b[k]), (nseg, k)
            assert a["nacc_total"] == b["nacc_total"] and a["phi_step"] == b["phi_step"]
            # f32 block partials are flushed at different points -> rounding-level differences,
            # measured against the size of the summed terms (sum r1 cancels to ~0)
            np.testing.assert_allclose(a["sums"], b["sums"], rtol=1e-5, atol=1e-6 * 9000 * 40)
        np.testing.assert_allclose(states["1avg"], states[nseg + "avg"], rtol=1e-6, atol=1e-5)


def test_batched_cases_match_single_case_handles(ps):
    base = dict(num_chains=128, precision=ps.F32, n=30, E0=1.0, K1=1.0, seed=4)
    cases = [ps.default_params(Fz=f, **base) for f in (0.0, 0.5, 2.0)]
    with ps.Ensemble(cases) as batch:
        batch.advance(3000)
        for i, c in enumerate(cases):
            with ps.Ensemble(c) as single:
                single.advance(3000)
                a, b = batch.chain_state(i * 128 + 7), single.chain_state(7)
                assert np.array_equal(a["theta"], b["theta"])
                avg_b, _ = batch.rolling(i)
                avg_s, _ = single.rolling(-1)
                np.testing.assert_allclose(avg_b, avg_s, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("rng,prec", [(0, 0), (1, 0), (0, 2)])
def test_f32_statistical_parity_config2(ps, oracle, golden, rng, prec):
    """BASELINE configs[1] at Fz=1: n=100 dielectric.  f32 kernel (both generators) and the
    lattice-state kernel (prec 2 = PSTAT_Q16) vs the CPU oracle under the SAME protocol (no burn-in,
    adaptation on): pooled means agree within 4 sigma."""
    nsteps, nch = 20000, 4096
    op, pp = both(nsteps, num_chains=nch, precision=prec, n=100, E0=1.0, K1=1.0, K2=0.0, Fz=1.0, seed=77, rng=rng)
    with ps.Ensemble(pp) as e:
        e.advance(nsteps)
        g_avg, g_se = e.rolling()
        s = e.summary()
    osums, onorm, onacc = oracle.run_many(op, 10_000_000, 256, nthreads=8, mode="fast")
    o_avg, o_se = pooled(osums, onorm)
    z = (g_avg - o_avg) / np.sqrt(g_se ** 2 + o_se ** 2 + 1e-300)
    assert np.all(np.abs(z) < 4.0), dict(zip(ps.OBS_NAMES, np.round(z, 2)))
    assert abs(s.acceptance_ratio - onacc.mean() / nsteps) < 4 * (s.ar_stderr + onacc.std() / nsteps / 16)
    # sanity against equilibrium; the estimator averages from step 1 of a random start, so at
    # N = 20 tau it still sits ~(tau/N) r_eq ~ 5-7 % below it (the oracle shows the same offset)
    eq = golden["cfg2_n100_E0_1_K1_1_Fz1"]["avg"]
    assert abs(g_avg[2] - eq["r3"]) < 0.10 * eq["r3"]
    assert abs(g_avg[9] - eq["p3"]) < 0.10 * eq["p3"]


@pytest.mark.parametrize("kw", [dict(n=64, E0=1.0, K1=1.0, Fz=0.5),
                                # the f32 Ising step sums its neighbour terms on n_i + n_j and scales by 1/|b/2|^3 once;
                                # r is carried in units of b: a bond length != 1, Fx != 0 and both chain types
                                dict(n=40, E0=1.0, K1=0.6, K2=0.2, Fz=0.3, Fx=0.2, b=1.3, kT=1.5, energy_type=2),
                                dict(n=40, E0=0.5, mu=0.8, Fz=0.4, b=0.7, kT=2.0, energy_type=2, chain_type=1),
                                dict(n=33, E0=1.2, mu=0.6, Fz=0.3, Fx=0.5, b=0.7, kT=0.8, chain_type=1, do_flips=1)],
                         ids=["noninteracting", "ising-dielectric-b1.3", "ising-polar-b0.7", "polar-fx-flips-b0.7"])
def test_f32_vs_f64_same_seeds(ps, kw):
    """Same seeds => same proposals, so the two precisions differ by arithmetic only."""
    nsteps, nch = 20000, 1024
    out = {}
    for prec in (ps.F32, ps.F64):
        pp = ps.default_params(num_chains=nch, precision=prec, seed=5, **kw)
        with ps.Ensemble(pp) as e:
            e.advance(nsteps)
            out[prec] = e.rolling()
    (a32, s32), (a64, s64) = out[ps.F32], out[ps.F64]
    z = (a32 - a64) / np.sqrt(s32 ** 2 + s64 ** 2 + 1e-300)
    assert np.all(np.abs(z) < 4.0), z


@pytest.mark.parametrize("prec", [0, 2])
def test_long_run_matches_closed_form_config3(ps, golden, prec):
    """BASELINE configs[2] point (polar, E0=1, mu=1, Fz=1): with N >> tau the protocol's transient
    bias is below the pooled standard error budgeted here (f32 and lattice-state kernels)."""
    nsteps, nch = 200000, 2048
    pp = ps.default_params(num_chains=nch, precision=prec, n=100, E0=1.0, mu=1.0, Fz=1.0,
                           chain_type=ps.POLAR, seed=123)
    with ps.Ensemble(pp) as e:
        e.advance(nsteps)
        avg, se = e.rolling()
    eq = golden["cfg3_polar_n100_E0_1_mu1_Fz1"]["avg"]
    # transient: (tau/N)(r_eq - r_0) with tau ~ 10 n = 1000 steps -> ~0.5% of r_eq
    assert abs(avg[2] - eq["r3"]) < 0.01 * eq["r3"] + 4 * se[2]
    assert abs(avg[9] - eq["p3"]) < 0.01 * eq["p3"] + 4 * se[9]
    assert abs(avg[14] - eq["U"]) < 0.01 * abs(eq["U"]) + 4 * se[14]


def test_q16_lattice_state_against_f64(ps, oracle):
    """PSTAT_Q16 vs the f64 kernel on options the other q16 tests do not reach: Fx != 0, --do-flips,
    Ising coupling, re-initialisation.  Different lattices => statistical comparison (4.5 sigma)."""
    cases = [dict(n=30, E0=1.5, K1=0.7, K2=0.3, Fz=0.4, Fx=0.3, kT=0.7, b=1.3, do_flips=1),
             dict(n=24, E0=1.0, K1=1.0, Fz=0.25, energy_type=2),
             dict(n=20, E0=1.0, mu=1.0, Fz=1.0, chain_type=1)]
    for kw in cases:
        out = {}
        for prec in (ps.F64, ps.Q16):
            pp = ps.default_params(num_chains=2048, precision=prec, seed=41 + prec, **kw)
            with ps.Ensemble(pp) as e:
                e.advance(15000)
                e.reinit(True)
                e.advance(15000)
                sm = e.summary()
                out[prec] = e.rolling() + (sm.acceptance_ratio, sm.ar_stderr)
        (a, sa, ara, sea), (b, sb, arb, seb) = out[ps.F64], out[ps.Q16]
        z = (a - b) / np.sqrt(sa ** 2 + sb ** 2 + 1e-300)
        assert np.all(np.abs(z) < 4.5), (kw, dict(zip(ps.OBS_NAMES, np.round(z, 2))))
        # after a forced re-init the reference's stale acceptor cache freezes many chains, so the
        # per-chain acceptance ratio is bimodal: compare with its across-chain standard error
        assert abs(ara - arb) < 4.5 * np.hypot(sea, seb) + 1e-4, (kw, ara, arb, sea, seb)
        # lattice angles really are on the lattice: theta = pi (k + 1/2) / 65536
        with ps.Ensemble(ps.default_params(num_chains=64, precision=ps.Q16, seed=3, **kw)) as e:
            e.advance(500)
            st = e.chain_state(5)
            k = st["theta"] * 65536 / np.pi - 0.5
            assert np.allclose(k, np.round(k), atol=1e-6) and k.min() >= 0 and k.max() <= 65535


def test_reinit_force_and_metropolis(ps, oracle):
    """--num-inits 3: f64 kernel vs oracle fast mode, including the acceptor's stale cache."""
    for force in (1, 0):
        nsteps, inits = 1500, 3
        op, pp = both(nsteps, num_chains=64, precision=ps.F64, num_inits=inits, force_init=force,
                      n=12, E0=1.0, K1=1.0, Fz=0.3, seed=31, steps_per_adjust=500)
        with ps.Ensemble(pp) as e:
            for k in range(inits):
                e.advance(nsteps)
                if k + 1 < inits:
                    e.reinit(bool(force))
            for c in (0, 13, 63):
                o = oracle.run(op, chain_id=c, mode="fast", trace=True)
                g = e.chain_state(c)
                assert np.array_equal(g["theta"], o.final_theta), (force, c)
                assert np.array_equal(g["rng"], o.rng)
                assert g["nacc_total"] == o.nacc_total
                np.testing.assert_allclose(g["sums"], o.sums, rtol=1e-9, atol=1e-7)


def test_reinit_under_umbrella_sampling_carries_the_weight_in_the_stale_cache(ps, oracle):
    """the acceptor's cached log-density includes the umbrella weight w = sum(u) * wscale
    (inc/acceptance.jl:13-16, inc/average.jl:104-124); after an adopted re-initialisation the comparisons are offset by
    lp_old - lp_new INCLUDING w.  With E0 = 3 the weight differs by O(10) between two random configurations, so a re-init
    kernel that left it out (round 1) takes different decisions from the oracle within a few steps.  Sweep kernels, LDS
    (n = 24) and state-in-memory (n = 60) variants, non-interacting and Ising, forced and Metropolis re-init."""
    for force, n, et in ((1, 24, 0), (0, 24, 0), (1, 60, 0), (1, 24, 2), (1, 60, 2)):
        nsteps, inits = 900, 4
        kw = dict(n=n, E0=3.0, K1=1.0 if et == 0 else 0.2, K2=0.1 if et == 0 else 0.02, Fz=0.2, umbrella=1, energy_type=et,
                  seed=77, steps_per_adjust=300)
        op, pp = both(nsteps, num_chains=16, precision=ps.F64, num_inits=inits, force_init=force, **kw)
        with ps.Ensemble(pp) as e:
            for k in range(inits):
                e.advance(nsteps)
                if k + 1 < inits:
                    e.reinit(bool(force))
            dw = []
            for c in range(16):
                o = oracle.run(op, chain_id=c, mode="fast", trace=True)
                g = e.chain_state(c)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (force, n, et, c)
                assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total, (force, n, et, c)
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-8, atol=1e-8)


def test_errors_are_loud(ps):
    with pytest.raises(ps.PstatError):
        ps.Ensemble(ps.default_params(n=0))
    with pytest.raises(ps.PstatError):
        ps.Ensemble(ps.default_params(kT=0.0))
    with pytest.raises(ps.PstatError) as ei:
        ps.Ensemble(ps.default_params(energy_type=ps.INTERACTING, n=600))   # > 8 monomers per lane
    assert ei.value.code == -4


def test_full_size_config2_properties(ps, golden):
    """BASELINE configs[1] at its full size (65 536 chains x 1e5 steps, n = 100, Fz = 1), checked
    through size-independent properties: closed-form equilibrium values within the no-burn-in
    transient (tau ~ 10 n = 1e3 steps => ~1 % low, ~2 % for the squares) plus 5 pooled standard errors; the sum rules
    <r.r> = sum_j <r_j^2>, <p.p> = sum_j <p_j^2>; symmetry <r_x> = <r_y> = 0; variances positive;
    every chain made exactly 1e5 attempts."""
    nsteps, nch = 100000, 65536
    pp = ps.default_params(num_chains=nch, precision=ps.F32, n=100, E0=1.0, K1=1.0, K2=0.0, Fz=1.0, seed=20260501)
    with ps.Ensemble(pp) as e:
        e.advance(nsteps)
        s = e.summary()
        st = e.chain_state(nch - 1)
    avg, se = np.array(s.avg), np.array(s.stderr)
    eq = golden["cfg2_n100_E0_1_K1_1_Fz1"]["avg"]
    assert s.num_chains == nch and s.steps_per_chain == nsteps and s.attempted_updates == float(nch) * nsteps
    assert st["steps_recorded"] == nsteps and 0 < st["nacc_total"] < nsteps
    for k, name in enumerate(ps.OBS_NAMES):
        want = eq[name]
        tol = (0.03 if name.endswith("sq") else 0.015) * (abs(want) + 1.0) + 5 * se[k]
        assert abs(avg[k] - want) < tol, (name, avg[k], want, se[k])
    assert avg[6] == pytest.approx(avg[3] + avg[4] + avg[5], rel=1e-12)
    assert avg[13] == pytest.approx(avg[10] + avg[11] + avg[12], rel=1e-12)
    assert abs(avg[0]) < 5 * se[0] and abs(avg[1]) < 5 * se[1]
    assert avg[5] > avg[2] ** 2 and avg[15] > avg[14] ** 2
    assert 0.5 < s.acceptance_ratio < 0.7


def test_full_size_config4_properties(ps):
    """BASELINE configs[3] at its full size (interacting dielectric, n = 64, 16 384 chains): launch
    splitting and checkpoint/restore are bit-invariant; with E0 = 0 a dielectric chain carries no dipoles, the
    pair energy vanishes and the all-pairs kernel must sit on the freely-jointed-chain closed form
    <r_z> = n b (coth x - 1/x), x = F b / kT (its O(n^2) machinery still runs)."""
    nch, n = 16384, 64
    pp = ps.default_params(num_chains=nch, precision=ps.F32, n=n, E0=1.0, K1=1.0, Fz=0.5, energy_type=ps.INTERACTING, seed=44)
    with ps.Ensemble(pp) as a, ps.Ensemble(pp) as b:
        a.advance(600)
        blob = a.checkpoint()
        a.advance(400)
        b.advance(250); b.advance(350)
        b.restore(blob)
        b.advance(400)
        for c in (0, 4097, nch - 1):
            ga, gb = a.chain_state(c), b.chain_state(c)
            assert np.array_equal(ga["theta"], gb["theta"]) and np.array_equal(ga["rng"], gb["rng"])
            assert np.array_equal(ga["sums"], gb["sums"]) and ga["nacc_total"] == gb["nacc_total"]
        assert a.summary().attempted_updates == float(nch) * 1000
    pp = ps.default_params(num_chains=nch, precision=ps.F32, n=n, E0=0.0, K1=1.0, Fz=1.0, energy_type=ps.INTERACTING, seed=45)
    with ps.Ensemble(pp) as e:
        e.advance(6000)
        e.reset_averages()
        e.advance(4000)
        s = e.summary()
    want = n * (1 / np.tanh(1.0) - 1.0)
    assert abs(s.avg[2] - want) < 5 * s.stderr[2] + 2e-3 * want, (s.avg[2], want, s.stderr[2])
    assert abs(s.avg[14] + want) < 5 * s.stderr[14] + 2e-3 * want          # U = -F r_z when E0 = 0
    assert s.avg[13] == 0.0                                                   # no dipoles at all


def test_full_size_config5_grid_properties(ps):
    """BASELINE configs[4]: the 546-point (E0, kT) grid at n = 200 in ONE launch.  A grid point inside the
    batch follows bit-identically the trajectory of the same case run alone (cases never interact); E0 = 0 columns sit on
    <r_z> = 0 and <r^2> = n b^2 for every temperature; the acceptance ratio stays inside the adaptation band
    or at its caps."""
    grid = [(0.2 * i, 10.0 ** (-2 + 0.2 * j)) for i in range(26) for j in range(21)]
    mk = lambda k: ps.default_params(n=200, E0=grid[k][0], kT=grid[k][1], K1=1.0, num_chains=64, seed=1000 + k,
                                     precision=ps.F32, energy_type=ps.ISING)
    with ps.Ensemble([mk(k) for k in range(len(grid))]) as e:
        e.advance(6000)
        picks = (0, 20, 273, 545)
        states = {k: e.chain_state(k * 64 + 63) for k in picks}
        sums = {k: e.summary(k) for k in (0, 10, 20)}           # E0 = 0, three temperatures
    for k in picks:
        with ps.Ensemble(mk(k)) as one:
            one.advance(6000)
            g = one.chain_state(63)
            assert np.array_equal(g["theta"], states[k]["theta"]) and np.array_equal(g["rng"], states[k]["rng"]), k
            # (the f32 block partials are folded, and the f32 running totals re-derived from the angles, at
            # segment boundaries, and the batch is cut into different time segments than a lone case: same
            # trajectory, sums equal to f32 rounding of the totals)
            np.testing.assert_allclose(g["sums"], states[k]["sums"], rtol=3e-5, atol=0.05)
    for k, s in sums.items():
        assert abs(s.avg[2]) < 5 * s.stderr[2] + 1e-9
        assert abs(s.avg[6] - 200.0) < 5 * s.stderr[6] + 2.0, (k, s.avg[6], s.stderr[6])
        assert 0.1 < s.acceptance_ratio <= 1.0


def test_edge_shapes(ps, oracle):
    """Ragged and degenerate shapes: one chain, chain counts that do not fill a wave, n = 1 and 2,
    one-step launches, zero-step launches, a field so strong that almost nothing is accepted."""
    for n, nch, nsteps, kw in [(1, 1, 1, {}), (2, 1, 7, {}), (2, 65, 333, dict(do_flips=1)), (5, 130, 1, {}),
                               (3, 7, 2501, dict(E0=30.0, K1=1.0, Fz=0.0, kT=0.05))]:
        op, pp = both(nsteps, num_chains=nch, precision=ps.F64, n=n, seed=77, **({"E0": 1.0, "Fz": 0.5} | kw))
        with ps.Ensemble(pp) as e:
            e.advance(0)
            e.advance(nsteps)
            for c in sorted({0, nch - 1}):
                o = oracle.run(op, chain_id=c, mode="fast", trace=True)
                g = e.chain_state(c)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (n, nch, c)
                assert g["nacc_total"] == o.nacc_total
            s = e.summary()
            assert s.num_chains == nch and np.all(np.isfinite(np.array(s.avg)))
            if nch == 1:
                assert np.all(np.array(s.stderr) == 0.0)     # one chain: no across-chain error


@pytest.mark.parametrize("prec", [0, 2, 1])
def test_equilibrium_after_burn_in_matches_closed_form_sharply(ps, golden, prec):
    """With a burn-in the estimator has no transient, so the pooled averages must equal the
    closed-form equilibrium values within their own (tiny) standard error: 65 536 chains x 5e4
    recorded steps give ~7e-5 relative resolution on <r_z> -- a bias test of the f32 and lattice
    arithmetic (and of the generator) three orders of magnitude sharper than the 1e-2-level tests."""
    nch = 65536 if prec != 1 else 8192
    pp = ps.default_params(num_chains=nch, precision=prec, n=100, E0=1.0, K1=1.0, K2=0.0, Fz=1.0, seed=424242)
    with ps.Ensemble(pp) as e:
        e.set_kT(3.0)
        e.advance(10000)           # a short hot rung, then the target temperature
        e.set_kT(1.0)
        e.advance(30000)
        e.reset_averages()
        e.advance(50000)
        s = e.summary()
    avg, se = np.array(s.avg), np.array(s.stderr)
    assert s.steps_per_chain == 50000
    eq = golden["cfg2_n100_E0_1_K1_1_Fz1"]["avg"]
    z = np.array([(avg[k] - eq[name]) / (se[k] + 1e-300) for k, name in enumerate(ps.OBS_NAMES)])
    assert np.all(np.abs(z) < 5.0), dict(zip(ps.OBS_NAMES, np.round(z, 2)))
    assert se[2] / eq["r3"] < (3e-4 if prec == 1 else 1e-4)    # the test really is that sharp


def test_f64_bit_parity_random_configurations(ps, oracle):
    """Seeded fuzz over the option space of the fixed-force main (chain type, energy type, field, forces,
    temperature, monomer length, flips, umbrella, adaptation cadence, generator, chain length, re-inits):
    every configuration must follow the oracle's trajectory bit for bit."""
    rng = np.random.default_rng(20260501)
    for trial in range(80):
        et = int(rng.choice([0, 0, 2, 1]))
        n = int(rng.integers(1, 40)) if et != 1 else int(rng.integers(2, 70))
        kw = dict(n=n, E0=float(rng.uniform(0, 2)), K1=float(rng.uniform(0, 1.2)), K2=float(rng.uniform(0, 0.5)),
                  mu=float(rng.uniform(0.01, 0.6)), kT=float(10 ** rng.uniform(-0.5, 0.7)), Fz=float(rng.uniform(-1, 2)),
                  Fx=float(rng.choice([0.0, rng.uniform(-1, 1)])), b=float(rng.uniform(0.5, 2.0)),
                  chain_type=int(rng.integers(0, 2)), energy_type=et, do_flips=int(rng.integers(0, 2)),
                  umbrella=int(rng.integers(0, 2)), steps_per_adjust=int(rng.choice([50, 137, 400, 2500])),
                  adj_scale=float(rng.choice([1.0, 1.1, 1.3])), rng=int(rng.integers(0, 2)), seed=int(rng.integers(0, 2 ** 40)))
        if et == 2:      # keep the Ising coupling weak: collapsed chains amplify rounding into decisions
            kw.update(K1=kw["K1"] * 0.3, K2=kw["K2"] * 0.2, mu=kw["mu"] * 0.3)
        nsteps = 400 if et == 1 else 1200
        inits = int(rng.choice([1, 1, 2]))
        force = int(rng.integers(0, 2))
        cid = int(rng.integers(0, 2 ** 33))
        if kw["rng"] == 0:
            cid %= (1 << 22) - 3      # MWC64X: ids below its stream-disjointness bound
        op, pp = both(nsteps, num_chains=3, precision=ps.F64, num_inits=inits, force_init=force, chain_id0=cid, **kw)
        with ps.Ensemble(pp) as e:
            for k in range(inits):
                e.advance(nsteps)
                if k + 1 < inits:
                    e.reinit(bool(force))
            e.sync()
            for c in range(3):
                o = oracle.run(op, chain_id=pp.chain_id0 + c, mode="fast", trace=True)
                g = e.chain_state(c)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (trial, kw)
                assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total, (trial, kw)
                assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step, (trial, kw)
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-8, atol=1e-8, err_msg=str((trial, kw)))


@pytest.mark.parametrize("moves", [0, 1])
def test_f32_running_totals_do_not_drift(ps, moves):
    """The f32 kernels carry r, p, U as running totals of accepted differences and re-derive them from the
    angles every 16 384 steps: after 3e6 steps the reported microstate must still be the microstate of the
    stored angles (without the refresh the rounding errors random-walk to ~5e-3 here)."""
    n = 60
    pp = ps.default_params(n=n, E0=1.0, K1=1.0, K2=0.2, Fz=0.7, Fx=0.3, b=1.3, num_chains=64, precision=ps.F32, seed=77,
                           move_set=moves, cluster_prob=0.7, bend_mod=0.3 if moves else 0.0)
    with ps.Ensemble(pp) as e:
        e.advance(3_000_000)
        worst = 0.0
        for c in (0, 31, 63):
            g = e.chain_state(c)
            th, ph = g["theta"], g["phi"]
            nh = np.c_[np.cos(ph) * np.sin(th), np.sin(ph) * np.sin(th), np.cos(th)]
            a = (pp.K1 - pp.K2) * pp.E0 * nh[:, 2]
            mu = np.c_[a * nh[:, 0], a * nh[:, 1], a * nh[:, 2] + pp.K2 * pp.E0]
            r, p = pp.b * nh.sum(0), mu.sum(0)
            U = -0.5 * pp.E0 * mu[:, 2].sum() - (pp.Fx * r[0] + pp.Fz * r[2])
            if moves:
                psi = np.arccos(np.clip((nh[1:] * nh[:-1]).sum(1), -1, 1))
                U += 0.5 * pp.bend_mod * ((psi - pp.bend_angle) ** 2).sum()
            m = e.microstate(c)
            worst = max(worst, np.abs(m[:3] - r).max(), np.abs(m[3:6] - p).max(), abs(m[6] - U))
        assert worst < 1.2e-3, worst


@pytest.mark.parametrize("Fz", [0.0, 1.0, 5.0])
def test_f64_bit_parity_headline_instantiation(ps, oracle, Fz):
    """BASELINE configs[1] on the exact instantiation bench.py times -- sweep_kernel<double, Mwc64x, dielectric,
    non-interacting, no Fx, no rare options, cells in memory> at n = 100 -- directly against the oracle (not through the
    LDS variant): three points of the Fz sweep of run/noninteracting-compare-with-clustering_2021-09-24.jl:21, 70 chains
    (one full wave and one 6-lane wave), adaptation active.  Reference lines: mcmc_eap_chain.jl:276-328."""
    op, pp = both(6000, num_chains=70, precision=ps.F64, n=100, E0=1.0, K1=1.0, K2=0.0, Fz=Fz, kT=1.0, seed=20260501)
    with ps.Ensemble(pp) as e:
        assert "state in L2" in e.launch_info().kernel.decode()
        e.advance(2500); e.advance(3500)
        e.sync()
        for c in range(70):
            o = oracle.run(op, chain_id=c, mode="fast", trace=True)
            g = e.chain_state(c)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (Fz, c)
            assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total
            assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step
            np.testing.assert_allclose(g["sums"], o.sums, rtol=1e-9, atol=1e-7)
            np.testing.assert_allclose(e.microstate(c), np.r_[o.r, o.p, o.U], rtol=1e-9, atol=1e-9)
        assert e.summary().nan_rejects == 0


def test_f64_bit_parity_phase_grid_point_n200(ps, oracle):
    """BASELINE configs[4] at the default precision against the oracle: grid points of run/K1_E0-kT-phase.jl:21-24 (n = 200,
    K1 = 1, K2 = 0, b = 1, F = 0, Ising energy) batched in one handle like tools/phase_scan.py does; warm enough that
    no chain collapses in 1500 steps.  Cells in memory, three rows fetched per step (run_segment, ST = 2, Ising)."""
    grid = [(0.2, 10.0 ** 0.6), (1.0, 10.0 ** 1.0), (0.4, 10.0 ** 2.0)]
    cases = [ps.default_params(n=200, E0=E0, kT=kT, K1=1.0, K2=0.0, b=1.0, num_chains=66, seed=20260501 + k, precision=ps.F64,
                               energy_type=ps.ISING) for k, (E0, kT) in enumerate(grid)]
    with ps.Ensemble(cases) as e:
        assert "state in L2" in e.launch_info().kernel.decode()
        e.advance(1500)
        e.sync()
        for k, (E0, kT) in enumerate(grid):
            op, _ = both(1500, num_chains=66, precision=ps.F64, n=200, E0=E0, kT=kT, K1=1.0, K2=0.0, b=1.0, seed=20260501 + k,
                         energy_type=ps.ISING)
            for c in (0, 63, 64, 65):
                o = oracle.run(op, chain_id=c, mode="fast", trace=True)
                g = e.chain_state(k * 66 + c)
                assert abs(o.U) < 1e3 * 200 * kT, "collapsed: not a trajectory-parity case"
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (k, c)
                assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-7, atol=1e-7)


def test_f64_sweep_fuzz_over_both_state_homes(ps, oracle, monkeypatch):
    """A seeded 150-configuration slice of tests/fuzz_f64.py (the 3 000-case run of round 2 was a builder-run script):
    non-interacting and Ising energies, every option, both generators, chain lengths 1 ... 140, the cells forced into LDS
    or into memory (ST = 2: the two-step-deep pipeline and its forwarding, hit hardest by SHORT chains) with 0 / 1 / 5 /
    39 rows kept in LDS, a launch split and a re-init in every run: the oracle's trajectory bit for bit."""
    rng = np.random.default_rng(20260503)
    homes = {"lds": 0, "global": 0, "auto": 0}
    for trial in range(150):
        et = int(rng.choice([0, 0, 2]))
        where = str(rng.choice(["lds", "global", "global", "auto"]))
        n = int(rng.integers(1, 12)) if rng.random() < 0.4 else int(rng.integers(12, 141))
        kw = dict(n=n, E0=float(rng.uniform(0, 2)), K1=float(rng.uniform(0, 1.2)), K2=float(rng.uniform(0, 0.5)),
                  mu=float(rng.uniform(0.01, 0.6)), kT=float(10 ** rng.uniform(-0.5, 0.7)), Fz=float(rng.uniform(-1, 2)),
                  Fx=float(rng.choice([0.0, rng.uniform(-1, 1)])), b=float(rng.uniform(0.5, 2.0)),
                  chain_type=int(rng.integers(0, 2)), energy_type=et, do_flips=int(rng.integers(0, 2)),
                  umbrella=int(rng.integers(0, 2)), steps_per_adjust=int(rng.choice([50, 137, 400, 2500])),
                  adj_scale=float(rng.choice([1.0, 1.1, 1.3])), rng=int(rng.integers(0, 2)), seed=int(rng.integers(0, 2 ** 40)))
        if et == 2:      # keep the Ising coupling weak: collapsed chains amplify rounding into decisions
            kw.update(K1=kw["K1"] * 0.3, K2=kw["K2"] * 0.2, mu=kw["mu"] * 0.3)
        nsteps = int(rng.choice([700, 1500, 3001]))
        inits = int(rng.choice([1, 1, 2]))
        force = int(rng.integers(0, 2))
        cid = int(rng.integers(0, 2 ** 33))
        if kw["rng"] == 0:
            cid %= (1 << 22) - 140
        nch = int(rng.choice([3, 65, 130]))
        if where == "auto":
            monkeypatch.delenv("PSTAT_F64_STATE", raising=False)
        else:
            monkeypatch.setenv("PSTAT_F64_STATE", where)
        rows = int(rng.choice([0, 1, 5, 39]))
        monkeypatch.setenv("PSTAT_F64_LDS_ROWS", str(rows))
        op, pp = both(nsteps, num_chains=nch, precision=ps.F64, num_inits=inits, force_init=force, chain_id0=cid, **kw)
        with ps.Ensemble(pp) as e:
            in_memory = "state in L2" in e.launch_info().kernel.decode()
            assert in_memory == (where == "global" or (where == "auto" and n > 40))
            homes[where] += 1
            for k in range(inits):
                half = nsteps // 3
                e.advance(half); e.advance(nsteps - half)          # a launch split in every run
                if k + 1 < inits:
                    e.reinit(bool(force))
            e.sync()
            for c in sorted(set([0, nch - 1, nch // 2])):
                o = oracle.run(op, chain_id=pp.chain_id0 + c, mode="fast", trace=True)
                g = e.chain_state(c)
                ctx = (trial, where, rows, nch, cid, nsteps, inits, force, kw)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), ctx
                assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total, ctx
                assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step, ctx
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-7, atol=1e-7, err_msg=str(ctx))
    assert min(homes.values()) > 20, homes


@pytest.mark.parametrize("n", [30, 60], ids=["cells-in-lds", "cells-in-memory"])
@pytest.mark.parametrize("x0_phi", [5.0e8, 3.0e9, -7.0e9], ids=["below-the-fold", "above", "far-below-zero"])
def test_f64_bit_parity_with_huge_unwrapped_phi(ps, oracle, n, x0_phi):
    """phi random-walks unwrapped (inc/eap_chain.jl:232).  The f64 sweep compiles its step loop twice: without the
    huge-argument fold of the phi sincos (taken whenever no chain of the wave can reach |phi| = 1e9 within the segment --
    always, in practice) and with it.  A start far out (the ABI's x0 start; the reference's --x0) drives both copies: the
    trajectory still equals the oracle's, whose libm reduces such arguments exactly."""
    kw = dict(n=n, E0=1.0, K1=1.0, K2=0.2, Fz=0.4, Fx=0.2, seed=23, use_x0=1, x0_phi=x0_phi, x0_theta=1.2, dx0_phi=3.0, dx0_theta=0.5)
    op, pp = both(2500, num_chains=64, precision=ps.F64, **kw)
    with ps.Ensemble(pp) as e:
        e.advance(1300)
        e.advance(1200)
        for c in (0, 31, 63):
            o = oracle.run(op, chain_id=c, mode="faithful", trace=True)
            g = e.chain_state(c)
            assert abs(g["phi"]).min() > 0.5 * abs(x0_phi)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (n, x0_phi, c)
            assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total
            np.testing.assert_allclose(e.microstate(c), np.r_[o.r, o.p, o.U], rtol=1e-7, atol=1e-7)
