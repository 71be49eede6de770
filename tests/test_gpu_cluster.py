"""GPU parity tests of the clustering main's step (mcmc_clustering_eap_chain.jl:268-311) -- run with -m gpu.

The checker is oracle.eap_run_cluster, a literal restatement of that main (trial chain = deep copy,
move!, cluster_flip!, full energy recomputation, acceptor caching log(pi) + log(alpha)).  The reference
holds no fixture for this path either: PARITY UNPINNED, as for the fixed-force main.

Tolerances: f64 kernel vs oracle on the same stream -- (theta, phi), generator state, acceptance counts
and step sizes BIT-EXACT; running sums 1e-9 relative (the kernel accumulates O(1) differences where
the oracle recomputes, and device libm differs from glibc in the last ulp).  f32 kernel: pooled means
within 4.5 standard errors of the oracle's (two-sample z).
"""
import numpy as np
import pytest

from helpers import both, pooled

pytestmark = pytest.mark.gpu

CLUSTER = dict(move_set=1)


@pytest.fixture(scope="module")
def ps():
    import polymer_stats_amd as ps
    assert ps._lib.load().pstat_device_count() >= 1, "no HIP device visible"
    return ps


@pytest.fixture(params=["lds", "memory", "wave"])
def f64_home(request, monkeypatch):
    """The three f64 kernels of the clustering main (non-interacting / Ising energies): one chain per lane with (theta, phi)
    cells in LDS (pstat_cluster.hip) or 40-byte cells with the reference's trigonometric cache in device memory
    (pstat_cluster_gm.hip; the default of large ensembles), and one chain per wavefront (pstat_cluster_cw.hip; the default
    of small ones, MWC64X only).  Every f64 parity test below runs on all three."""
    monkeypatch.setenv("PSTAT_F64_STATE", {"memory": "global", "lds": "lds", "wave": "wave"}[request.param])
    return request.param


def _check_home(e, home):
    k = e.launch_info().kernel.decode()
    if home == "wave":
        # (xoshiro128++ has no skip-ahead: those handles fall back to the chain-per-lane kernel; the all-pairs energies have
        # their own chain-per-wavefront kernel)
        assert "cluster_chain_wave_kernel" in k or e.cases[0].rng == 1 or "cluster_wave_kernel" in k, k
    elif "cluster_kernel<double" in k and home is not None:
        assert ("state in memory" in k) == (home == "memory"), (k, home)


def _pair(ps, nsteps, nchains, precision, burn_sched=(), burn_in=0, **kw):
    op, pp = both(nsteps, num_chains=nchains, precision=precision, **kw)
    if burn_sched:
        op.burn_nsched = len(burn_sched)
        op.burn_in = burn_in
        for i, v in enumerate(burn_sched):
            op.burn_sched[i] = v
    pp.move_set = ps.MOVES_CLUSTER
    return op, pp


def _run_gpu(e, pp, nsteps, burn_sched=(), burn_in=0):
    """The main's driver: every rung of the burn-in ladder and the production run are fresh mcmc() calls
    (mcmc_clustering_eap_chain.jl:365-392)."""
    for f in burn_sched:
        e.set_kT(pp.kT * f)
        e.reset_sampler()
        e.reset_averages()
        e.advance(burn_in)
    e.set_kT(pp.kT)
    e.reset_sampler()
    e.reset_averages()
    e.advance(nsteps)
    e.sync()


def _bit_parity(ps, oracle, nsteps, nchains, burn_sched=(), burn_in=0, home=None, **kw):
    op, pp = _pair(ps, nsteps, nchains, ps.F64, burn_sched, burn_in, **kw)
    with ps.Ensemble(pp) as e:
        _check_home(e, home)
        _run_gpu(e, pp, nsteps, burn_sched, burn_in)
        for c in range(nchains):
            o = oracle.run(op, chain_id=c, mode="cluster", trace=True)
            g = e.chain_state(c)
            x = e.chain_extras(c)
            assert np.array_equal(g["theta"], o.final_theta), f"theta differs, chain {c}"
            assert np.array_equal(g["phi"], o.final_phi), f"phi differs, chain {c}"
            assert np.array_equal(g["rng"], o.rng), f"rng differs, chain {c}"
            assert g["nacc_total"] == o.nacc_total
            assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step
            # value / normalizer: with --umbrella-sampling the gauge constant of the weights cancels
            np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(x["sums"] / g["normalizer"], o.extra_sums / o.norm, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(e.microstate(c), np.r_[o.r, o.p, o.U], rtol=1e-9, atol=1e-8)


@pytest.mark.parametrize("rng", [0, 1])
def test_f64_bit_parity_cluster_dielectric(ps, oracle, rng, f64_home):
    # BASELINE configs[4] family: bending stiffness + cluster flips, adaptation active
    _bit_parity(ps, oracle, 6000, 66, home=f64_home, n=20, E0=1.2, K1=1.0, K2=0.2, Fz=0.7, kT=1.0, seed=21, rng=rng,
                bend_mod=0.5, bend_angle=0.3, cluster_prob=0.5, steps_per_adjust=500)


def test_f64_bit_parity_cluster_polar_fx(ps, oracle, f64_home):
    _bit_parity(ps, oracle, 4000, 64, home=f64_home, n=17, E0=0.8, mu=0.9, Fz=0.3, Fx=0.25, kT=0.8, b=1.2, chain_type=1,
                seed=22, bend_mod=0.2, bend_angle=0.0, cluster_prob=0.3, steps_per_adjust=400)


def test_f64_bit_parity_cluster_ising(ps, oracle, f64_home):
    # Weak coupling and a short run on purpose: under the reference's Ising energy a reflected cluster
    # end can land anti-parallel to its neighbour, r = x_i - x_{i+1} -> 0 and U -> -1e7 within a few
    # thousand steps (polar chains at any coupling).  In that collapsed state a 1e-12 relative
    # difference in r (cumsum positions vs. b/2 (n_i + n_j)) already moves accept decisions, so
    # bit parity is only meaningful before the collapse.
    _bit_parity(ps, oracle, 2000, 64, home=f64_home, n=17, E0=1.0, K1=0.3, K2=0.02, Fz=0.3, Fx=0.25, kT=0.8, b=1.2,
                energy_type=2, seed=22, bend_mod=0.2, bend_angle=0.0, cluster_prob=0.3, steps_per_adjust=400)


def test_f64_bit_parity_cluster_always_and_never(ps, oracle, f64_home):
    # cluster_prob = 0: a cluster flip rides on every proposal; = 1: never (the plain single move + bending)
    _bit_parity(ps, oracle, 3000, 64, home=f64_home, n=12, E0=1.0, Fz=0.5, seed=23, cluster_prob=0.0, steps_per_adjust=300)
    _bit_parity(ps, oracle, 3000, 64, home=f64_home, n=12, E0=1.0, Fz=0.5, seed=23, cluster_prob=1.0, bend_mod=1.0,
                bend_angle=0.5, steps_per_adjust=300)
    _bit_parity(ps, oracle, 2000, 64, home=f64_home, n=2, E0=1.0, Fz=0.5, seed=24, cluster_prob=0.2)
    # aligned start, no field: every link joins with probability ~1, so clusters run to the chain ends (many growth
    # rounds, several member passes in the commit)
    _bit_parity(ps, oracle, 1500, 64, home=f64_home, n=37, E0=0.2, Fz=0.1, seed=29, cluster_prob=0.1, use_x0=1, x0_phi=0.3,
                x0_theta=0.4, dx0_phi=0.05, dx0_theta=0.05, steps_per_adjust=300)


def test_f64_bit_parity_burn_in_ladder_and_x0(ps, oracle, f64_home):
    # the annealing ladder: each rung restarts step sizes / acceptor / averagers at kT * factor
    _bit_parity(ps, oracle, 3000, 64, burn_sched=(10.0, 2.0, 1.0), burn_in=1500, home=f64_home, n=16, E0=1.0, Fz=0.6, kT=0.9,
                seed=25, bend_mod=0.3, cluster_prob=0.5, steps_per_adjust=500)
    # --x0 "[phi; theta]" --dx0: start near a given orientation instead of uniformly
    _bit_parity(ps, oracle, 2000, 64, home=f64_home, n=16, E0=1.0, Fz=0.6, seed=26, cluster_prob=0.5, use_x0=1, x0_phi=0.3,
                x0_theta=1.2, dx0_phi=2 * np.pi, dx0_theta=0.1)


def test_f64_bit_parity_per_monomer_x0(ps, oracle, f64_home):
    """--x0 of length 2n: [phi1, theta1, phi2, theta2, ...] + Uniform(0, dx0) (inc/eap_chain.jl:73-75), for the
    chain-per-lane kernel and the chain-per-wavefront one."""
    for n, extra in ((11, {}), (9, dict(energy_type=1, K1=0.5))):
        rng = np.random.default_rng(n)
        x0 = np.c_[rng.uniform(0, 6, n), rng.uniform(0.3, 2.8, n)].reshape(-1)
        kw = dict(n=n, E0=1.0, Fz=0.5, seed=28, cluster_prob=0.5, dx0_phi=0.4, dx0_theta=0.05, **extra)
        op, pp = _pair(ps, 800, 8, ps.F64, **kw)
        from oracle import binding as ob
        op = ob.make_params(num_steps=800, x0_vec=x0, **kw)
        with ps.Ensemble(pp) as e:
            _check_home(e, f64_home)
            e.advance(50)                                   # whatever happened before is discarded
            e.restart_from_x0(x0, 0.4, 0.05)
            g0 = e.chain_state(3)
            assert np.all((g0["phi"] >= x0[0::2]) & (g0["phi"] < x0[0::2] + 0.4))
            assert np.all((g0["theta"] >= x0[1::2]) & (g0["theta"] < x0[1::2] + 0.05))
            _run_gpu(e, pp, 800)
            for c in range(8):
                o = oracle.run(op, chain_id=c, mode="cluster", trace=True)
                g = e.chain_state(c)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi)
                assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-9, atol=1e-9)
            with pytest.raises(ps.PstatError):
                e.restart_from_x0(x0[:5], 0.4, 0.05)


def test_f64_bit_parity_cluster_umbrella(ps, oracle, f64_home):
    _bit_parity(ps, oracle, 3000, 64, home=f64_home, n=14, E0=1.5, K1=1.0, K2=0.0, Fz=0.2, seed=27, umbrella=1,
                bend_mod=0.4, bend_angle=0.2, cluster_prob=0.5, steps_per_adjust=500)


@pytest.mark.parametrize("n,kw", [
    (20, dict(energy_type=1, E0=1.0, K1=1.0, K2=0.1, Fz=0.4, Fx=0.2, bend_mod=0.3, bend_angle=0.2)),   # M = 1
    (23, dict(energy_type=1, E0=0.8, mu=0.3, chain_type=1, Fz=0.5, rng=1)),
    (70, dict(energy_type=1, E0=1.0, K1=0.8, K2=0.0, Fz=0.3, bend_mod=0.2)),                          # M = 2
    (130, dict(energy_type=1, E0=1.0, K1=0.5, K2=0.0, Fz=0.3)),                                      # M = 4
    (20, dict(energy_type=3, cutoff_radius=2.5, E0=1.0, K1=0.6, K2=0.1, Fz=0.4, bend_mod=0.3)),
    (70, dict(energy_type=3, cutoff_radius=7.5, E0=1.0, K1=0.6, K2=0.0, Fz=0.4, umbrella=1)),
])
def test_f64_bit_parity_cluster_all_pairs(ps, oracle, n, kw):
    """interacting / cutoff energies with cluster moves: the one-chain-per-wavefront kernel against the
    oracle's literal restatement (short runs: these energies share the 1/r^3 collapse of the Ising one)."""
    _bit_parity(ps, oracle, 700, 5, n=n, kT=1.0, seed=51, cluster_prob=0.5, steps_per_adjust=200, **kw)


def test_f64_all_pairs_burn_in_ladder(ps, oracle):
    _bit_parity(ps, oracle, 500, 4, burn_sched=(10.0, 1.0), burn_in=300, n=24, E0=1.0, K1=0.7, Fz=0.4, kT=0.9,
                seed=52, energy_type=1, cluster_prob=0.4, steps_per_adjust=150)


def test_f32_cutoff_long_chain_statistics(ps, oracle):
    """n = 300: 8 monomers per lane, the regime of the reference's only cutoff-energy sweep
    (run/phases-big_2023-05-18.jl: n = 400, --energy-type cutoff).  Short run against the literal oracle."""
    kw = dict(n=300, E0=1.0, K1=0.3, K2=0.03, Fz=0.3, kT=1.0, seed=54, cluster_prob=0.5, energy_type=3, cutoff_radius=7.5)
    nsteps = 600
    op, pp = _pair(ps, nsteps, 512, ps.F32, **kw)
    with ps.Ensemble(pp) as e:
        _run_gpu(e, pp, nsteps)
        s = e.summary()
        assert e.launch_info().blocks == 512
    gm, gs = np.array(s.avg), np.array(s.stderr)
    sums, norm, nacc = oracle.run_many(op, 1 << 20, 96, nthreads=8, mode="cluster")
    om, os_ = pooled(sums, norm)
    keep = [i for i, k in enumerate(oracle.OBS_NAMES) if k not in ("Usq", "U")]     # U: heavy 1/r^3 tails
    z = ((gm - om) / np.sqrt(gs ** 2 + os_ ** 2 + 1e-300))[keep]
    assert np.all(np.abs(z) < 4.5), (z, gm, om)
    oar = nacc / nsteps
    assert abs(s.acceptance_ratio - oar.mean()) < 4.5 * np.hypot(s.ar_stderr, oar.std(ddof=1) / np.sqrt(len(oar))) + 1e-3
    with pytest.raises(ps.PstatError) as ei:       # 8 monomers per lane is the limit, in either precision
        ps.Ensemble(ps.default_params(move_set=ps.MOVES_CLUSTER, n=513, energy_type=3, precision=ps.F64))
    assert ei.value.code == -4


@pytest.mark.parametrize("et", [3, 1], ids=["cutoff", "interacting"])
def test_f64_long_chain_all_pairs_bit_parity(ps, oracle, et):
    """f64 (the hosts' default) at 8 monomers per lane: n = 300 and the reference's n = 400 (run/phases-big_2023-05-18.jl),
    trajectories bit for bit against the oracle's literal clustering main (weak coupling: no collapse in 150 steps)."""
    for n in (300, 400):
        _bit_parity(ps, oracle, 150, 3, n=n, E0=1.0, K1=0.3, K2=0.03, Fz=0.3, kT=1.0, seed=56, cluster_prob=0.5,
                    energy_type=et, cutoff_radius=7.5, steps_per_adjust=50)


def test_f32_all_pairs_cluster_statistics(ps, oracle):
    kw = dict(n=16, E0=1.0, K1=0.5, K2=0.05, Fz=0.6, kT=1.0, seed=53, bend_mod=0.3, cluster_prob=0.5, energy_type=1)
    nsteps, burn = 3000, 1000
    op, pp = _pair(ps, nsteps, 2048, ps.F32, (1.0,), burn, **kw)
    with ps.Ensemble(pp) as e:
        _run_gpu(e, pp, nsteps, (1.0,), burn)
        s = e.summary()
        assert b"cluster_wave_kernel" in e.launch_info().kernel
    gm, gs = np.array(s.avg), np.array(s.stderr)
    sums, norm, nacc = oracle.run_many(op, 1 << 20, 512, nthreads=8, mode="cluster")
    om, os_ = pooled(sums, norm)
    # U^2 is dominated by rare close approaches (1/r^3): compare everything else
    keep = [i for i, k in enumerate(oracle.OBS_NAMES) if k != "Usq"]
    z = ((gm - om) / np.sqrt(gs ** 2 + os_ ** 2 + 1e-300))[keep]
    assert np.all(np.abs(z) < 4.5), (z, gm, om)


@pytest.fixture(params=["lds", "memory"])
def f32_home(request, monkeypatch):
    """The f32 chain-per-lane cluster kernel has both homes too: cells in LDS (the default while the ensemble is at most
    twice what LDS seats) or 20-byte cells with cached n-hat in device memory (pstat_cluster_gm.hip, beyond)."""
    monkeypatch.setenv("PSTAT_F32_STATE", "global" if request.param == "memory" else "lds")
    return request.param


@pytest.mark.parametrize("energy_type,K1", [(0, 1.0), (2, 0.2)])
def test_f32_cluster_statistics(ps, oracle, energy_type, K1, f32_home):
    kw = dict(n=20, E0=1.0, K1=K1, K2=0.1 * K1, Fz=0.8, kT=1.0, seed=31, bend_mod=0.5, bend_angle=0.3,
              cluster_prob=0.5, energy_type=energy_type)
    nsteps, burn = 20000, 3000
    op, pp = _pair(ps, nsteps, 4096, ps.F32, (1.0,), burn, **kw)
    with ps.Ensemble(pp) as e:
        assert ("state in memory" in e.launch_info().kernel.decode()) == (f32_home == "memory")
        _run_gpu(e, pp, nsteps, (1.0,), burn)
        s = e.summary()
    gm, gs = np.array(s.avg), np.array(s.stderr)
    gx, gxs = np.array(s.extra_avg), np.array(s.extra_stderr)
    nref = 768
    sums, norm, nacc = oracle.run_many(op, 1 << 20, nref, nthreads=8, mode="cluster")
    om, os_ = pooled(sums, norm)
    z = (gm - om) / np.sqrt(gs ** 2 + os_ ** 2 + 1e-300)
    assert np.all(np.abs(z) < 4.5), (z, gm, om)
    oar = nacc / nsteps
    zar = (s.acceptance_ratio - oar.mean()) / np.hypot(s.ar_stderr, oar.std(ddof=1) / np.sqrt(nref))
    assert abs(zar) < 4.5, (s.acceptance_ratio, oar.mean())
    # the two extra averagers against single-chain oracle runs
    ex = np.array([oracle.run(op, chain_id=(1 << 20) + c, mode="cluster").extra_sums / nsteps for c in range(96)])
    zx = (gx - ex.mean(0)) / np.sqrt(gxs ** 2 + ex.var(0, ddof=1) / len(ex))
    assert np.all(np.abs(zx) < 4.5), (zx, gx, ex.mean(0))


@pytest.mark.parametrize("energy_type,K1", [(0, 1.0), (2, 0.2)])
def test_q16_cluster_matches_f64_kernel(ps, energy_type, K1):
    """Opt-in lattice state in the cluster kernel: pooled averages against the f64 cluster kernel (itself
    bit-exact vs the oracle) under the same protocol, two-sample z < 4.5; the state stays on the lattice."""
    out = {}
    for prec in (ps.F64, ps.Q16):
        pp = ps.default_params(n=20, E0=1.0, K1=K1, K2=0.1 * K1, Fz=0.8, Fx=0.2, kT=1.0, seed=33 + prec, bend_mod=0.5,
                               bend_angle=0.3, cluster_prob=0.5, energy_type=energy_type, num_chains=4096,
                               precision=prec, move_set=ps.MOVES_CLUSTER)
        with ps.Ensemble(pp) as e:
            _run_gpu(e, pp, 20000, (1.0,), 3000)
            s = e.summary()
            out[prec] = (np.r_[s.avg, s.extra_avg, s.acceptance_ratio], np.r_[s.stderr, s.extra_stderr, s.ar_stderr])
            if prec == ps.Q16:
                assert b"q16" in e.launch_info().kernel
                g = e.chain_state(5)
                k = g["theta"] / np.pi * 65536 - 0.5
                assert np.allclose(k, np.round(k), atol=1e-6) and k.min() >= 0 and k.max() <= 65535
    z = (out[ps.Q16][0] - out[ps.F64][0]) / np.sqrt(out[ps.Q16][1] ** 2 + out[ps.F64][1] ** 2 + 1e-300)
    assert np.all(np.abs(z) < 4.5), z


def test_cluster_full_size_properties(ps):
    """Reference-scale ensemble (no oracle): segment invariance, checkpoint round trip, cluster_prob = 1
    leaves the equilibrium of the plain sampler (closed form for E0 = 0, kappa = 0)."""
    pp = ps.default_params(n=100, E0=0.0, Fz=1.0, kT=1.0, num_chains=16384, seed=41, precision=ps.F64,
                           move_set=ps.MOVES_CLUSTER, cluster_prob=0.5)
    with ps.Ensemble(pp) as a, ps.Ensemble(pp) as b:
        a.advance(3000)
        blob = a.checkpoint()
        a.advance(2000)
        b.advance(1000); b.advance(2000)
        b.restore(blob)
        b.advance(2000)
        a.sync(); b.sync()
        for c in (0, 777, 16383):
            ga, gb = a.chain_state(c), b.chain_state(c)
            assert np.array_equal(ga["theta"], gb["theta"]) and np.array_equal(ga["rng"], gb["rng"])
            assert np.array_equal(ga["sums"], gb["sums"])
        info = a.launch_info()
        assert b"cluster_kernel" in info.kernel
    # Langevin: <r3>/(n b) = coth(F b/kT) - kT/(F b)
    pp = ps.default_params(n=100, E0=0.0, Fz=1.0, kT=1.0, num_chains=16384, seed=42, precision=ps.F32,
                           move_set=ps.MOVES_CLUSTER, cluster_prob=1.0)
    with ps.Ensemble(pp) as e:
        e.advance(40000)
        e.reset_averages()
        e.advance(20000)
        s = e.summary()
    want = 100 * (1 / np.tanh(1.0) - 1.0)
    assert abs(s.avg[2] - want) < 5 * s.stderr[2] + 1e-3 * want, (s.avg[2], want, s.stderr[2])


def test_cluster_errors(ps):
    for bad in (dict(energy_type=1, n=600), dict(energy_type=1, n=600, precision=ps.F32), dict(energy_type=1, precision=ps.Q16),
                dict(energy_type=3, precision=ps.Q16)):
        with pytest.raises(ps.PstatError) as ei:
            ps.Ensemble(ps.default_params(move_set=ps.MOVES_CLUSTER, **{"n": 16, **bad}))
        assert ei.value.code == -4
    with ps.Ensemble(ps.default_params(move_set=ps.MOVES_CLUSTER, n=16, num_chains=64)) as e:
        with pytest.raises(ps.PstatError):
            e.reinit(True)


def test_scale_kT_is_a_ladder_rung_for_every_case(ps):
    """pstat_scale_kT: kT_i <- kT0_i * mult for all cases of a sweep grid at once (a phase scan's ladder)."""
    def cases(mult):
        return [ps.default_params(n=12, E0=1.0, Fz=0.3, kT=kT * mult, num_chains=64, seed=61 + i, precision=ps.F64,
                                  move_set=ps.MOVES_CLUSTER, cluster_prob=0.5) for i, kT in enumerate((0.5, 2.0))]
    with ps.Ensemble(cases(1.0)) as a, ps.Ensemble(cases(10.0)) as b:
        a.scale_kT(10.0)
        a.advance(800); b.advance(800)
        a.sync(); b.sync()
        for c in (0, 63, 64, 127):
            ga, gb = a.chain_state(c), b.chain_state(c)
            assert np.array_equal(ga["theta"], gb["theta"]) and ga["nacc_total"] == gb["nacc_total"]
        a.scale_kT(1.0)
        with pytest.raises(ps.PstatError):
            a.scale_kT(0.0)


def test_f64_bit_parity_random_cluster_configurations(ps, oracle, f64_home):
    """Seeded fuzz over the clustering main's options (all four energies, bending, cluster_prob, ladder,
    both forms of --x0, umbrella, generator): the device trajectory equals the oracle's bit for bit."""
    rng = np.random.default_rng(20260502)
    for trial in range(60):
        et = int(rng.choice([0, 0, 2, 1, 3]))
        n = int(rng.integers(2, 30)) if et in (0, 2) else int(rng.integers(2, 80))
        weak = 0.25 if et != 0 else 1.0          # all 1/r^3 energies: stay away from the collapse (see the Ising test)
        kw = dict(n=n, E0=float(rng.uniform(0, 1.5)), K1=float(rng.uniform(0, 1.0)) * weak, K2=float(rng.uniform(0, 0.4)) * weak,
                  mu=float(rng.uniform(0.01, 0.5)) * weak, kT=float(10 ** rng.uniform(-0.3, 0.6)), Fz=float(rng.uniform(-1, 1.5)),
                  Fx=float(rng.choice([0.0, rng.uniform(-1, 1)])), b=float(rng.uniform(0.6, 1.8)),
                  chain_type=int(rng.integers(0, 2)), energy_type=et, umbrella=int(rng.integers(0, 2)),
                  bend_mod=float(rng.choice([0.0, rng.uniform(0, 1.5)])), bend_angle=float(rng.uniform(0, 1.0)),
                  cluster_prob=float(rng.choice([0.0, 0.3, 0.5, 0.8, 1.0])), steps_per_adjust=int(rng.choice([60, 200, 2500])),
                  rng=int(rng.integers(0, 2)), seed=int(rng.integers(0, 2 ** 40)), cutoff_radius=float(rng.uniform(1.5, 9.0)))
        nsteps = 300 if et in (1, 3) else 900
        sched = tuple(float(x) for x in rng.choice([100.0, 10.0, 2.0, 1.0], size=int(rng.integers(0, 3))))
        burn = int(rng.integers(50, 300))
        x0mode = int(rng.integers(0, 3))
        x0 = None
        if x0mode == 1:
            kw.update(use_x0=1, x0_phi=float(rng.uniform(0, 6)), x0_theta=float(rng.uniform(0.3, 2.5)), dx0_phi=0.7, dx0_theta=0.1)
        op, pp = _pair(ps, nsteps, 3, ps.F64, sched, burn, **kw)
        if x0mode == 2:
            x0 = np.c_[rng.uniform(0, 6, n), rng.uniform(0.3, 2.5, n)].reshape(-1)
            from oracle import binding as ob
            op = ob.make_params(num_steps=nsteps, x0_vec=x0, dx0_phi=0.7, dx0_theta=0.1, burn_in=burn,
                                burn_sched=list(sched), **kw)
        with ps.Ensemble(pp) as e:
            _check_home(e, f64_home)
            if x0 is not None:
                e.restart_from_x0(x0, 0.7, 0.1)
            _run_gpu(e, pp, nsteps, sched, burn)
            for c in range(3):
                o = oracle.run(op, chain_id=c, mode="cluster", trace=True)
                g = e.chain_state(c)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (trial, kw, sched, x0mode)
                assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total, (trial, kw)
                np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-8, atol=1e-8, err_msg=str((trial, kw)))


@pytest.mark.parametrize("n,kw", [
    (100, dict(E0=1.0, K1=0.0, K2=1.0, Fz=0.0, adj_ub=0.40)),                       # run/Ising_2025-12-18.jl's chain, no coupling
    (100, dict(E0=1.0, K1=0.3, K2=0.05, Fz=0.2, energy_type=2, bend_mod=0.3, bend_angle=0.2)),
    (200, dict(E0=0.6, K1=0.25, K2=0.0, Fz=0.0, kT=2.5, energy_type=2, adj_ub=0.40)),   # a point of run/K1_E0-kT-phase.jl's grid
    (41, dict(E0=1.0, mu=0.2, chain_type=1, Fz=0.4, Fx=0.3, energy_type=2, umbrella=1, rng=1)),
])
@pytest.mark.parametrize("forced", [None, "global"])
def test_f64_cluster_long_chains_bit_parity_in_the_default_home(ps, oracle, monkeypatch, n, kw, forced):
    """The chain lengths the reference's sweeps launch this main with (n = 100, 200) on the kernel pstat_create picks for a
    small ensemble (one chain per wavefront; xoshiro128++ has no skip-ahead and stays with the chain-per-lane kernel in
    device memory) and on the in-memory chain-per-lane kernel, the default of large ensembles: trajectories against the
    oracle's literal clustering main, 70 chains = one full and one 6-lane wave there, a launch split included (weak coupling:
    no 1/r^3 collapse in 1200 steps)."""
    nsteps = 1200
    if forced:
        monkeypatch.setenv("PSTAT_F64_STATE", forced)
    op, pp = _pair(ps, nsteps, 70, ps.F64, n=n, seed=71, cluster_prob=0.5, steps_per_adjust=400, **kw)
    with ps.Ensemble(pp) as e:
        k = e.launch_info().kernel.decode()
        assert ("cluster_chain_wave_kernel" in k) if (forced is None and not kw.get("rng")) else ("state in memory" in k), k
        e.advance(500); e.advance(nsteps - 500)
        e.sync()
        for c in (0, 1, 31, 63, 64, 69):
            o = oracle.run(op, chain_id=c, mode="cluster", trace=True)
            g = e.chain_state(c)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (n, c)
            assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total
            assert g["phi_step"] == o.phi_step and g["theta_step"] == o.theta_step
            np.testing.assert_allclose(g["sums"] / g["normalizer"], o.avg, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(e.microstate(c), np.r_[o.r, o.p, o.U], rtol=1e-9, atol=1e-8)


def test_f64_cluster_state_in_memory_matches_state_in_lds(ps, monkeypatch):
    """Two homes for a chain, one step: forced either way, every compared chain's angles, generator, counters and step
    sizes are bit-identical, the running sums agree to 1e-10 (a reflected monomer's cached n-hat is the mapped one, the
    LDS kernel's is recomputed from the reflected angle: an ulp), also across launch splits, an annealing rung and a
    checkpoint written by one variant and restored into the other."""
    for kw in (dict(n=100, E0=1.0, K1=0.0, K2=1.0, adj_ub=0.40),
               dict(n=17, E0=0.5, mu=0.7, Fz=0.3, Fx=0.4, chain_type=ps.POLAR, bend_mod=0.4, bend_angle=0.3, umbrella=1),
               dict(n=64, E0=1.0, K1=0.3, K2=0.05, Fz=0.3, Fx=0.2, energy_type=ps.ISING, cluster_prob=0.2),
               dict(n=33, E0=0.2, Fz=0.1, cluster_prob=0.1, use_x0=1, x0_phi=0.3, x0_theta=0.4, dx0_phi=0.05, dx0_theta=0.05)):
        res, blob = {}, None
        for where in ("lds", "global", "wave"):
            monkeypatch.setenv("PSTAT_F64_STATE", where)
            pp = ps.default_params(num_chains=200, precision=ps.F64, seed=78, steps_per_adjust=300,
                                   move_set=ps.MOVES_CLUSTER, **kw)
            with ps.Ensemble(pp) as e:
                k = e.launch_info().kernel.decode()
                assert ("state in memory" in k) == (where == "global") and ("chain_wave" in k) == (where == "wave"), k
                e.scale_kT(3.0); e.advance(400)
                e.scale_kT(1.0); e.reset_sampler(); e.reset_averages()
                e.advance(501)
                if blob is None:
                    blob = e.checkpoint()
                else:
                    e.restore(blob)          # the other variant's checkpoint: the layout is the angles-only one
                e.advance(299)
                res[where] = [e.chain_state(c) for c in (0, 63, 64, 199)] + [e.reduce_host()]
        for other in ("global", "wave"):
            for x, y in zip(res["lds"][:-1], res[other][:-1]):
                for key in ("theta", "phi", "rng"):
                    assert np.array_equal(x[key], y[key]), (kw, key, other)
                np.testing.assert_allclose(x["sums"], y["sums"], rtol=1e-10, atol=1e-8)
                assert (x["nacc_total"], x["phi_step"], x["theta_step"]) == (y["nacc_total"], y["phi_step"], y["theta_step"])
                np.testing.assert_allclose(x["normalizer"], y["normalizer"], rtol=1e-12)
            np.testing.assert_allclose(res["lds"][-1], res[other][-1], rtol=1e-10, atol=1e-8)
    monkeypatch.delenv("PSTAT_F64_STATE")
    # defaults: a wave per chain for small ensembles and for sweeps of many small cases, the chain-per-lane kernel in device
    # memory (at every chain length: the cache of n-hat pays everywhere) for large ensembles, xoshiro128++ and n > 256
    P = lambda **kw: ps.default_params(precision=ps.F64, move_set=ps.MOVES_CLUSTER, **kw)
    for cases, wave in (([P(num_chains=64, n=2)], True), ([P(num_chains=4096, n=41)], True), ([P(num_chains=4097, n=40)], False),
                        ([P(num_chains=16, n=100, seed=i) for i in range(2730)], True),
                        ([P(num_chains=16, n=200, seed=i) for i in range(2000)], False),
                        ([P(num_chains=64, n=100, seed=i) for i in range(546)], False),
                        ([P(num_chains=64, n=40, rng=1)], False), ([P(num_chains=64, n=257)], False)):
        with ps.Ensemble(cases) as e:
            k = e.launch_info().kernel.decode()
            assert ("cluster_chain_wave_kernel" in k) == wave and ("state in memory" in k) == (not wave), (k, len(cases))


@pytest.mark.parametrize("kw", [
    dict(n=100, E0=1.0, K1=0.0, K2=1.0, kT=1.0, adj_ub=0.40),                                  # the chain of run/Ising_2025-12-18.jl, disordered
    dict(n=100, E0=1.0, K1=0.3, K2=0.05, kT=0.05, energy_type=2, bend_mod=0.5, adj_ub=0.40),  # cold and stiff: clusters run on for tens of monomers
], ids=["disordered", "aligned"])
def test_f64_cluster_homes_agree_at_scale(ps, monkeypatch, kw):
    """Full-size ensembles of the three f64 kernels of the clustering main under one protocol (annealing rung, then a recorded
    run): independent samples of the SAME algorithm, so every pooled average -- 16 observables, acceptance ratio, <cos^2>,
    <psi> -- must agree within 4.5 combined standard errors.  The aligned case drives what the bit-parity cases reach with
    a handful of chains only: the ring-fed growth beyond the window and the member passes of long clusters, on 16 384
    chains at once."""
    out = {}
    for where in ("lds", "global", "wave"):
        monkeypatch.setenv("PSTAT_F64_STATE", where)
        pp = ps.default_params(num_chains=16384, precision=ps.F64, seed=101 + len(out), move_set=ps.MOVES_CLUSTER,
                               cluster_prob=0.5, **kw)
        with ps.Ensemble(pp) as e:
            k = e.launch_info().kernel.decode()
            assert ("state in memory" in k) == (where == "global") and ("chain_wave" in k) == (where == "wave"), k
            _run_gpu(e, pp, 12000, (10.0, 1.0), 4000)
            s = e.summary()
            assert s.nan_rejects == 0
            out[where] = (np.r_[s.avg, s.extra_avg, s.acceptance_ratio], np.r_[s.stderr, s.extra_stderr, s.ar_stderr])
    for other in ("global", "wave"):
        z = (out[other][0] - out["lds"][0]) / np.sqrt(out[other][1] ** 2 + out["lds"][1] ** 2 + 1e-300)
        assert np.all(np.abs(z) < 4.5), (other, z, out[other][0], out["lds"][0])


def test_f64_cluster_in_memory_equilibrium_without_flips(ps, golden):
    """cluster_prob = 1 (no flips): the clustering main is the plain single-move sampler plus two averagers, so after a
    burn-in the default f64 kernel (chains in memory, cached n-hat) must sit on the closed form of BASELINE configs[1] at
    Fz = 1 -- all 16 pooled averages within 5 of their own standard errors (~1e-4 relative on <r_z>)."""
    eq = golden["cfg2_n100_E0_1_K1_1_Fz1"]["avg"]
    pp = ps.default_params(n=100, E0=1.0, K1=1.0, K2=0.0, Fz=1.0, kT=1.0, num_chains=32768, seed=55, precision=ps.F64,
                           move_set=ps.MOVES_CLUSTER, cluster_prob=1.0)
    with ps.Ensemble(pp) as e:
        assert "state in memory" in e.launch_info().kernel.decode()
        e.advance(20000)
        e.reset_averages()
        e.advance(30000)
        s = e.summary()
    z = np.array([(s.avg[k] - eq[nm]) / (s.stderr[k] + 1e-12 * (1 + abs(eq[nm]))) for k, nm in enumerate(ps.OBS_NAMES)])
    assert np.all(np.abs(z) < 5.0), dict(zip(ps.OBS_NAMES, np.round(z, 2)))


def test_f64_cluster_in_memory_time_segments(ps, monkeypatch):
    """The persistent (block, segment) queue under the in-memory kernel: every job co-resident (64 blocks x 3 or 7 segments),
    later segments wait on their predecessors, each boundary spills the angles and refills the working buffer (the cached
    n-hat is re-derived there).  Trajectories must equal the unsegmented launch's; the running sums agree to 1e-10 (a
    reflected monomer's cached n_z is the mapped one until the next refill: an ulp)."""
    pp = ps.default_params(num_chains=4096, precision=ps.F64, n=60, E0=1.0, K1=0.3, K2=0.05, Fz=0.4, seed=12, energy_type=ps.ISING,
                           move_set=ps.MOVES_CLUSTER, cluster_prob=0.4, bend_mod=0.3, steps_per_adjust=700)
    monkeypatch.setenv("PSTAT_MAX_SPINS", str(1 << 19))     # fail within ~1 s instead of hanging
    monkeypatch.setenv("PSTAT_F64_STATE", "global")        # (an ensemble this small would run one chain per wavefront)
    states = {}
    for nseg in ("1", "3", "7"):
        monkeypatch.setenv("PSTAT_SEGMENTS", nseg)
        with ps.Ensemble(pp) as e:
            assert "state in memory" in e.launch_info().kernel.decode()
            e.advance(4200)
            e.sync()
            states[nseg] = [e.chain_state(c) for c in (0, 63, 64, 1000, 4095)] + [e.reduce_host()]
    for nseg in ("3", "7"):
        for a, b in zip(states["1"][:-1], states[nseg][:-1]):
            for k in ("theta", "phi", "rng"):
                assert np.array_equal(a[k], b[k]), (nseg, k)
            assert a["nacc_total"] == b["nacc_total"] and a["phi_step"] == b["phi_step"]
            np.testing.assert_allclose(a["sums"], b["sums"], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(states["1"][-1], states[nseg][-1], rtol=1e-10, atol=1e-8)


def test_f32_cluster_homes_agree_at_scale_and_default_home(ps, monkeypatch):
    """f32 clustering main: the LDS kernel and the in-memory one are independent samples of one algorithm -- every pooled
    average within 4.5 combined standard errors at n = 120, long runs included (the f32 running totals are re-derived at every segment start in both)."""
    out = {}
    kw = dict(n=120, E0=1.0, K1=0.0, K2=1.0, kT=1.0, adj_ub=0.40, Fz=0.3, Fx=0.2, bend_mod=0.3)
    for where in ("lds", "global"):
        monkeypatch.setenv("PSTAT_F32_STATE", where)
        pp = ps.default_params(num_chains=16384, precision=ps.F32, seed=201 + len(out), move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, **kw)
        with ps.Ensemble(pp) as e:
            assert ("state in memory" in e.launch_info().kernel.decode()) == (where == "global")
            _run_gpu(e, pp, 40000, (10.0, 1.0), 4000)
            s = e.summary()
            out[where] = (np.r_[s.avg, s.extra_avg, s.acceptance_ratio], np.r_[s.stderr, s.extra_stderr, s.ar_stderr])
            # the reported microstate is the microstate of the stored angles (no drift of the f32 totals)
    z = (out["global"][0] - out["lds"][0]) / np.sqrt(out["global"][1] ** 2 + out["lds"][1] ** 2 + 1e-300)
    assert np.all(np.abs(z) < 4.5), (z, out["global"][0], out["lds"][0])
    monkeypatch.delenv("PSTAT_F32_STATE")
    # default home: LDS while the ensemble is at most twice what LDS seats (160 KiB / 8 n chains per CU), memory beyond
    for n, chains, mem in ((200, 64, False), (200, 32768, False), (200, 65536, True), (100, 65536, False), (20, 131072, False)):
        with ps.Ensemble(ps.default_params(num_chains=chains, precision=ps.F32, n=n, move_set=ps.MOVES_CLUSTER)) as e:
            assert ("state in memory" in e.launch_info().kernel.decode()) == mem, (n, chains)


def test_f64_chain_per_wavefront_kernel_fuzz(ps, monkeypatch):
    """pstat_cluster_cw.hip over every chain length it takes (n = 2 ... 256: one, two and four monomers per lane and the
    lengths either side of each boundary), hot to cold (aligned starts: clusters that run to the chain ends, one end drawing
    alone for tens of rounds), both chain types, Ising and non-interacting, bending, umbrella, both eps contracts, a launch
    split and an annealing rung in every run: 60 seeded configurations of tests/fuzz_cluster_wave.py (the long form: 600
    more, profiles/r04/fuzz_cluster_wave.txt), three chains each bit-equal to the oracle."""
    import fuzz_cluster_wave
    monkeypatch.setenv("PSTAT_F64_STATE", "wave")      # (run() sets it too; restored by monkeypatch afterwards)
    lines = []
    assert fuzz_cluster_wave.run(60, 20261005, log=lambda *a: lines.append(" ".join(str(x) for x in a))) == 0, "\n".join(lines)
