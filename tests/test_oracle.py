"""CPU tests of the oracle (oracle/eap_oracle.c) -- the checker must itself be checked.

Pins available (the reference ships none -- "parity unpinned", see oracle/eap_oracle.h):
  * Philox4x32-10 known-answer vectors of Random123 (kat_vectors), xoshiro128++ against an
    independent pure-Python transcription of the published algorithm;
  * closed-form equilibrium averages (tests/golden/ni_closed_form.json);
  * hand-computable dipole-dipole energies (inc/eap_chain.jl:200-207);
  * literal ("faithful") and incremental ("fast") restatements making identical decisions.
"""
import numpy as np
import pytest

from helpers import pooled

M32 = 0xFFFFFFFF


def test_philox_known_answers(oracle):
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([M32] * 4, [M32] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_xoshiro128pp_matches_published_algorithm(oracle):
    def rotl(x, k):
        return ((x << k) | (x >> (32 - k))) & M32
    seed, chain = 0x1234_5678_9abc_def0, 42
    s, got = oracle.xoshiro_stream(seed, chain, 1000)
    # seeding contract: Philox(key=seed, ctr=(chain_lo, chain_hi, 0x5eed, 0))
    assert s == oracle.philox([chain & M32, chain >> 32, 0x5eed, 0], [seed & M32, seed >> 32])
    want = []
    for _ in range(1000):
        want.append((rotl((s[0] + s[3]) & M32, 7) + s[0]) & M32)
        t = (s[1] << 9) & M32
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 11)
    assert got == want
    # distinct chains get distinct streams
    assert oracle.xoshiro_stream(seed, chain + 1, 4)[1] != got[:4]


def test_mwc64x_matches_big_integer_reference(oracle):
    """MWC64X recurrence, the Philox-selected base state and the 2^40-output skip-ahead per chain,
    against exact Python integers (independent of the oracle's 128-bit C arithmetic)."""
    A = 4294883355
    M = A * (1 << 32) - 1
    assert M == 0xFFFEB81AFFFFFFFF and pow(A, 1 << 40, M) == 0x82A211110E454078
    for seed, chain in [(0, 0), (0x1234567890abcdef, 77), (20260501, 65535), (5, 2 ** 40 + 3)]:
        ph = oracle.philox([0, 0, 0x5eed, 1], [seed & M32, seed >> 32])
        base = 1 + (ph[0] | ph[1] << 32) % (M - 2)
        st = base * pow(A, chain * (1 << 40), M) % M
        x, c = st & M32, st >> 32
        init, outs = oracle.mwc64x_stream(seed, chain, 500)
        assert init == (x, c)
        for got in outs:
            assert got == x ^ c
            t = A * x + c
            x, c = t & M32, t >> 32
    # skip-ahead == stepping: chain k+1 starts exactly 2^40 outputs after chain k
    st = (123456789 << 32) | 987654321
    assert oracle.lib().eap_mwc64x_skip(st, 1000) == st * pow(A, 1000, M) % M
    (x0, c0), _ = oracle.mwc64x_stream(9, 4, 1)
    (x1, c1), _ = oracle.mwc64x_stream(9, 5, 1)
    assert oracle.lib().eap_mwc64x_skip((c0 << 32) | x0, 1 << 40) == (c1 << 32) | x1


def test_uniform_contract(oracle):
    L = oracle.lib()
    assert L.eap_u01(0) == 0.0
    assert L.eap_u01(M32) == 1.0 - 2.0 ** -23
    assert L.eap_u01(1 << 9) == 2.0 ** -23


def test_pair_energy_by_hand(oracle):
    # two dipoles along z, separated along z by d: head-to-tail, U = (1 - 3)/(4 pi d^3) * m1 m2
    d, m1, m2 = 1.7, 0.8, 1.3
    xs = np.array([[0, 0, 0], [0, 0, d]], float)
    mus = np.array([[0, 0, m1], [0, 0, m2]], float)
    assert oracle.pair_energy(xs, mus) == pytest.approx(-2 * m1 * m2 / (4 * np.pi * d ** 3), rel=1e-14)
    # side by side (separated along x), parallel: U = + m1 m2 / (4 pi d^3)
    xs = np.array([[0, 0, 0], [d, 0, 0]], float)
    assert oracle.pair_energy(xs, mus) == pytest.approx(m1 * m2 / (4 * np.pi * d ** 3), rel=1e-14)
    # three collinear along x, dipoles along x: pairs (1,2),(2,3) at d and (1,3) at 2d
    xs = np.array([[0, 0, 0], [d, 0, 0], [2 * d, 0, 0]], float)
    mus = np.array([[m1, 0, 0]] * 3, float)
    want = -2 * m1 * m1 / (4 * np.pi) * (2 / d ** 3 + 1 / (2 * d) ** 3)
    assert oracle.pair_energy(xs, mus) == pytest.approx(want, rel=1e-14)
    # Ising = nearest neighbours only
    assert oracle.pair_energy(xs, mus, ising=True) == pytest.approx(-2 * m1 * m1 / (4 * np.pi) * 2 / d ** 3, rel=1e-14)


def test_chain_energy_straight_chain(oracle):
    # all monomers along z: theta = 0 -> r = n b z, dielectric mu_i = K1 E0 z, u_i = -E0^2 K1 / 2
    n, E0, K1, K2, b, Fz = 5, 1.5, 0.7, 0.3, 1.3, 0.4
    P = oracle.make_params(n=n, E0=E0, K1=K1, K2=K2, b=b, Fz=Fz, Fx=0.2)
    U, r, p = oracle.chain_energy(P, np.zeros(n), np.zeros(n))
    np.testing.assert_allclose(r, [0, 0, n * b], atol=1e-14)
    np.testing.assert_allclose(p, [0, 0, n * K1 * E0], atol=1e-14)
    assert U == pytest.approx(-0.5 * E0 * n * K1 * E0 - Fz * n * b, rel=1e-14)
    # interacting: + sum over pairs of collinear head-to-tail dipoles m = K1 E0 at distance |i-j| b
    Pi = oracle.make_params(n=n, E0=E0, K1=K1, K2=K2, b=b, Fz=Fz, energy_type=oracle.INTERACTING)
    Ui, _, _ = oracle.chain_energy(Pi, np.zeros(n), np.zeros(n))
    m = K1 * E0
    pairs = sum(-2 * m * m / (4 * np.pi * ((j - i) * b) ** 3) for i in range(n) for j in range(i + 1, n))
    assert Ui - U == pytest.approx(pairs, rel=1e-12)
    # polar: mu = mu n, u = -E0 mu cos(theta) / 2 (the 1/2 applies to polar chains too: eap_chain.jl:53)
    Pp = oracle.make_params(n=n, E0=E0, mu=0.9, chain_type=oracle.POLAR, b=b)
    Up, _, pp = oracle.chain_energy(Pp, np.zeros(n), np.full(n, np.pi / 3))
    assert Up == pytest.approx(-0.5 * E0 * 0.9 * n * 0.5, rel=1e-13)
    assert pp[2] == pytest.approx(0.9 * n * 0.5, rel=1e-13)


CASES = [
    dict(n=20, E0=0.0, Fz=1.0),
    dict(n=33, E0=1.5, K1=0.7, K2=0.3, Fz=0.4, Fx=0.3, kT=0.7, b=1.3, do_flips=1, steps_per_adjust=500),
    dict(n=16, E0=1.0, mu=1.0, Fz=1.0, chain_type=1),
    dict(n=12, E0=1.0, K1=1.0, Fz=0.25, energy_type=2, steps_per_adjust=400),
    dict(n=10, E0=1.0, K1=1.0, Fz=0.5, energy_type=1, steps_per_adjust=400),
    dict(n=14, E0=1.0, K1=1.0, K2=0.2, Fz=0.5, umbrella=1),
    dict(n=9, E0=1.0, mu=0.5, Fz=0.2, chain_type=1, umbrella=1, energy_type=2),
    dict(n=1, E0=2.0, K1=1.0, Fz=0.5, adj_scale=1.0),
]


@pytest.mark.parametrize("rng", [0, 1])
@pytest.mark.parametrize("kw", CASES)
def test_faithful_and_fast_agree(oracle, kw, rng):
    """Same stream -> same accept/reject sequence, same final angles, same generator state; running
    sums equal to rounding.  This is what licenses the O(1)-energy form the kernels use."""
    for inits, force in ((1, 0), (3, 1), (3, 0)):
        P = oracle.make_params(num_steps=4000, num_inits=inits, force_init=force, seed=17, stepout=500, rng=rng, **kw)
        a = oracle.run(P, 3, "faithful", trace=True, rows=True)
        b = oracle.run(P, 3, "fast", trace=True, rows=True)
        assert np.array_equal(a.accepted, b.accepted)
        assert np.array_equal(a.final_theta, b.final_theta) and np.array_equal(a.final_phi, b.final_phi)
        assert np.array_equal(a.rng, b.rng)
        assert a.nacc_total == b.nacc_total and (a.phi_step, a.theta_step) == (b.phi_step, b.theta_step)
        np.testing.assert_allclose(a.avg, b.avg, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(a.rolling, b.rolling, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(a.traj, b.traj, rtol=1e-9, atol=1e-9)
        assert a.rolling.shape == (inits * 8, 17) and a.rolling[0, 0] == 500.0


def test_interacting_reduces_to_noninteracting_without_dipoles(oracle):
    # E0 = 0 -> all dipoles vanish -> pair energy identically 0 -> same chain as non-interacting
    kw = dict(n=12, E0=0.0, Fz=0.7, num_steps=3000, seed=4)
    a = oracle.run(oracle.make_params(**kw), 0, "fast", trace=True)
    b = oracle.run(oracle.make_params(energy_type=oracle.INTERACTING, **kw), 0, "faithful", trace=True)
    assert np.array_equal(a.accepted, b.accepted)
    np.testing.assert_allclose(a.avg, b.avg, rtol=1e-10, atol=1e-12)


def test_adaptation_saturates_in_weak_field(oracle):
    # config 1 accepts ~60 % > ub, so the step sizes grow by 1.1 eleven times and cap
    P = oracle.make_params(n=20, E0=0.0, Fz=1.0, num_steps=100000, seed=1)
    r = oracle.run(P, 0, "fast")
    assert r.phi_step == np.pi and r.theta_step == np.pi / 2
    assert 0.55 < r.ar < 0.65


def test_rejected_when_clamped_to_zero(oracle):
    # theta' clamped to exactly 0 -> sin = 0 -> log-density -inf -> always rejected:
    # with a huge theta step half the proposals clamp, so the acceptance ratio must stay < 0.5+
    P = oracle.make_params(n=4, E0=0.0, Fz=0.0, num_steps=20000, theta_step=50.0, phi_step=1.0,
                           adj_scale=1.0, seed=2)
    r = oracle.run(P, 0, "fast", trace=True)
    assert r.ar < 0.05
    assert np.all(r.final_theta > 0) and np.all(r.final_theta <= np.pi)


@pytest.mark.parametrize("name,nsteps,nch", [
    ("cfg1_n20_E0_0_Fz1", 100000, 48),
    ("diel_n8_E0_15_K1_07_K2_03_Fz04_Fx03_kT07_b13", 100000, 48),
])
def test_oracle_against_closed_form(oracle, golden, name, nsteps, nch):
    """Pooled oracle averages vs single-monomer quadrature.  The estimator has no burn-in, so a
    transient of relative size ~tau/N (tau ~ 10 n steps) is allowed on top of 4.5 standard errors."""
    g = golden[name]
    c = g["params"]
    P = oracle.make_params(n=c["n"], E0=c["E0"], K1=c["K1"], K2=c["K2"], mu=c["mu"], kT=c["kT"], Fz=c["Fz"],
                           Fx=c["Fx"], b=c["b"], chain_type=oracle.POLAR if c["chain"] == "polar" else oracle.DIELECTRIC,
                           num_steps=nsteps, seed=99)
    sums, norm, _ = oracle.run_many(P, 0, nch, nthreads=8, mode="fast")
    avg, se = pooled(sums, norm)
    tau_over_N = 10.0 * c["n"] / nsteps
    for k, name_k in enumerate(oracle.OBS_NAMES):
        want = g["avg"][name_k]
        tol = 4.5 * se[k] + 3 * tau_over_N * (abs(want) + 1.0)
        assert abs(avg[k] - want) < tol, (name_k, avg[k], want, se[k])


def test_umbrella_and_standard_estimate_the_same_averages(oracle):
    """The reference's own validation idea (run/noninteracting-compare-with-clustering_2021-09-24.jl):
    umbrella-weighted and plain sampling must agree."""
    kw = dict(n=10, E0=0.8, K1=1.0, K2=0.0, Fz=0.3, num_steps=60000, seed=8)
    s0, n0, _ = oracle.run_many(oracle.make_params(**kw), 0, 48, nthreads=8, mode="fast")
    s1, n1, _ = oracle.run_many(oracle.make_params(umbrella=1, **kw), 1000, 48, nthreads=8, mode="fast")
    a0, e0 = pooled(s0, n0)
    a1, e1 = pooled(s1, n1)
    z = (a0 - a1) / np.sqrt(e0 ** 2 + e1 ** 2 + 1e-300)
    assert np.all(np.abs(z) < 4.5), z


# ------------------------------------------------------------------ clustering main's restatement

def test_cluster_oracle_without_flips_matches_closed_form(oracle, golden):
    """cluster_prob = 1 (never flip), kappa = 0: the clustering main reduces to a plain Metropolis
    walk with a burn-in -- pooled averages must sit on the closed form (cfg1: n = 20, E0 = 0, Fz = 1)."""
    case = golden["cfg1_n20_E0_0_Fz1"]
    P = oracle.make_params(n=20, E0=0.0, Fz=1.0, kT=1.0, num_steps=20000, seed=71, cluster_prob=1.0,
                           burn_in=4000, burn_sched=[10.0, 1.0])
    sums, norm, nacc = oracle.run_many(P, 0, 256, nthreads=8, mode="cluster")
    m = sums / norm[:, None]
    mean, se = m.mean(0), m.std(0, ddof=1) / np.sqrt(m.shape[0])
    want = np.array([case["avg"][k] for k in oracle.OBS_NAMES])
    z = (mean - want) / (se + 1e-12)
    assert np.all(np.abs(z) < 4.5), z


def test_cluster_oracle_reproduces_the_references_bias(oracle, golden):
    """Documented in DESIGN.md 3.7: with cluster flips on, the literal algorithm (single move and
    cluster flip in one proposal, acceptor caching log alpha) is biased -- <r_z> comes out 2.5-4.5 % low
    (it depends on the run length through the step-size adaptation).
    Pinned here so that a change of the restatement that 'fixes' the reference is noticed."""
    case = golden["cfg1_n20_E0_0_Fz1"]
    P = oracle.make_params(n=20, E0=0.0, Fz=1.0, kT=1.0, num_steps=20000, seed=72, cluster_prob=0.5,
                           burn_in=4000, burn_sched=[10.0, 1.0])
    sums, norm, _ = oracle.run_many(P, 0, 256, nthreads=8, mode="cluster")
    r3 = (sums / norm[:, None])[:, 2]
    want = case["avg"]["r3"]
    se = r3.std(ddof=1) / np.sqrt(len(r3))
    assert (want - r3.mean()) > 8 * se and 0.015 < (want - r3.mean()) / want < 0.07, (r3.mean(), want, se)


def test_cluster_oracle_bookkeeping(oracle):
    """extra averagers are sums of cos^2(theta) and of the mean bond angle of the CURRENT chain; the
    ladder's rungs only contribute their final configuration; --x0 start is x0 + U(0, dx0)."""
    P = oracle.make_params(n=9, E0=1.0, Fz=0.4, num_steps=0, seed=73, cluster_prob=0.5, use_x0=1, x0_phi=0.3,
                           x0_theta=1.2, dx0_phi=0.5, dx0_theta=0.1)
    o = oracle.run(P, chain_id=3, mode="cluster", trace=True)
    assert np.all((o.final_phi >= 0.3) & (o.final_phi < 0.8)) and np.all((o.final_theta >= 1.2) & (o.final_theta < 1.3))
    P = oracle.make_params(n=9, E0=1.0, Fz=0.4, num_steps=1, seed=73, cluster_prob=0.5, bend_mod=0.7, bend_angle=0.2)
    o = oracle.run(P, chain_id=3, mode="cluster", trace=True)
    th, ph = o.final_theta, o.final_phi
    nh = np.c_[np.cos(ph) * np.sin(th), np.sin(ph) * np.sin(th), np.cos(th)]
    psi = np.arccos(np.clip((nh[1:] * nh[:-1]).sum(1), -1, 1))
    np.testing.assert_allclose(o.extra_sums, [np.sum(np.cos(th) ** 2), psi.mean()], rtol=1e-12)
    # U of the final chain = field terms + bending - F.r   (inc/energy.jl:7-9, inc/eap_chain.jl:53-58)
    mu_z = 1.0 * np.cos(th) ** 2                      # K1 = 1, K2 = 0, E0 = 1
    U = -0.5 * mu_z.sum() + 0.35 * ((psi - 0.2) ** 2).sum() - 0.4 * np.cos(th).sum()
    assert o.U == pytest.approx(U, rel=1e-12)


def test_metropolis_eps_contracts(oracle):
    """eap_eps: 23 bits = (w >> 9) 2^-23; 53 bits (the default, uniform_bits = 0 | 53) = the eps word followed by the low 9
    bits of the dtheta word, the low 9 of the dphi word and the low 3 of the index word -- exact in a double, in [u23, u23 +
    2^-23), and 0 only if all 53 bits are.  faithful == fast under both, and the two contracts walk one stream."""
    L = oracle.lib()
    w_eps, w_idx, w_phi, w_th = 0x89ABCDEF, 0xFFFFFFFD, 0x123451FF, 0x765430AA
    assert L.eap_eps(23, w_eps, w_idx, w_phi, w_th) == (w_eps >> 9) / 2.0 ** 23
    want = ((w_eps << 21) | ((w_th & 511) << 12) | ((w_phi & 511) << 3) | (w_idx & 7)) / 2.0 ** 53
    assert L.eap_eps(53, w_eps, w_idx, w_phi, w_th) == want == L.eap_eps(0, w_eps, w_idx, w_phi, w_th)
    assert (w_eps >> 9) / 2.0 ** 23 <= want < ((w_eps >> 9) + 1) / 2.0 ** 23
    assert L.eap_eps(53, 0, 0, 0, 0) == 0.0 and L.eap_eps(53, 0, 1, 0, 0) == 2.0 ** -53
    assert L.eap_eps(53, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -53
    kw = dict(n=16, E0=1.2, K1=0.8, K2=0.1, Fz=0.5, Fx=0.2, num_steps=4000, seed=19, do_flips=1, stepout=0)
    runs = {}
    for bits in (23, 53):
        for mode in ("faithful", "fast"):
            runs[bits, mode] = oracle.run(oracle.make_params(uniform_bits=bits, **kw), chain_id=2, mode=mode, trace=True)
        a, b = runs[bits, "faithful"], runs[bits, "fast"]
        assert np.array_equal(a.accepted, b.accepted) and np.array_equal(a.final_theta, b.final_theta) and np.array_equal(a.rng, b.rng)
    assert np.array_equal(runs[23, "fast"].rng, runs[53, "fast"].rng)       # no extra draw under 53 bits
    with pytest.raises(RuntimeError):
        oracle.run(oracle.make_params(uniform_bits=24, **kw), chain_id=0, mode="fast")


def test_eps_word_scan_finds_the_23_bit_zeros(oracle):
    """eap_find_eps23_zero against the stream itself: word 2n + 4s + 3 of the chain's stream is step s's Metropolis word."""
    P = oracle.make_params(n=7, seed=123, num_steps=1, stepout=0)
    hits = oracle.find_eps23_zero(P, chain_id=3, nsteps=3_000_000)
    assert 0 <= len(hits) <= 4
    _, words = oracle.mwc64x_stream(123, 3, 2 * 7 + 4 * 2000)
    assert [s for s in range(2000) if words[14 + 4 * s + 3] >> 9 == 0] == [h for h in hits if h < 2000]
