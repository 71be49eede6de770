"""Resolution of the Metropolis uniform (include/pstat.h, "RESOLUTION OF THE UNIFORM DRAWS").

The reference tests `rand() < exp(delta)` with a Float64 rand() of 52-53 random bits (mcmc_eap_chain.jl:287,
inc/acceptance.jl:29-39).  A 23-bit eps -- all the f32 kernels can compare -- is 0 once per 2^23 proposals, and then ANY
proposal with exp(delta) > 0 is accepted: a floor of 2^-23 = 1.2e-7 under every acceptance probability.  The f64 kernels
therefore build eps from 53 bits by default (`uniform_bits`), out of the step's own words.  These tests pin both
behaviours on a cold, strongly coupled chain -- where uphill proposals of tens to hundreds of kT are the rule, the one
regime in which the floor is visible -- and check the device against the oracle under either contract."""
import numpy as np
import pytest

from helpers import both

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ps():
    import polymer_stats_amd as ps
    assert ps._lib.load().pstat_device_count() >= 1, "no HIP device visible"
    return ps


# A cold chain with the adaptation switched off (--step-adjust-scale 1): the monomers sit at the poles (theta ~ sqrt(kT)) and
# a full-size proposal is uphill by up to E0^2 K1 / (2 kT) = 100 kT -- far beyond -log(2^-23) = 16, far from the 745 at which
# exp() underflows.  (With adaptation on, the step sizes shrink until proposals are uphill by O(1) kT and the floor is invisible.)
COLD = dict(n=12, E0=1.0, K1=1.0, K2=0.0, kT=0.005, Fz=0.0, adj_scale=1.0, seed=77)


def test_the_23_bit_floor_is_real_and_the_53_bit_default_removes_it(ps, oracle):
    """Scan the streams of 256 chains x 400 000 steps for Metropolis words whose 23 leading bits are zero (expected: 256 x
    4e5 / 2^23 = 12).  Under uniform_bits = 23 such a step accepts ANY proposal whose exp(delta) is positive, however far
    uphill (here: every proposal that the theta clamp does not send to sin(theta') = 0); under 53 bits the same step
    accepts only if eps' low bits allow, i.e. hardly ever.  The device reproduces the oracle's trajectory up to and
    including that step under both settings."""
    nsteps, nchains = 400000, 256
    op23, _ = both(nsteps, uniform_bits=23, **COLD)
    hits = []
    for c in range(nchains):
        h = oracle.find_eps23_zero(op23, c, nsteps)
        if h:
            hits.append((c, h[0]))                 # a chain's first such step: both contracts have made the same decisions before it
    assert 3 <= len(hits) <= 40, len(hits)
    acc = {23: 0, 53: 0}
    ordinary = []
    for c, s in hits:
        for bits in (23, 53):
            op, pp = both(s + 1, num_chains=1, chain_id0=c, precision=ps.F64, uniform_bits=bits, **COLD)
            o = oracle.run(op, chain_id=c, mode="fast", trace=True)
            acc[bits] += int(o.accepted[s])
            ordinary.append(o.accepted[:s].mean() if s else 0.0)
            with ps.Ensemble(pp) as e:            # the device under the same contract: same trajectory, same decision
                e.advance(s + 1)
                g = e.chain_state(0)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (c, s, bits)
            assert g["nacc_total"] == o.nacc_total and np.array_equal(g["rng"], o.rng), (c, s, bits)
    # the chain's ordinary acceptance ratio is a few percent; at the scanned steps the 23-bit contract accepts about every
    # second proposal (all but the clamped ones), the 53-bit contract about as many as anywhere else
    assert max(ordinary) < 0.2, max(ordinary)
    assert acc[23] >= max(3, len(hits) // 4), (acc, len(hits))
    assert acc[53] <= acc[23] - 2, (acc, len(hits))


@pytest.mark.parametrize("kw", [dict(n=20, E0=1.0, K1=1.0, Fz=1.0, seed=3),
                                dict(n=30, E0=2.0, K1=0.2, K2=0.5, Fx=0.3, Fz=0.2, do_flips=1, umbrella=1, seed=4),
                                dict(n=60, E0=1.5, K1=0.3, energy_type=2, kT=0.7, seed=5),
                                dict(n=64, E0=1.0, K1=0.5, Fz=0.5, energy_type=1, seed=6)],
                         ids=["sweep", "sweep-rare", "ising-in-memory", "all-pairs"])
@pytest.mark.parametrize("bits", [23, 53])
def test_f64_bit_parity_under_either_contract(ps, oracle, kw, bits):
    nsteps = 1500 if kw.get("energy_type") == 1 else 6000
    op, pp = both(nsteps, num_chains=16, precision=ps.F64, uniform_bits=bits, **kw)
    with ps.Ensemble(pp) as e:
        e.advance(nsteps)
        for c in (0, 7, 15):
            o = oracle.run(op, chain_id=c, mode="fast", trace=True)
            g = e.chain_state(c)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (bits, c)
            assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total


@pytest.mark.parametrize("et", [0, 2, 1], ids=["non-interacting", "Ising", "interacting"])
@pytest.mark.parametrize("bits", [23, 53])
def test_f64_cluster_bit_parity_under_either_contract(ps, oracle, et, bits):
    kw = dict(n=24, E0=1.0, K1=0.3 if et else 1.0, K2=0.1, Fz=0.4, seed=8, energy_type=et, cluster_prob=0.5, bend_mod=0.3)
    op, pp = both(1200, num_chains=8, precision=ps.F64, uniform_bits=bits, **kw)
    pp.move_set = ps.MOVES_CLUSTER
    with ps.Ensemble(pp) as e:
        e.advance(1200)
        for c in (0, 7):
            o = oracle.run(op, chain_id=c, mode="cluster", trace=True)
            g = e.chain_state(c)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (et, bits, c)
            assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total


def test_the_two_contracts_share_one_stream_and_the_default_is_53_for_f64(ps):
    """No extra draw: the generator state after N steps is the same under 23 and 53 bits as long as no decision differed,
    and it does not in 2 000 warm steps (a differing decision needs eps23 = 0-ish: ~1e-7 per step).  uniform_bits = 0 means
    53 for f64; f32 accepts 0 and 23 only."""
    kw = dict(n=20, E0=1.0, K1=1.0, Fz=0.5, seed=12, num_chains=64, precision=ps.F64)
    st = {}
    for bits in (0, 23, 53):
        with ps.Ensemble(ps.default_params(uniform_bits=bits, **kw)) as e:
            e.advance(2000)
            st[bits] = e.chain_state(5)
    for bits in (23, 53):
        assert np.array_equal(st[0]["rng"], st[bits]["rng"]) and np.array_equal(st[0]["theta"], st[bits]["theta"])
    with pytest.raises(ps._lib.PstatError) as ei:
        ps.Ensemble(ps.default_params(uniform_bits=53, n=20, num_chains=64, precision=ps.F32))
    assert ei.value.code == -1 and "uniform_bits" in str(ei.value)
    with ps.Ensemble(ps.default_params(uniform_bits=23, n=20, num_chains=64, precision=ps.F32)) as e:
        e.advance(10)
