"""pstat_checkpoint / pstat_restore (include/pstat.h): the image continues exactly the run it was taken from, on a handle
created with the same options -- and on nothing else.  The reference has no counterpart; this is the north
star's "spill only at checkpoint" made safe: format 4 names the generator, move set, umbrella, do-flips, uniform_bits,
seed / chain ids, ABI version and a fingerprint of every other option, carries every case's current kT, and
pstat_restore refuses whatever does not match with PSTAT_ERR_BAD_CHECKPOINT, leaving the handle untouched."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BAD_CHECKPOINT, TOO_SMALL = -6, -7


@pytest.fixture(scope="module")
def ps():
    import polymer_stats_amd as ps
    assert ps._lib.load().pstat_device_count() >= 1, "no HIP device visible"
    return ps


BASE = dict(n=24, E0=1.0, K1=1.0, K2=0.1, Fz=0.4, seed=31, num_chains=64, precision=1)


def _refused(ps, handle, blob, needle):
    before = handle.chain_state(3)
    with pytest.raises(ps._lib.PstatError) as ei:
        handle.restore(blob)
    assert ei.value.code == BAD_CHECKPOINT and needle in str(ei.value), str(ei.value)
    after = handle.chain_state(3)                 # untouched
    assert np.array_equal(before["theta"], after["theta"]) and np.array_equal(before["rng"], after["rng"])


def test_round_trip_and_every_mismatch_is_refused(ps):
    with ps.Ensemble(ps.default_params(**BASE)) as a:
        a.advance(700)
        blob = a.checkpoint()
        a.advance(300)
        want = a.chain_state(9)
        with ps.Ensemble(ps.default_params(**BASE)) as b:          # a fresh handle with the same options resumes exactly
            b.restore(blob)
            b.advance(300)
            got = b.chain_state(9)
            for k in ("theta", "phi", "rng", "sums"):
                assert np.array_equal(want[k], got[k]), k
            assert want["nacc_total"] == got["nacc_total"] and got["steps_recorded"] == 1000
            # truncated, foreign and too-short buffers
            _refused(ps, b, blob[:len(blob) // 2], "truncated")
            _refused(ps, b, blob[:40], "not even a header")
            _refused(ps, b, b"\0" * len(blob), "magic")
            _refused(ps, b, blob[:8] + (5).to_bytes(4, "little") + blob[12:], "ABI version 5")
    others = [
        (dict(rng=1), "generator"),                                 # MWC64X words would be read as xoshiro128++ state
        (dict(move_set=1, cluster_prob=0.5), "move_set"),
        (dict(umbrella=1), "umbrella"),
        (dict(do_flips=1), "do-flips"),
        (dict(uniform_bits=23), "uniform_bits"),
        (dict(seed=32), "seed"),
        (dict(chain_id0=64), "chain_id0"),
        (dict(n=25), "num-monomers"),
        (dict(num_chains=128), "number of chains"),
        (dict(precision=0), "precision"),
        (dict(chain_type=1), "chain-type"),
        (dict(energy_type=2), "energy-type"),
        (dict(E0=1.5), "physics scalar"),
        (dict(Fz=0.41), "physics scalar"),
        (dict(phi_step=1.0), "adaptation option"),
        (dict(steps_per_adjust=1000), "adaptation option"),
    ]
    for change, needle in others:
        with ps.Ensemble(ps.default_params(**dict(BASE, **change))) as other:
            other.advance(10)
            _refused(ps, other, blob, needle)
    # a batch: the number of cases and every case's scalars count
    cases = [ps.default_params(**dict(BASE, kT=1.0 + 0.5 * i, seed=40 + i)) for i in range(3)]
    with ps.Ensemble(cases) as batch:
        batch.advance(100)
        bblob = batch.checkpoint()
        _refused(ps, batch, blob, "number of c")
        cases[2].Fz = 0.5
        with ps.Ensemble(cases) as batch2:
            _refused(ps, batch2, bblob, "physics scalar")


def test_checkpoint_buffer_too_small_reports_the_size(ps):
    lib = ps._lib.load()
    with ps.Ensemble(ps.default_params(**BASE)) as e:
        e.advance(50)
        need = C.c_size_t(0)
        assert lib.pstat_checkpoint(e._h, None, C.byref(need)) == 0 and need.value > 64 * 24 * 16
        small = C.create_string_buffer(need.value - 1)
        size = C.c_size_t(need.value - 1)
        assert lib.pstat_checkpoint(e._h, small, C.byref(size)) == TOO_SMALL
        assert size.value == need.value and b"needs" in lib.pstat_last_error()
        size = C.c_size_t(0)
        assert lib.pstat_checkpoint(e._h, small, C.byref(size)) == TOO_SMALL and size.value == need.value
        big = C.create_string_buffer(need.value + 100)
        size = C.c_size_t(need.value + 100)
        assert lib.pstat_checkpoint(e._h, big, C.byref(size)) == 0 and size.value == need.value


def test_a_rung_of_the_burn_in_ladder_is_restored_with_its_temperature(ps, oracle):
    """mcmc_clustering_eap_chain.jl:365-386: the burn-in runs on a kT ladder.  A checkpoint taken on the 10 x kT rung carries
    that temperature: restored into a fresh handle (which sits at the creation-time kT) it continues on the rung, bit for
    bit what the uninterrupted handle does; without the restore the fresh handle's chains differ."""
    kw = dict(n=20, E0=1.0, K1=0.0, K2=1.0, kT=0.5, seed=5, num_chains=32, precision=1, move_set=1, cluster_prob=0.5)
    cases = [ps.default_params(**dict(kw, kT=0.5 * (1 + i), seed=5 + i)) for i in range(2)]
    with ps.Ensemble(cases) as a, ps.Ensemble(cases) as b:
        a.scale_kT(10.0)
        a.advance(400)
        blob = a.checkpoint()
        a.advance(400)
        b.restore(blob)               # b was created at kT = 0.5, 1.0 and never told about the rung
        b.advance(400)
        for c in (0, 31, 32, 63):
            x, y = a.chain_state(c), b.chain_state(c)
            assert np.array_equal(x["theta"], y["theta"]) and np.array_equal(x["rng"], y["rng"]) and x["nacc_total"] == y["nacc_total"]
        # and the ladder goes on from there: back to the base temperature on both
        a.scale_kT(1.0); b.scale_kT(1.0)
        a.advance(200); b.advance(200)
        assert np.array_equal(a.chain_state(40)["theta"], b.chain_state(40)["theta"])
        # per-case pstat_set_kT is carried the same way
        a.set_kT(3.0, icase=1)
        blob2 = a.checkpoint()
        a.advance(300)
        b.restore(blob2)
        b.advance(300)
        assert np.array_equal(a.chain_state(50)["theta"], b.chain_state(50)["theta"])
        assert np.array_equal(a.chain_state(5)["theta"], b.chain_state(5)["theta"])
