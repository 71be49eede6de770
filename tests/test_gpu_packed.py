"""Packed cases: when a case has fewer chains than a wave has lanes, pstat_create lets a workgroup hold `lanes` consecutive
GLOBAL chains, whichever cases they belong to (run_job_queue<true>, csrc/pstat_device.h) -- what the reference's own sweep
drivers need: one chain per case, 5-25 repeats (run/K1_E0-kT-phase.jl:19-45, run/interacting_dielectric_study.jl:37-47).
The case's physics scalars then travel per lane (VGPRs) instead of per wave (SGPRs); nothing else may change: a chain's
trajectory depends on its (seed, chain id) and its case's options, never on which lanes share its wave."""
import numpy as np
import pytest

from helpers import both

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ps():
    import polymer_stats_amd as ps
    assert ps._lib.load().pstat_device_count() >= 1, "no HIP device visible"
    return ps


def _state_equal(a, b, what):
    for k in ("theta", "phi", "rng"):
        assert np.array_equal(a[k], b[k]), (what, k)
    assert a["nacc_total"] == b["nacc_total"] and a["phi_step"] == b["phi_step"] and a["theta_step"] == b["theta_step"], what


# (family, parameters that select the kernel, steps).  n = 30 keeps the f64 sweep's cells in LDS, n = 60 in memory.
FAMILIES = [
    ("f64 sweep, cells in LDS", dict(n=30, precision=1), 2500),
    ("f64 sweep, cells in memory", dict(n=60, precision=1), 2500),
    ("f64 Ising sweep, cells in memory", dict(n=60, precision=1, energy_type=2, K1=0.3), 2000),
    ("f64 sweep, rare options", dict(n=60, precision=1, do_flips=1, umbrella=1, Fx=0.3), 2000),
    ("f64 clustering main", dict(n=40, precision=1, move_set=1, cluster_prob=0.5, bend_mod=0.3, bend_angle=0.2, K2=0.2), 1500),
    ("f64 clustering main, Ising", dict(n=40, precision=1, move_set=1, cluster_prob=0.4, energy_type=2, K1=0.2), 1500),
    ("f64 clustering main, polar", dict(n=24, precision=1, move_set=1, cluster_prob=0.5, chain_type=1, mu=0.7), 1500),
]


def _cases(ps, fam_kw, ncases, nchains):
    out = []
    for i in range(ncases):
        kw = dict(E0=0.5 + 0.25 * i, K1=1.0, K2=0.0, kT=0.6 + 0.2 * i, Fz=0.1 * i, b=1.0 + 0.05 * i, seed=300 + i,
                  chain_id0=1000 * i, num_chains=nchains)
        kw.update(fam_kw)
        if fam_kw.get("move_set"):
            kw["cluster_prob"] = min(0.9, fam_kw["cluster_prob"] + 0.05 * i)      # per-case options of the clustering main
            kw["bend_mod"] = fam_kw.get("bend_mod", 0.0) * (1 + 0.1 * i)
        out.append(kw)
    return out


@pytest.mark.parametrize("name,fam_kw,nsteps", FAMILIES, ids=[f[0] for f in FAMILIES])
def test_packed_batch_equals_single_case_handles_and_the_oracle(ps, oracle, monkeypatch, name, fam_kw, nsteps):
    """8 cases x 16 chains in one handle, packed four cases to a wave: every chain bit-identical to the same chain in a
    single-case handle (blocks inside the case, scalars in SGPRs) and to the oracle; per-case averages equal."""
    cases = _cases(ps, fam_kw, 8, 16)
    monkeypatch.setenv("PSTAT_PACK", "1")
    batch = ps.Ensemble([ps.default_params(**kw) for kw in cases])
    info = batch.launch_info()
    assert info.packed_cases == 1 and "[packed cases]" in info.kernel.decode(), info.kernel.decode()
    assert info.blocks == -(-8 * 16 // info.lanes_per_block)
    batch.advance(nsteps)
    batch.sync()
    monkeypatch.setenv("PSTAT_PACK", "0")
    mode = "cluster" if fam_kw.get("move_set") else "fast"
    for i, kw in enumerate(cases):
        with ps.Ensemble(ps.default_params(**kw)) as single:
            assert single.launch_info().packed_cases == 0
            single.advance(nsteps)
            for k in (0, 5, 15):
                _state_equal(batch.chain_state(i * 16 + k), single.chain_state(k), (name, i, k))
                np.testing.assert_allclose(batch.chain_state(i * 16 + k)["sums"], single.chain_state(k)["sums"], rtol=1e-12, atol=1e-9)
            a, _ = batch.rolling(i)
            b, _ = single.rolling(-1)
            np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-10)
        okw = {k: v for k, v in kw.items() if k not in ("num_chains", "precision", "move_set", "chain_id0")}
        op, _ = both(nsteps, **okw)
        for k in (0, 15):
            o = oracle.run(op, chain_id=kw["chain_id0"] + k, mode=mode, trace=True)
            g = batch.chain_state(i * 16 + k)
            assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (name, i, k)
            assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total, (name, i, k)
    batch.close()


@pytest.mark.parametrize("nchains,ncases", [(1, 200), (5, 77), (25, 13), (100, 7)])
def test_packed_ragged_shapes_and_block_boundaries(ps, monkeypatch, nchains, ncases):
    """Chains per case that do not divide the wave (1, 5, 25 -- the reference's repeat counts -- and 100): cases straddle
    waves, the last wave is partly empty.  Packed and unpacked handles agree chain by chain; so does a handle advanced in
    several launches with time segments (every job refills its block from the checkpoint layout)."""
    cases = [ps.default_params(n=48, E0=0.3 * (i % 7), K1=1.0, kT=0.5 + 0.1 * (i % 5), Fz=0.05 * (i % 11), seed=900 + i,
                               num_chains=nchains, precision=ps.F64) for i in range(ncases)]
    states = {}
    for pack, segs in (("0", "1"), ("1", "1"), ("1", "3")):
        monkeypatch.setenv("PSTAT_PACK", pack)
        monkeypatch.setenv("PSTAT_SEGMENTS", segs)
        with ps.Ensemble(cases) as e:
            assert e.launch_info().packed_cases == int(pack)
            e.advance(700)
            e.advance(500)
            total = nchains * ncases
            states[pack + segs] = [e.chain_state(c) for c in sorted({0, 1, 63, 64, 65, total // 2, total - 1} & set(range(total)))]
            states[pack + segs + "avg"] = np.array([e.rolling(i)[0] for i in (0, ncases // 2, ncases - 1)])
    for key in ("11", "13"):
        for a, b in zip(states["01"], states[key]):
            _state_equal(a, b, (nchains, ncases, key))
        np.testing.assert_allclose(states["01avg"], states[key + "avg"], rtol=1e-11, atol=1e-10)


def test_pstat_create_packs_when_it_shortens_the_launch_and_only_then(ps, monkeypatch):
    """The chooser itself (no PSTAT_PACK): an ensemble of many few-chain cases that overflows the chip's resident waves
    is packed; full-wave cases, a single case, and ensembles that fit the chip either way keep their SGPR scalars."""
    monkeypatch.delenv("PSTAT_PACK", raising=False)
    P = lambda **kw: ps.default_params(n=100, E0=1.0, K1=0.0, K2=1.0, precision=ps.F64, **kw)
    grid = lambda m, nc, **kw: [P(kT=0.5 + 0.001 * i, num_chains=nc, seed=i, **kw) for i in range(m)]
    cl = dict(move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, energy_type=ps.ISING)
    # (sweeps of many small f64 cases run the clustering main one chain per wavefront, pstat_cluster_cw.hip: nothing to pack; the
    # chain-per-lane kernel's own chooser is asked for below)
    for m, nc in ((2730, 16), (2730, 5), (2730, 1)):
        with ps.Ensemble(grid(m, nc, **cl)) as e:
            info = e.launch_info()
            assert info.packed_cases == 0 and info.blocks == m * nc and "cluster_chain_wave_kernel" in info.kernel.decode()
    monkeypatch.setenv("PSTAT_F64_STATE", "global")
    with ps.Ensemble(grid(2730, 16, **cl)) as e:         # run/K1_E0-kT-phase.jl: 546 points x 5 runs, 16 chains each here: the
        assert e.launch_info().packed_cases == 0         # clustering main is paced by its cold cases: no four-case waves
    with ps.Ensemble(grid(2730, 5, **cl)) as e:          # ... but the idle lanes of a 16-lane wave are filled with further cases
        info = e.launch_info()
        assert info.packed_cases == 1 and info.lanes_per_block == 16 and info.blocks == -(-2730 * 5 // 16)
    with ps.Ensemble(grid(9000, 8, **cl)) as e:          # ... and full waves win once workgroups queue many deep
        info = e.launch_info()
        assert info.packed_cases == 1 and info.lanes_per_block == 64 and info.blocks == -(-9000 * 8 // 64)
    with ps.Ensemble(grid(546, 64, **cl)) as e:          # the phase scan: 64 chains per point divide the wave
        assert e.launch_info().packed_cases == 0
    monkeypatch.delenv("PSTAT_F64_STATE")
    with ps.Ensemble(grid(3000, 16)) as e:               # fixed-force main, f64 cells in memory: 3 000 quarter waves > 1 024 slots
        info = e.launch_info()
        assert info.packed_cases == 1 and info.blocks == 750 and "state in L2" in info.kernel.decode()
    with ps.Ensemble(grid(174, 16)) as e:                # fits the chip either way: nothing to gain
        assert e.launch_info().packed_cases == 0
    with ps.Ensemble(P(num_chains=1000, seed=1)) as e:   # one case
        assert e.launch_info().packed_cases == 0
    with ps.Ensemble(grid(3000, 16, energy_type=ps.INTERACTING)[:40]) as e:    # all-pairs: a chain per wavefront, nothing to pack
        assert e.launch_info().packed_cases == 0


def test_packed_f32_sweep_statistics_and_unpacked_agreement(ps, monkeypatch):
    """The f32 fast path packs too (its LDS-sized workgroups hold 51 lanes at n = 100): same chains, same trajectories
    as unpacked (f32 arithmetic is deterministic per chain)."""
    cases = [ps.default_params(n=100, E0=1.0, K1=1.0, Fz=0.1 * i, seed=50 + i, num_chains=10, precision=ps.F32) for i in range(40)]
    out = {}
    for pack in ("0", "1"):
        monkeypatch.setenv("PSTAT_PACK", pack)
        with ps.Ensemble(cases) as e:
            assert e.launch_info().packed_cases == int(pack)
            e.advance(3000)
            out[pack] = ([e.chain_state(c) for c in (0, 9, 10, 199, 399)], np.array([e.rolling(i)[0] for i in range(40)]))
    for a, b in zip(out["0"][0], out["1"][0]):
        assert np.array_equal(a["theta"], b["theta"]) and np.array_equal(a["phi"], b["phi"]) and a["nacc_total"] == b["nacc_total"]
    np.testing.assert_allclose(out["0"][1], out["1"][1], rtol=1e-5, atol=1e-4)


def test_packed_handles_through_the_rest_of_the_abi(ps, oracle, monkeypatch):
    """What else a host does with a batched handle, on packed blocks: re-initialisation between inits (fixed-force main),
    a burn-in rung (pstat_scale_kT / reset_sampler / reset_averages) and a per-case pstat_set_kT (clustering main),
    checkpoint + restore into a fresh packed handle, per-case reductions and pstat_chain_means -- each equal to the same
    calls on unpacked handles, chain by chain."""
    mk = lambda i, **kw: ps.default_params(n=50, E0=0.4 + 0.2 * i, K1=1.0, K2=0.1, Fz=0.1 * i, kT=0.7 + 0.1 * i, seed=4000 + i,
                                           num_chains=12, precision=ps.F64, **kw)
    results = {}
    for pack in ("0", "1"):
        monkeypatch.setenv("PSTAT_PACK", pack)
        out = []
        # fixed-force main: two inits with a Metropolis re-init between them, then checkpoint / restore
        with ps.Ensemble([mk(i) for i in range(9)]) as e:
            assert e.launch_info().packed_cases == int(pack)
            e.advance(900)
            e.reinit(False)
            e.advance(400)
            blob = e.checkpoint()
            e.advance(300)
            out.append([e.chain_state(c) for c in (0, 11, 12, 60, 107)])
            out.append(np.array([e.reduce_host(i) for i in range(9)]))
            out.append(e.chain_means(4))
            with ps.Ensemble([mk(i) for i in range(9)]) as f:
                f.restore(blob)
                f.advance(300)
                for c in (0, 11, 12, 60, 107):
                    _state_equal(e.chain_state(c), f.chain_state(c), ("restore", pack, c))
        # clustering main: one rung at 10 kT, back to kT, then case 3 alone at another temperature
        cl = dict(move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, energy_type=ps.ISING)
        with ps.Ensemble([mk(i, **cl) for i in range(9)]) as e:
            assert e.launch_info().packed_cases == int(pack)
            e.scale_kT(10.0); e.advance(300); e.reset_sampler(); e.reset_averages()
            e.scale_kT(1.0); e.advance(500)
            e.set_kT(2.5, icase=3); e.advance(300)
            out.append([e.chain_state(c) for c in (0, 35, 36, 47, 48, 107)])
            out.append(np.array([e.rolling(i)[0] for i in range(9)]))
        results[pack] = out
    a, b = results["0"], results["1"]
    for x, y in zip(a[0] + a[3], b[0] + b[3]):
        _state_equal(x, y, "packed vs unpacked")
    np.testing.assert_allclose(a[1], b[1], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(a[2], b[2], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a[4], b[4], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("prec", [0, 2], ids=["f32", "q16"])
@pytest.mark.parametrize("main", ["sweep", "cluster"])
def test_every_chain_per_lane_kernel_packs(ps, monkeypatch, prec, main):
    """The f32 / q16 kernels (cells in LDS, LDS-sized workgroups of e.g. 51 lanes) pack too: the same chains, the same
    trajectories as unpacked -- per chain their f32 arithmetic does not depend on what shares the wave."""
    kw = dict(n=60, K1=1.0, K2=0.1, seed=7, num_chains=6, precision=prec)
    if main == "cluster":
        kw.update(move_set=ps.MOVES_CLUSTER, cluster_prob=0.5, bend_mod=0.2)
        monkeypatch.setenv("PSTAT_F32_STATE", "lds")
    cases = [ps.default_params(E0=0.3 + 0.05 * i, Fz=0.02 * i, kT=0.8 + 0.01 * i, chain_id0=100 * i, **kw) for i in range(37)]
    out = {}
    for pack in ("0", "1"):
        monkeypatch.setenv("PSTAT_PACK", pack)
        with ps.Ensemble(cases) as e:
            info = e.launch_info()
            assert info.packed_cases == int(pack) and ("[packed cases]" in info.kernel.decode()) == (pack == "1")
            assert "state in memory" not in info.kernel.decode()
            e.advance(1500)
            e.advance(700)
            out[pack] = ([e.chain_state(c) for c in (0, 5, 6, 100, 221)], np.array([e.rolling(i)[0] for i in range(37)]))
    for a, b in zip(out["0"][0], out["1"][0]):
        assert np.array_equal(a["theta"], b["theta"]) and np.array_equal(a["phi"], b["phi"]) and np.array_equal(a["rng"], b["rng"])
        assert a["nacc_total"] == b["nacc_total"] and a["phi_step"] == b["phi_step"]
    np.testing.assert_allclose(out["0"][1], out["1"][1], rtol=2e-5, atol=2e-4)


def test_packed_f64_cluster_with_cells_in_lds_against_the_oracle(ps, oracle, monkeypatch):
    """The f64 clustering main's LDS home (the literal second witness of the variant tests) in packed blocks: bit parity."""
    monkeypatch.setenv("PSTAT_F64_STATE", "lds")
    monkeypatch.setenv("PSTAT_PACK", "1")
    kws = [dict(n=20, E0=0.5 + 0.1 * i, K1=1.0, K2=0.1, Fz=0.1 * i, kT=0.9 + 0.05 * i, seed=60 + i, cluster_prob=0.5, bend_mod=0.3)
           for i in range(9)]
    params = []
    for i, kw in enumerate(kws):
        _, pp = both(1200, num_chains=7, precision=ps.F64, chain_id0=50 * i, **kw)
        pp.move_set = ps.MOVES_CLUSTER
        params.append(pp)
    with ps.Ensemble(params) as e:
        info = e.launch_info()
        assert info.packed_cases == 1 and info.kernel.decode() == "cluster_kernel<double> [packed cases]"
        e.advance(1200)
        for i in (0, 4, 8):
            op, _ = both(1200, **kws[i])
            for k in (0, 6):
                o = oracle.run(op, chain_id=50 * i + k, mode="cluster", trace=True)
                g = e.chain_state(i * 7 + k)
                assert np.array_equal(g["theta"], o.final_theta) and np.array_equal(g["phi"], o.final_phi), (i, k)
                assert np.array_equal(g["rng"], o.rng) and g["nacc_total"] == o.nacc_total
