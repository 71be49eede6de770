"""Throughput floors on the GPU box: not a benchmark (bench.py is), a tripwire.  Each floor is about half of what the
configuration measured on MI355X in rounds 2 and 3, so box-to-box differences cannot trip it, while the
kind of accident that does happen -- a launch-shape chooser picking a bad lane count, a kernel falling back to a slow
variant, state no longer where it should be -- costs more than that.

Two tests per configuration, under two markers:
  * `gpu`  -- test_kernel_variant: which kernel and launch shape pstat_create picked.  No clock is read, so a slow or shared
              box cannot fail it; it runs inside the parity suite (`pytest -m gpu`).
  * `perf` -- test_throughput_floor: the wall-clock floor.  NOT part of `-m gpu` (under `pytest -x` one slow box would stop the
              run before the closed-form and oracle tests that sort after it); run it with `pytest -m perf` on the GPU box."""
import time

import pytest



@pytest.fixture(scope="module")
def ps():
    import polymer_stats_amd as ps
    assert ps._lib.load().pstat_device_count() >= 1, "no HIP device visible"
    return ps


def _rate(ps, steps, **kw):
    """Best of three timed launches after a warm-up one (first-touch allocation, clocks): a floor must not trip on a
    box that was busy for one of them."""
    with ps.Ensemble(ps.default_params(**kw)) as e:
        e.advance(max(500, steps // 10))
        e.sync()
        dt = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            e.advance(steps)
            e.sync()
            dt = min(dt, time.perf_counter() - t0)
        return kw["num_chains"] * steps / dt, e.launch_info()


CASES = [
    # name, floor (updates or proposals per second), steps, expected kernel substring, parameters
    ("f64 sweep n=100 (bench workload)", 4.5e10, 40000, "state in L2",
     dict(n=100, E0=1.0, K1=1.0, Fz=1.0, num_chains=65536, precision=1, seed=1)),
    ("f32 sweep n=100 (fast path)", 1.6e11, 100000, "sweep_kernel<float>",
     dict(n=100, E0=1.0, K1=1.0, Fz=1.0, num_chains=65536, precision=0, seed=1)),
    ("q16 sweep n=100", 1.6e11, 100000, "q16",
     dict(n=100, E0=1.0, K1=1.0, Fz=1.0, num_chains=65536, precision=2, seed=1)),
    ("f64 Ising sweep n=200", 2.0e10, 20000, "state in L2",      # round 3: chain-contiguous working buffer, 2.6e10 -> 4e10
     dict(n=200, E0=1.0, K1=1.0, kT=1.0, energy_type=2, num_chains=65536, precision=1, seed=1)),
    ("f64 all-pairs n=64", 1.6e8, 2000, "interacting_kernel<double>",
     dict(n=64, E0=1.0, K1=1.0, Fz=0.5, energy_type=1, num_chains=16384, precision=1, seed=1)),
    ("f32 all-pairs n=64", 4.0e8, 4000, "interacting_kernel<float>",
     dict(n=64, E0=1.0, K1=1.0, Fz=0.5, energy_type=1, num_chains=16384, precision=0, seed=1)),
    ("f32 clustering main n=100", 8.0e9, 5000, "cluster_kernel<float>",
     dict(n=100, E0=1.0, K1=0.0, K2=1.0, num_chains=65536, precision=0, seed=1, move_set=1, cluster_prob=0.5, adj_ub=0.40)),
    ("q16 clustering main n=100", 9.0e9, 5000, "cluster_kernel",
     dict(n=100, E0=1.0, K1=0.0, K2=1.0, num_chains=65536, precision=2, seed=1, move_set=1, cluster_prob=0.5, adj_ub=0.40)),
    # round 3: the f64 clustering main keeps its chains in device memory (pstat_cluster_gm.hip): 4.1e9 -> 1.2e10 / 5.8e9 -> 1.5e10
    ("f64 clustering main n=100", 6.0e9, 3000, "cluster_kernel<double, state in memory>",
     dict(n=100, E0=1.0, K1=0.0, K2=1.0, num_chains=65536, precision=1, seed=1, move_set=1, cluster_prob=0.5, adj_ub=0.40)),
    ("f64 clustering main n=100 Ising", 7.5e9, 3000, "cluster_kernel<double, state in memory>",
     dict(n=100, E0=1.0, K1=0.0, K2=1.0, energy_type=2, num_chains=65536, precision=1, seed=1, move_set=1, cluster_prob=0.5,
          adj_ub=0.40)),
    ("f64 clustering main n=200", 4.5e9, 2000, "cluster_kernel<double, state in memory>",
     dict(n=200, E0=1.0, K1=0.0, K2=1.0, num_chains=65536, precision=1, seed=1, move_set=1, cluster_prob=0.5, adj_ub=0.40)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,floor,steps,kernel,kw", CASES, ids=[c[0] for c in CASES])
def test_kernel_variant(ps, name, floor, steps, kernel, kw):
    """The fallback-variant tripwire without a clock: the kernel pstat_create picks for the configuration, that it has a
    resident slot on every CU, and that a short launch of it completes."""
    with ps.Ensemble(ps.default_params(**kw)) as e:
        info = e.launch_info()
        assert kernel in info.kernel.decode(), info.kernel.decode()
        assert info.blocks_per_cu >= 1 and info.blocks >= 1 and 1 <= info.lanes_per_block <= 64
        e.advance(64)
        e.sync()
        assert e.summary().steps_per_chain == 64


@pytest.mark.perf
@pytest.mark.parametrize("name,floor,steps,kernel,kw", CASES, ids=[c[0] for c in CASES])
def test_throughput_floor(ps, name, floor, steps, kernel, kw):
    rate, info = _rate(ps, steps, **kw)
    assert kernel in info.kernel.decode(), info.kernel.decode()
    assert rate > floor, f"{name}: {rate:.3e} per second, floor {floor:.1e} ({info.kernel.decode()}, " \
                         f"{info.lanes_per_block} lanes x {info.blocks_per_cu} workgroups per CU)"


@pytest.mark.perf
def test_packing_pays_where_it_is_chosen(ps, monkeypatch):
    """2 730 cases x 16 chains of the fixed-force main: pstat_create packs four cases to a wave on its own, and the launch is
    at least 1.6 x faster than with every workgroup inside one case (measured 2.25 x)."""
    cases = [ps.default_params(n=100, E0=0.2 * (i // 21 % 26), K1=1.0, kT=10 ** (-2 + 0.2 * (i % 21)), num_chains=16, precision=1,
                               seed=1000 + i) for i in range(2730)]
    rate = {}
    for pack in (None, "0"):
        if pack is None:
            monkeypatch.delenv("PSTAT_PACK", raising=False)
        else:
            monkeypatch.setenv("PSTAT_PACK", pack)
        with ps.Ensemble(cases) as e:
            assert e.launch_info().packed_cases == (1 if pack is None else 0)
            e.advance(4000); e.sync()
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter(); e.advance(40000); e.sync()
                best = min(best, time.perf_counter() - t0)
            rate[pack] = 2730 * 16 * 40000 / best
    assert rate[None] > 1.6 * rate["0"], rate


def _phase_grid(ps, per):
    """run/K1_E0-kT-phase.jl:17-45: 26 x 21 (E0, kT) points x 5 runs, clustering main, Ising, n = 100."""
    return [ps.default_params(n=100, E0=0.2 * (i // 21 % 26), K1=1.0, K2=0.0, kT=10 ** (-2 + 0.2 * (i % 21)), num_chains=per, precision=1,
                              seed=1000 + i, move_set=1, cluster_prob=0.5, energy_type=2) for i in range(2730)]


@pytest.mark.gpu
def test_phase_scan_kernel_variant(ps):
    """The reference's own phase scan (one chain per case) runs one chain per wavefront, four waves to a SIMD."""
    with ps.Ensemble(_phase_grid(ps, 1)) as e:
        info = e.launch_info()
        assert info.kernel.decode() == "cluster_chain_wave_kernel<double>" and info.blocks == 2730 and info.packed_cases == 0
        assert info.blocks_per_cu >= 16, info.blocks_per_cu
        e.advance(64); e.sync()
        assert e.summary(0).steps_per_chain == 64


@pytest.mark.perf
def test_chain_per_wavefront_pays_on_a_phase_scan(ps, monkeypatch):
    """The reference's phase scan, one chain per case: the chain-per-wavefront kernel steps the 2 730 chains in under 6 us
    (measured 2.8) and at least 4 x faster than the chain-per-lane kernel does (measured 10 x)."""
    t = {}
    for home in (None, "global"):
        if home is None:
            monkeypatch.delenv("PSTAT_F64_STATE", raising=False)
        else:
            monkeypatch.setenv("PSTAT_F64_STATE", home)
        with ps.Ensemble(_phase_grid(ps, 1)) as e:
            assert ("chain_wave" in e.launch_info().kernel.decode()) == (home is None)
            e.advance(2000); e.sync()
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter(); e.advance(10000); e.sync()
                best = min(best, time.perf_counter() - t0)
            t[home] = best / 10000
    assert t[None] < 6e-6 and t["global"] > 4 * t[None], t
