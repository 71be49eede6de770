"""Ensemble: Python handle on a libpstat ensemble of chains (thin wrapper, no arithmetic of its own)."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Sequence

import numpy as np

from . import _lib
from ._lib import NOBS, NRED, Params, Summary, LaunchInfo, check


class Ensemble:
    """`cases`: one Params or a list of Params that differ only in physics scalars (a sweep grid).
    `stream`: raw hipStream_t (int) to launch on, e.g. torch.cuda.current_stream().cuda_stream."""

    def __init__(self, cases: Params | Sequence[Params], stream: int | None = None):
        self._L = _lib.load()
        if isinstance(cases, Params):
            cases = [cases]
        self.cases = list(cases)
        self.ncases = len(self.cases)
        arr = (Params * self.ncases)(*self.cases)
        self._h = C.c_void_p()
        check(self._L.pstat_create(arr, self.ncases, C.c_void_p(stream) if stream else None,
                                   C.byref(self._h)))
        self.n = int(self.cases[0].n)
        self.num_chains = int(self.cases[0].num_chains)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pstat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # --- the step loop
    def advance(self, nsteps: int):
        check(self._L.pstat_advance(self._h, int(nsteps)))

    def sync(self):
        check(self._L.pstat_sync(self._h))

    def reinit(self, force_init: bool):
        check(self._L.pstat_reinit(self._h, 1 if force_init else 0))

    def reset_averages(self):
        """Discard what has been recorded so far (burn-in); chains, generators, step sizes stay."""
        check(self._L.pstat_reset_averages(self._h))

    def reset_sampler(self):
        """What a fresh mcmc(nsteps, pargs, chain) call of the reference resets besides the averagers:
        step sizes, adaptation window, the acceptor's cache, the in-run step counter."""
        check(self._L.pstat_reset_sampler(self._h))

    def set_kT(self, kT: float, icase: int = -1):
        check(self._L.pstat_set_kT(self._h, icase, float(kT)))

    def restart_from_x0(self, x0, dx0_phi: float, dx0_theta: float):
        """Start every chain over from x0 ([phi, theta] or interleaved per-monomer angles) + Uniform(0, dx0)."""
        a = np.ascontiguousarray(x0, dtype=np.float64)
        check(self._L.pstat_restart_from_x0(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), len(a),
                                            float(dx0_phi), float(dx0_theta)))

    def scale_kT(self, mult: float):
        """kT of every case <- its creation-time kT * mult (one rung of the burn-in ladder for a grid)."""
        check(self._L.pstat_scale_kT(self._h, float(mult)))

    # --- read-outs
    def reduce_into(self, dev_ptr: int, icase: int = -1):
        """Device-side reduction into a caller-owned device buffer of NRED doubles (async)."""
        check(self._L.pstat_reduce_device(self._h, icase, C.c_void_p(dev_ptr)))

    def reduce_host(self, icase: int = -1) -> np.ndarray:
        red = np.zeros(NRED)
        check(self._L.pstat_reduce_host(self._h, icase, red.ctypes.data_as(C.POINTER(C.c_double))))
        return red

    def summary(self, icase: int = -1) -> Summary:
        s = Summary()
        check(self._L.pstat_summary_get(self._h, icase, C.byref(s)))
        return s

    def rolling(self, icase: int = -1):
        avg = np.zeros(NOBS)
        se = np.zeros(NOBS)
        dp = C.POINTER(C.c_double)
        check(self._L.pstat_rolling(self._h, icase, avg.ctypes.data_as(dp), se.ctypes.data_as(dp)))
        return avg, se

    def chain_means(self, icase: int = -1) -> np.ndarray:
        """Per-chain running means, shape [NQ, chains]: what the device reduction folds (pstat_chain_means)."""
        m = self.num_chains * (self.ncases if icase < 0 else 1)
        out = np.zeros((_lib.NQ, m))
        check(self._L.pstat_chain_means(self._h, icase, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def microstate(self, chain: int = 0) -> np.ndarray:
        out = np.zeros(7)
        check(self._L.pstat_microstate(self._h, chain, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def chain_state(self, chain: int) -> dict:
        ang = np.zeros(2 * self.n)
        sums = np.zeros(NOBS)
        cnt = np.zeros(4, dtype=np.int64)
        steps = np.zeros(3)
        rng = np.zeros(4, dtype=np.uint32)
        dp = C.POINTER(C.c_double)
        check(self._L.pstat_chain_state(self._h, chain, ang.ctypes.data_as(dp), sums.ctypes.data_as(dp),
                                        cnt.ctypes.data_as(C.POINTER(C.c_int64)), steps.ctypes.data_as(dp),
                                        rng.ctypes.data_as(C.POINTER(C.c_uint32))))
        return dict(theta=ang[:self.n], phi=ang[self.n:], sums=sums, nacc_total=int(cnt[0]),
                    steps_recorded=int(cnt[1]), nacc_window=int(cnt[2]), natt_window=int(cnt[3]),
                    phi_step=steps[0], theta_step=steps[1], normalizer=steps[2], rng=rng)

    def chain_extras(self, chain: int) -> dict:
        """Clustering main only: running sums and current values of sum cos^2(theta) and <psi>."""
        sums = np.zeros(2)
        now = np.zeros(2)
        dp = C.POINTER(C.c_double)
        check(self._L.pstat_chain_extras(self._h, chain, sums.ctypes.data_as(dp), now.ctypes.data_as(dp)))
        return dict(sums=sums, now=now)

    def checkpoint(self) -> bytes:
        size = C.c_size_t(0)
        check(self._L.pstat_checkpoint(self._h, None, C.byref(size)))
        buf = C.create_string_buffer(size.value)
        check(self._L.pstat_checkpoint(self._h, buf, C.byref(size)))
        return buf.raw[:size.value]

    def restore(self, blob: bytes):
        buf = C.create_string_buffer(blob, len(blob))
        check(self._L.pstat_restore(self._h, buf, len(blob)))

    def launch_info(self) -> LaunchInfo:
        info = LaunchInfo()
        check(self._L.pstat_launch_info_get(self._h, C.byref(info)))
        return info


def summary_from_reduction(red: Iterable[float], steps_per_chain: int) -> Summary:
    """Host arithmetic on an (all-reduced) NRED vector -> pooled averages and standard errors."""
    a = np.ascontiguousarray(np.asarray(list(red), dtype=np.float64))
    assert a.shape == (NRED,)
    s = Summary()
    check(_lib.load().pstat_summary_from_reduction(a.ctypes.data_as(C.POINTER(C.c_double)),
                                                   int(steps_per_chain), C.byref(s)))
    return s
