"""ctypes binding of libpstat.so (include/pstat.h).  No CPU fallback: if the HIP library is missing
or no GPU is visible, calls fail loudly."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PSTAT_LIB: another build of the same library (kernel experiments: tools/build_variant.sh); never needed in production
LIB_PATH = os.environ.get("PSTAT_LIB") or os.path.join(HERE, "libpstat.so")

DIELECTRIC, POLAR = 0, 1
NONINTERACTING, INTERACTING, ISING, CUTOFF = 0, 1, 2, 3
F32, F64, Q16 = 0, 1, 2
RNG_MWC64X, RNG_XOSHIRO128PP = 0, 1
NOBS = 16
NQ = 19
NX = 2
NRED = 1 + 2 * NQ + NX
MWC64X_MAX_CHAINS = 1 << 22
MOVES_SINGLE, MOVES_CLUSTER = 0, 1
OBS_NAMES = ["r1", "r2", "r3", "r1sq", "r2sq", "r3sq", "rsq",
             "p1", "p2", "p3", "p1sq", "p2sq", "p3sq", "psq", "U", "Usq"]

# every symbol include/pstat.h declares (tests check the built library exports all of them)
SYMBOLS = [
    "pstat_abi_version", "pstat_strerror", "pstat_last_error", "pstat_device_count",
    "pstat_default_params", "pstat_create", "pstat_destroy", "pstat_advance", "pstat_sync",
    "pstat_reinit", "pstat_reset_averages", "pstat_set_kT", "pstat_scale_kT", "pstat_reset_sampler", "pstat_reduce_device", "pstat_reduce_host", "pstat_rolling", "pstat_microstate",
    "pstat_summary_get", "pstat_summary_from_reduction", "pstat_chain_state", "pstat_chain_extras", "pstat_restart_from_x0",
    "pstat_checkpoint", "pstat_restore", "pstat_launch_info_get", "pstat_chain_means",
]
ABI_VERSION = 6


class PstatError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str):
        super().__init__(f"libpstat: {what} ({code}): {detail}")
        self.code = code


class Params(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("E0", "K1", "K2", "mu", "kT", "Fz", "Fx", "b",
                 "phi_step", "theta_step", "adj_lb", "adj_ub", "adj_scale")] + \
               [("steps_per_adjust", C.c_int64), ("n", C.c_int64), ("num_chains", C.c_int64),
                ("seed", C.c_uint64), ("chain_id0", C.c_uint64)] + \
               [(k, C.c_int32) for k in
                ("chain_type", "energy_type", "do_flips", "umbrella", "precision", "device", "rng", "move_set")] + \
               [(k, C.c_double) for k in
                ("bend_mod", "bend_angle", "cluster_prob", "x0_phi", "x0_theta", "dx0_phi", "dx0_theta")] + \
               [("use_x0", C.c_int32), ("uniform_bits", C.c_int32), ("cutoff_radius", C.c_double)]


class Summary(C.Structure):
    _fields_ = [("avg", C.c_double * NOBS), ("stderr", C.c_double * NOBS),
                ("acceptance_ratio", C.c_double), ("ar_stderr", C.c_double),
                ("num_chains", C.c_int64), ("steps_per_chain", C.c_int64),
                ("attempted_updates", C.c_double),
                ("extra_avg", C.c_double * 2), ("extra_stderr", C.c_double * 2),
                ("nan_rejects", C.c_int64), ("chains_collapsed", C.c_int64)]


class LaunchInfo(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("lds_bytes", C.c_int32), ("threads_per_block", C.c_int32),
                ("lanes_per_block", C.c_int32), ("blocks", C.c_int64), ("blocks_per_cu", C.c_int32),
                ("num_cus", C.c_int32), ("packed_cases", C.c_int32), ("reserved", C.c_int32)]


_lib = None


def load():
    """Loads libpstat.so (built by `make -C polymer_stats_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C polymer_stats_amd/csrc` "
                          "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dp = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_double)
    L.pstat_abi_version.restype = C.c_int
    L.pstat_strerror.argtypes = [C.c_int]
    L.pstat_strerror.restype = C.c_char_p
    L.pstat_last_error.restype = C.c_char_p
    L.pstat_device_count.restype = C.c_int
    L.pstat_default_params.argtypes = [C.POINTER(Params)]
    L.pstat_default_params.restype = None
    L.pstat_create.argtypes = [C.POINTER(Params), i32, vp, C.POINTER(vp)]
    L.pstat_destroy.argtypes = [vp]
    L.pstat_destroy.restype = None
    L.pstat_advance.argtypes = [vp, i64]
    L.pstat_sync.argtypes = [vp]
    L.pstat_reinit.argtypes = [vp, i32]
    L.pstat_reset_averages.argtypes = [vp]
    L.pstat_set_kT.argtypes = [vp, i32, C.c_double]
    L.pstat_reset_sampler.argtypes = [vp]
    L.pstat_scale_kT.argtypes = [vp, C.c_double]
    L.pstat_restart_from_x0.argtypes = [vp, dp, i64, C.c_double, C.c_double]
    L.pstat_chain_extras.argtypes = [vp, i64, dp, dp]
    L.pstat_reduce_device.argtypes = [vp, i32, vp]
    L.pstat_reduce_host.argtypes = [vp, i32, dp]
    L.pstat_rolling.argtypes = [vp, i32, dp, dp]
    L.pstat_microstate.argtypes = [vp, i64, dp]
    L.pstat_summary_get.argtypes = [vp, i32, C.POINTER(Summary)]
    L.pstat_summary_from_reduction.argtypes = [dp, i64, C.POINTER(Summary)]
    L.pstat_chain_state.argtypes = [vp, i64, dp, dp, C.POINTER(C.c_int64), dp, C.POINTER(C.c_uint32)]
    L.pstat_checkpoint.argtypes = [vp, vp, C.POINTER(C.c_size_t)]
    L.pstat_restore.argtypes = [vp, vp, C.c_size_t]
    L.pstat_launch_info_get.argtypes = [vp, C.POINTER(LaunchInfo)]
    L.pstat_chain_means.argtypes = [vp, i32, dp]
    if L.pstat_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {L.pstat_abi_version()}, this binding needs {ABI_VERSION}: "
                          "rebuild it with `make -C polymer_stats_amd/csrc`")
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        L = load()
        raise PstatError(rc, L.pstat_strerror(rc).decode(), L.pstat_last_error().decode())


def default_params(**kw) -> Params:
    p = Params()
    load().pstat_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p
