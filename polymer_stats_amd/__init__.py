"""polymer_stats_amd -- MI355X (gfx950) implementation of the fixed-force-ensemble MCMC hot path of
grasingerm/polymer-stats (mcmc_eap_chain.jl), behind the C ABI of include/pstat.h."""
from ._lib import (DIELECTRIC, POLAR, NONINTERACTING, INTERACTING, ISING, CUTOFF, F32, F64, Q16, RNG_MWC64X, RNG_XOSHIRO128PP, NOBS, NRED, NQ, MOVES_SINGLE, MOVES_CLUSTER,
                   OBS_NAMES, Params, PstatError, default_params)
from .ensemble import Ensemble, summary_from_reduction

__all__ = ["DIELECTRIC", "POLAR", "NONINTERACTING", "INTERACTING", "ISING", "F32", "F64", "Q16", "RNG_MWC64X", "RNG_XOSHIRO128PP", "NOBS",
           "NRED", "NQ", "MOVES_SINGLE", "MOVES_CLUSTER", "OBS_NAMES", "Params", "PstatError", "default_params", "Ensemble",
           "summary_from_reduction"]
