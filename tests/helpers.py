"""Shared helpers: build matching parameter sets for the CPU oracle and for libpstat."""
import numpy as np

PHYS = ("E0", "K1", "K2", "mu", "kT", "Fz", "Fx", "b")
SHARED = PHYS + ("phi_step", "theta_step", "adj_lb", "adj_ub", "adj_scale", "n", "steps_per_adjust",
                 "seed", "chain_type", "energy_type", "do_flips", "umbrella", "rng",
                 # clustering main
                 "bend_mod", "bend_angle", "cluster_prob", "use_x0", "x0_phi", "x0_theta", "dx0_phi", "dx0_theta",
                 "cutoff_radius", "uniform_bits")


def both(num_steps, num_chains=64, precision=1, chain_id0=0, num_inits=1, force_init=0, stepout=0, **kw):
    """Returns (oracle_params, pstat_params) describing the same ensemble."""
    from oracle import binding as ob
    import polymer_stats_amd as ps
    shared = {k: v for k, v in kw.items() if k in SHARED}
    unknown = set(kw) - set(SHARED)
    assert not unknown, unknown
    op = ob.make_params(num_steps=num_steps, num_inits=num_inits, force_init=force_init,
                        stepout=stepout, **shared)
    pp = ps.default_params(num_chains=num_chains, precision=precision, chain_id0=chain_id0, **shared)
    return op, pp


def pooled(sums, norm):
    """Per-chain means -> (pooled mean, standard error) per observable."""
    m = sums / norm[:, None]
    return m.mean(axis=0), m.std(axis=0, ddof=1) / np.sqrt(m.shape[0])
