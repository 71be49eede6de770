"""Generates tests/golden/ni_closed_form.json: equilibrium ensemble averages of NON-INTERACTING
chains by single-monomer quadrature (scipy).  This is our own derivation, not reference output --
the reference holds no fixtures, so parity stays "unpinned".

A non-interacting chain factorises into iid monomers with density (inc/acceptance.jl:18-22,
inc/energy.jl:7-9, inc/eap_chain.jl:53, inc/dipole_response.jl:7-29)

    rho(theta, phi) ~ sin(theta) * exp(-[u(theta) - b (Fx sin(theta) cos(phi) + Fz cos(theta))] / kT)
    u = -1/2 E0 mu_z,  mu = (K1-K2) E0 cos(theta) n + K2 E0 z   (dielectric)   or   mu * n  (polar)

For the dielectric chain this is the density the reference's own mean-field solver integrates,
rho = c exp(-omega0 cos^2(theta) + lambda cos(theta) + alpha cos(phi) sin(theta)) with the measure sin(theta)
(inc/solvers.jl:88-98, mean_field.jl:136): its force-ensemble reading has lambda = Fz b / kT, alpha = Fx b / kT,
omega0 = -(K1-K2) E0^2 / (2 kT) (the K2 term of u is a constant).  The reference cross-checks its MCMC against that
solver for non-interacting chains (run/noninteracting-compare-with-clustering_2021-09-24.jl); these fixtures are the
same check with the integrals done here.

and then   <r> = n b <n>,  <r_j^2> = n b^2 Var(n_j) + <r_j>^2,  <p> = n <mu>,
           <p_j^2> = n Var(mu_j) + <p_j>^2,  <U> = n <w>,  <U^2> = n Var(w) + <U>^2,
with w = u - b F.n the one-monomer energy.  Cross terms vanish by independence.

Run:  python tests/golden/make_closed_form.py   (rewrites the JSON next to this file)
"""
import json
import os

import numpy as np
from scipy import integrate

CASES = {
    # BASELINE.json configs[0]
    "cfg1_n20_E0_0_Fz1": dict(chain="dielectric", n=20, E0=0.0, K1=1.0, K2=0.0, mu=0.01, kT=1.0, Fz=1.0, Fx=0.0, b=1.0),
    # configs[1] at Fz = 1 (the bench workload) and two more points of its Fz sweep
    "cfg2_n100_E0_1_K1_1_Fz1": dict(chain="dielectric", n=100, E0=1.0, K1=1.0, K2=0.0, mu=0.01, kT=1.0, Fz=1.0, Fx=0.0, b=1.0),
    "cfg2_n100_E0_1_K1_1_Fz0": dict(chain="dielectric", n=100, E0=1.0, K1=1.0, K2=0.0, mu=0.01, kT=1.0, Fz=0.0, Fx=0.0, b=1.0),
    "cfg2_n100_E0_1_K1_1_Fz5": dict(chain="dielectric", n=100, E0=1.0, K1=1.0, K2=0.0, mu=0.01, kT=1.0, Fz=5.0, Fx=0.0, b=1.0),
    "diel_n100_E0_2_K2_1_Fz05": dict(chain="dielectric", n=100, E0=2.0, K1=0.0, K2=1.0, mu=0.01, kT=1.0, Fz=0.5, Fx=0.0, b=1.0),
    "diel_n100_E0_1_K1_1_Fx1": dict(chain="dielectric", n=100, E0=1.0, K1=1.0, K2=0.0, mu=0.01, kT=1.0, Fz=0.0, Fx=1.0, b=1.0),
    # configs[2]
    "cfg3_polar_n100_E0_1_mu1_Fz1": dict(chain="polar", n=100, E0=1.0, K1=1.0, K2=0.0, mu=1.0, kT=1.0, Fz=1.0, Fx=0.0, b=1.0),
    "cfg3_polar_n100_E0_10_mu1_Fz025": dict(chain="polar", n=100, E0=10.0, K1=1.0, K2=0.0, mu=1.0, kT=1.0, Fz=0.25, Fx=0.0, b=1.0),
    # a short, cold, anisotropic chain with b != 1 to exercise every parameter
    "diel_n8_E0_15_K1_07_K2_03_Fz04_Fx03_kT07_b13": dict(chain="dielectric", n=8, E0=1.5, K1=0.7, K2=0.3, mu=0.01, kT=0.7, Fz=0.4, Fx=0.3, b=1.3),
}


def one_monomer_moments(c):
    E0, K1, K2, mu, kT, Fz, Fx, b = (c[k] for k in ("E0", "K1", "K2", "mu", "kT", "Fz", "Fx", "b"))

    def fields(th, ph):
        st, ct, sp, cp = np.sin(th), np.cos(th), np.sin(ph), np.cos(ph)
        nh = np.array([cp * st, sp * st, ct])
        if c["chain"] == "dielectric":
            m = (K1 - K2) * E0 * ct * nh + np.array([0.0, 0.0, K2 * E0])
        else:
            m = mu * nh
        u = -0.5 * E0 * m[2]
        w = u - b * (Fx * nh[0] + Fz * nh[2])
        return nh, m, w

    def weight(th, ph):
        return np.sin(th) * np.exp(-fields(th, ph)[2] / kT)

    def integral(f):
        val, _ = integrate.dblquad(lambda ph, th: f(th, ph) * weight(th, ph), 0.0, np.pi, 0.0, 2 * np.pi,
                                   epsabs=1e-13, epsrel=1e-13)
        return val

    Z = integral(lambda th, ph: 1.0)
    out = {}
    for j in range(3):
        out[f"n{j}"] = integral(lambda th, ph: fields(th, ph)[0][j]) / Z
        out[f"n{j}sq"] = integral(lambda th, ph: fields(th, ph)[0][j] ** 2) / Z
        out[f"m{j}"] = integral(lambda th, ph: fields(th, ph)[1][j]) / Z
        out[f"m{j}sq"] = integral(lambda th, ph: fields(th, ph)[1][j] ** 2) / Z
    out["w"] = integral(lambda th, ph: fields(th, ph)[2]) / Z
    out["wsq"] = integral(lambda th, ph: fields(th, ph)[2] ** 2) / Z
    return out


def chain_averages(c):
    m = one_monomer_moments(c)
    n, b = c["n"], c["b"]
    avg = {}
    rsq = psq = 0.0
    for j in range(3):
        rj = n * b * m[f"n{j}"]
        rj2 = n * b * b * (m[f"n{j}sq"] - m[f"n{j}"] ** 2) + rj * rj
        pj = n * m[f"m{j}"]
        pj2 = n * (m[f"m{j}sq"] - m[f"m{j}"] ** 2) + pj * pj
        avg[f"r{j+1}"], avg[f"r{j+1}sq"], avg[f"p{j+1}"], avg[f"p{j+1}sq"] = rj, rj2, pj, pj2
        rsq += rj2
        psq += pj2
    avg["rsq"], avg["psq"] = rsq, psq
    avg["U"] = n * m["w"]
    avg["Usq"] = n * (m["wsq"] - m["w"] ** 2) + avg["U"] ** 2
    # one-chain variances of the instantaneous observables, handy for z-tests
    var = {"r3": n * b * b * (m["n2sq"] - m["n2"] ** 2), "p3": n * (m["m2sq"] - m["m2"] ** 2),
           "U": n * (m["wsq"] - m["w"] ** 2)}
    return avg, var


def main():
    out = {"_generator": "tests/golden/make_closed_form.py (scipy.integrate.dblquad; not reference output)",
           "cases": {}}
    for name, c in CASES.items():
        avg, var = chain_averages(c)
        out["cases"][name] = {"params": c, "avg": avg, "var1": var}
        print(name, {k: round(v, 6) for k, v in avg.items() if k in ("r3", "r1sq", "r3sq", "p3", "U", "Usq")})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ni_closed_form.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
