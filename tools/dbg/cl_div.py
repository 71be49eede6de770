import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from helpers import both
from oracle import binding as ob
import polymer_stats_amd as ps
kw=dict(n=17, E0=1.0, K1=0.3, K2=0.02, Fz=0.3, Fx=0.25, kT=0.8, b=1.2, energy_type=2, seed=22, bend_mod=0.2, bend_angle=0.0, cluster_prob=0.3, steps_per_adjust=400)
op, pp = both(2000, num_chains=64, precision=ps.F64, **kw)
pp.move_set=1
np.set_printoptions(precision=6, linewidth=200)
with ps.Ensemble(pp) as e:
    g0=e.chain_state(49)
    print("gpu U0", e.microstate(49)[6])
    for k in range(1,11):
        e.advance(1)
        op.num_steps=k
        o=ob.run(op, chain_id=49, mode="cluster", trace=True)
        g=e.chain_state(49)
        ch=np.nonzero(g["theta"]!=g0["theta"])[0]
        print(k, "gpuU", e.microstate(49)[6], "orU", o.U, "gpu changed", ch, "eq", np.array_equal(g["theta"],o.final_theta), "rng eq", np.array_equal(g["rng"],o.rng))
        if not np.array_equal(g["theta"],o.final_theta):
            print(" gpu th", g["theta"][:4], "ph", g["phi"][:4]); print(" or  th", o.final_theta[:4], "ph", o.final_phi[:4])
            print(" prev th", g0["theta"][:4], "ph", g0["phi"][:4])
            break
        g0=g
