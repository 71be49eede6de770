import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import polymer_stats_amd as ps
from oracle import binding as ob
from helpers import both
kw = dict(n=23, E0=1.2, mu=0.8, Fz=0.3, Fx=0.2, chain_type=1, energy_type=1, do_flips=1, seed=14, steps_per_adjust=250)
for c in range(5):
    op, pp = both(1500, num_chains=1, precision=ps.F64, chain_id0=c, **kw)
    with ps.Ensemble(pp) as e:
        e.advance(1500)
        s = e.summary()
        red = e.reduce_host()
        o = ob.run(op, chain_id=c, mode="fast")
        print(c, "gpu nan", s.nan_rejects, "collapsed", s.chains_collapsed, "oracle", o.nan_rejects, "U", e.microstate(0)[6], o.U, "red tail", red[-3:])
kw = dict(n=12, E0=1.0, mu=2.0, Fz=0.2, chain_type=ps.POLAR, energy_type=ps.ISING, seed=8)
for prec in (ps.F32, ps.F64):
    with ps.Ensemble(ps.default_params(num_chains=512, precision=prec, **kw)) as e:
        e.advance(20000)
        s = e.summary()
        print("ising", prec, s.nan_rejects, s.chains_collapsed, s.avg[14])
