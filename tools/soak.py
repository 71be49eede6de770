#!/usr/bin/env python3
"""Robustness soak on the GPU box: (1) many create/advance/summary/destroy cycles -- no error, no device-memory
growth; (2) one very long launch (thousands of time segments per chain block through the job queue), checked
against the closed form; (3) the same for the cluster kernel with a long, low-temperature (whole-chain cluster)
case, which is the slowest job the predecessor wait has to sit through."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polymer_stats_amd as ps

hip = C.CDLL("libamdhip64.so")


def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value


t0 = time.time()
base = None
for k in range(300):
    prec = (ps.F32, ps.F64, ps.Q16)[k % 3]
    p = ps.default_params(n=10 + (k % 50), E0=1.0, Fz=0.5, num_chains=1000 + 37 * (k % 11), precision=prec, seed=k,
                          energy_type=(0, 2)[k % 2], move_set=(0, 1)[(k // 2) % 2] if prec != ps.Q16 or True else 0)
    with ps.Ensemble(p) as e:
        e.advance(1500)
        s = e.summary()
        assert np.isfinite(s.avg[2]) and s.steps_per_chain == 1500
    if k == 20:
        base = free_bytes()
leak = base - free_bytes()
print(f"(1) 300 create/advance/destroy cycles in {time.time() - t0:.1f} s; device memory drift since cycle 20: {leak} bytes", flush=True)
assert abs(leak) < 64 << 20

t0 = time.time()
p = ps.default_params(n=100, E0=1.0, K1=1.0, Fz=1.0, num_chains=65536, precision=ps.F32, seed=5)
with ps.Ensemble(p) as e:
    e.advance(200_000)
    e.reset_averages()
    e.advance(40_000_000)            # one call: 1221 segments per block
    e.sync()
    s = e.summary()
want = 35.106379
print(f"(2) 65536 chains x 4e7 steps in ONE advance: {time.time() - t0:.1f} s, <r_z> = {s.avg[2]:.6f} +- {s.stderr[2]:.1e} "
      f"(closed form {want}), z = {(s.avg[2] - want) / s.stderr[2]:.2f}", flush=True)
assert abs(s.avg[2] - want) < 5 * s.stderr[2] + 1e-4

t0 = time.time()
p = ps.default_params(n=200, E0=3.0, K1=1.0, kT=0.02, num_chains=4096, precision=ps.F32, seed=6, energy_type=ps.NONINTERACTING,
                      move_set=ps.MOVES_CLUSTER, cluster_prob=0.5)
with ps.Ensemble(p) as e:
    e.advance(300_000)
    e.sync()
    s = e.summary()
print(f"(3) cluster kernel, aligned chains (kT = 0.02, whole-chain clusters): 4096 x 3e5 steps in {time.time() - t0:.1f} s, "
      f"AR = {s.acceptance_ratio:.3f}, <cos^2> = {s.extra_avg[0]:.2f}", flush=True)
print("soak ok")
