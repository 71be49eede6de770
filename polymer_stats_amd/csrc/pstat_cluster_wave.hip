// pstat_cluster_wave.hip -- the step of mcmc_clustering_eap_chain.jl:268-311 for the energies that need
// every pair: --energy-type interacting (U_interaction, inc/eap_chain.jl:196-211) and cutoff (UCutoff,
// :165-192).
//
// Mapping is the interacting kernel's (pstat_interacting.hip): ONE CHAIN PER WAVEFRONT, lane l owns M = 1, 2,
// 4 or (f32 only) 8 consecutive monomers, chain-level quantities are wave-uniform, the n(n-1)/2 pair terms of a
// configuration are met through a 128-entry LDS ring at compile-time offsets.  One chain per wave also
// makes the cluster move cheap to express: growth is a wave-uniform loop (no lane divergence at all),
// members are a contiguous index range, and the proposal is evaluated the way the reference does it --
// the whole trial configuration is re-derived from its angles (trig, positions by a wave scan, bond
// angles, u_i + bending, r, p, log sin) and its pair energy summed afresh, so sin(theta) after a
// reflection is the recomputed one and the reference's corner cases (theta = fl(pi) -> 0) need no
// special handling.
//
// Reference behaviour kept: the acceptor caches log(pi) + log(alpha) (`lag`); with energy-type cutoff
// the energy functor returns the truncated pair sum ALONE (inc/eap_chain.jl:171-192: no sum(us), no
// -F.r), so neither the field nor the force enters U there.
#include <hip/hip_runtime.h>
#include <math.h>

#include <type_traits>

#include "../../include/pstat.h"
#include "pstat_device.h"
#include "pstat_math.h"
#include "pstat_wave.h"

namespace pstat {

namespace {

__device__ __forceinline__ float cw_acos(float x) { return acosf(x); }
__device__ __forceinline__ double cw_acos(double x) { return acos_r(x); }   // pstat_math.h: 1.2 ulp, ~35 instructions
template <typename R> __device__ __forceinline__ R cw_log(R x) { return log_r(x); }

template <typename R, int M>
struct Cfg {  // one configuration of the chain, M monomers per lane, and its chain totals
  R th[M], ph[M], st[M], nx[M], ny[M], nz[M], mx[M], my[M], mz[M], xx[M], xy[M], xz[M];
  R rx, ry, rz, px, py, pz, usum, psisum, c2sum, upair, U;
};

// Waves per SIMD asked of the register allocator.  Measured, 16 384 chains: f32 M = 1 at 4 instead of the 2 it
// takes by itself +41 % at n = 64 (3: +25 %; 5 spills: -10 %), M = 4 at 2 instead of 1 +24 % at n = 200, M = 2
// stays at 2 (3: -4 % at n = 100), M = 8 needs one wave's 512 registers; f64 M = 1, 2 at 2 instead of 1:
// +38 % at n = 64, +23 % at n = 100 (a few spilled registers outside the pair loop).
#ifndef PSTAT_WOCC_M1
#define PSTAT_WOCC_M1 4
#endif
#ifndef PSTAT_WOCC_M2
#define PSTAT_WOCC_M2 2
#endif
#ifndef PSTAT_WOCC_M4
#define PSTAT_WOCC_M4 2
#endif
template <typename R, typename G, int CT, int M>
#ifndef PSTAT_WOCC_F64M1
#define PSTAT_WOCC_F64M1 2
#endif
#ifndef PSTAT_WOCC_F64M2
#define PSTAT_WOCC_F64M2 2
#endif
__global__ __launch_bounds__(64, sizeof(R) == 8 ? (M == 1 ? PSTAT_WOCC_F64M1 : (M == 2 ? PSTAT_WOCC_F64M2 : 1)) : (M == 1 ? PSTAT_WOCC_M1 : (M == 2 ? PSTAT_WOCC_M2 : (M == 4 ? PSTAT_WOCC_M4 : 1))))
void cluster_wave_kernel(SweepArgs A, DevState S, const CaseConst *__restrict__ cases,
                                                          int umb, int cutoff) {
  using AG = Ang<R>;
  using R4 = typename std::conditional<sizeof(R) == 4, float4, double4>::type;
  using R2 = typename Vec2<R>::type;
  __shared__ R4 ringA[128 * M];   // (x, y, z, mu_x) of monomer e mod 64M at entry e
  __shared__ R2 ringB[128 * M * (sizeof(R) == 4 ? 2 : 1)];   // (mu_y, mu_z); f32: 16-byte entries (pstat_wave.h)
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  const int64_t C = S.C;
  const int n = (int)A.n;
  const CaseConst cc = cases[c / A.chains_per_case];
  const R Fz = (R)cc.Fz, Fx = (R)cc.Fx, b = (R)cc.b, kT = (R)cc.kT;
  const R a_or_mu = (CT == PSTAT_DIELECTRIC) ? (R)((cc.K1 - cc.K2) * cc.E0) : (R)cc.mu;
  const R k2e = (R)(cc.K2 * cc.E0);
  const R mhalfE0 = (R)(-0.5 * cc.E0);
  const R khalf = (R)(cc.kappa / 2), psi0 = (R)cc.psi0, cprob = (R)cc.cluster_prob;
  // UCutoff(cutoff-radius * mlen): pairs beyond it contribute nothing; +inf = plain U_interaction
  const R crad2 = cutoff ? (R)((cc.cutoff_radius * cc.b) * (cc.cutoff_radius * cc.b)) : (R)INFINITY;

  bool real[M];
#pragma unroll
  for (int j = 0; j < M; ++j) real[j] = lane * M + j < n;

  // ---- everything the reference caches in an EAPChain, from the angles (inc/eap_chain.jl:109-134)
  auto derive = [&](Cfg<R, M> &q) {
    R tnx = 0, tny = 0, tnz = 0, tmx = 0, tmy = 0, tmz = 0, tc2 = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      R ct, sp, cp;
      AG::sc_theta(q.th[j], &q.st[j], &ct);
      AG::sc_phi(q.ph[j], &sp, &cp);
      q.nx[j] = real[j] ? cp * q.st[j] : (R)0; q.ny[j] = real[j] ? sp * q.st[j] : (R)0; q.nz[j] = real[j] ? ct : (R)0;
      dipole<R, CT>(a_or_mu, k2e, q.nx[j], q.ny[j], q.nz[j], q.mx[j], q.my[j], q.mz[j]);
      if (!real[j]) { q.mx[j] = 0; q.my[j] = 0; q.mz[j] = 0; }
      tnx += q.nx[j]; tny += q.ny[j]; tnz += q.nz[j];
      tmx += q.mx[j]; tmy += q.my[j]; tmz += q.mz[j];
      tc2 += q.nz[j] * q.nz[j];
    }
    R cx = wave_excl_scan<R>(tnx, lane), cy = wave_excl_scan<R>(tny, lane), cz = wave_excl_scan<R>(tnz, lane);
#pragma unroll
    for (int j = 0; j < M; ++j) {
      cx += q.nx[j]; cy += q.ny[j]; cz += q.nz[j];
      q.xx[j] = b * (cx - (R)0.5 * q.nx[j]); q.xy[j] = b * (cy - (R)0.5 * q.ny[j]); q.xz[j] = b * (cz - (R)0.5 * q.nz[j]);
    }
    // bond angles psi_i = angle(n_i, n_{i+1}) and their bending energy (inc/eap_chain.jl:45-47,54-58);
    // the successor of my last monomer is the first monomer of the next lane
    const R fx = __shfl_down(q.nx[0], 1, 64), fy = __shfl_down(q.ny[0], 1, 64), fz = __shfl_down(q.nz[0], 1, 64);
    R tpsi = 0, tbend = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const R sx = j + 1 < M ? q.nx[j + 1 < M ? j + 1 : j] : fx;
      const R sy = j + 1 < M ? q.ny[j + 1 < M ? j + 1 : j] : fy;
      const R sz = j + 1 < M ? q.nz[j + 1 < M ? j + 1 : j] : fz;
      const R psi = cw_acos(fmin((R)1, fmax((R)-1, q.nx[j] * sx + q.ny[j] * sy + q.nz[j] * sz)));
      const bool bonded = lane * M + j + 1 < n;
      tpsi += bonded ? psi : (R)0;
      tbend += bonded ? khalf * (psi - psi0) * (psi - psi0) : (R)0;
    }
    q.rx = b * wave_allsum<R>(tnx); q.ry = b * wave_allsum<R>(tny); q.rz = b * wave_allsum<R>(tnz);
    q.px = wave_allsum<R>(tmx); q.py = wave_allsum<R>(tmy); q.pz = wave_allsum<R>(tmz);
    q.usum = wave_allsum<R>(mhalfE0 * tmz + tbend);
    q.psisum = wave_allsum<R>(tpsi);
    q.c2sum = wave_allsum<R>(tc2);
  };

  // sum over all pairs (within the cutoff) of configuration q: pstat_wave.h
  auto pair_sum = [&](const Cfg<R, M> &q) -> R {
    if constexpr (sizeof(R) == 4 && M >= 2) return ring_pair_sum_pk<M, true>(ringA, ringB, lane, n, crad2, q.xx, q.xy, q.xz, q.mx, q.my, q.mz);
    else return ring_pair_sum<R, M, true>(ringA, ringB, lane, n, crad2, q.xx, q.xy, q.xz, q.mx, q.my, q.mz);
  };
  // inc/energy.jl:13-16, or UCutoff's functor (pair sum only)
  auto total_U = [&](Cfg<R, M> &q) {
    q.upair = pair_sum(q);
    q.U = cutoff ? q.upair : q.usum + q.upair - (q.rx * Fx + q.rz * Fz);
  };
  // per-monomer array value at monomer i (wave-uniform result)
  auto at = [&](const R (&a)[M], int i) -> R {
    const int owner = i / M, slot = i % M;
    R v = 0;
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (slot == j) v = lane_value<R>(a[j], owner);
    return v;
  };

  // ---- fill
  Cfg<R, M> cur;
  {
    const R *gth = (const R *)S.ang, *gph = (const R *)S.ang + (int64_t)n * C;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int i = lane * M + j;
      cur.th[j] = real[j] ? gth[(int64_t)i * C + c] : (R)0;
      cur.ph[j] = real[j] ? gph[(int64_t)i * C + c] : (R)0;
    }
  }
  G g;
  {
    uint32_t w[4];
    for (int q = 0; q < 4; ++q) w[q] = __builtin_amdgcn_readfirstlane(S.rng[q * C + c]);
    g.load(w, 1);
  }
  double phistep_d = S.stepsz[0 * C + c], thstep_d = S.stepsz[1 * C + c];
  R phistep = (R)(phistep_d / AG::unit), thstep = (R)(thstep_d / AG::unit);
  int64_t nacc_off = S.win[0 * C + c], natt_off = S.win[1 * C + c];
  int nacc_seg = 0, steps_seg = 0;
  int nnan_seg = 0;   // proposals whose trial energy was NaN or +-Inf (1/r^3 at r -> 0)
  R lag = (R)S.lag[c];
  const R wscale = umb ? (R)((0.2 + 0.8 * exp(-(cc.Fx * cc.Fx + cc.Fz * cc.Fz) / cc.kT)) / cc.kT) : (R)0;
  R uref = umb ? (R)S.uref[c] : (R)0;
  bool regauged = false;
  double wnorm = umb ? S.wnorm[c] : 0.0;
  // the f64 running sums stay in HBM: lane 0 adds a block of steps to them every FLUSH steps
  const R inv_nm1 = n > 1 ? (R)(1.0 / (double)(n - 1)) : (R)0;

  derive(cur);
  total_U(cur);

  int64_t step = A.step0;
  int64_t remaining = A.nsteps;
  const int64_t spa = A.steps_per_adjust;
  int64_t to_adj = A.adaptive ? spa - (step % spa) : 0;
  constexpr int FLUSH = 128;

  while (remaining > 0) {
    int64_t chunk = remaining < FLUSH ? remaining : FLUSH;
    if (A.adaptive && to_adj < chunk) chunk = to_adj;
    R acc1[9], acc2[7], accw = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) acc1[q] = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) acc2[q] = 0;

    for (int k = 0; k < (int)chunk; ++k) {
      // ---- the single-monomer part of the proposal, mcmc_clustering_eap_chain.jl:269-272
      const uint32_t w0 = g.next();
      const int idx = (int)__umulhi(w0, (uint32_t)n);
      // (the trajectory itself: each product rounded before its sum -- through an opaque register, so that no build flag
      // can fuse them; cf. run_segment)
      auto rounded = [](R v) __attribute__((always_inline)) -> R { asm volatile("" : "+v"(v)); return v; };
      const uint32_t wphi = g.next(), wth = g.next();
      const R dphi = rounded(phistep * sym11<R>(wphi));
      const R dth = rounded(thstep * sym11<R>(wth));
      const R th0 = at(cur.th, idx), ph0 = at(cur.ph, idx);
      const R ph1 = AG::wrap(ph0 + dphi);
      const R th1 = fmin(AG::theta_max, fmax((R)0, th0 + dth));
      Cfg<R, M> tr;
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const bool mine = lane * M + j == idx;
        tr.th[j] = mine ? th1 : cur.th[j];
        tr.ph[j] = mine ? ph1 : cur.ph[j];
      }

      // ---- cluster_flip!(trial, idx), inc/eap_chain.jl:269-333 -- wave-uniform
      R alpha = 1;
      bool edge = false;
      if (!(u01<R>(g.next()) <= cprob)) {                                   // :276
        // theta clamped to exactly 0 and then reflected: the reference's Omega goes -inf then +inf = NaN
        edge = th1 == (R)0;
        R st1, ct1, sp1, cp1;
        AG::sc_theta(th1, &st1, &ct1);
        AG::sc_phi(ph1, &sp1, &cp1);
        const R n1x = cp1 * st1, n1y = sp1 * st1, n1z = ct1;
        int upper = idx, lower = idx;
        R upper_p = 0, lower_p = 0;
        R ux = n1x, uy = n1y, uz = n1z, lx = n1x, ly = n1y, lz = n1z;   // n-hat of the two end members
        bool gu = upper < n - 1, gl = lower > 0;
        while (gu || gl) {   // rounds: one link above, then one below (the stream contract, pstat_cluster.hip)
          if (gu) {
            const R bx = at(cur.nx, upper + 1), by = at(cur.ny, upper + 1), bz = at(cur.nz, upper + 1);
            upper_p = (1 + (ux * bx + uy * by + uz * bz)) / 2;
            if (u01<R>(g.next()) <= upper_p) {
              ++upper; ux = bx; uy = by; uz = bz;
              if (upper >= n - 1) { upper_p = 0; gu = false; }
            } else gu = false;
          }
          if (gl) {
            const R bx = at(cur.nx, lower - 1), by = at(cur.ny, lower - 1), bz = at(cur.nz, lower - 1);
            lower_p = (1 + (lx * bx + ly * by + lz * bz)) / 2;
            if (u01<R>(g.next()) <= lower_p) {
              --lower; lx = bx; ly = by; lz = bz;
              if (lower <= 0) { lower_p = 0; gl = false; }
            } else gl = false;
          }
        }
        // the boundary links after the reflection n -> (n_x, n_y, -n_z), :318-327
        R new_upper_p = 0, new_lower_p = 0;
        if (upper < n - 1) {
          const R bx = at(cur.nx, upper + 1), by = at(cur.ny, upper + 1), bz = at(cur.nz, upper + 1);
          new_upper_p = (1 + (ux * bx + uy * by + -uz * bz)) / 2;
        }
        if (lower > 0) {
          const R bx = at(cur.nx, lower - 1), by = at(cur.ny, lower - 1), bz = at(cur.nz, lower - 1);
          new_lower_p = (1 + (lx * bx + ly * by + -lz * bz)) / 2;
        }
        alpha = ((1 - new_upper_p) * (1 - new_lower_p)) / ((1 - upper_p) * (1 - lower_p));   // :328-329
        // refl_n! on every member, :263-265,314-316
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const int i = lane * M + j;
          if (i >= lower && i <= upper) {
            if constexpr (sizeof(R) == 8) tr.th[j] = fmin(AG::theta_max, fmax((R)0, tr.th[j] + (AG::theta_max - 2 * tr.th[j])));
            else tr.th[j] = AG::theta_max - tr.th[j];
          }
        }
      }
      const uint32_t weps = g.next();
      R eps;       // f64: 53 random bits by default (eps_uniform, pstat_math.h); f32: the 23 its mantissa holds
      if constexpr (sizeof(R) == 8) eps = eps_uniform(A.wide_eps != 0, weps, w0, wphi, wth);
      else eps = u01<R>(weps);

      // ---- the trial chain, re-derived (what move!/refl_n! do per touched monomer, :230-257) and its energy
      derive(tr);
      total_U(tr);
      R lr = 0;   // Omega(trial) - Omega(current) = sum over touched monomers of log(sin'/sin), :238,117
#pragma unroll
      for (int j = 0; j < M; ++j) lr += (real[j] && (tr.th[j] != cur.th[j])) ? cw_log(tr.st[j] / cur.st[j]) : (R)0;
      const R domega = wave_allsum<R>(lr);

      // ---- Metropolis-Hastings, inc/acceptance.jl:29-39 (NaN from 1/r^3 or log(0/0) => rejected)
      const R dw = (tr.usum - cur.usum) * wscale;
      const R dlt = -(tr.U - cur.U) / kT + domega + dw + cw_log(alpha) - lag;
      bool ok = ((dlt >= 0) || (eps < exp_r(dlt))) && !edge;
      ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
      nnan_seg += not_finite(tr.U) ? 1 : 0;
      if (ok) {
        cur = tr;
        lag = cw_log(alpha);
        ++nacc_seg;
      }

      // ---- record! x 10, mcmc_clustering_eap_chain.jl:243-244,310-311
      R wgt = 1;
      if (umb) {
        bool raise;
        R wrel = umbrella_logw(cur.usum, uref, wscale, raise);
        if (raise) {   // (wave-uniform) the gauge rises to this configuration (pstat_math.h)
          const double f = exp_f64(-(double)wrel);
          const R fr = (R)f;
#pragma unroll
          for (int q = 0; q < 9; ++q) acc1[q] *= fr;
#pragma unroll
          for (int q = 0; q < 7; ++q) acc2[q] *= fr;
          accw *= fr;
          if (lane == 0)
            for (int q = 0; q < NSUMS; ++q) S.sums[(int64_t)q * C + c] *= f;
          wnorm *= f;
          uref = cur.usum; regauged = true; wrel = 0;
        }
        wgt = exp_r(wrel);
      }
      const R obs[9] = {cur.rx, cur.ry, cur.rz, cur.px, cur.py, cur.pz, cur.U, cur.c2sum, cur.psisum * inv_nm1};
      accw += wgt;
#pragma unroll
      for (int q = 0; q < 9; ++q) acc1[q] = fma_r(wgt, obs[q], acc1[q]);
#pragma unroll
      for (int q = 0; q < 7; ++q) acc2[q] = fma_r(wgt * obs[q], obs[q], acc2[q]);
    }

    if (lane == 0) {
      auto add = [&](const int q, const R v) { S.sums[q * C + c] += (double)v; };
      add(S_R1, acc1[0]); add(S_R2, acc1[1]); add(S_R3, acc1[2]);
      add(S_P1, acc1[3]); add(S_P2, acc1[4]); add(S_P3, acc1[5]);
      add(S_U, acc1[6]); add(S_C2, acc1[7]); add(S_PSI, acc1[8]);
      add(S_R1SQ, acc2[0]); add(S_R2SQ, acc2[1]); add(S_R3SQ, acc2[2]);
      add(S_P1SQ, acc2[3]); add(S_P2SQ, acc2[4]); add(S_P3SQ, acc2[5]);
      add(S_USQ, acc2[6]);
    }
    wnorm += (double)accw;
    step += chunk;
    remaining -= chunk;
    steps_seg += (int)chunk;

    if (A.adaptive) {  // mcmc_clustering_eap_chain.jl:287-308
      to_adj -= chunk;
      if (to_adj == 0) {
        to_adj = spa;
        const int64_t nacc = nacc_off + nacc_seg, natt = natt_off + steps_seg;
        const double ratio = (double)nacc / (double)natt;
        if (ratio > A.adj_ub && phistep_d != K<double>::pi && thstep_d != K<double>::half_pi) {
          nacc_off = -nacc_seg; natt_off = -steps_seg;
          phistep_d = fmin(K<double>::pi, phistep_d * A.adj_scale);
          thstep_d = fmin(K<double>::half_pi, thstep_d * A.adj_scale);
        } else if (ratio < A.adj_lb) {
          nacc_off = -nacc_seg; natt_off = -steps_seg;
          phistep_d /= A.adj_scale;
          thstep_d /= A.adj_scale;
        }
        phistep = (R)(phistep_d / AG::unit); thstep = (R)(thstep_d / AG::unit);
      }
    }
  }

  // ---- spill
  {
    R *wth = (R *)S.ang, *wph = (R *)S.ang + (int64_t)n * C;
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (real[j]) {
        wth[(int64_t)(lane * M + j) * C + c] = cur.th[j];
        wph[(int64_t)(lane * M + j) * C + c] = cur.ph[j];
      }
  }
  if (lane == 0) {
    g.store(S.rng + c, C);
    S.stepsz[0 * C + c] = phistep_d; S.stepsz[1 * C + c] = thstep_d;
    S.win[0 * C + c] = nacc_off + nacc_seg; S.win[1 * C + c] = natt_off + steps_seg;
    S.nacc_total[c] += nacc_seg;
    S.nanrej[c] += nnan_seg;
    S.obs[OBS_R1 * C + c] = cur.rx; S.obs[OBS_R2 * C + c] = cur.ry; S.obs[OBS_R3 * C + c] = cur.rz;
    S.obs[OBS_P1 * C + c] = cur.px; S.obs[OBS_P2 * C + c] = cur.py; S.obs[OBS_P3 * C + c] = cur.pz;
    S.obs[OBS_U * C + c] = cur.U; S.obs[OBS_USUM * C + c] = cur.usum;
    S.obs[OBS_C2 * C + c] = cur.c2sum; S.obs[OBS_PSI * C + c] = cur.psisum;
    S.lag[c] = lag;
    if (umb) S.wnorm[c] = wnorm;
    if (regauged) S.uref[c] = (double)uref;
  }
}

using WaveFn = void (*)(SweepArgs, DevState, const CaseConst *, int, int);

template <typename R, typename G>
WaveFn pick_m(const LaunchCfg &cfg, int64_t n) {
  const bool diel = cfg.chain_type == PSTAT_DIELECTRIC;
  if (n <= 64) return diel ? cluster_wave_kernel<R, G, PSTAT_DIELECTRIC, 1> : cluster_wave_kernel<R, G, PSTAT_POLAR, 1>;
  if (n <= 128) return diel ? cluster_wave_kernel<R, G, PSTAT_DIELECTRIC, 2> : cluster_wave_kernel<R, G, PSTAT_POLAR, 2>;
  if (n <= 256) return diel ? cluster_wave_kernel<R, G, PSTAT_DIELECTRIC, 4> : cluster_wave_kernel<R, G, PSTAT_POLAR, 4>;
  // 8 monomers per lane (n <= 512; the reference's only cutoff-energy sweep, run/phases-big_2023-05-18.jl, uses n = 400)
  return diel ? cluster_wave_kernel<R, G, PSTAT_DIELECTRIC, 8> : cluster_wave_kernel<R, G, PSTAT_POLAR, 8>;
}

}  // namespace

// Two objects are built from this file (csrc/Makefile): -DPSTAT_WPART=1 holds the f32 instantiations
// (statistical parity only), -DPSTAT_WPART=2 the f64 ones (bit parity with the oracle) and the launchers; both with
// -ffp-contract=fast -- the f64 proposal arithmetic is fenced, see the note in the Makefile.  Without the macro: everything in one object.
#if !defined(PSTAT_WPART) || PSTAT_WPART == 1
WaveFn pick_wave_f32(const LaunchCfg &cfg, int64_t n) {
  return cfg.rng == PSTAT_RNG_XOSHIRO128PP ? pick_m<float, Xoshiro128pp>(cfg, n) : pick_m<float, Mwc64x>(cfg, n);
}
#endif
#if !defined(PSTAT_WPART) || PSTAT_WPART == 2
WaveFn pick_wave_f32(const LaunchCfg &cfg, int64_t n);

static WaveFn pick_wave(const LaunchCfg &cfg, int64_t n) {
  if (cfg.precision != PSTAT_F64) return pick_wave_f32(cfg, n);
  return cfg.rng == PSTAT_RNG_XOSHIRO128PP ? pick_m<double, Xoshiro128pp>(cfg, n) : pick_m<double, Mwc64x>(cfg, n);
}

hipError_t launch_cluster_wave(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s, const CaseConst *cases,
                               hipStream_t stream) {
  WaveFn fn = pick_wave(cfg, a.n);
  hipLaunchKernelGGL(fn, dim3((unsigned)s.C), dim3(64), 0, stream, a, s, cases, cfg.umbrella,
                     cfg.energy_type == PSTAT_CUTOFF ? 1 : 0);
  return hipGetLastError();
}

hipError_t cluster_wave_kernel_info(const LaunchCfg &cfg, int64_t n, int *blocks_per_cu, const char **name) {
  WaveFn fn = pick_wave(cfg, n);
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)fn, 64, 0);
  if (e != hipSuccess) return e;
  if (blocks_per_cu) *blocks_per_cu = nb;
  if (name) *name = cfg.precision == PSTAT_F64 ? "cluster_wave_kernel<double>" : "cluster_wave_kernel<float>";
  return hipSuccess;
}
#endif

}  // namespace pstat
