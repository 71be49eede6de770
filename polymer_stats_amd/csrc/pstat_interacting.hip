// pstat_interacting.hip -- gfx950 kernel for --energy-type interacting (inc/energy.jl:11-16,
// U_interaction inc/eap_chain.jl:196-211): every trial move needs the O(n^2) dipole-dipole sum.
//
// Mapping: ONE CHAIN PER WAVEFRONT, lane l owns M = 1, 2, 4 or 8 consecutive monomers (n <= 64 M = 512; the
// reference's own interacting sweeps use n = 100 and 200, run/interacting_dielectric_study.jl:26).  Per lane in registers: the
// angles, n-hat, dipole and position of its monomer.  Everything that is one-per-chain (generator,
// proposal, r, p, U, running sums) is wave-uniform: the generator lives in SGPRs and runs on the
// scalar ALU beside the vector work.
//   * trial positions: a move of monomer idx shifts x_idx by b/2 dn and every x_j, j > idx, by b dn
//     (x_i = b (sum_{k<=i} n_k - n_i/2), inc/eap_chain.jl:49-51): three FMAs per lane, no scan;
//   * the n(n-1)/2 pair terms (ring_pair_sum, pstat_wave.h): the trial (x, mu) of every monomer is staged
//     once per step in an LDS ring of the L = ceil(n/M) lanes that carry the chain, so "lane i meets
//     lane i-k" is one ds_read_b128 + one ds_read_b64 per partner monomer at an offset from the lane's own
//     slot and no VALU work for data movement (a DPP rotation costs 6 quarter-rate moves per partner,
//     measured in tools/ubench).  Rotations k = 1..L/2-1 meet every pair of lanes once; rotation L/2 meets
//     its pairs from both ends and is weighted 1/2: 2016 pairs in 32 rotations at n = 64, 4950 pairs in
//     25 rotations of 4 terms at n = 100.  Unused slots carry zero dipoles at distinct far-away positions,
//     so they contribute exactly 0 without a per-pair select;
//   * one butterfly reduction gives the new pair energy to all lanes.
// Full recomputation per step is the reference's own cost model (it recomputes U from scratch,
// inc/eap_chain.jl:254); an exact incremental form would still touch ~n^2/6 pairs twice.
// One wave per workgroup: no barrier, no atomics; HBM only at launch start/end.
#include "pstat_device.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>

#include "../../include/pstat.h"
#include "pstat_math.h"
#include "pstat_wave.h"

namespace pstat {

// M consecutive monomers per lane: lane l owns monomers l*M .. l*M + M-1 (n <= 64 M).
// (f64, M = 1 sits at the 256-VGPR boundary: ask for two waves per SIMD so that it stays on the good side)
// UMB: --umbrella-sampling (its weights and the rising gauge, pstat_math.h) is compiled into its own instantiation: as a run-time
// option its few extra live values cost the plain kernel 3 % at n = 64 (three waves per SIMD, a register budget with no slack)
template <typename R, typename G, int CT, int M, bool UMB>
// f32 waves per SIMD asked of the register allocator for M = 1, 2, 4 (measured, 16 384 chains: M = 1 at
// 4 instead of the 3 it takes by itself +2 % at n = 64, +10 % at n = 23; M = 4 at 2 instead of 1 +16 % at
// n = 200; M = 2 at 3 instead of 2 loses 4 % at n = 100)
#ifndef PSTAT_IOCC_M1
#define PSTAT_IOCC_M1 4
#endif
#ifndef PSTAT_IOCC_M2
#define PSTAT_IOCC_M2 2
#endif
#ifndef PSTAT_IOCC_M4
#define PSTAT_IOCC_M4 2
#endif
#ifndef PSTAT_IOCC_F64M2
#define PSTAT_IOCC_F64M2 2
#endif
#ifndef PSTAT_IOCC_F64M4
#define PSTAT_IOCC_F64M4 1   // (measured round 2 with the rsq-Newton term: F64M2 / F64M4 = 2/1 1.34e8 | 3.38e7 at n = 100 | 200;
                             //  2/2 1.34e8 | 3.23e7; 3/1 1.20e8 | 3.38e7: the round-1 choice stands)
#endif
#ifndef PSTAT_IOCC_M8
#define PSTAT_IOCC_M8 1      // 257 <= n <= 512 (eight monomers per lane: f64 spills a few dozen registers there, like the
#endif                       //  clustering main's all-pairs kernel at that size)
#ifndef PSTAT_IOCC_F64M1
#define PSTAT_IOCC_F64M1 3   // (measured round 2, n = 64: 2 waves 3.47e8, 3 waves see DESIGN, 4 waves 2.68e8 -- spills)
#endif
__global__ __launch_bounds__(64, M == 8 ? PSTAT_IOCC_M8 :
                                 sizeof(R) == 8 ? (M == 1 ? PSTAT_IOCC_F64M1 : (M == 2 ? PSTAT_IOCC_F64M2 : PSTAT_IOCC_F64M4))
                                                : (M == 1 ? PSTAT_IOCC_M1 : (M == 2 ? PSTAT_IOCC_M2 : PSTAT_IOCC_M4))) void interacting_kernel(SweepArgs A, DevState S,
                                                         const CaseConst *__restrict__ cases,
                                                         int do_flips, int use_lag,
                                                         int reinit_mode /* 0 | 1 metropolis | 2 forced */) {
  constexpr bool umb = UMB;
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
  const R nbeta_log2e = (R)(-1.4426950408889634 / cc.kT);
  (void)kT; (void)nbeta_log2e;

  // ---- fill: my monomers' angles; chain-level scalars are wave-uniform
  const R *gth = (const R *)S.ang, *gph = (const R *)S.ang + (int64_t)n * C;
  R th[M], ph[M];
  bool real[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int i = lane * M + j;
    real[j] = i < n;
    th[j] = real[j] ? gth[(int64_t)i * C + c] : (R)0;
    ph[j] = real[j] ? gph[(int64_t)i * C + c] : (R)0;
  }
  G g;   // wave-uniform: one stream per chain, kept in SGPRs
  {
    uint32_t w[4];
    for (int q = 0; q < 4; ++q) w[q] = __builtin_amdgcn_readfirstlane(S.rng[q * C + c]);
    g.load(w, 1);
  }
  double phistep_d = S.stepsz[0 * C + c], thstep_d = S.stepsz[1 * C + c];
  R phistep = (R)(phistep_d / AG::unit), thstep = (R)(thstep_d / AG::unit);
  int64_t nacc_off = S.win[0 * C + c], natt_off = S.win[1 * C + c];
  int nacc_seg = 0, steps_seg = 0;
  int nnan_seg = 0;   // proposals whose trial energy was NaN or +-Inf (1/r^3 at r -> 0), rejected as in the reference
  R lag = use_lag ? (R)S.lag[c] : (R)0;
  // umbrella sampling (inc/average.jl:104-124), as in the sweep kernel: only w - w(first config) matters
  const R wscale = umb ? (R)((0.2 + 0.8 * exp(-(cc.Fx * cc.Fx + cc.Fz * cc.Fz) / cc.kT)) / cc.kT) : (R)0;
  R uref = umb ? (R)S.uref[c] : (R)0;
  bool regauged = false;
  double wnorm = umb ? S.wnorm[c] : 0.0;
  // the f64 running sums stay in HBM: lane 0 adds a block of steps to them every FLUSH steps (28 VGPRs
  // that the pair loop can use instead)

  // ---- derive my monomers and the chain totals (inc/eap_chain.jl:109-134)
  R st[M], nx[M], ny[M], nz[M], mx[M], my[M], mz[M], xx[M], xy[M], xz[M];
  R rx, ry, rz, px, py, pz, usum, upair, U;
  auto derive = [&]() {
    R tnx = 0, tny = 0, tnz = 0, tmx = 0, tmy = 0, tmz = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      R ct, sp, cp;
      AG::sc_theta(th[j], &st[j], &ct);
      AG::sc_phi(ph[j], &sp, &cp);
      nx[j] = real[j] ? cp * st[j] : (R)0; ny[j] = real[j] ? sp * st[j] : (R)0; nz[j] = real[j] ? ct : (R)0;
      dipole<R, CT>(a_or_mu, k2e, nx[j], ny[j], nz[j], mx[j], my[j], mz[j]);
      if (!real[j]) { mx[j] = 0; my[j] = 0; mz[j] = 0; }
      tnx += nx[j]; tny += ny[j]; tnz += nz[j];
      tmx += mx[j]; tmy += my[j]; tmz += mz[j];
    }
    // x_i = b (sum_{k<=i} n_k - n_i / 2): lane-exclusive prefix + running sum inside the lane
    R cx = wave_excl_scan<R>(tnx, lane), cy = wave_excl_scan<R>(tny, lane), cz = wave_excl_scan<R>(tnz, lane);
#pragma unroll
    for (int j = 0; j < M; ++j) {
      cx += nx[j]; cy += ny[j]; cz += nz[j];
      xx[j] = b * (cx - (R)0.5 * nx[j]); xy[j] = b * (cy - (R)0.5 * ny[j]); xz[j] = b * (cz - (R)0.5 * nz[j]);
    }
    rx = b * wave_allsum<R>(tnx); ry = b * wave_allsum<R>(tny); rz = b * wave_allsum<R>(tnz);
    px = wave_allsum<R>(tmx); py = wave_allsum<R>(tmy); pz = wave_allsum<R>(tmz);
    usum = wave_allsum<R>(mhalfE0 * tmz);
  };

  // sum over all pairs of a configuration (tx, tm) held M monomers per lane: pstat_wave.h
  auto pair_sum = [&](const R (&tx)[M], const R (&ty)[M], const R (&tz)[M], const R (&tmx)[M],
                      const R (&tmy)[M], const R (&tmz)[M]) -> R {
    if constexpr (sizeof(R) == 4 && M >= 2) return ring_pair_sum_pk<M, false>(ringA, ringB, lane, n, 0.0f, tx, ty, tz, tmx, tmy, tmz);
    else return ring_pair_sum<R, M, false>(ringA, ringB, lane, n, (R)0, tx, ty, tz, tmx, tmy, tmz);
  };
  // value of per-monomer array `a` at monomer idx (wave-uniform result)
  auto at_idx = [&](const R (&a)[M], int owner, int slot) -> R {
    R v = 0;
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (slot == j) v = lane_value<R>(a[j], owner);
    return v;
  };

  derive();
  upair = pair_sum(xx, xy, xz, mx, my, mz);
  U = usum + upair - (rx * Fx + rz * Fz);

  if (reinit_mode) {
    // mcmc_eap_chain.jl:352-361: draw a fresh configuration (all phi, then all theta, inc/eap_chain.jl:
    // 61-62); adopt it if forced or by metropolis_acc (inc/acceptance.jl:1-3).  The acceptor's cached
    // log-density is NOT refreshed by the reference, hence the `lag` offset (see reinit_kernel).
    R oth[M], oph[M];
#pragma unroll
    for (int j = 0; j < M; ++j) { oth[j] = th[j]; oph[j] = ph[j]; }
    const double U_old = (double)U, usum_old = (double)usum;
    double lsin = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) lsin += real[j] ? log((double)st[j]) : 0.0;
    const double logp_old = wave_allsum<double>(lsin);
    for (int i = 0; i < n; ++i) {
      const R v = store_phi<R>(u01<double>(g.next()));
#pragma unroll
      for (int j = 0; j < M; ++j) if (lane * M + j == i) ph[j] = v;
    }
    for (int i = 0; i < n; ++i) {
      const R v = store_theta<R>(u01<double>(g.next()));
#pragma unroll
      for (int j = 0; j < M; ++j) if (lane * M + j == i) th[j] = v;
    }
    derive();
    upair = pair_sum(xx, xy, xz, mx, my, mz);
    U = usum + upair - (rx * Fx + rz * Fz);
    lsin = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) lsin += real[j] ? log((double)st[j]) : 0.0;
    const double logp_new = wave_allsum<double>(lsin);
    bool adopt = reinit_mode == 2;
    if (!adopt) {
      const double eps = u01<double>(g.next());
      adopt = eps <= exp(-((double)U - U_old) / cc.kT + (logp_new - logp_old));
    }
    adopt = __builtin_amdgcn_readfirstlane(adopt ? 1 : 0) != 0;
    if (adopt) {
      const double ws = (double)wscale;
      const double lp_old = -U_old / cc.kT + logp_old + usum_old * ws;
      const double lp_new = -(double)U / cc.kT + logp_new + (double)usum * ws;
      lag = (R)((lp_old + (double)lag) - lp_new);
    } else {
#pragma unroll
      for (int j = 0; j < M; ++j) { th[j] = oth[j]; ph[j] = oph[j]; }
      derive();
      upair = pair_sum(xx, xy, xz, mx, my, mz);
      U = usum + upair - (rx * Fx + rz * Fz);
    }
  }

  int64_t step = A.step0;
  int64_t remaining = A.nsteps;
  const int64_t spa = A.steps_per_adjust;
  int64_t to_adj = A.adaptive ? spa - (step % spa) : 0;
  constexpr int FLUSH = 128;

  while (remaining > 0) {
    int64_t chunk = remaining < FLUSH ? remaining : FLUSH;
    if (A.adaptive && to_adj < chunk) chunk = to_adj;
    R acc1[7], acc2[7], accw = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) { acc1[q] = 0; acc2[q] = 0; }

    for (int k = 0; k < (int)chunk; ++k) {
      // ---- proposal (wave-uniform), mcmc_eap_chain.jl:277-280
      const uint32_t w0 = g.next();
      const int idx = (int)__umulhi(w0, (uint32_t)n);
      const int owner = idx / M, slot = idx % M;
      // (the trajectory itself: each product rounded before its sum, as the oracle and Julia round them -- through an
      // opaque register, so that no build flag can fuse them; cf. run_segment)
      auto rounded = [](R v) __attribute__((always_inline)) -> R { asm volatile("" : "+v"(v)); return v; };
      const uint32_t wphi = g.next();
      const R dphi = rounded(phistep * sym11<R>(wphi));
      const R th0 = at_idx(th, owner, slot), ph0 = at_idx(ph, owner, slot);
      R flip = 0;
      if (do_flips && (g.next() >> 31)) flip = AG::theta_max - rounded(2 * th0);
      const uint32_t wth = g.next();
      const R dth = flip + rounded(thstep * sym11<R>(wth));
      const uint32_t weps = g.next();
      const R eps = u01<R>(weps);
      (void)eps;
      const R st0 = at_idx(st, owner, slot);
      const R n0x = at_idx(nx, owner, slot), n0y = at_idx(ny, owner, slot), n0z = at_idx(nz, owner, slot);
      const R m0x = at_idx(mx, owner, slot), m0y = at_idx(my, owner, slot), m0z = at_idx(mz, owner, slot);

      // ---- move!, inc/eap_chain.jl:232-253
      const R ph1 = AG::wrap(ph0 + dphi);
      const R th1 = fmin(AG::theta_max, fmax((R)0, th0 + dth));
      R st1, ct1, sp1, cp1;
      AG::sc_theta(th1, &st1, &ct1);
      AG::sc_phi(ph1, &sp1, &cp1);
      const R n1x = cp1 * st1, n1y = sp1 * st1, n1z = ct1;
      R m1x, m1y, m1z;
      dipole<R, CT>(a_or_mu, k2e, n1x, n1y, n1z, m1x, m1y, m1z);
      const R dnx = n1x - n0x, dny = n1y - n0y, dnz = n1z - n0z;
      R tx[M], ty[M], tz[M], tmx[M], tmy[M], tmz[M];
      bool mine[M];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const int i = lane * M + j;
        mine[j] = i == idx;
        const R w = i > idx ? b : (mine[j] ? (R)0.5 * b : (R)0);   // shift of x_i in units of dn
        tx[j] = fma_r(w, dnx, xx[j]); ty[j] = fma_r(w, dny, xy[j]); tz[j] = fma_r(w, dnz, xz[j]);
        tmx[j] = mine[j] ? m1x : mx[j]; tmy[j] = mine[j] ? m1y : my[j]; tmz[j] = mine[j] ? m1z : mz[j];
      }

      // ---- energy, inc/energy.jl:13-16: sum(us) + U_interaction - F.r
      const R upair1 = pair_sum(tx, ty, tz, tmx, tmy, tmz);
      const R du = mhalfE0 * (m1z - m0z);
      const R drx = b * dnx, dry = b * dny, drz = b * dnz;
      const R dpair = upair1 - upair;
      const R dU = du + dpair - (Fx * drx + Fz * drz);

      // ---- Metropolis, inc/acceptance.jl:18-39 (1/r^3 singularities give NaN => rejected)
      bool ok;
      const R dw = du * wscale;   // change of the umbrella weight function (0 if off)
      if constexpr (sizeof(R) == 8) {
        ok = metropolis_f64(dU, kT, -1.0 / kT, st1, st0, dw - lag, A.wide_eps != 0, weps, w0, wphi, wth);
      } else {
        const R e = __builtin_amdgcn_exp2f((R)1.44269504f * (dw - lag) + dU * nbeta_log2e);
        ok = eps * st0 < st1 * e;
      }
      ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;   // one decision per chain
      nnan_seg += not_finite(dU) ? 1 : 0;
      if (ok) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          if (mine[j]) {
            th[j] = th1; ph[j] = ph1; st[j] = st1;
            nx[j] = n1x; ny[j] = n1y; nz[j] = n1z; mx[j] = m1x; my[j] = m1y; mz[j] = m1z;
          }
          xx[j] = tx[j]; xy[j] = ty[j]; xz[j] = tz[j];
        }
        rx += drx; ry += dry; rz += drz;
        px += m1x - m0x; py += m1y - m0y; pz += m1z - m0z;
        usum += du; upair = upair1; U += dU;
        lag = 0;
        ++nacc_seg;
      }
      // ---- record! x 8, mcmc_eap_chain.jl:327-328 (UmbrellaAverager: value += v / e^w, inc/average.jl:63-67)
      const R obs[7] = {rx, ry, rz, px, py, pz, U};
      if constexpr (UMB) {
        bool raise;
        R wrel = umbrella_logw(usum, uref, wscale, raise);
        if (raise) {   // (wave-uniform) the gauge rises to this configuration (pstat_math.h)
          const double f = exp_f64(-(double)wrel);
          const R fr = (R)f;
#pragma unroll
          for (int q = 0; q < 7; ++q) { acc1[q] *= fr; acc2[q] *= fr; }
          accw *= fr;
          if (lane == 0)
            for (int q = 0; q < NSUMS_BASE; ++q) S.sums[(int64_t)q * C + c] *= f;
          wnorm *= f;
          uref = usum; regauged = true; wrel = 0;
        }
        const R wgt = exp_r(wrel);
        accw += wgt;
#pragma unroll
        for (int q = 0; q < 7; ++q) { acc1[q] = fma_r(wgt, obs[q], acc1[q]); acc2[q] = fma_r(wgt * obs[q], obs[q], acc2[q]); }
      } else {
#pragma unroll
        for (int q = 0; q < 7; ++q) { acc1[q] += obs[q]; acc2[q] = fma_r(obs[q], obs[q], acc2[q]); }
      }
    }

    if (lane == 0) {
      auto add = [&](const int q, const R v) { S.sums[q * C + c] += (double)v; };
      add(S_R1, acc1[0]); add(S_R2, acc1[1]); add(S_R3, acc1[2]);
      add(S_P1, acc1[3]); add(S_P2, acc1[4]); add(S_P3, acc1[5]);
      add(S_U, acc1[6]);
      add(S_R1SQ, acc2[0]); add(S_R2SQ, acc2[1]); add(S_R3SQ, acc2[2]);
      add(S_P1SQ, acc2[3]); add(S_P2SQ, acc2[4]); add(S_P3SQ, acc2[5]);
      add(S_USQ, acc2[6]);
    }
    wnorm += (double)accw;
    step += chunk;
    remaining -= chunk;
    steps_seg += (int)chunk;

    if constexpr (sizeof(R) == 4) {
      // f32 only: positions and totals are updated incrementally; re-derive them from the angles
      // once per block of steps so rounding cannot drift (the pair energy is always a fresh sum)
      const R keep_pair = upair;
      derive();
      upair = keep_pair;
      U = usum + upair - (rx * Fx + rz * Fz);
    }

    if (A.adaptive) {  // mcmc_eap_chain.jl:301-322
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
        wth[(int64_t)(lane * M + j) * C + c] = th[j];
        wph[(int64_t)(lane * M + j) * C + c] = ph[j];
      }
  }
  if (lane == 0) {
    g.store(S.rng + c, C);
    S.stepsz[0 * C + c] = phistep_d; S.stepsz[1 * C + c] = thstep_d;
    S.win[0 * C + c] = nacc_off + nacc_seg; S.win[1 * C + c] = natt_off + steps_seg;
    S.nacc_total[c] += nacc_seg;
    S.nanrej[c] += nnan_seg;
    S.obs[OBS_R1 * C + c] = rx; S.obs[OBS_R2 * C + c] = ry; S.obs[OBS_R3 * C + c] = rz;
    S.obs[OBS_P1 * C + c] = px; S.obs[OBS_P2 * C + c] = py; S.obs[OBS_P3 * C + c] = pz;
    S.obs[OBS_U * C + c] = U; S.obs[OBS_USUM * C + c] = usum;
    S.lag[c] = lag;
    if (umb) S.wnorm[c] = wnorm;
    if (regauged) S.uref[c] = (double)uref;
  }
}

using InterFn = void (*)(SweepArgs, DevState, const CaseConst *, int, int, int);

template <typename R, typename G, bool UMB>
static InterFn pick_interacting_mu(const LaunchCfg &cfg, int64_t n) {
  const bool diel = cfg.chain_type == PSTAT_DIELECTRIC;
  if (n <= 64) return diel ? interacting_kernel<R, G, PSTAT_DIELECTRIC, 1, UMB> : interacting_kernel<R, G, PSTAT_POLAR, 1, UMB>;
  if (n <= 128) return diel ? interacting_kernel<R, G, PSTAT_DIELECTRIC, 2, UMB> : interacting_kernel<R, G, PSTAT_POLAR, 2, UMB>;
  if (n <= 256) return diel ? interacting_kernel<R, G, PSTAT_DIELECTRIC, 4, UMB> : interacting_kernel<R, G, PSTAT_POLAR, 4, UMB>;
  return diel ? interacting_kernel<R, G, PSTAT_DIELECTRIC, 8, UMB> : interacting_kernel<R, G, PSTAT_POLAR, 8, UMB>;
}
template <typename R, bool UMB>
static InterFn pick_interacting_ru(const LaunchCfg &cfg, int64_t n) {
  return cfg.rng == PSTAT_RNG_XOSHIRO128PP ? pick_interacting_mu<R, Xoshiro128pp, UMB>(cfg, n)
                                           : pick_interacting_mu<R, Mwc64x, UMB>(cfg, n);
}

// Four objects are built from this file (csrc/Makefile, parallel build): -DPSTAT_IPART=1 holds the f32 instantiations
// (statistical parity only), =2 the f64 ones (bit parity with the oracle) and the launchers, =3 / =4 their umbrella-sampling
// twins; all with -ffp-contract=fast -- the f64 proposal arithmetic is fenced, see the note in the Makefile.  Without the
// macro: everything in one object.
#if !defined(PSTAT_IPART) || PSTAT_IPART == 1
InterFn pick_interacting_f32(const LaunchCfg &cfg, int64_t n) { return pick_interacting_ru<float, false>(cfg, n); }
#endif
#if !defined(PSTAT_IPART) || PSTAT_IPART == 3
InterFn pick_interacting_f32_umb(const LaunchCfg &cfg, int64_t n) { return pick_interacting_ru<float, true>(cfg, n); }
#endif
#if !defined(PSTAT_IPART) || PSTAT_IPART == 4
InterFn pick_interacting_f64_umb(const LaunchCfg &cfg, int64_t n) { return pick_interacting_ru<double, true>(cfg, n); }
#endif
#if !defined(PSTAT_IPART) || PSTAT_IPART == 2
InterFn pick_interacting_f32(const LaunchCfg &cfg, int64_t n);
InterFn pick_interacting_f32_umb(const LaunchCfg &cfg, int64_t n);
InterFn pick_interacting_f64_umb(const LaunchCfg &cfg, int64_t n);

static InterFn pick_interacting(const LaunchCfg &cfg, int64_t n) {
  if (cfg.precision != PSTAT_F64) return cfg.umbrella ? pick_interacting_f32_umb(cfg, n) : pick_interacting_f32(cfg, n);
  return cfg.umbrella ? pick_interacting_f64_umb(cfg, n) : pick_interacting_ru<double, false>(cfg, n);
}

hipError_t launch_interacting(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                              const CaseConst *cases, int reinit_mode, hipStream_t stream) {
  InterFn fn = pick_interacting(cfg, a.n);
  hipLaunchKernelGGL(fn, dim3((unsigned)s.C), dim3(64), 0, stream, a, s, cases, cfg.do_flips,
                     (cfg.lag || reinit_mode) ? 1 : 0, reinit_mode);
  return hipGetLastError();
}

hipError_t interacting_kernel_info(const LaunchCfg &cfg, int64_t n, int *blocks_per_cu, const char **name) {
  InterFn fn = pick_interacting(cfg, n);
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)fn, 64, 0);
  if (e != hipSuccess) return e;
  if (blocks_per_cu) *blocks_per_cu = nb;
  if (name) *name = cfg.precision == PSTAT_F64 ? "interacting_kernel<double>" : "interacting_kernel<float>";
  return hipSuccess;
}
#endif

}  // namespace pstat
