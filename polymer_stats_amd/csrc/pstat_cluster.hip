// pstat_cluster.hip -- the step of mcmc_clustering_eap_chain.jl:268-311 on gfx950.
//
// One step of that main is ONE proposal made of two parts applied to the trial chain: the
// single-monomer move of mcmc_eap_chain.jl (move!, inc/eap_chain.jl:232-258) and then cluster_flip!
// (inc/eap_chain.jl:269-333): with probability 1 - cluster_prob a cluster is grown from the moved
// monomer -- link (i, i+1) joins with probability (1 + n_i . n_{i+1}) / 2 -- and every member is
// reflected through the plane normal to the field (theta -> pi - theta).  The proposal is accepted by
// Metropolis-Hastings with the ratio alpha of the boundary probabilities (inc/acceptance.jl:29-39).
// The energy also carries the bending term kappa/2 (psi - psi0)^2 of every bond (eap_chain.jl:54-58)
// and two more observables are recorded: sum cos^2(theta) and the mean bond angle.
//
// Layout is the sweep kernel's: one chain per lane, angles in LDS as [monomer][lane], everything
// else in registers, the same persistent (block, segment) job queue.  What differs is the step:
//   * the n-hats of the two neighbours are always needed (bond angles), and the cluster growth walks
//     outwards from idx reading one LDS row per accepted link -- a per-lane loop, so the wave runs
//     as long as its longest cluster;
//   * a reflection leaves every interior bond (angle, Ising pair energy) unchanged, so the energy
//     difference of the whole proposal is the single move's O(1) difference plus two boundary bonds
//     plus the members' field terms, which are accumulated while the cluster grows;
//   * an accepted proposal rewrites theta of every member in LDS.
// Reference behaviour kept on purpose: the acceptor caches log(pi) + log(alpha) of the last accepted
// proposal (acceptance.jl:33-36), so later comparisons are offset by that log(alpha) -- `lag` below.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pstat_cluster_common.h"
#include "pstat_device.h"
#include "pstat_math.h"

namespace pstat {

namespace {

// ST = 0: the LDS cell is the (theta, phi) pair in R.  ST = 1 (PSTAT_Q16, R = float): one 32-bit word,
// theta lattice index in the low half and phi index in the high half (pstat_math.h); the reflection
// theta -> pi - theta is k -> 65535 - k, exact on the midpoint lattice.
template <typename R, typename G, int CT, int EN, int ST>
__device__ __forceinline__ void run_cluster_segment(const SweepArgs &A, const DevState &S, const CaseConst &cc,
                                                    const int umb_on, unsigned char *smem, const int lane,
                                                    const int64_t c, int64_t step, int64_t remaining) {
  using R2 = typename Vec2<R>::type;
  using AG = Ang<R>;
  using T3 = V3<R>;
  constexpr bool Q = ST == 1;
  static_assert(!Q || sizeof(R) == 4, "the lattice state runs on f32 arithmetic");
  using Cell = typename std::conditional<Q, uint32_t, R2>::type;
  Cell *ang = reinterpret_cast<Cell *>(smem);  // [n][lanes]
  auto dec = [](const Cell v) __attribute__((always_inline)) -> R2 {   // cell -> (theta, phi) in the unit of Ang<R>
    if constexpr (Q) { R2 a; a.x = q16_theta_turns(v & 0xFFFFu); a.y = q16_phi_turns(v >> 16); return a; }
    else return v;
  };
  auto refl_cell = [](const Cell v) __attribute__((always_inline)) -> Cell {   // refl_n! on a stored monomer
    if constexpr (Q) return (v & 0xFFFF0000u) | (65535u - (v & 0xFFFFu));
    else {
      Cell o = v;
      if constexpr (sizeof(R) == 8) o.x = fmin(AG::theta_max, fmax((R)0, v.x + (AG::theta_max - 2 * v.x)));
      else o.x = AG::theta_max - v.x;
      return o;
    }
  };
  const int lanes = A.lanes;
  const int64_t C = S.C;
  const int n = (int)A.n;

  const R Fz = (R)cc.Fz, Fx = (R)cc.Fx, b = (R)cc.b, kT = (R)cc.kT;
  const R a_or_mu = (CT == PSTAT_DIELECTRIC) ? (R)((cc.K1 - cc.K2) * cc.E0) : (R)cc.mu;
  const R k2e = (R)(cc.K2 * cc.E0);
  const R mhalfE0 = (R)(-0.5 * cc.E0);
  const R hb = (R)(-cc.b / 2);
  const R nbeta_log2e = (R)(-1.4426950408889634 / cc.kT);
  const R khalf = (R)(cc.kappa / 2), psi0 = (R)cc.psi0;
  const R cprob = (R)cc.cluster_prob;
  (void)hb; (void)nbeta_log2e; (void)kT;

  if constexpr (Q) {  // ---- fill
    const uint16_t *gth = (const uint16_t *)S.ang, *gph = (const uint16_t *)S.ang + (int64_t)n * C;
#pragma unroll 8
    for (int i = 0; i < n; ++i)
      ang[i * lanes + lane] = (uint32_t)gth[(int64_t)i * C + c] | ((uint32_t)gph[(int64_t)i * C + c] << 16);
  } else {
    const R *gth = (const R *)S.ang, *gph = (const R *)S.ang + (int64_t)n * C;
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
      R2 v;
      v.x = gth[(int64_t)i * C + c];
      v.y = gph[(int64_t)i * C + c];
      ang[i * lanes + lane] = v;
    }
  }
  // step sizes in the unit the proposal is added in: radians (f64), turns (f32), lattice cells (q16)
  constexpr double th_unit = Q ? 3.14159265358979323846 / 65536.0 : AG::unit;
  constexpr double ph_unit = Q ? 6.28318530717958647692 / 65536.0 : AG::unit;
  G g;
  g.load(S.rng + c, C);
  double phistep_d = S.stepsz[0 * C + c], thstep_d = S.stepsz[1 * C + c];
  R phistep = (R)(phistep_d / ph_unit), thstep = (R)(thstep_d / th_unit);
  int64_t nacc_off = S.win[0 * C + c], natt_off = S.win[1 * C + c];
  int nacc_seg = 0, steps_seg = 0;
  int nnan_seg = 0;   // proposals with a non-finite energy difference (Ising pair terms at r -> 0)
  R Orx = (R)S.obs[OBS_R1 * C + c], Ory = (R)S.obs[OBS_R2 * C + c], Orz = (R)S.obs[OBS_R3 * C + c];
  R Opx = (R)S.obs[OBS_P1 * C + c], Opy = (R)S.obs[OBS_P2 * C + c], Opz = (R)S.obs[OBS_P3 * C + c];
  R OU = (R)S.obs[OBS_U * C + c];
  R usum = (R)S.obs[OBS_USUM * C + c];      // sum of u_i INCLUDING the bending terms (eap_chain.jl:53-58)
  R c2sum = (R)S.obs[OBS_C2 * C + c], psisum = (R)S.obs[OBS_PSI * C + c];
  R lag = (R)S.lag[c];                      // log(alpha) of the last accepted proposal of this mcmc() call
  const bool umb = umb_on != 0;
  const R wscale = umb ? (R)((0.2 + 0.8 * exp(-(cc.Fx * cc.Fx + cc.Fz * cc.Fz) / cc.kT)) / cc.kT) : (R)0;
  R uref = umb ? (R)S.uref[c] : (R)0;
  bool regauged = false;
  double wnorm = umb ? S.wnorm[c] : 0.0;
  double sums[NSUMS];
#pragma unroll
  for (int q = 0; q < NSUMS; ++q) sums[q] = S.sums[q * C + c];
  const R inv_nm1 = n > 1 ? (R)(1.0 / (double)(n - 1)) : (R)0;

  const int64_t spa = A.steps_per_adjust;
  int64_t to_adj = A.adaptive ? spa - (step % spa) : 0;
  constexpr int FLUSH = 128;
  int left = (int)remaining;

  // n-hat and dipole of monomer i from its stored angles
  // (returns whether theta sits exactly on a clamp value, see `edge` below)
  auto nm_of = [&](const R2 a, T3 &nh, T3 &mu) __attribute__((always_inline)) -> bool {
    R s, co, sp, cp;
    AG::sc_theta(a.x, &s, &co);
    AG::sc_phi(a.y, &sp, &cp);
    nh.x = cp * s; nh.y = sp * s; nh.z = co;
    dipole<R, CT>(a_or_mu, k2e, nh.x, nh.y, nh.z, mu.x, mu.y, mu.z);
    return a.x == (R)0 || a.x == AG::theta_max;
  };
  auto nhat_of = [&](const R2 a, T3 &nh) __attribute__((always_inline)) -> bool {
    R s, co, sp, cp;
    AG::sc_theta(a.x, &s, &co);
    AG::sc_phi(a.y, &sp, &cp);
    nh.x = cp * s; nh.y = sp * s; nh.z = co;
    return a.x == (R)0 || a.x == AG::theta_max;
  };
  // a monomer joins the cluster: its n_z and the components of its dipole that the reflection flips
  // (dielectric: mu_x, mu_y = (K1-K2) E0 n_z (n_x, n_y); polar: mu_z = mu n_z) enter the member sums
  auto member = [&](const bool acc, const T3 &nh, R &snz, T3 &sm) __attribute__((always_inline)) {
    const R z = acc ? nh.z : (R)0;
    snz += z;
    const R q = a_or_mu * z;
    if constexpr (CT == PSTAT_DIELECTRIC) { sm.x += q * nh.x; sm.y += q * nh.y; }
    else sm.z += q;
  };
  auto load_nm = [&](const int i, T3 &nh, T3 &mu) __attribute__((always_inline)) -> bool {
    return nm_of(dec(ang[i * lanes + lane]), nh, mu);
  };
  // reflection through the plane normal to the field: refl_n!, inc/eap_chain.jl:263-265
  auto refl_theta = [&](const R th) __attribute__((always_inline)) -> R {
    if constexpr (sizeof(R) == 8) return fmin(AG::theta_max, fmax((R)0, th + (AG::theta_max - 2 * th)));
    else return AG::theta_max - th;
  };
  auto refl_n = [](const T3 &v) __attribute__((always_inline)) -> T3 { return T3{v.x, v.y, -v.z}; };
  auto refl_mu = [](const T3 &m) __attribute__((always_inline)) -> T3 {
    if constexpr (CT == PSTAT_DIELECTRIC) return T3{-m.x, -m.y, m.z};   // a nz (nx, ny, nz) + k2e z
    else return T3{m.x, m.y, -m.z};
  };
  // what bond (a, b) contributes: its angle, bending energy and (Ising) dipole-dipole energy
  auto bond = [&](const T3 &na, const T3 &ma, const T3 &nb, const T3 &mb, R &psi, R &ebend, R &epair)
      __attribute__((always_inline)) {
    psi = bond_angle<R>(na, nb);
    ebend = khalf * (psi - psi0) * (psi - psi0);
    if constexpr (EN == PSTAT_ISING)
      epair = pair_term_fast(hb * (na.x + nb.x), hb * (na.y + nb.y), hb * (na.z + nb.z),
                             ma.x, ma.y, ma.z, mb.x, mb.y, mb.z);
    else epair = 0;
  };
  // f32 keeps r, p, U, sum(u), sum(psi), sum cos^2 as running totals of accepted differences, whose rounding
  // errors random-walk; at every segment start (<= 32 768 steps apart in f32) they are re-derived from the
  // angles just filled into LDS (cf. the sweep kernel)
  auto refresh_totals = [&]() {
    double tx = 0, ty = 0, tz = 0, qx = 0, qy = 0, qz = 0, tu = 0, tp = 0, tpsi = 0, tc2 = 0;
    T3 pn{0, 0, 1}, pm{0, 0, 0};
    for (int i = 0; i < n; ++i) {
      T3 ni, mi;
      load_nm(i, ni, mi);
      tx += (double)ni.x; ty += (double)ni.y; tz += (double)ni.z;
      qx += (double)mi.x; qy += (double)mi.y; qz += (double)mi.z;
      tu += (double)(mhalfE0 * mi.z);
      tc2 += (double)(ni.z * ni.z);
      if (i > 0) {
        R psi, eb, ep;
        bond(pn, pm, ni, mi, psi, eb, ep);
        tpsi += (double)psi; tu += (double)eb; tp += (double)ep;
      }
      pn = ni; pm = mi;
    }
    const double bd = (double)b;
    Orx = (R)(bd * tx); Ory = (R)(bd * ty); Orz = (R)(bd * tz);
    Opx = (R)qx; Opy = (R)qy; Opz = (R)qz;
    usum = (R)tu; psisum = (R)tpsi; c2sum = (R)tc2;
    OU = (R)(tu + tp - ((double)Fx * bd * tx + (double)Fz * bd * tz));
  };

  if constexpr (sizeof(R) == 4) refresh_totals();

  while (left > 0) {
    int chunk = left < FLUSH ? left : FLUSH;
    if (A.adaptive && to_adj < chunk) chunk = (int)to_adj;
    R a1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, a2[7] = {0, 0, 0, 0, 0, 0, 0};
    R accw = 0;

    for (int s = 0; s < chunk; ++s) {
      // ---- the single-monomer part, mcmc_clustering_eap_chain.jl:269-272
      const uint32_t w0 = g.next();
      const int idx = (int)__umulhi(w0, (uint32_t)n);
      const uint32_t wphi = g.next(), wth = g.next();
      const int cell = idx * lanes + lane;
      const Cell c0 = ang[cell];
      const R2 a0 = dec(c0);
      const R th0 = a0.x, ph0 = a0.y;
      R th1, ph1;
      bool inside = true;       // q16: the trial theta stayed on the lattice (else: the reference's clamp => rejected)
      uint32_t k1c = 0, j1 = 0;
      if constexpr (Q) {
        const int k1 = (int)(c0 & 0xFFFFu) + q16_disp(thstep, sym11<R>(wth));
        j1 = ((c0 >> 16) + (uint32_t)q16_disp(phistep, sym11<R>(wphi))) & 0xFFFFu;
        inside = (uint32_t)k1 < 65536u;
        k1c = (uint32_t)min(max(k1, 0), 65535);
        th1 = q16_theta_turns(k1c);
        ph1 = q16_phi_turns(j1);
      } else {
        ph1 = AG::wrap(ph0 + phistep * sym11<R>(wphi));
        th1 = fmin(AG::theta_max, fmax((R)0, th0 + thstep * sym11<R>(wth)));
      }
      R st0, ct0, sp0, cp0, st1, ct1, sp1, cp1;
      AG::sc_theta(th0, &st0, &ct0);
      AG::sc_theta(th1, &st1, &ct1);
      AG::sc_phi(ph0, &sp0, &cp0);
      AG::sc_phi(ph1, &sp1, &cp1);
      const T3 n0{cp0 * st0, sp0 * st0, ct0}, n1{cp1 * st1, sp1 * st1, ct1};
      T3 m0, m1;
      dipole<R, CT>(a_or_mu, k2e, n0.x, n0.y, n0.z, m0.x, m0.y, m0.z);
      dipole<R, CT>(a_or_mu, k2e, n1.x, n1.y, n1.z, m1.x, m1.y, m1.z);
      const bool hasL = idx > 0, hasR = idx + 1 < n;
      // the two neighbours, branch-free: at a chain end the clamped index re-reads the monomer itself and the
      // bond's contributions are masked out
      T3 nL, mL, nR, mR;
      const bool edgeL = load_nm(max(idx - 1, 0), nL, mL) && hasL;
      const bool edgeR = load_nm(min(idx + 1, n - 1), nR, mR) && hasR;
      const R du_field = mhalfE0 * (m1.z - m0.z);
      R dpsi = 0, dbend = 0, dpair = 0;
      {
        R p0, e0, q0, p1, e1, q1;
        bond(nL, mL, n0, m0, p0, e0, q0);
        bond(nL, mL, n1, m1, p1, e1, q1);
        dpsi += hasL ? p1 - p0 : (R)0; dbend += hasL ? e1 - e0 : (R)0; dpair += hasL ? q1 - q0 : (R)0;
      }
      {
        R p0, e0, q0, p1, e1, q1;
        bond(n0, m0, nR, mR, p0, e0, q0);
        bond(n1, m1, nR, mR, p1, e1, q1);
        dpsi += hasR ? p1 - p0 : (R)0; dbend += hasR ? e1 - e0 : (R)0; dpair += hasR ? q1 - q0 : (R)0;
      }

      // ---- cluster_flip!(trial, idx), inc/eap_chain.jl:269-333
      // `edge`: a member's theta is exactly 0 or pi (only a clamp produces those).  The reference
      // re-derives sin(theta) after the reflection and fl(pi) -> 0 turns 1.2e-16 into an exact 0, so
      // its log-density of such a proposal is -inf (or NaN): never accepted.
      R alpha = 1;
      bool flipped = false, edge = false;
      int upper = idx, lower = idx;
      R drz_flip = 0, du_flip = 0, dpair_flip = 0, dpsi_flip = 0;
      T3 dp_flip{0, 0, 0};
      // (the region below is predicated by `flipped`, not branched around: with ~50 lanes some lane flips in
      // practically every step, so the wave runs it anyway, and without the branch the scheduler can overlap
      // it with the single-move arithmetic)
      flipped = !(u01<R>(g.next()) <= cprob);                               // :276
      if (__builtin_amdgcn_ballot_w64(flipped) != 0) {     // wave-uniform: skipped only if no lane flips at all
        edge = flipped && (th1 == (R)0 || th1 == AG::theta_max);
        R snz = n1.z;                 // sums over the members (the moved monomer enters as proposed)
        T3 sm = m1;
        R upper_p = 0, lower_p = 0, new_upper_p = 0, new_lower_p = 0;
        // Both ends grow in ONE loop: round t tests the link above the cluster, (idx+t, idx+t+1), then
        // the link below it, (idx-t, idx-t-1), each with its own draw while that end is still growing
        // (the stream contract; the reference runs the two loops one after the other, :281-309 -- the
        // links are disjoint and the draws iid, so the law of (lower, upper) is the same).
        // Every lane that is still growing at round t has accepted exactly t links on that side, so
        // the rows visited depend on t only: the n-hats shift down a window (A <- B <- prefetched)
        // without selects, the row two away is read and converted speculatively, and the only
        // predicated state is the generator, the extents, the member sums and the clamp flag.
        T3 Au = n1, Bu = nR, Al = n1, Bl = nL;
        bool eBu = edgeR, eBl = edgeL;
        bool gu = flipped && hasR, gl = flipped && hasL;
        int rowu = min(idx + 2, n - 1), rowl = max(idx - 2, 0);
        Cell au = ang[rowu * lanes + lane], al = ang[rowl * lanes + lane];
        auto round = [&]() __attribute__((always_inline)) {
          T3 Cu, Cl;
          const bool eCu = nhat_of(dec(au), Cu), eCl = nhat_of(dec(al), Cl);
          rowu = min(rowu + 1, n - 1); rowl = max(rowl - 1, 0);
          au = ang[rowu * lanes + lane]; al = ang[rowl * lanes + lane];
          {
            const R p = (1 + dot3(Au, Bu)) / 2;
            G g2 = g;
            const bool acc = gu && (u01<R>(g2.next()) <= p);
            g.pick(gu, g2);
            upper_p = gu ? p : upper_p;
            upper += acc ? 1 : 0;
            edge = edge || (acc && eBu);
            member(acc, Bu, snz, sm);
            gu = acc && upper < n - 1;
            Au = Bu; Bu = Cu; eBu = eCu;
          }
          {
            const R p = (1 + dot3(Al, Bl)) / 2;
            G g2 = g;
            const bool acc = gl && (u01<R>(g2.next()) <= p);
            g.pick(gl, g2);
            lower_p = gl ? p : lower_p;
            lower -= acc ? 1 : 0;
            edge = edge || (acc && eBl);
            member(acc, Bl, snz, sm);
            gl = acc && lower > 0;
            Al = Bl; Bl = Cl; eBl = eCl;
          }
        };
        // two rounds per trip: the body is branch-free (a finished end neither draws nor moves), so the second
        // round is harmless when everything stopped in the first, and the loop test is paid half as often
        while (gu || gl) { round(); round(); }
        upper_p = upper >= n - 1 ? (R)0 : upper_p;   // ran into the chain end: no link to test, :282-284
        lower_p = lower <= 0 ? (R)0 : lower_p;       // :299-301
        // the two boundary bonds, before and after the reflection (:318-326); their monomers are read
        // back from LDS (the moved monomer enters as proposed)
        {
          T3 cu, cum, nu, num, cl, clm, nl, nlm;
          load_nm(upper, cu, cum);
          load_nm(min(upper + 1, n - 1), nu, num);
          load_nm(lower, cl, clm);
          load_nm(max(lower - 1, 0), nl, nlm);
          const bool selfu = upper == idx, selfl = lower == idx;
          cu.x = selfu ? n1.x : cu.x; cu.y = selfu ? n1.y : cu.y; cu.z = selfu ? n1.z : cu.z;
          cum.x = selfu ? m1.x : cum.x; cum.y = selfu ? m1.y : cum.y; cum.z = selfu ? m1.z : cum.z;
          cl.x = selfl ? n1.x : cl.x; cl.y = selfl ? n1.y : cl.y; cl.z = selfl ? n1.z : cl.z;
          clm.x = selfl ? m1.x : clm.x; clm.y = selfl ? m1.y : clm.y; clm.z = selfl ? m1.z : clm.z;
          {
            const bool on = flipped && upper < n - 1;
            const T3 rf = refl_n(cu), rfm = refl_mu(cum);
            R p0, e0, q0, p1, e1, q1;
            bond(cu, cum, nu, num, p0, e0, q0);
            bond(rf, rfm, nu, num, p1, e1, q1);
            new_upper_p = on ? (1 + dot3(rf, nu)) / 2 : (R)0;
            dpsi_flip += on ? p1 - p0 : (R)0; du_flip += on ? e1 - e0 : (R)0; dpair_flip += on ? q1 - q0 : (R)0;
          }
          {
            const bool on = flipped && lower > 0;
            const T3 rf = refl_n(cl), rfm = refl_mu(clm);
            R p0, e0, q0, p1, e1, q1;
            bond(nl, nlm, cl, clm, p0, e0, q0);
            bond(nl, nlm, rf, rfm, p1, e1, q1);
            new_lower_p = on ? (1 + dot3(rf, nl)) / 2 : (R)0;
            dpsi_flip += on ? p1 - p0 : (R)0; du_flip += on ? e1 - e0 : (R)0; dpair_flip += on ? q1 - q0 : (R)0;
          }
        }
        R ratio;
        if constexpr (sizeof(R) == 8)
          ratio = ((1 - new_upper_p) * (1 - new_lower_p)) / ((1 - upper_p) * (1 - lower_p));   // :328-329
        else
          ratio = ((1 - new_upper_p) * (1 - new_lower_p)) * __builtin_amdgcn_rcpf((1 - upper_p) * (1 - lower_p));
        alpha = flipped ? ratio : (R)1;
        // members' own terms: n_z -> -n_z; dielectric mu -> (-mu_x, -mu_y, mu_z), polar mu_z -> -mu_z
        const R f2 = flipped ? (R)-2 : (R)0;
        drz_flip = b * (f2 * snz);
        if constexpr (CT == PSTAT_DIELECTRIC) { dp_flip.x = f2 * sm.x; dp_flip.y = f2 * sm.y; }
        else { dp_flip.z = f2 * sm.z; du_flip += mhalfE0 * dp_flip.z; }
      }
      const uint32_t weps = g.next();   // the acceptance draw comes after the cluster's draws

      // ---- energy difference of the whole proposal, inc/energy.jl:7-23
      const R drx = b * (n1.x - n0.x), dry = b * (n1.y - n0.y), drz = b * (n1.z - n0.z) + drz_flip;
      const R dus = du_field + dbend + du_flip;        // change of sum(u), bending included
      const R dU = dus + (dpair + dpair_flip) - (Fx * drx + Fz * drz);

      // ---- Metropolis-Hastings, inc/acceptance.jl:29-39
      bool ok;
      const R dw = umb ? dus * wscale : (R)0;
      if constexpr (sizeof(R) == 8) {
        // (the f32 filter of pstat_math.h decides all but ~1e-5 of the draws; the literal expression the rest)
        ok = metropolis_filter(dU * (-1.0 / kT) + (dw - lag), st1 * alpha, st0, weps, [&]() -> bool {
          const R delta = -dU / kT + log_r(st1 / st0) + dw + log_r(alpha) - lag;
          const R eps = (R)eps_uniform(A.wide_eps != 0, weps, w0, wphi, wth);
          return (delta >= 0) || (eps < exp_r(delta));
        });
      } else {
        const R e = __builtin_amdgcn_exp2f((R)1.44269504f * (dw - lag) + dU * nbeta_log2e) * alpha;
        ok = bits12(weps) * st0 < fma_r(st1, e, st0);   // (1 + u) sin0 < sin1 e alpha + sin0
      }

      ok = ok && !edge && inside;
      if constexpr (EN == PSTAT_ISING) nnan_seg += not_finite(dU) ? 1 : 0;

      // ---- commit
      if (ok) {
        Cell a1;
        if constexpr (Q) a1 = (flipped ? 65535u - k1c : k1c) | (j1 << 16);
        else { a1.x = flipped ? refl_theta(th1) : th1; a1.y = ph1; }
        ang[cell] = a1;
        if (flipped) {
          // four members per pass: the reads are independent, slots past `upper` alias `upper` and
          // write the same value again; the moved monomer (already stored) is passed through
          for (int i = lower; i <= upper; i += 4) {
            Cell v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ang[min(i + j, upper) * lanes + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int m = min(i + j, upper);
              ang[m * lanes + lane] = m == idx ? v[j] : refl_cell(v[j]);
            }
          }
        }
        Orx += drx; Ory += dry; Orz += drz;
        Opx += (m1.x - m0.x) + dp_flip.x; Opy += (m1.y - m0.y) + dp_flip.y; Opz += (m1.z - m0.z) + dp_flip.z;
        OU += dU;
        usum += dus;
        psisum += dpsi + dpsi_flip;
        c2sum += ct1 * ct1 - ct0 * ct0;
        lag = log_r(alpha);
        ++nacc_seg;
      }

      // ---- record! x 10, mcmc_clustering_eap_chain.jl:243-244,310-311
      R wgt = 1;
      if (umb) {
        bool raise;
        R wrel = umbrella_logw(usum, uref, wscale, raise);
        if (__builtin_amdgcn_ballot_w64(raise) != 0) {   // the gauge rises to this configuration (pstat_math.h)
          if (raise) {
            const double f = exp_f64(-(double)wrel);
            const R fr = (R)f;
#pragma unroll
            for (int q = 0; q < 9; ++q) a1[q] *= fr;
#pragma unroll
            for (int q = 0; q < 7; ++q) a2[q] *= fr;
            accw *= fr;
#pragma unroll
            for (int q = 0; q < NSUMS; ++q) sums[q] *= f;
            wnorm *= f;
            uref = usum; regauged = true; wrel = 0;
          }
        }
        wgt = exp_r(wrel);
      }
      const R psim = psisum * inv_nm1;
      accw += wgt;
      a1[0] = fma_r(wgt, Orx, a1[0]); a1[1] = fma_r(wgt, Ory, a1[1]); a1[2] = fma_r(wgt, Orz, a1[2]);
      a1[3] = fma_r(wgt, Opx, a1[3]); a1[4] = fma_r(wgt, Opy, a1[4]); a1[5] = fma_r(wgt, Opz, a1[5]);
      a1[6] = fma_r(wgt, OU, a1[6]); a1[7] = fma_r(wgt, c2sum, a1[7]); a1[8] = fma_r(wgt, psim, a1[8]);
      a2[0] = fma_r(wgt * Orx, Orx, a2[0]); a2[1] = fma_r(wgt * Ory, Ory, a2[1]); a2[2] = fma_r(wgt * Orz, Orz, a2[2]);
      a2[3] = fma_r(wgt * Opx, Opx, a2[3]); a2[4] = fma_r(wgt * Opy, Opy, a2[4]); a2[5] = fma_r(wgt * Opz, Opz, a2[5]);
      a2[6] = fma_r(wgt * OU, OU, a2[6]);
    }

    sums[S_R1] += (double)a1[0]; sums[S_R2] += (double)a1[1]; sums[S_R3] += (double)a1[2];
    sums[S_P1] += (double)a1[3]; sums[S_P2] += (double)a1[4]; sums[S_P3] += (double)a1[5];
    sums[S_U] += (double)a1[6]; sums[S_C2] += (double)a1[7]; sums[S_PSI] += (double)a1[8];
    sums[S_R1SQ] += (double)a2[0]; sums[S_R2SQ] += (double)a2[1]; sums[S_R3SQ] += (double)a2[2];
    sums[S_P1SQ] += (double)a2[3]; sums[S_P2SQ] += (double)a2[4]; sums[S_P3SQ] += (double)a2[5];
    sums[S_USQ] += (double)a2[6];
    wnorm += (double)accw;
    step += chunk;
    left -= chunk;
    steps_seg += chunk;

    // ---- step-size adaptation, mcmc_clustering_eap_chain.jl:287-308
    if (A.adaptive) {
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
        phistep = (R)(phistep_d / ph_unit); thstep = (R)(thstep_d / th_unit);
      }
    }
  }

  if constexpr (Q) {  // ---- spill
    uint16_t *gth = (uint16_t *)S.ang, *gph = (uint16_t *)S.ang + (int64_t)n * C;
    for (int i = 0; i < n; ++i) {
      const uint32_t v = ang[i * lanes + lane];
      gth[(int64_t)i * C + c] = (uint16_t)(v & 0xFFFFu);
      gph[(int64_t)i * C + c] = (uint16_t)(v >> 16);
    }
  } else {
    R *gth = (R *)S.ang, *gph = (R *)S.ang + (int64_t)n * C;
    for (int i = 0; i < n; ++i) {
      const R2 v = ang[i * lanes + lane];
      gth[(int64_t)i * C + c] = v.x;
      gph[(int64_t)i * C + c] = v.y;
    }
  }
  g.store(S.rng + c, C);
  S.stepsz[0 * C + c] = phistep_d; S.stepsz[1 * C + c] = thstep_d;
  S.win[0 * C + c] = nacc_off + nacc_seg; S.win[1 * C + c] = natt_off + steps_seg;
  S.nacc_total[c] += nacc_seg;
  if constexpr (EN == PSTAT_ISING) S.nanrej[c] += nnan_seg;
  S.obs[OBS_R1 * C + c] = Orx; S.obs[OBS_R2 * C + c] = Ory; S.obs[OBS_R3 * C + c] = Orz;
  S.obs[OBS_P1 * C + c] = Opx; S.obs[OBS_P2 * C + c] = Opy; S.obs[OBS_P3 * C + c] = Opz;
  S.obs[OBS_U * C + c] = OU; S.obs[OBS_USUM * C + c] = usum;
  S.obs[OBS_C2 * C + c] = c2sum; S.obs[OBS_PSI * C + c] = psisum;
  S.lag[c] = lag;
  if (umb) S.wnorm[c] = wnorm;
  if (regauged) S.uref[c] = (double)uref;
#pragma unroll
  for (int q = 0; q < NSUMS; ++q) S.sums[q * C + c] = sums[q];
}

// the persistent (block, segment) job loop of pstat_device.h around run_cluster_segment
// (PACKED: chain blocks straddle cases, the case's scalars are per-lane values -- run_job_queue, pstat_device.h)
template <typename R, typename G, int CT, int EN, int ST, bool PACKED>
__global__ __launch_bounds__(64) void cluster_kernel(SweepArgs A, DevState S, const CaseConst *__restrict__ cases,
                                                     int umbrella, int *__restrict__ queue) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  run_job_queue<PACKED>(A, queue, lane, [&](const CaseConst &cc, int64_t chain, int64_t first, int64_t len, int) {
    run_cluster_segment<R, G, CT, EN, ST>(A, S, cc, umbrella, smem, lane, chain, first, len);
  }, cases);
}

using ClusterFn = void (*)(SweepArgs, DevState, const CaseConst *, int, int *);

template <typename R, typename G, int ST, bool PACKED>
ClusterFn pick_ct_en_p(const LaunchCfg &cfg) {
  const bool ising = cfg.energy_type == PSTAT_ISING;
  if (cfg.chain_type == PSTAT_DIELECTRIC)
    return ising ? cluster_kernel<R, G, PSTAT_DIELECTRIC, PSTAT_ISING, ST, PACKED>
                 : cluster_kernel<R, G, PSTAT_DIELECTRIC, PSTAT_NONINTERACTING, ST, PACKED>;
  return ising ? cluster_kernel<R, G, PSTAT_POLAR, PSTAT_ISING, ST, PACKED>
               : cluster_kernel<R, G, PSTAT_POLAR, PSTAT_NONINTERACTING, ST, PACKED>;
}
template <typename R, typename G, int ST>
ClusterFn pick_ct_en(const LaunchCfg &cfg) {
  return cfg.packed ? pick_ct_en_p<R, G, ST, true>(cfg) : pick_ct_en_p<R, G, ST, false>(cfg);
}

}  // namespace

// Two objects are built from this file (csrc/Makefile): -DPSTAT_CPART=1 holds the f32 and q16 instantiations
// and is compiled with -ffp-contract=fast (statistical parity only), -DPSTAT_CPART=2 the f64 ones (LDS home: the literal witness, see the Makefile) and the
// launchers with -ffp-contract=off (bit parity with the oracle).  Without the macro: everything in one object.
#if !defined(PSTAT_CPART) || PSTAT_CPART == 1
ClusterFn pick_cluster_f32(const LaunchCfg &cfg) {
  const bool xo = cfg.rng == PSTAT_RNG_XOSHIRO128PP;
  if (cfg.precision == PSTAT_Q16) return xo ? pick_ct_en<float, Xoshiro128pp, 1>(cfg) : pick_ct_en<float, Mwc64x, 1>(cfg);
  return xo ? pick_ct_en<float, Xoshiro128pp, 0>(cfg) : pick_ct_en<float, Mwc64x, 0>(cfg);
}
#endif
#if !defined(PSTAT_CPART) || PSTAT_CPART == 2
ClusterFn pick_cluster_f32(const LaunchCfg &cfg);

ClusterFn pick_cluster_f64(const LaunchCfg &cfg) {
  return cfg.rng == PSTAT_RNG_XOSHIRO128PP ? pick_ct_en<double, Xoshiro128pp, 0>(cfg) : pick_ct_en<double, Mwc64x, 0>(cfg);
}

static ClusterFn pick_cluster(const LaunchCfg &cfg) {
  return cfg.precision == PSTAT_F64 ? pick_cluster_f64(cfg) : pick_cluster_f32(cfg);
}

static int cluster_lds_bytes(const LaunchCfg &cfg, const SweepArgs &a) {
  return (int)(a.n * a.lanes * (cfg.precision == PSTAT_F64 ? 16 : (cfg.precision == PSTAT_Q16 ? 4 : 8)));
}

hipError_t cluster_kernel_info(const LaunchCfg &cfg, const SweepArgs &a, int *lds_bytes, int *blocks_per_cu,
                               const char **name) {
  if (cfg.state_global) return cluster_gm_kernel_info(cfg, a, lds_bytes, blocks_per_cu, name);   // pstat_cluster_gm.hip
  ClusterFn fn = pick_cluster(cfg);
  const int lds = cluster_lds_bytes(cfg, a);
  hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  int nb = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)fn, 64, lds);
  if (e != hipSuccess) return e;
  if (lds_bytes) *lds_bytes = lds;
  if (blocks_per_cu) *blocks_per_cu = nb;
  if (name) *name = cfg.precision == PSTAT_F64 ? (cfg.packed ? "cluster_kernel<double> [packed cases]" : "cluster_kernel<double>")
                 : (cfg.precision == PSTAT_Q16 ? (cfg.packed ? "cluster_kernel<float, q16 state> [packed cases]" : "cluster_kernel<float, q16 state>")
                                               : (cfg.packed ? "cluster_kernel<float> [packed cases]" : "cluster_kernel<float>"));
  return hipSuccess;
}

hipError_t launch_cluster(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s, const CaseConst *cases,
                          int *queue, unsigned grid, hipStream_t stream) {
  if (cfg.state_global) return launch_cluster_gm(cfg, a, s, cases, queue, grid, stream);
  ClusterFn fn = pick_cluster(cfg);
  const int lds = cluster_lds_bytes(cfg, a);
  hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(queue + 1, 0, sizeof(int) * (sweep_queue_ints(a) - 1), stream);   // queue[0]: sticky error word
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, stream, a, s, cases, cfg.umbrella, queue);
  return hipGetLastError();
}
#endif

}  // namespace pstat
