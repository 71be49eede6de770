// pstat_cluster_gm.hip -- the f64 step of mcmc_clustering_eap_chain.jl:268-311 with the chain state in DEVICE MEMORY.
//
// Same step, stream contract and results as cluster_kernel<double> of pstat_cluster.hip (one chain per lane, the
// persistent (block, segment) job queue); what differs is where a chain lives while a segment runs and what a cell holds.
//
//   * An f64 (theta, phi) cell is 16 bytes, so LDS seats 160 KiB / (16 n) chains per CU: 102 at n = 100, 51 at n = 200 --
//     a quarter (an eighth) of the 256 lanes of a CU's four SIMDs.  Here the cells live in DevState::work, laid out
//     CHAIN-CONTIGUOUS, [chain block][lane][monomer], and every SIMD carries a full wave.  A proposal touches the moved
//     monomer, its two neighbours and the monomers the cluster grows over -- one contiguous run of the chain, i.e. two
//     or three 128-byte lines per proposal whatever the lane's random monomer index is.
//   * A cell is the reference's own per-monomer cache (inc/eap_chain.jl:22-28: cphi, sphi, ctheta, stheta, n-hat), 48
//     bytes: [n_x, n_y | n_z, theta | phi, sin(theta)].  The LDS kernel re-derives n-hat from the angles of every row it
//     visits (two sincos per row: ~65 f64 instructions, ~50 rows per wave-step, two thirds of its instruction stream);
//     LDS capacity forbade the cache there, device memory does not.  A reflection (refl_n!, inc/eap_chain.jl:263-265)
//     maps a cached cell exactly: n_z -> -n_z, theta -> clamp(theta + (pi - 2 theta)), sin(theta) kept.  The reference
//     recomputes sin and cos of the reflected angle, which agrees with the mapped values to an ulp or two (theta +
//     (pi - 2 theta) is pi - theta rounded); that is the same class of difference as device sincos vs glibc (see
//     sincos_fast_f64) and moves a decision with probability ~1e-9 per decision: the trajectories of the bit-parity
//     tests are unchanged.  theta itself is stored and updated exactly as the reference does.
//   * Loads and stores go through a buffer resource and are steered by ADDRESS: an offset past the buffer makes a load
//     return 0 and drops a store, so speculative rows of lanes that do not grow, and the stores of rejected proposals,
//     cost no traffic and no branch.
//   * Order inside a step: all rows whose addresses are known from the draws (the moved monomer, its neighbours and two
//     rows further out on either side) are requested at once; growth requests two rounds ahead; the cluster's boundary
//     rows and its members (for the read-modify-write of an accepted reflection) are requested as soon as the extents
//     are known and land while the single move's four bond terms are computed.
// HBM holds the checkpoint layout (DevState::ang, angles only); the working buffer is filled from it when a segment
// starts and spilled to it when it ends, exactly as the LDS variants do.
#include <hip/hip_runtime.h>

#include "pstat_cluster_common.h"
#include "pstat_device.h"
#include "pstat_math.h"

namespace pstat {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2dd __attribute__((ext_vector_type(2)));

constexpr uint32_t CELL = PSTAT_CLUSTER_GM_CELL;   // bytes per monomer
constexpr uint32_t OOB = 0x80000000u;  // past every working buffer (num_records < 2^31, checked by the host): no access

template <typename G, int CT, int EN>
__device__ __forceinline__ void run_cluster_segment_gm(const SweepArgs &A, const DevState &S, const CaseConst &cc,
                                                       const int umb_on, const int lane, const int64_t c, int64_t step,
                                                       int64_t remaining, const int blk) {
  using R = double;
  using AG = Ang<R>;
  using T3 = V3<R>;
  const int lanes = A.lanes;
  const int64_t C = S.C;
  const int n = (int)A.n;
  constexpr R PI = AG::theta_max;

  const R Fz = cc.Fz, Fx = cc.Fx, b = cc.b, kT = cc.kT;
  const R a_or_mu = (CT == PSTAT_DIELECTRIC) ? (cc.K1 - cc.K2) * cc.E0 : cc.mu;
  const R k2e = cc.K2 * cc.E0;
  const R mhalfE0 = -0.5 * cc.E0;
  const R hb = -cc.b / 2;
  const R khalf = cc.kappa / 2, psi0 = cc.psi0;
  const R cprob = cc.cluster_prob;
  (void)hb;

  // ---- the wave's working buffer: [lane][monomer] cells of this chain block
  const uint32_t chain_bytes = (uint32_t)n * CELL;
  unsigned char *const wbase = reinterpret_cast<unsigned char *>(S.work) + (size_t)blk * (size_t)lanes * chain_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wbase, 0, (int)((uint32_t)lanes * chain_bytes), 0x00020000);
  const uint32_t lb = (uint32_t)lane * chain_bytes;
  auto ld = [&](const uint32_t off) __attribute__((always_inline)) -> v2dd {
    return __builtin_bit_cast(v2dd, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
  };
  auto st = [&](const uint32_t off, const R x, const R y) __attribute__((always_inline)) {
    const v2dd v = {x, y};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), rsrc, off, 0, 0);
  };
  auto row_off = [&](const int row) __attribute__((always_inline)) -> uint32_t { return lb + (uint32_t)row * CELL; };

  // ---- fill: angles from the checkpoint planes (coalesced over the lanes), the cached trigonometry derived from them
  // with the same functions the step uses for a moved monomer
  {
    const R *gth = (const R *)S.ang, *gph = (const R *)S.ang + (int64_t)n * C;
#pragma unroll 2
    for (int i = 0; i < n; ++i) {
      const R th = gth[(int64_t)i * C + c], ph = gph[(int64_t)i * C + c];
      R s, co, sp, cp;
      AG::sc_theta(th, &s, &co);
      AG::sc_phi(ph, &sp, &cp);
      const uint32_t o = row_off(i);
      st(o, cp * s, sp * s);
      st(o + 16, co, th);
      st(o + 32, ph, s);
    }
  }
  G g;
  g.load(S.rng + c, C);
  double phistep = S.stepsz[0 * C + c], thstep = S.stepsz[1 * C + c];
  int64_t nacc_off = S.win[0 * C + c], natt_off = S.win[1 * C + c];
  int nacc_seg = 0, steps_seg = 0;
  int nnan_seg = 0;   // proposals with a non-finite energy difference (Ising pair terms at r -> 0)
  R Orx = S.obs[OBS_R1 * C + c], Ory = S.obs[OBS_R2 * C + c], Orz = S.obs[OBS_R3 * C + c];
  R Opx = S.obs[OBS_P1 * C + c], Opy = S.obs[OBS_P2 * C + c], Opz = S.obs[OBS_P3 * C + c];
  R OU = S.obs[OBS_U * C + c];
  R usum = S.obs[OBS_USUM * C + c];      // sum of u_i INCLUDING the bending terms (eap_chain.jl:53-58)
  R c2sum = S.obs[OBS_C2 * C + c], psisum = S.obs[OBS_PSI * C + c];
  // log(alpha) of the last accepted proposal of this mcmc() call (inc/acceptance.jl:33-36), kept as alpha itself
  // (`lag_alpha`, lag_pending) until a literal evaluation or the spill needs the logarithm
  R lag = S.lag[c], lag_alpha = 1;
  bool lag_pending = false;
  const bool umb = umb_on != 0;
  const R wscale = umb ? (0.2 + 0.8 * exp(-(cc.Fx * cc.Fx + cc.Fz * cc.Fz) / cc.kT)) / cc.kT : 0.0;
  const R uref = umb ? S.uref[c] : 0.0;
  double wnorm = umb ? S.wnorm[c] : 0.0;
  double sums[NSUMS];
#pragma unroll
  for (int q = 0; q < NSUMS; ++q) sums[q] = S.sums[q * C + c];
  const R inv_nm1 = n > 1 ? 1.0 / (double)(n - 1) : 0.0;
  const R ninv_kT = -1.0 / kT;

  const int64_t spa = A.steps_per_adjust;
  int64_t to_adj = A.adaptive ? spa - (step % spa) : 0;
  constexpr int FLUSH = 128;
  int left = (int)remaining;

  auto is_edge = [](const R th) __attribute__((always_inline)) -> bool { return th == (R)0 || th == PI; };
  auto mu_of = [&](const T3 &nh) __attribute__((always_inline)) -> T3 {
    T3 m;
    dipole<R, CT>(a_or_mu, k2e, nh.x, nh.y, nh.z, m.x, m.y, m.z);
    return m;
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
  // reflection through the plane normal to the field: refl_n!, inc/eap_chain.jl:263-265
  auto refl_theta = [&](const R th) __attribute__((always_inline)) -> R { return fmin(PI, fmax((R)0, th + (PI - 2 * th))); };
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

  while (left > 0) {
    int chunk = left < FLUSH ? left : FLUSH;
    if (A.adaptive && to_adj < chunk) chunk = (int)to_adj;
    R a1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, a2[7] = {0, 0, 0, 0, 0, 0, 0};
    R accw = 0;

    for (int s = 0; s < chunk; ++s) {
      // ---- every draw whose place in the stream is fixed: mcmc_clustering_eap_chain.jl:269-272 and the skip draw of
      // cluster_flip! (inc/eap_chain.jl:276)
      const int idx = (int)__umulhi(g.next(), (uint32_t)n);
      const uint32_t wphi = g.next(), wth = g.next();
      const bool flipped = !(u01<R>(g.next()) <= cprob);
      const bool hasL = idx > 0, hasR = idx + 1 < n;
      // ---- every row whose address follows from them.  At a chain end the missing neighbour's slot re-reads the
      // monomer itself and its bond is masked out; rows past the chain, and the outer rows of lanes that grow no
      // cluster, are steered off the buffer.
      const uint32_t off0 = row_off(idx);
      const v2dd c0a = ld(off0), c0b = ld(off0 + 16), c0c = ld(off0 + 32);
      const uint32_t offL = hasL ? off0 - CELL : off0, offR = hasR ? off0 + CELL : off0;
      const v2dd La = ld(offL), Lb = ld(offL + 16), Ra = ld(offR), Rb = ld(offR + 16);
      auto up_off = [&](const int row, const bool want) __attribute__((always_inline)) -> uint32_t {
        return (want && row <= n - 1) ? row_off(row) : OOB;
      };
      auto dn_off = [&](const int row, const bool want) __attribute__((always_inline)) -> uint32_t {
        return (want && row >= 0) ? row_off(row) : OOB;
      };
      uint32_t o;
      o = up_off(idx + 2, flipped); v2dd Xua = ld(o), Xub = ld(o + 16);
      o = dn_off(idx - 2, flipped); v2dd Xla = ld(o), Xlb = ld(o + 16);
      o = up_off(idx + 3, flipped); v2dd Yua = ld(o), Yub = ld(o + 16);
      o = dn_off(idx - 3, flipped); v2dd Yla = ld(o), Ylb = ld(o + 16);

      // ---- the single-monomer part
      const R th0 = c0b.y, ph0 = c0c.x, st0 = c0c.y;
      const T3 n0{c0a.x, c0a.y, c0b.x};
      const R ct0 = n0.z;
      const R ph1 = ph0 + phistep * sym11<R>(wphi);
      const R th1 = fmin(PI, fmax((R)0, th0 + thstep * sym11<R>(wth)));
      R st1, ct1, sp1, cp1;
      AG::sc_theta(th1, &st1, &ct1);
      AG::sc_phi(ph1, &sp1, &cp1);
      const T3 n1{cp1 * st1, sp1 * st1, ct1};
      const T3 m0 = mu_of(n0), m1 = mu_of(n1);
      const T3 nL{La.x, La.y, Lb.x}, nR{Ra.x, Ra.y, Rb.x};
      const T3 mL = mu_of(nL), mR = mu_of(nR);
      const bool edgeL = is_edge(Lb.y) && hasL, edgeR = is_edge(Rb.y) && hasR;

      // ---- cluster_flip!(trial, idx), inc/eap_chain.jl:269-333 (see pstat_cluster.hip for the scheme: both ends grow
      // in one loop of uniform rounds, each with its own draw while that end still grows)
      R alpha = 1;
      bool edge = false;
      int upper = idx, lower = idx;
      R drz_flip = 0, du_flip = 0, dpair_flip = 0, dpsi_flip = 0;
      T3 dp_flip{0, 0, 0};
      R snz = n1.z;                 // sums over the members (the moved monomer enters as proposed)
      T3 sm = m1;
      R upper_p = 0, lower_p = 0, new_upper_p = 0, new_lower_p = 0;
      v2dd cua{0, 0}, cub{0, 0}, nua{0, 0}, nub{0, 0}, cla{0, 0}, clb{0, 0}, nla{0, 0}, nlb{0, 0};
      v2dd mv[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
      const bool any_flip = __builtin_amdgcn_ballot_w64(flipped) != 0;   // wave-uniform
      if (any_flip) {
        edge = flipped && is_edge(th1);
        T3 Au = n1, Bu = nR, Al = n1, Bl = nL;
        bool eBu = edgeR, eBl = edgeL;
        bool gu = flipped && hasR, gl = flipped && hasL;
        int rowu = idx + 3, rowl = idx - 3;   // the outermost rows requested so far
        auto round = [&](v2dd &ua, v2dd &ub, v2dd &la, v2dd &lb_) __attribute__((always_inline)) {
          const T3 Cu{ua.x, ua.y, ub.x}, Cl{la.x, la.y, lb_.x};
          const bool eCu = is_edge(ub.y), eCl = is_edge(lb_.y);
          rowu += 1; rowl -= 1;       // this register set is free again: request the row two rounds out
          uint32_t q;
          q = up_off(rowu, gu); ua = ld(q); ub = ld(q + 16);
          q = dn_off(rowl, gl); la = ld(q); lb_ = ld(q + 16);
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
        while (gu || gl) { round(Xua, Xub, Xla, Xlb); round(Yua, Yub, Yla, Ylb); }
        upper_p = upper >= n - 1 ? (R)0 : upper_p;   // ran into the chain end: no link to test, :282-284
        lower_p = lower <= 0 ? (R)0 : lower_p;       // :299-301
        // the extents are known: request the monomers of the two boundary bonds (:318-326; the moved monomer enters as
        // proposed) and the first four members, whose cells an accepted proposal rewrites
        const bool selfu = upper == idx, selfl = lower == idx;
        o = (flipped && !selfu) ? row_off(upper) : OOB; cua = ld(o); cub = ld(o + 16);
        o = (flipped && upper < n - 1) ? row_off(upper + 1) : OOB; nua = ld(o); nub = ld(o + 16);
        o = (flipped && !selfl) ? row_off(lower) : OOB; cla = ld(o); clb = ld(o + 16);
        o = (flipped && lower > 0) ? row_off(lower - 1) : OOB; nla = ld(o); nlb = ld(o + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = lower + j;
          mv[j] = ld((flipped && m <= upper && m != idx) ? row_off(m) + 16 : OOB);
        }
      }

      // ---- the single move's two bonds, before and after (they need nothing of the above: its loads land meanwhile)
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

      if (any_flip) {
        // the two boundary bonds, before and after the reflection
        const bool selfu = upper == idx, selfl = lower == idx;
        T3 cu{cua.x, cua.y, cub.x}, cl{cla.x, cla.y, clb.x};
        const T3 nu{nua.x, nua.y, nub.x}, nl{nla.x, nla.y, nlb.x};
        cu.x = selfu ? n1.x : cu.x; cu.y = selfu ? n1.y : cu.y; cu.z = selfu ? n1.z : cu.z;
        cl.x = selfl ? n1.x : cl.x; cl.y = selfl ? n1.y : cl.y; cl.z = selfl ? n1.z : cl.z;
        const T3 cum = mu_of(cu), clm = mu_of(cl), num = mu_of(nu), nlm = mu_of(nl);
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
        const R ratio = ((1 - new_upper_p) * (1 - new_lower_p)) / ((1 - upper_p) * (1 - lower_p));   // :328-329
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

      // ---- Metropolis-Hastings, inc/acceptance.jl:29-39.  The f32 filter of pstat_math.h decides all but ~1e-5 of the
      // draws, the literal expression the rest.  The cached log(alpha) of the last acceptance enters the filter as the
      // factor alpha itself (no logarithm on the common path).
      const R dw = umb ? dus * wscale : (R)0;
      bool ok = metropolis_filter(dU * ninv_kT + (dw - (lag_pending ? (R)0 : lag)), st1 * alpha,
                                  st0 * (lag_pending ? lag_alpha : (R)1), weps, [&]() -> bool {
        const R lg = lag_pending ? log_r(lag_alpha) : lag;
        const R delta = -dU / kT + log_r(st1 / st0) + dw + log_r(alpha) - lg;
        const R eps = u01<R>(weps);
        return (delta >= 0) || (eps < exp_r(delta));
      });
      ok = ok && !edge;
      if constexpr (EN == PSTAT_ISING) nnan_seg += not_finite(dU) ? 1 : 0;

      // ---- commit: stores steered by address (a rejected proposal stores nothing)
      {
        const bool okf = ok && flipped;
        const uint32_t os = ok ? off0 : OOB;
        st(os, n1.x, n1.y);
        st(os + 16, flipped ? -n1.z : n1.z, flipped ? refl_theta(th1) : th1);
        st(os + 32, ph1, st1);
        if (any_flip) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int m = lower + j;
            st((okf && m <= upper && m != idx) ? row_off(m) + 16 : OOB, -mv[j].x, refl_theta(mv[j].y));
          }
          // longer clusters: four members per pass
          for (int i = lower + 4; __builtin_amdgcn_ballot_w64(okf && i <= upper) != 0; i += 4) {
            v2dd v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int m = i + j;
              v[j] = ld((okf && m <= upper && m != idx) ? row_off(m) + 16 : OOB);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int m = i + j;
              st((okf && m <= upper && m != idx) ? row_off(m) + 16 : OOB, -v[j].x, refl_theta(v[j].y));
            }
          }
        }
        Orx = ok ? Orx + drx : Orx; Ory = ok ? Ory + dry : Ory; Orz = ok ? Orz + drz : Orz;
        Opx = ok ? Opx + ((m1.x - m0.x) + dp_flip.x) : Opx;
        Opy = ok ? Opy + ((m1.y - m0.y) + dp_flip.y) : Opy;
        Opz = ok ? Opz + ((m1.z - m0.z) + dp_flip.z) : Opz;
        OU = ok ? OU + dU : OU;
        usum = ok ? usum + dus : usum;
        psisum = ok ? psisum + (dpsi + dpsi_flip) : psisum;
        c2sum = ok ? c2sum + (ct1 * ct1 - ct0 * ct0) : c2sum;
        lag_alpha = ok ? alpha : lag_alpha;
        lag_pending = lag_pending || ok;
        nacc_seg += ok ? 1 : 0;
      }

      // ---- record! x 10, mcmc_clustering_eap_chain.jl:243-244,310-311
      const R wgt = umb ? exp_r(-(usum - uref) * wscale) : (R)1;
      const R psim = psisum * inv_nm1;
      accw += wgt;
      a1[0] = fma_r(wgt, Orx, a1[0]); a1[1] = fma_r(wgt, Ory, a1[1]); a1[2] = fma_r(wgt, Orz, a1[2]);
      a1[3] = fma_r(wgt, Opx, a1[3]); a1[4] = fma_r(wgt, Opy, a1[4]); a1[5] = fma_r(wgt, Opz, a1[5]);
      a1[6] = fma_r(wgt, OU, a1[6]); a1[7] = fma_r(wgt, c2sum, a1[7]); a1[8] = fma_r(wgt, psim, a1[8]);
      a2[0] = fma_r(wgt * Orx, Orx, a2[0]); a2[1] = fma_r(wgt * Ory, Ory, a2[1]); a2[2] = fma_r(wgt * Orz, Orz, a2[2]);
      a2[3] = fma_r(wgt * Opx, Opx, a2[3]); a2[4] = fma_r(wgt * Opy, Opy, a2[4]); a2[5] = fma_r(wgt * Opz, Opz, a2[5]);
      a2[6] = fma_r(wgt * OU, OU, a2[6]);
    }

    sums[S_R1] += a1[0]; sums[S_R2] += a1[1]; sums[S_R3] += a1[2];
    sums[S_P1] += a1[3]; sums[S_P2] += a1[4]; sums[S_P3] += a1[5];
    sums[S_U] += a1[6]; sums[S_C2] += a1[7]; sums[S_PSI] += a1[8];
    sums[S_R1SQ] += a2[0]; sums[S_R2SQ] += a2[1]; sums[S_R3SQ] += a2[2];
    sums[S_P1SQ] += a2[3]; sums[S_P2SQ] += a2[4]; sums[S_P3SQ] += a2[5];
    sums[S_USQ] += a2[6];
    wnorm += accw;
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
        if (ratio > A.adj_ub && phistep != K<double>::pi && thstep != K<double>::half_pi) {
          nacc_off = -nacc_seg; natt_off = -steps_seg;
          phistep = fmin(K<double>::pi, phistep * A.adj_scale);
          thstep = fmin(K<double>::half_pi, thstep * A.adj_scale);
        } else if (ratio < A.adj_lb) {
          nacc_off = -nacc_seg; natt_off = -steps_seg;
          phistep /= A.adj_scale;
          thstep /= A.adj_scale;
        }
      }
    }
  }

  // ---- spill: the angles back to the checkpoint planes
  {
    R *gth = (R *)S.ang, *gph = (R *)S.ang + (int64_t)n * C;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      const uint32_t o = row_off(i);
      const v2dd b1 = ld(o + 16), b2 = ld(o + 32);
      gth[(int64_t)i * C + c] = b1.y;
      gph[(int64_t)i * C + c] = b2.x;
    }
  }
  g.store(S.rng + c, C);
  S.stepsz[0 * C + c] = phistep; S.stepsz[1 * C + c] = thstep;
  S.win[0 * C + c] = nacc_off + nacc_seg; S.win[1 * C + c] = natt_off + steps_seg;
  S.nacc_total[c] += nacc_seg;
  if constexpr (EN == PSTAT_ISING) S.nanrej[c] += nnan_seg;
  S.obs[OBS_R1 * C + c] = Orx; S.obs[OBS_R2 * C + c] = Ory; S.obs[OBS_R3 * C + c] = Orz;
  S.obs[OBS_P1 * C + c] = Opx; S.obs[OBS_P2 * C + c] = Opy; S.obs[OBS_P3 * C + c] = Opz;
  S.obs[OBS_U * C + c] = OU; S.obs[OBS_USUM * C + c] = usum;
  S.obs[OBS_C2 * C + c] = c2sum; S.obs[OBS_PSI * C + c] = psisum;
  S.lag[c] = lag_pending ? log_r(lag_alpha) : lag;
  if (umb) S.wnorm[c] = wnorm;
#pragma unroll
  for (int q = 0; q < NSUMS; ++q) S.sums[q * C + c] = sums[q];
}

template <typename G, int CT, int EN>
__global__ __launch_bounds__(64) void cluster_gm_kernel(SweepArgs A, DevState S, const CaseConst *__restrict__ cases,
                                                        int umbrella, int *__restrict__ queue) {
  const int lane = threadIdx.x;
  run_job_queue(A, queue, lane, [&](const CaseConst &cc, int64_t chain, int64_t first, int64_t len, int blk) {
    run_cluster_segment_gm<G, CT, EN>(A, S, cc, umbrella, lane, chain, first, len, blk);
  }, cases);
}

using ClusterFn = void (*)(SweepArgs, DevState, const CaseConst *, int, int *);

template <typename G>
ClusterFn pick_ct_en(const LaunchCfg &cfg) {
  const bool ising = cfg.energy_type == PSTAT_ISING;
  if (cfg.chain_type == PSTAT_DIELECTRIC)
    return ising ? cluster_gm_kernel<G, PSTAT_DIELECTRIC, PSTAT_ISING> : cluster_gm_kernel<G, PSTAT_DIELECTRIC, PSTAT_NONINTERACTING>;
  return ising ? cluster_gm_kernel<G, PSTAT_POLAR, PSTAT_ISING> : cluster_gm_kernel<G, PSTAT_POLAR, PSTAT_NONINTERACTING>;
}

ClusterFn pick(const LaunchCfg &cfg) {
  return cfg.rng == PSTAT_RNG_XOSHIRO128PP ? pick_ct_en<Xoshiro128pp>(cfg) : pick_ct_en<Mwc64x>(cfg);
}

}  // namespace

size_t cluster_gm_work_bytes(const SweepArgs &a) {
  return (size_t)(a.blocks_per_case * a.ncases) * (size_t)a.lanes * (size_t)a.n * CELL;
}

hipError_t cluster_gm_kernel_info(const LaunchCfg &cfg, const SweepArgs &a, int *lds_bytes, int *blocks_per_cu,
                                  const char **name) {
  (void)a;
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)pick(cfg), 64, 0);
  if (e != hipSuccess) return e;
  if (lds_bytes) *lds_bytes = 0;
  if (blocks_per_cu) *blocks_per_cu = nb;
  if (name) *name = "cluster_kernel<double, state in memory>";
  return hipSuccess;
}

hipError_t launch_cluster_gm(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s, const CaseConst *cases,
                             int *queue, unsigned grid, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(queue + 1, 0, sizeof(int) * (sweep_queue_ints(a) - 1), stream);   // queue[0]: sticky error word
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(pick(cfg), dim3(grid), dim3(64), 0, stream, a, s, cases, cfg.umbrella, queue);
  return hipGetLastError();
}

}  // namespace pstat
