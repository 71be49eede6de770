// pstat_cluster_gm.hip -- the step of mcmc_clustering_eap_chain.jl:268-311 with the chain state in DEVICE MEMORY: the f64 kernel
// of that main (every chain length), and its f32 sibling for ensembles LDS cannot seat (f64_state_global(), pstat_kernels.hip).
//
// Same step, stream contract and results as cluster_kernel<R> of pstat_cluster.hip (one chain per lane, the
// persistent (block, segment) job queue); what differs is where a chain lives while a segment runs and what a cell holds.
// The text below describes the f64 instantiation; f32 keeps the same five values per cell in turns (20 bytes), its sincos
// are two instructions (v_sin/v_cos) and its running totals are re-derived at every segment start like the LDS kernel's.
//
//   * An f64 (theta, phi) cell is 16 bytes, so LDS seats 160 KiB / (16 n) chains per CU: 102 at n = 100, 51 at n = 200 --
//     a quarter (an eighth) of the 256 lanes of a CU's four SIMDs.  Here the cells live in DevState::work, laid out
//     CHAIN-CONTIGUOUS, [chain block][lane][monomer], and every SIMD carries a full wave.  A proposal touches the moved
//     monomer, its two neighbours and the monomers the cluster grows over -- one contiguous run of the chain, i.e. three
//     or four 128-byte lines per proposal whatever the lane's random monomer index is.
//   * A cell is the reference's own per-monomer cache (inc/eap_chain.jl:22-28: the trigonometry of every monomer is kept
//     beside its angles), 40 bytes: [n_x, n_y | n_z, theta | phi].  The LDS kernel re-derives n-hat from the angles of
//     every row it visits (two sincos per row: ~65 f64 instructions, ~50 rows per wave-step, two thirds of its
//     instruction stream); LDS capacity forbade the cache there, device memory does not.  A reflection (refl_n!,
//     inc/eap_chain.jl:263-265) maps a cached cell exactly: n_z -> -n_z, theta -> clamp(theta + (pi - 2 theta)).  The
//     reference recomputes sin and cos of the reflected angle, which agrees with the mapped values to an ulp or two
//     (theta + (pi - 2 theta) is pi - theta rounded); that is the same class of difference as device sincos vs glibc
//     (see sincos_fast_f64): a link or Metropolis decision compares against a 23-bit uniform, so it moves with
//     probability |delta p| ~ 1e-16 per decision.  theta itself is stored and updated exactly as the reference does.
//     40 bytes and not 48 (with sin(theta)): 65 536 chains x 100 monomers are 262 MB against 315 MB, and the 256 MiB
//     Infinity Cache decides the latency of every access (measured: +14 % at n = 100; a chain length whose working set
//     fits entirely, n <= 72, runs another 10 % faster).
//   * Loads and stores go through a buffer resource and are steered by ADDRESS: an offset past the buffer makes a load
//     return 0 and drops a store, so the rows of lanes that do not grow, and the stores of rejected proposals, cost no
//     traffic and no branch.
//   * The step is LATENCY-bound (a dependent access costs ~1 300 cycles from the Infinity Cache, ~3 000 from HBM, one wave
//     per SIMD), so it is organised to have as few memory phases as possible: a WINDOW of the moved monomer and W rows on
//     either side is requested at once and kept in registers for the whole step -- the single move's bonds, the first W
//     growth rounds, the boundary bonds and the members' new cells of 7 clusters in 8 come out of it; the ends still
//     growing after round XREQ (one in four) request E more rows, which serve the next E rounds the same way; only one
//     end in 2^(W+E) goes on row by row.  An accepted reflection rewrites its members from those registers (no
//     read-modify-write).  Requesting the next step's window a step ahead (with the commit forwarded into it) was built
//     and measured: no gain, the forwarding costs what the latency saved.
// HBM holds the checkpoint layout (DevState::ang, angles only); the working buffer is filled from it when a segment
// starts and spilled to it when it ends, exactly as the LDS variants do.
#include <hip/hip_runtime.h>

#include <type_traits>

// (the three-quadrant tail of the theta sincos pays in the sweep's step; here, where only sin(theta) of the old angle is kept,
// it costs the register allocator 17 moves per step: the general tail stays)
#define PSTAT_THETA_GENERIC_TAIL 1
#include "pstat_cluster_common.h"
#include "pstat_device.h"
#include "pstat_math.h"

namespace pstat {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef double v2dd __attribute__((ext_vector_type(2)));
typedef float v2ff __attribute__((ext_vector_type(2)));

#ifndef PSTAT_GM_LDAUX
#define PSTAT_GM_LDAUX 0   // cache-policy bits of the working buffer's loads / stores (sc0 = 1, nt = 2, sc1 = 16)
#endif
#ifndef PSTAT_GM_STAUX
#define PSTAT_GM_STAUX 0
#endif
// bytes per monomer: f64 [n_x, n_y | n_z, theta | phi] = 40; f32 the same five values in turns = 20
template <typename R> constexpr uint32_t cell_bytes() { return sizeof(R) == 8 ? PSTAT_CLUSTER_GM_CELL : 20u; }
#ifndef PSTAT_GM_W
#define PSTAT_GM_W 3
#endif
#ifndef PSTAT_GM_E
#define PSTAT_GM_E 2
#endif
#ifndef PSTAT_GM_XREQ
#define PSTAT_GM_XREQ 1
#endif
constexpr int W = PSTAT_GM_W;        // rows on either side of the moved monomer that every step requests up front
constexpr int E = PSTAT_GM_E;        // further rows, requested after growth round XREQ by the ends still growing then
constexpr int XREQ = PSTAT_GM_XREQ;
#ifndef PSTAT_GM_D
#define PSTAT_GM_D 2
#endif
constexpr int D = PSTAT_GM_D;        // rows in flight per end once a cluster has outgrown those
constexpr int NS = D + 1;            // register sets of the ring: D rows in flight + the set of the previous round's row
#ifndef PSTAT_GM_CAPT
#define PSTAT_GM_CAPT 9
#endif
constexpr int CAPT = PSTAT_GM_CAPT;  // ring rounds that still capture the boundary monomers (a multiple of NS)
static_assert(CAPT % NS == 0, "whole ring trips");
static_assert(XREQ < W, "the outer rows are requested inside the window rounds");
constexpr uint32_t OOB = 0x80000000u;  // past every working buffer (num_records < 2^31, checked by the host): no access

template <typename R, typename G, int CT, int EN>
__device__ __forceinline__ void run_cluster_segment_gm(const SweepArgs &A, const DevState &S, const CaseConst &cc,
                                                       const int umb_on, const int lane, const int64_t c, int64_t step,
                                                       int64_t remaining, const int blk) {
  using AG = Ang<R>;
  using T3 = V3<R>;
  using P2 = typename std::conditional<sizeof(R) == 8, v2dd, v2ff>::type;   // half a row: (n_x, n_y) or (n_z, theta)
  constexpr uint32_t CELL = cell_bytes<R>();
  constexpr uint32_t HB = 2 * sizeof(R);                                      // bytes of such a half
  const int lanes = A.lanes;
  const int64_t C = S.C;
  const int n = (int)A.n;
  constexpr R PI = AG::theta_max;

  const R Fz = (R)cc.Fz, Fx = (R)cc.Fx, b = (R)cc.b, kT = (R)cc.kT;
  const R a_or_mu = (CT == PSTAT_DIELECTRIC) ? (R)((cc.K1 - cc.K2) * cc.E0) : (R)cc.mu;
  const R k2e = (R)(cc.K2 * cc.E0);
  const R mhalfE0 = (R)(-0.5 * cc.E0);
  const R hb = (R)(-cc.b / 2);
  const R khalf = (R)(cc.kappa / 2), psi0 = (R)cc.psi0;
  const R cprob = (R)cc.cluster_prob;
  const R nbeta_log2e = (R)(-1.4426950408889634 / cc.kT);
  (void)hb; (void)nbeta_log2e; (void)kT;

  // ---- the wave's working buffer: [lane][monomer] cells of this chain block
  const uint32_t chain_bytes = (uint32_t)n * CELL;
  unsigned char *const wbase = reinterpret_cast<unsigned char *>(S.work) + (size_t)blk * (size_t)lanes * chain_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wbase, 0, (int)((uint32_t)lanes * chain_bytes), 0x00020000);
  const uint32_t lb = (uint32_t)lane * chain_bytes;
  auto ld = [&](const uint32_t off) __attribute__((always_inline)) -> P2 {     // half a row
    if constexpr (sizeof(R) == 8) return __builtin_bit_cast(P2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, PSTAT_GM_LDAUX));
    else return __builtin_bit_cast(P2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, PSTAT_GM_LDAUX));
  };
  auto st = [&](const uint32_t off, const R x, const R y) __attribute__((always_inline)) {
    const P2 v = {x, y};
    if constexpr (sizeof(R) == 8) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), rsrc, off, 0, PSTAT_GM_STAUX);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), rsrc, off, 0, PSTAT_GM_STAUX);
  };
  auto ld8 = [&](const uint32_t off) __attribute__((always_inline)) -> R {      // phi
    if constexpr (sizeof(R) == 8) return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, PSTAT_GM_LDAUX));
    else return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, PSTAT_GM_LDAUX));
  };
  auto st8 = [&](const uint32_t off, const R x) __attribute__((always_inline)) {
    if constexpr (sizeof(R) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, x), rsrc, off, 0, PSTAT_GM_STAUX);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, x), rsrc, off, 0, PSTAT_GM_STAUX);
  };
  auto row_off = [&](const int row) __attribute__((always_inline)) -> uint32_t { return lb + (uint32_t)row * CELL; };

  G g;
  g.load(S.rng + c, C);
  double phistep = S.stepsz[0 * C + c], thstep = S.stepsz[1 * C + c];    // radians, f64, like the reference (adaptation)
  R phs = (R)(phistep / AG::unit), ths = (R)(thstep / AG::unit);          // in the unit the angles are stored in
  int64_t nacc_off = S.win[0 * C + c], natt_off = S.win[1 * C + c];
  int nacc_seg = 0, steps_seg = 0;
  int nnan_seg = 0;   // proposals with a non-finite energy difference (Ising pair terms at r -> 0)
  R Orx = (R)S.obs[OBS_R1 * C + c], Ory = (R)S.obs[OBS_R2 * C + c], Orz = (R)S.obs[OBS_R3 * C + c];
  R Opx = (R)S.obs[OBS_P1 * C + c], Opy = (R)S.obs[OBS_P2 * C + c], Opz = (R)S.obs[OBS_P3 * C + c];
  R OU = (R)S.obs[OBS_U * C + c];
  R usum = (R)S.obs[OBS_USUM * C + c];      // sum of u_i INCLUDING the bending terms (eap_chain.jl:53-58)
  R c2sum = (R)S.obs[OBS_C2 * C + c], psisum = (R)S.obs[OBS_PSI * C + c];
  // log(alpha) of the last accepted proposal of this mcmc() call (inc/acceptance.jl:33-36), kept as alpha itself
  // (`lag_alpha`, lag_pending) until a literal evaluation or the spill needs the logarithm
  R lag = (R)S.lag[c], lag_alpha = 1;
  bool lag_pending = false;
  const bool umb = umb_on != 0;
  const R wscale = umb ? (R)((0.2 + 0.8 * exp(-(cc.Fx * cc.Fx + cc.Fz * cc.Fz) / cc.kT)) / cc.kT) : (R)0;
  R uref = umb ? (R)S.uref[c] : (R)0;
  bool regauged = false;
  double wnorm = umb ? S.wnorm[c] : 0.0;
  // (the f64 running sums stay in HBM: a block of FLUSH steps is added to them at a time -- 32 registers that the
  // step's window needs more)
  const R inv_nm1 = n > 1 ? (R)(1.0 / (double)(n - 1)) : (R)0;
  const R ninv_kT = (R)(-1.0 / cc.kT);
  (void)ninv_kT;

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
  auto refl_theta = [&](const R th) __attribute__((always_inline)) -> R {
    if constexpr (sizeof(R) == 8) return fmin(PI, fmax((R)0, th + (PI - 2 * th)));    // the reference's arithmetic
    else return PI - th;
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

  // ---- fill: angles from the checkpoint planes (coalesced over the lanes), the cached trigonometry derived from them
  // with the same functions the step uses for a moved monomer.  f32 carries r, p, U, sum(u), sum(psi), sum cos^2 as running
  // totals of accepted differences, whose rounding errors random-walk: they are re-derived here, at every segment start
  // (<= 32 768 steps apart in f32), from the angles being filled in, as the LDS kernel does.
  {
    const R *gth = (const R *)S.ang, *gph = (const R *)S.ang + (int64_t)n * C;
    double tx = 0, ty = 0, tz = 0, qx = 0, qy = 0, qz = 0, tu = 0, tp = 0, tpsi = 0, tc2 = 0;
    T3 pn{0, 0, 1}, pm{0, 0, 0};
    (void)pn; (void)pm;
#pragma unroll 2
    for (int i = 0; i < n; ++i) {
      const R th = gth[(int64_t)i * C + c], ph = gph[(int64_t)i * C + c];
      R s, co, sp, cp;
      AG::sc_theta(th, &s, &co);
      AG::sc_phi(ph, &sp, &cp);
      const uint32_t o = row_off(i);
      st(o, cp * s, sp * s);
      st(o + HB, co, th);
      st8(o + 2 * HB, ph);
      if constexpr (sizeof(R) == 4) {
        const T3 ni{cp * s, sp * s, co}, mi = mu_of(ni);
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
    }
    if constexpr (sizeof(R) == 4) {
      const double bd = (double)b;
      Orx = (R)(bd * tx); Ory = (R)(bd * ty); Orz = (R)(bd * tz);
      Opx = (R)qx; Opy = (R)qy; Opz = (R)qz;
      usum = (R)tu; psisum = (R)tpsi; c2sum = (R)tc2;
      OU = (R)(tu + tp - ((double)Fx * bd * tx + (double)Fz * bd * tz));
    }
  }
  auto up_off = [&](const int row, const bool want) __attribute__((always_inline)) -> uint32_t {
    return (want && row <= n - 1) ? row_off(row) : OOB;
  };
  auto dn_off = [&](const int row, const bool want) __attribute__((always_inline)) -> uint32_t {
    return (want && row >= 0) ? row_off(row) : OOB;
  };

  // ---- the draws whose place in the stream is fixed (mcmc_clustering_eap_chain.jl:269-272 and the skip draw of
  // cluster_flip!, inc/eap_chain.jl:276) and the WINDOW they address
  struct Draw { int idx; uint32_t w0, wphi, wth; bool flipped; };
  struct Win { P2 c0a, c0b, c0c, ua[W + 1], ub[W + 1], da[W + 1], db[W + 1]; };
  auto draw_next = [&]() __attribute__((always_inline)) -> Draw {
    Draw d;
    d.w0 = g.next();
    d.idx = (int)__umulhi(d.w0, (uint32_t)n);
    d.wphi = g.next(); d.wth = g.next();
    d.flipped = !(u01<R>(g.next()) <= cprob);
    return d;
  };
  auto request = [&](const Draw &d, Win &w) __attribute__((always_inline)) {
    const uint32_t off0 = row_off(d.idx);
    w.c0a = ld(off0); w.c0b = ld(off0 + HB);
    w.c0c = P2{ld8(off0 + 2 * HB), (R)0};
    const uint32_t offR = d.idx + 1 < n ? off0 + CELL : off0, offL = d.idx > 0 ? off0 - CELL : off0;
    w.ua[1] = ld(offR); w.ub[1] = ld(offR + HB);
    w.da[1] = ld(offL); w.db[1] = ld(offL + HB);
#pragma unroll
    for (int k = 2; k <= W; ++k) {
      uint32_t o = up_off(d.idx + k, d.flipped);
      w.ua[k] = ld(o); w.ub[k] = ld(o + HB);
      o = dn_off(d.idx - k, d.flipped);
      w.da[k] = ld(o); w.db[k] = ld(o + HB);
    }
  };
#ifdef PSTAT_GM_PROF   // (timing experiment: wave clocks of the step's phases, printed by one wave per launch)
  uint64_t pf_t[6] = {0, 0, 0, 0, 0, 0};
  auto pf_now = []() __attribute__((always_inline)) -> uint64_t { return __builtin_readcyclecounter(); };
#define PF_MARK(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const uint64_t t_ = pf_now(); pf_t[i] += t_ - pf_last; pf_last = t_; } while (0)
#else
#define PF_MARK(i) do {} while (0)
#endif
  while (left > 0) {
    int chunk = left < FLUSH ? left : FLUSH;
    if (A.adaptive && to_adj < chunk) chunk = (int)to_adj;
    R a1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, a2[7] = {0, 0, 0, 0, 0, 0, 0};
    R accw = 0;

    for (int s = 0; s < chunk; ++s) {
      // ---- every draw whose place in the stream is fixed: mcmc_clustering_eap_chain.jl:269-272 and the skip draw of
      // cluster_flip! (inc/eap_chain.jl:276)
      const Draw d = draw_next();
      Win w;
      request(d, w);
#ifdef PSTAT_GM_PROF
      uint64_t pf_last = pf_now();
#endif
      const int idx = d.idx;
      const uint32_t wphi = d.wphi, wth = d.wth;
      const bool flipped = d.flipped;
      const bool hasL = idx > 0, hasR = idx + 1 < n;
      // ---- the WINDOW: the moved monomer and W rows on either side, requested at once and kept in registers for the
      // whole step.  The single move's bonds, the first W growth rounds, the boundary bonds and the members' cells of
      // 7 clusters in 8 come out of it.  At a chain end the missing neighbour's slot re-reads the monomer itself (its
      // bond is masked out); rows past the chain, and the outer rows of lanes that grow no cluster, are steered off
      // the buffer.  Rows W + 1 .. W + E (`x*`) follow after growth round XREQ, asked for by the ends still growing
      // then (one in four), and serve rounds W .. W + E - 1 the same way.
      const uint32_t off0 = row_off(idx);
      const P2 c0a = w.c0a, c0b = w.c0b, c0c = w.c0c;
      P2 ua[W + E + 1], ub[W + E + 1], da[W + E + 1], db[W + E + 1];   // [k]: row idx + k / idx - k (a = n_x, n_y; b = n_z, theta)
#pragma unroll
      for (int k = 1; k <= W; ++k) { ua[k] = w.ua[k]; ub[k] = w.ub[k]; da[k] = w.da[k]; db[k] = w.db[k]; }
#pragma unroll
      for (int k = W + 1; k <= W + E; ++k) { ua[k] = ub[k] = da[k] = db[k] = P2{0, 0}; }

      PF_MARK(0);   // window requested and landed (behind the preceding commit's stores)
      // ---- the single-monomer part
      const R th0 = c0b.y, ph0 = c0c.x;
      R st0 = c0c.y;
      { R ct0_; AG::sc_theta(th0, &st0, &ct0_); }   // (the cell carries no sin(theta): 8 bytes per monomer matter, see the header)
      const T3 n0{c0a.x, c0a.y, c0b.x};
      const R ct0 = n0.z;
      // (the trajectory itself: each product rounded before its sum, as the oracle and Julia round them -- through an opaque
      // register, so that no build flag can fuse them; cf. run_segment)
      auto rounded = [](R v) __attribute__((always_inline)) -> R { asm volatile("" : "+v"(v)); return v; };
      const R ph1 = AG::wrap(ph0 + rounded(phs * sym11<R>(wphi)));
      const R th1 = fmin(PI, fmax((R)0, th0 + rounded(ths * sym11<R>(wth))));
      R st1, ct1, sp1, cp1;
      AG::sc_theta(th1, &st1, &ct1);
      AG::sc_phi(ph1, &sp1, &cp1);
      const T3 n1{cp1 * st1, sp1 * st1, ct1};
      const T3 m0 = mu_of(n0), m1 = mu_of(n1);
      auto nhat = [](const P2 &a, const P2 &b_) __attribute__((always_inline)) -> T3 { return T3{a.x, a.y, b_.x}; };
      const T3 nL = nhat(da[1], db[1]), nR = nhat(ua[1], ub[1]);
      const T3 mL = mu_of(nL), mR = mu_of(nR);

      // ---- cluster_flip!(trial, idx), inc/eap_chain.jl:269-333 (see pstat_cluster.hip for the scheme: both ends grow
      // in one sequence of uniform rounds, each with its own draw while that end still grows)
      R alpha = 1;
      bool edge = false;
      int upper = idx, lower = idx;
      R drz_flip = 0, du_flip = 0, dpair_flip = 0, dpsi_flip = 0;
      T3 dp_flip{0, 0, 0};
      R snz = n1.z;                 // sums over the members (the moved monomer enters as proposed)
      T3 sm = m1;
      R upper_p = 0, lower_p = 0, new_upper_p = 0, new_lower_p = 0;
      T3 cu = n1, nu = nR, cl = n1, nl = nL;      // the monomers of the two boundary bonds
      const bool any_flip = __builtin_amdgcn_ballot_w64(flipped) != 0;   // wave-uniform
      if (any_flip) {
        edge = flipped && is_edge(th1);
        T3 Au = n1, Al = n1;
        bool gu = flipped && hasR, gl = flipped && hasL;
        // one end's link test of one round: A = the cluster's outermost member, B = the candidate beyond it
        auto half_round = [&](const bool up, bool &gx, const T3 &A_, const T3 &B_, const bool eB, R &xp, int &ext)
            __attribute__((always_inline)) {
          const R p = (1 + dot3(A_, B_)) / 2;
          G g2 = g;
          const bool acc = gx && (u01<R>(g2.next()) <= p);
          g.pick(gx, g2);
          xp = gx ? p : xp;
          ext += acc ? (up ? 1 : -1) : 0;
          edge = edge || (acc && eB);
          member(acc, B_, snz, sm);
          gx = acc && (up ? ext < n - 1 : ext > 0);
        };
        // an end still growing when a round begins has (A, B) as its boundary bond unless it grows on: overwritten every
        // round it is alive, the last one stays
        auto capture = [&](const bool gx, const T3 &A_, const T3 &B_, T3 &c_, T3 &n_) __attribute__((always_inline)) {
          c_.x = gx ? A_.x : c_.x; c_.y = gx ? A_.y : c_.y; c_.z = gx ? A_.z : c_.z;
          n_.x = gx ? B_.x : n_.x; n_.y = gx ? B_.y : n_.y; n_.z = gx ? B_.z : n_.z;
        };
#pragma unroll
        for (int t = 0; t < W; ++t) {      // rounds inside the window: link (idx + t, idx + t + 1), then (idx - t, idx - t - 1)
          const T3 Bu = nhat(ua[t + 1], ub[t + 1]), Bl = nhat(da[t + 1], db[t + 1]);
          half_round(true, gu, Au, Bu, is_edge(ub[t + 1].y) && (t > 0 || hasR), upper_p, upper);
          Au = Bu;
          half_round(false, gl, Al, Bl, is_edge(db[t + 1].y) && (t > 0 || hasL), lower_p, lower);
          Al = Bl;
          if (t == XREQ) {
#pragma unroll
            for (int k = W + 1; k <= W + E; ++k) {
              uint32_t q = up_off(idx + k, gu);
              ua[k] = ld(q); ub[k] = ld(q + HB);
              q = dn_off(idx - k, gl);
              da[k] = ld(q); db[k] = ld(q + HB);
            }
          }
        }
        // the boundary monomers of an end that stopped inside the window (ku accepted links: rows ku and ku + 1)
        {
          const int ku = upper - idx, kl = idx - lower;
#pragma unroll
          for (int k = 1; k < W; ++k) {
            const T3 a_ = nhat(ua[k], ub[k]), a1_ = nhat(ua[k + 1], ub[k + 1]), b_ = nhat(da[k], db[k]), b1_ = nhat(da[k + 1], db[k + 1]);
            capture(ku >= k, a_, a1_, cu, nu);
            capture(kl >= k, b_, b1_, cl, nl);
          }
        }
        if (__builtin_amdgcn_ballot_w64(gu || gl) != 0) {
          // (the ring of the rounds beyond W + E - 1, see below: rows W + E + 1 .. W + E + D, asked for by the ends that enter round W)
          P2 ra[NS], rb[NS], sa[NS], sb[NS];
          int ring = W + E + 1;      // row offset of the next ring round's candidate row
#pragma unroll
          for (int k = 0; k < D; ++k) {
            uint32_t q = up_off(idx + ring + k, gu);
            ra[k] = ld(q); rb[k] = ld(q + HB);
            q = dn_off(idx - ring - k, gl);
            sa[k] = ld(q); sb[k] = ld(q + HB);
          }
          // ---- rounds W .. W + E - 1, out of the rows requested after round XREQ
#pragma unroll
          for (int t = W; t < W + E; ++t) {
            const T3 Bu = nhat(ua[t + 1], ub[t + 1]), Bl = nhat(da[t + 1], db[t + 1]);
            capture(gu, Au, Bu, cu, nu);
            half_round(true, gu, Au, Bu, is_edge(ub[t + 1].y), upper_p, upper);
            Au = Bu;
            capture(gl, Al, Bl, cl, nl);
            half_round(false, gl, Al, Bl, is_edge(db[t + 1].y), lower_p, lower);
            Al = Bl;
          }
          // ---- beyond: one end in 2^(W + E) gets here -- rarely in a disordered chain, every step in an aligned one
          // (cold or stiff: every link joins, clusters run to the chain ends).  A ring of NS = D + 1 register sets: each row is
          // requested D rounds before its round (the first D when these ends entered round W), and the set of the round
          // before still holds this round's outermost member.
          // The boundary monomers are captured round by round for the first CAPT rounds (a disordered chain's tail is a
          // round or two: no further memory phase); an end that grows on drops the capture -- a fifth of the round's
          // instructions, for tens of rounds -- and reads its two boundary monomers back when it has stopped.
          // A round's outermost member A is the candidate B of the round before: it is read out of THAT round's register set
          // (no copy: a lone wave pays four cycles for every move), which is then free and requests the row D rounds out.
          // The first ring round finds the outermost member of the rounds so far in the spare set.
          ra[D] = P2{Au.x, Au.y}; rb[D] = P2{Au.z, (R)0};
          sa[D] = P2{Al.x, Al.y}; sb[D] = P2{Al.z, (R)0};
          auto ring_round = [&](const int k, const bool cap) __attribute__((always_inline)) {
            const int ka = (k + NS - 1) % NS;
            const T3 Au_ = nhat(ra[ka], rb[ka]), Al_ = nhat(sa[ka], sb[ka]);
            const T3 Bu = nhat(ra[k], rb[k]), Bl = nhat(sa[k], sb[k]);
            const bool eBu = is_edge(rb[k].y), eBl = is_edge(sb[k].y);
            if (cap) capture(gu, Au_, Bu, cu, nu);
            half_round(true, gu, Au_, Bu, eBu, upper_p, upper);
            if (cap) capture(gl, Al_, Bl, cl, nl);
            half_round(false, gl, Al_, Bl, eBl, lower_p, lower);
            uint32_t q = up_off(idx + ring + D, gu);      // set ka is free: its next row is D rounds out
            ra[ka] = ld(q); rb[ka] = ld(q + HB);
            q = dn_off(idx - ring - D, gl);
            sa[ka] = ld(q); sb[ka] = ld(q + HB);
            ring += 1;
          };
          for (int trip = 0; trip < CAPT / NS && (gu || gl); ++trip) {
#pragma unroll
            for (int k = 0; k < NS; ++k) ring_round(k, true);
          }
          const bool longu = gu, longl = gl;     // still growing after the captured rounds
          if (__builtin_amdgcn_ballot_w64(longu || longl) != 0) {
            while (gu || gl) {
#pragma unroll
              for (int k = 0; k < NS; ++k) ring_round(k, false);
            }
            uint32_t o = longu ? row_off(upper) : OOB;
            P2 ta = ld(o), tb = ld(o + HB);
            cu.x = longu ? ta.x : cu.x; cu.y = longu ? ta.y : cu.y; cu.z = longu ? tb.x : cu.z;
            o = (longu && upper < n - 1) ? row_off(upper + 1) : OOB; ta = ld(o); tb = ld(o + HB);
            nu.x = longu ? ta.x : nu.x; nu.y = longu ? ta.y : nu.y; nu.z = longu ? tb.x : nu.z;
            o = longl ? row_off(lower) : OOB; ta = ld(o); tb = ld(o + HB);
            cl.x = longl ? ta.x : cl.x; cl.y = longl ? ta.y : cl.y; cl.z = longl ? tb.x : cl.z;
            o = (longl && lower > 0) ? row_off(lower - 1) : OOB; ta = ld(o); tb = ld(o + HB);
            nl.x = longl ? ta.x : nl.x; nl.y = longl ? ta.y : nl.y; nl.z = longl ? tb.x : nl.z;
          }
        }
        upper_p = upper >= n - 1 ? (R)0 : upper_p;   // ran into the chain end: no link to test, :282-284
        lower_p = lower <= 0 ? (R)0 : lower_p;       // :299-301
      }
      const uint32_t weps = g.next();   // the acceptance draw comes after the cluster's draws
      PF_MARK(1);   // single-move trigonometry + growth

      // ---- the single move's two bonds, before and after
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
        // the two boundary bonds, before and after the reflection (:318-326; the moved monomer enters as proposed)
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

      // ---- energy difference of the whole proposal, inc/energy.jl:7-23
      const R drx = b * (n1.x - n0.x), dry = b * (n1.y - n0.y), drz = b * (n1.z - n0.z) + drz_flip;
      const R dus = du_field + dbend + du_flip;        // change of sum(u), bending included
      const R dU = dus + (dpair + dpair_flip) - (Fx * drx + Fz * drz);

      // ---- Metropolis-Hastings, inc/acceptance.jl:29-39.  The f32 filter of pstat_math.h decides all but ~1e-5 of the
      // draws, the literal expression the rest.  The cached log(alpha) of the last acceptance enters the filter as the
      // factor alpha itself (no logarithm on the common path).
      const R dw = umb ? dus * wscale : (R)0;
      bool ok;
      if constexpr (sizeof(R) == 8) {
        ok = metropolis_filter(dU * ninv_kT + (dw - (lag_pending ? (R)0 : lag)), st1 * alpha,
                               st0 * (lag_pending ? lag_alpha : (R)1), weps, [&]() -> bool {
          const R lg = lag_pending ? log_r(lag_alpha) : lag;
          const R delta = -dU / kT + log_r(st1 / st0) + dw + log_r(alpha) - lg;
          const R eps = (R)eps_uniform(A.wide_eps != 0, weps, d.w0, wphi, wth);
          return (delta >= 0) || (eps < exp_r(delta));
        });
      } else {   // f32: the same test with the logarithm folded away, (1 + u) sin0 < sin1 e alpha + sin0 (cf. pstat_cluster.hip)
        const R lg = lag_pending ? log_r(lag_alpha) : lag;
        const R e = __builtin_amdgcn_exp2f((R)1.44269504f * (dw - lg) + dU * nbeta_log2e) * alpha;
        ok = bits12(weps) * st0 < fma_r(st1, e, st0);
      }
      ok = ok && !edge;
      if constexpr (EN == PSTAT_ISING) nnan_seg += not_finite(dU) ? 1 : 0;

      PF_MARK(2);   // bonds + Metropolis
      // ---- commit: stores steered by address (a rejected proposal stores nothing)
      {
        const bool okf = ok && flipped;
#ifdef PSTAT_GM_NOSTORE   // (timing experiment: nothing is ever committed to memory)
        const uint32_t os = OOB;
#else
        const uint32_t os = ok ? off0 : OOB;
#endif
        st(os, n1.x, n1.y);
        st(os + HB, flipped ? -n1.z : n1.z, flipped ? refl_theta(th1) : th1);
        st8(os + 2 * HB, ph1);
        if (any_flip) {
          // the members inside the window come out of its registers: n_z -> -n_z, theta reflected
          const int ku = upper - idx, kl = idx - lower;
#pragma unroll
          for (int k = 1; k <= W; ++k) {
            st((okf && k <= ku) ? off0 + (uint32_t)k * CELL + HB : OOB, -ub[k].x, refl_theta(ub[k].y));
            st((okf && k <= kl) ? off0 - (uint32_t)k * CELL + HB : OOB, -db[k].x, refl_theta(db[k].y));
          }
          if (__builtin_amdgcn_ballot_w64(okf && (ku > W || kl > W)) != 0) {
#pragma unroll
            for (int k = W + 1; k <= W + E; ++k) {
              st((okf && k <= ku) ? off0 + (uint32_t)k * CELL + HB : OOB, -ub[k].x, refl_theta(ub[k].y));
              st((okf && k <= kl) ? off0 - (uint32_t)k * CELL + HB : OOB, -db[k].x, refl_theta(db[k].y));
            }
            // members beyond the requested rows: read-modify-write, four rows of either side per pass
            for (int i = W + E + 1; __builtin_amdgcn_ballot_w64(okf && (i <= ku || i <= kl)) != 0; i += 4) {
              P2 v[8];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                v[j] = ld((okf && i + j <= ku) ? off0 + (uint32_t)(i + j) * CELL + HB : OOB);
                v[4 + j] = ld((okf && i + j <= kl) ? off0 - (uint32_t)(i + j) * CELL + HB : OOB);
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                st((okf && i + j <= ku) ? off0 + (uint32_t)(i + j) * CELL + HB : OOB, -v[j].x, refl_theta(v[j].y));
                st((okf && i + j <= kl) ? off0 - (uint32_t)(i + j) * CELL + HB : OOB, -v[4 + j].x, refl_theta(v[4 + j].y));
              }
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

      PF_MARK(3);   // commit, its stores acknowledged
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
            for (int q = 0; q < NSUMS; ++q) S.sums[(int64_t)q * C + c] *= f;
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
      PF_MARK(4);   // record
    }

    {
      double *const sm_ = S.sums + c;
      auto add = [&](const int q, const R v) __attribute__((always_inline)) { sm_[(int64_t)q * C] += (double)v; };
      add(S_R1, a1[0]); add(S_R2, a1[1]); add(S_R3, a1[2]);
      add(S_P1, a1[3]); add(S_P2, a1[4]); add(S_P3, a1[5]);
      add(S_U, a1[6]); add(S_C2, a1[7]); add(S_PSI, a1[8]);
      add(S_R1SQ, a2[0]); add(S_R2SQ, a2[1]); add(S_R3SQ, a2[2]);
      add(S_P1SQ, a2[3]); add(S_P2SQ, a2[4]); add(S_P3SQ, a2[5]);
      add(S_USQ, a2[6]);
    }
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
        if (ratio > A.adj_ub && phistep != K<double>::pi && thstep != K<double>::half_pi) {
          nacc_off = -nacc_seg; natt_off = -steps_seg;
          phistep = fmin(K<double>::pi, phistep * A.adj_scale);
          thstep = fmin(K<double>::half_pi, thstep * A.adj_scale);
        } else if (ratio < A.adj_lb) {
          nacc_off = -nacc_seg; natt_off = -steps_seg;
          phistep /= A.adj_scale;
          thstep /= A.adj_scale;
        }
        phs = (R)(phistep / AG::unit); ths = (R)(thstep / AG::unit);
      }
    }
  }

#ifdef PSTAT_GM_PROF
  if (blk == 3 && lane == 0 && remaining > 100)
    printf("gm prof (cycles per step): window %.0f  move+growth %.0f  bonds+accept %.0f  commit %.0f  record %.0f\n",
           (double)pf_t[0] / remaining, (double)pf_t[1] / remaining, (double)pf_t[2] / remaining, (double)pf_t[3] / remaining,
           (double)pf_t[4] / remaining);
#endif
  // ---- spill: the angles back to the checkpoint planes
  {
    R *gth = (R *)S.ang, *gph = (R *)S.ang + (int64_t)n * C;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      const uint32_t o = row_off(i);
      const P2 b1 = ld(o + HB);
      const P2 b2 = P2{ld8(o + 2 * HB), (R)0};
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
  if (regauged) S.uref[c] = (double)uref;
}

#ifndef PSTAT_GM_WAVES
#define PSTAT_GM_WAVES 1   // waves per SIMD the register allocator is asked for: f64 needs the whole file (at 2: ~250 spilled registers)
#endif
#ifndef PSTAT_GM_WAVES_F32
#define PSTAT_GM_WAVES_F32 2   // f32 fits 256 registers with ~20 spilled ones: a second wave per SIMD hides latency for ensembles >= 131 072 chains (+10...19 %)
#endif
template <typename R, typename G, int CT, int EN, bool PACKED>
__global__ __launch_bounds__(64, sizeof(R) == 4 ? PSTAT_GM_WAVES_F32 : PSTAT_GM_WAVES) void cluster_gm_kernel(SweepArgs A, DevState S, const CaseConst *__restrict__ cases,
                                                        int umbrella, int *__restrict__ queue) {
  const int lane = threadIdx.x;
  run_job_queue<PACKED>(A, queue, lane, [&](const CaseConst &cc, int64_t chain, int64_t first, int64_t len, int blk) {
    run_cluster_segment_gm<R, G, CT, EN>(A, S, cc, umbrella, lane, chain, first, len, blk);
  }, cases);
}

using ClusterFn = void (*)(SweepArgs, DevState, const CaseConst *, int, int *);

template <typename R, typename G, bool PACKED>
ClusterFn pick_ct_en(const LaunchCfg &cfg) {
  const bool ising = cfg.energy_type == PSTAT_ISING;
  if (cfg.chain_type == PSTAT_DIELECTRIC)
    return ising ? cluster_gm_kernel<R, G, PSTAT_DIELECTRIC, PSTAT_ISING, PACKED> : cluster_gm_kernel<R, G, PSTAT_DIELECTRIC, PSTAT_NONINTERACTING, PACKED>;
  return ising ? cluster_gm_kernel<R, G, PSTAT_POLAR, PSTAT_ISING, PACKED> : cluster_gm_kernel<R, G, PSTAT_POLAR, PSTAT_NONINTERACTING, PACKED>;
}

template <bool PACKED>
ClusterFn pick_p(const LaunchCfg &cfg) {
  const bool xo = cfg.rng == PSTAT_RNG_XOSHIRO128PP;
  if (cfg.precision == PSTAT_F64) return xo ? pick_ct_en<double, Xoshiro128pp, PACKED>(cfg) : pick_ct_en<double, Mwc64x, PACKED>(cfg);
  return xo ? pick_ct_en<float, Xoshiro128pp, PACKED>(cfg) : pick_ct_en<float, Mwc64x, PACKED>(cfg);
}
ClusterFn pick(const LaunchCfg &cfg) { return cfg.packed ? pick_p<true>(cfg) : pick_p<false>(cfg); }

}  // namespace

size_t cluster_gm_work_bytes(const LaunchCfg &cfg, const SweepArgs &a) {
  return (size_t)a.nblocks * (size_t)a.lanes * (size_t)a.n *
         (cfg.precision == PSTAT_F64 ? cell_bytes<double>() : cell_bytes<float>());
}

hipError_t cluster_gm_kernel_info(const LaunchCfg &cfg, const SweepArgs &a, int *lds_bytes, int *blocks_per_cu,
                                  const char **name) {
  (void)a;
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)pick(cfg), 64, 0);
  if (e != hipSuccess) return e;
  if (lds_bytes) *lds_bytes = 0;
  if (blocks_per_cu) *blocks_per_cu = nb;
  if (name) *name = cfg.precision == PSTAT_F64 ? (cfg.packed ? "cluster_kernel<double, state in memory> [packed cases]" : "cluster_kernel<double, state in memory>")
                                                : (cfg.packed ? "cluster_kernel<float, state in memory> [packed cases]" : "cluster_kernel<float, state in memory>");
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
