// pstat_cluster_cw.hip -- the step of mcmc_clustering_eap_chain.jl:268-311 (non-interacting and Ising energies, f64, MWC64X)
// with ONE CHAIN PER WAVEFRONT: the kernel of small ensembles and of sweeps of many small cases, i.e. of the reference's own
// phase scans, which run one chain per case (run/K1_E0-kT-phase.jl:19-45: 546 grid points x 5 runs).
//
// Why a second mapping.  The chain-per-lane kernels (pstat_cluster_gm.hip) process 64 chains per instruction, but a step costs
// them a fixed 4-5 us (dependent memory phases at one wave per SIMD) and, on an aligned chain (low kT, strong field: clusters
// run over tens of monomers), 25-33 us -- a wave grows its clusters link by link and runs as long as the longest one among its
// lanes, and a launch as long as its slowest wave.  A sweep of a few thousand chains over millions of steps is therefore bound
// by that SEQUENTIAL step time with most of the chip idle.  Here a step is ~400 vector instructions of one wave whatever the
// cluster length, four waves share a SIMD, and the chip steps 2 730 chains in 2.8 us (chain per lane: 29 us):
//   * the chain lives in LDS as cells (n_x, n_y | n_z, theta | phi) -- the reference's per-monomer cache, inc/eap_chain.jl:22-28
//     -- and lane l owns monomers / links l, l + 64, ... (M = 1, 2 or 4 of them: n <= 256);
//   * cluster_flip! (inc/eap_chain.jl:269-333) grows in ONE pass: every lane forms the link probability of its own links and
//     tests it against ITS draw of the stream.  The stream contract (oracle/eap_oracle.c cluster_flip) hands out the draws
//     round by round, upper link then lower link while that end still grows, so a link's draw sits at a position that depends
//     only on the link's distance from the moved monomer and on WHEN THE OTHER END STOPPED: position 2r (upper) / 2r + 1 (lower)
//     in round r while both grow, consecutive positions for the survivor afterwards.  Two passes settle it: the first assumes
//     both ends alive and finds the end that stops first (its answer is exact: until then the assumption holds), the second
//     re-reads the survivor's draws at their shifted positions.  First failures are found with ballots;
//   * the draws themselves are produced 64 (M + 1) at a time: MWC64X is the LCG s <- A s mod M (pstat_device.h), lane l keeps
//     the state (M + 1) l outputs down the chain's stream, emits M + 1 words per step into LDS, and after the step every lane
//     skips ahead by the number of words the step consumed -- one 64 x 64 -> 128-bit product with A^(d - 2) mod M from a
//     constant table and two reduction steps T -> hi(T) + A lo32(T) (= T / 2^32 mod M, since A 2^32 = M + 1);
//   * everything a step computes once per chain is spread over lanes instead of being repeated in all of them: lanes 0-2 take
//     the three sincos, lanes 0-7 of every group of eight the eight bonds of the proposal (bond angle, bending and Ising pair
//     term of (L,0) (L,1) (0,R) (1,R) and of the two boundary bonds before / after the reflection), gathered from LDS by
//     per-lane cell index, and ONE signed tree sum over eight lanes delivers the three differences;
//   * the ten observables of the microstate live across lanes 0-8 of one register: record! is two fused multiply-adds;
//   * member sums (sum n_z; the dipole components a reflection flips only when the proposal is accepted) are DPP tree sums;
//   * what a step needs but the register file should not hold: polynomial coefficients are read from an LDS table where a
//     fused multiply-add wants them (a double constant otherwise costs two vector moves or a parked scalar pair), the object
//     is built with machine LICM off (csrc/Makefile: immediates are re-formed at their use instead of being hoisted into ~90
//     scalar registers that spill the step's own), pointers of the flush / spill sit in LDS, and the literal acceptance
//     test is a function call.  Result: 128 registers, four waves per SIMD (measured, per step of 43 680 chains: 34 us when
//     the allocator is asked for two or three waves, 30 for four, 58 for five -- spills; 33 for four with machine LICM on;
//     the first version, before any of this, 45; tools/ab_cw_waves.sh -> profiles/r04/experiments/ab_cw_waves.txt).
// Results: the same trajectories as the oracle and the chain-per-lane kernels bit for bit (angles, generator state, acceptance
// counts, step sizes; tests/fuzz_cluster_wave.py); running sums differ in their last bits (order of the member sums), like
// every kernel pair here.  Chosen by cluster_chain_wave() below (PSTAT_F64_STATE=wave|lds|global overrides, for tests and
// experiments); xoshiro128++ has no cheap skip-ahead and keeps the chain-per-lane kernels.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#define PSTAT_THETA_GENERIC_TAIL 1   // (one sincos form for all three angles of the step: they share an instruction stream)
#include "pstat_cluster_common.h"
#include "pstat_device.h"
#include "pstat_math.h"
#include "pstat_wave.h"

namespace pstat {

namespace {

// ---- MWC64X skip-ahead.  POW.v[d] = A^(d - 2) mod M (v[0] = A^-2 = 2^64 mod M, v[1] = A^-1 = 2^32): mwc_skip(s, v[d]) = s A^d.
constexpr int NPOW = 64 * 5 + 16;
struct PowTable { uint64_t v[NPOW]; };
constexpr uint64_t cmulmod(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % Mwc64x::M); }
constexpr PowTable make_pow() {
  PowTable t{};
  t.v[0] = (uint64_t)((((unsigned __int128)1) << 64) % Mwc64x::M);
  t.v[1] = 1ull << 32;
  uint64_t p = 1;
  for (int d = 2; d < NPOW; ++d) { t.v[d] = p; p = cmulmod(p, Mwc64x::A); }
  return t;
}
__constant__ PowTable POW = make_pow();

// y g A^2 mod M for canonical y, g < M: the 128-bit product, then twice T -> hi(T) + A lo32(T) (each divides by 2^32 mod M)
__device__ __forceinline__ uint64_t mwc_skip(const uint64_t y, const uint64_t g) {
  constexpr uint64_t A = Mwc64x::A, Mm = Mwc64x::M;
  const uint32_t yl = (uint32_t)y, yh = (uint32_t)(y >> 32), gl = (uint32_t)g, gh = (uint32_t)(g >> 32);
  const uint64_t p00 = (uint64_t)yl * gl;
  const uint64_t p01 = (uint64_t)yl * gh + (p00 >> 32);
  const uint64_t p10 = (uint64_t)yh * gl + (uint32_t)p01;
  const uint64_t p11 = (uint64_t)yh * gh + (p01 >> 32) + (p10 >> 32);      // < M^2 / 2^64 < M
  const uint64_t t = A * (uint64_t)(uint32_t)p00 + (uint32_t)p10;
  const uint64_t Q = p11 + (t >> 32);                                        // < M + A < 2^64
  uint64_t r = A * (uint64_t)(uint32_t)t + Q;                                // true value < 2 M: one carry bit
  if (r < Q || r >= Mm) r -= Mm;
  return r;
}

// A lane's value as seen from its DPP partner (callers keep it out of the operands of a per-lane ?: -- see the bond lanes)
template <int CTRL> __device__ __forceinline__ double dpp_f64(const double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// b in the lanes of the constant mask, a elsewhere (the mask is an immediate of the scalar unit: no compare, no live register)
__device__ __forceinline__ double sel_lanes(const double a, const double b, const uint64_t mask) {
  int lo, hi;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(a)), "v"(__double2loint(b)), "s"(mask));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(a)), "v"(__double2hiint(b)), "s"(mask));
  return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
// sum over the 8 lanes of an aligned group (every lane of the group ends with it)
__device__ __forceinline__ double sum8(double v) {
  v += dpp_f64<DPP_XOR1>(v);
  v += dpp_f64<DPP_XOR2>(v);
  v += dpp_f64<DPP_HALF_MIRROR>(v);
  return v;
}
// sum over the wave, wave-uniform
__device__ __forceinline__ double sum64(double v) {
  v = sum8(v);
  v += dpp_f64<DPP_ROW_MIRROR>(v);
  return (lane_value<double>(v, 0) + lane_value<double>(v, 16)) + (lane_value<double>(v, 32) + lane_value<double>(v, 48));
}

// ---- Polynomial coefficients of the step's trigonometry, read from LDS where they are used.  A double constant cannot be an
// operand of an f64 instruction: it is built in a register pair first -- by two vector moves if the instruction wants it as
// its accumulator, which is what the Horner forms below compile to, or parked in scalar registers across the loop, which
// spills the step's own scalars (the kernel needs ~60 such constants).  One broadcast ds_read_b128 delivers two coefficients
// to where the fused multiply-add wants them and costs the vector ALU nothing.  The arithmetic is that of sincos_fast_f64
// (general form), pair_term_fast and acos_r in pstat_math.h, operation for operation.
enum { K_2OPI = 0, K_PIO2_HI, K_PIO2_MID, K_S6, K_S5, K_S4, K_S3, K_S2, K_S1, K_C6, K_C5, K_C4, K_C3, K_C2, K_C1, K_R375, K_M3, K_INV4PI,
       K_A0, K_PIO2 = K_A0 + 13, K_PI, NK };
__constant__ double KINIT[NK] = {
    6.36619772367581382433e-01, 1.57079632679489655800e+00, 6.12323399573676603587e-17,
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04,
    8.33333333332248946124e-03, -1.66666666666666324348e-01,
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, 2.48015872894767294178e-05,
    -1.38888888888741095749e-03, 4.16666666666666019037e-02,
    0.375, -3.0, 0.0795774715459476679,
    2.87578513674215663354e-02, -1.48518870712472036977e-02, 1.74008794426940213707e-02, 5.45750671864035814818e-03,
    1.03228143501857792808e-02, 1.14791774151849056834e-02, 1.39712129735529329289e-02, 1.73523927208699725588e-02,
    2.23721729421498885526e-02, 3.03819441385312465076e-02, 4.46428571463554288434e-02, 7.49999999999843292020e-02,
    1.66666666666666685170e-01,
    1.57079632679489661923, 3.14159265358979323846};

__device__ __forceinline__ void sincos_tab(double x, const double *kt, double *s, double *c) {
  if (!(fabs(x) < PSTAT_PHI_FOLD)) {   // fold by whole turns (no chain gets here: see sincos_fast_f64)
    const double t = rint(x * 1.59154943091895345609e-01);
    x = __builtin_fma(-t, 2.44929359829470641435e-16, __builtin_fma(-t, 6.28318530717958623200e+00, x));
  }
  const double k = rint(x * kt[K_2OPI]);
  const double r = __builtin_fma(-k, kt[K_PIO2_MID], __builtin_fma(-k, kt[K_PIO2_HI], x));
  const double z = r * r;
  const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, kt[K_S6], kt[K_S5]), kt[K_S4]), kt[K_S3]), kt[K_S2]), kt[K_S1]);
  const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, kt[K_C6], kt[K_C5]), kt[K_C4]), kt[K_C3]), kt[K_C2]), kt[K_C1]);
  const uint32_t q = (uint32_t)(int)k;
  const uint32_t odd = q << 31;
  const uint32_t m_kc = (q >> 1) << 31;
  const uint32_t m_ks = m_kc ^ odd;
  const double rs = __hiloint2double((int)(__double2hiint(r) ^ m_ks), __double2loint(r));
  const double ks = __builtin_fma(rs * z, ps, rs);
  const double kc0 = __builtin_fma(z, __builtin_fma(z, pc, -0.5), 1.0);
  const double kc = __hiloint2double((int)(__double2hiint(kc0) ^ m_kc), __double2loint(kc0));
  const bool swap = (int)odd < 0;
  *s = swap ? kc : ks;
  *c = swap ? ks : kc;
}
__device__ __forceinline__ double pair_term_tab(double rx, double ry, double rz, double mix, double miy, double miz, double mjx,
                                                double mjy, double mjz, const double *kt) {
  const double r2 = __builtin_fma(rz, rz, __builtin_fma(ry, ry, rx * rx));
  const double y0 = __builtin_amdgcn_rsq(r2);
  const double e = __builtin_fma(-(r2 * y0), y0, 1.0);
  const double y = __builtin_fma(y0 * e, __builtin_fma(e, kt[K_R375], 0.5), y0);       // rsqrt_f64
  const double ir2 = y * y;
  const double mimj = __builtin_fma(miz, mjz, __builtin_fma(miy, mjy, mix * mjx));
  const double mir = __builtin_fma(miz, rz, __builtin_fma(miy, ry, mix * rx));
  const double mjr = __builtin_fma(mjz, rz, __builtin_fma(mjy, ry, mjx * rx));
  const double num = __builtin_fma(kt[K_M3] * ir2, mir * mjr, mimj);
  return num * (ir2 * y) * kt[K_INV4PI];
}
__device__ __forceinline__ double acos_tab(const double x, const double *kt) {
  const double a = fabs(x);
  const bool big = a >= 0.5;
  const double z = big ? __builtin_fma(a, -0.5, 0.5) : a * a;
  double p = kt[K_A0];
#pragma unroll
  for (int i = 1; i < 13; ++i) p = __builtin_fma(p, z, kt[K_A0 + i]);
  const double zp = z * p;
  const double y = __builtin_amdgcn_rsq(z);
  const double s0 = z * y, h0 = 0.5 * y;
  const double r = __builtin_fma(-s0, h0, 0.5);
  const double s1 = __builtin_fma(s0, r, s0), h1 = __builtin_fma(h0, r, h0);
  double sq = __builtin_fma(__builtin_fma(-s1, s1, z), h1, s1);
  sq = z == 0.0 ? 0.0 : sq;
  const double t = big ? sq : x;
  const double as = __builtin_fma(t, zp, t);
  const double small = kt[K_PIO2] - as;
  const double two = as + as;
  const double bigv = x < 0.0 ? kt[K_PI] - two : two;
  return big ? bigv : small;
}

// The literal acceptance test of inc/acceptance.jl:29-39 with the Hastings ratio alpha = anum / aden and the cached log(alpha)
// of the last acceptance: what metropolis_filter (pstat_math.h) evaluates for the one draw in ~1e3 that falls inside its
// margin.  A function of its own, not inlined: its two dozen polynomial coefficients would otherwise be parked in scalar
// registers across the whole step loop.
__device__ __attribute__((noinline)) bool literal_accept(const double dU, const double kT, const double st1, const double st0,
                                                         const double dw, const double anum, const double aden,
                                                         const bool lag_pending, const double lag_num, const double lag_den,
                                                         const double lag, const bool wide, const uint32_t weps, const uint32_t w0,
                                                         const uint32_t wphi, const uint32_t wth) {
  const double lg = lag_pending ? log_r(lag_num / lag_den) : lag;
  const double delta = -dU / kT + log_r(st1 / st0) + dw + log_r(anum / aden) - lg;
  const double eps = eps_uniform(wide, weps, w0, wphi, wth);
  return (delta >= 0) || (eps < exp_r(delta));
}

#ifndef PSTAT_CW_WAVES
#define PSTAT_CW_WAVES 4   // waves per SIMD asked of the register allocator (see the header)
#endif

template <int CT, int EN, int M>
__global__ __launch_bounds__(64, PSTAT_CW_WAVES) void cluster_cw_kernel(SweepArgs A, DevState S, const CaseConst *__restrict__ cases,
                                                                         int umb_on) {
  using R = double;
  using AG = Ang<R>;
  using T3 = V3<R>;
  constexpr int WPL = M + 1;          // draws per lane and step: 64 (M + 1) >= n + 5 stream positions
  constexpr int NPOS = 64 * WPL;
  constexpr int NC = 64 * M;
  constexpr R PI = AG::theta_max;
  // cell of monomer k at entry k + 1; entries 0 and n + 1 are pads (a neighbour that does not exist: read, never used)
  __shared__ double2 cA[NC + 3];      // (n_x, n_y); entry NC + 2: the moved monomer BEFORE the move (the bonds before it read that)
  __shared__ double2 cB[NC + 3];      // (n_z, theta)
  __shared__ double cP[NC + 2];       // phi
  __shared__ uint32_t draws[NPOS];
  __shared__ double KT[NK];

  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  const int64_t C = S.C;
  const int n = (int)A.n;
  const CaseConst cc = cases[c / A.chains_per_case];
  const R Fz = cc.Fz, Fx = cc.Fx, b = cc.b, kT = cc.kT;
  const R a_or_mu = (CT == PSTAT_DIELECTRIC) ? (cc.K1 - cc.K2) * cc.E0 : cc.mu;
  const R k2e = cc.K2 * cc.E0;
  const R mhalfE0 = -0.5 * cc.E0;
  const R hb = -cc.b / 2;
  const R khalf = cc.kappa / 2, psi0 = cc.psi0;
  const R cprob = cc.cluster_prob;
  const R ninv_kT = -1.0 / cc.kT;
  const bool bend = cc.kappa != 0;
  (void)hb;

  auto is_edge = [](const R th) __attribute__((always_inline)) -> bool { return th == (R)0 || th == PI; };
  auto mu_of = [&](const T3 &nh) __attribute__((always_inline)) -> T3 {
    T3 m;
    dipole<R, CT>(a_or_mu, k2e, nh.x, nh.y, nh.z, m.x, m.y, m.z);
    return m;
  };
  // refl_n!, inc/eap_chain.jl:263-265, in the reference's arithmetic
  auto refl_theta = [&](const R th) __attribute__((always_inline)) -> R { return fmin(PI, fmax((R)0, th + (PI - 2 * th))); };

  // ---- fill: angles from the checkpoint planes, the cached trigonometry derived with the functions the step uses
  {
    const R *gth = (const R *)S.ang, *gph = (const R *)S.ang + (int64_t)n * C;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int k = lane + 64 * j;
      if (k < n) {
        const R th = gth[(int64_t)k * C + c], ph = gph[(int64_t)k * C + c];
        R s, co, sp, cp;
        AG::sc_theta(th, &s, &co);
        AG::sc_phi(ph, &sp, &cp);
        cA[k + 1] = double2{cp * s, sp * s};
        cB[k + 1] = double2{co, th};
        cP[k + 1] = ph;
      } else {   // entries past the chain are read by the lanes that own no monomer there (masked, but 0 x garbage is NaN)
        cA[k + 1] = double2{0, 0}; cB[k + 1] = double2{0, 0}; cP[k + 1] = 0;
      }
    }
    if (lane < NK) KT[lane] = KINIT[lane];
    if (lane == 0) {
      cA[0] = double2{0, 0}; cB[0] = double2{0, 0}; cP[0] = 0;
      cA[NC + 1] = double2{0, 0}; cB[NC + 1] = double2{0, 0}; cP[NC + 1] = 0;
    }
  }
  // Pointers and options that only the flush every FLUSH steps, the adaptation and the spill touch are parked in LDS: held in
  // registers across the step loop they cost ~40 scalar registers, and the step's own scalars spill around them.
  struct Cold {
    double *sums, *obs, *stepsz, *lag, *wnorm;
    uint32_t *rng;
    int64_t *win, *nacc, *nanrej;
    R *ang;
    double adj_lb, adj_ub, adj_scale;
    int64_t spa, C;
  };
  __shared__ Cold cold_;
  if (lane == 0) {
    cold_.sums = S.sums + c; cold_.obs = S.obs + c; cold_.stepsz = S.stepsz + c; cold_.lag = S.lag + c; cold_.wnorm = S.wnorm + c;
    cold_.rng = S.rng + c; cold_.win = S.win + c; cold_.nacc = S.nacc_total + c; cold_.nanrej = S.nanrej + c;
    cold_.ang = (R *)S.ang + c;
    cold_.adj_lb = A.adj_lb; cold_.adj_ub = A.adj_ub; cold_.adj_scale = A.adj_scale;
    cold_.spa = A.steps_per_adjust; cold_.C = S.C;
  }
  __builtin_amdgcn_wave_barrier();
  volatile Cold *const cold = &cold_;     // (volatile: re-read at every use, never kept in a register across the loop)
  double *const obs_c = S.obs + c;
  uint32_t *const rng_c = S.rng + c;
  double *const stepsz_c = S.stepsz + c;
  int64_t *const win_c = S.win + c;

  // the chain's generator: lane l keeps the state WPL * l outputs down the stream
  uint64_t base;
  {
    Mwc64x g;
    g.load(rng_c, C);
    const uint64_t s0 = ((uint64_t)g.c << 32) | g.x;
    base = mwc_skip(s0, POW.v[WPL * lane]);
  }
  double phistep = stepsz_c[0 * C], thstep = stepsz_c[1 * C];
  R phs = phistep, ths = thstep;
  int64_t nacc_off = win_c[0 * C], natt_off = win_c[1 * C];
  int nacc_seg = 0, steps_seg = 0, nnan_seg = 0;
  // The current microstate's observables live ACROSS lanes: lane q < 9 of `obsv` holds r_x, r_y, r_z, p_x, p_y, p_z, U,
  // sum cos^2(theta), sum psi -- record! (ten averagers, mcmc_clustering_eap_chain.jl:243-244,310-311) is then two
  // fused multiply-adds for the whole chain instead of one per averager
  const int orow = lane < 7 ? lane : lane + 1;          // row of DevState::obs (OBS_USUM sits between U and C2)
  R obsv = lane < 9 ? obs_c[(int64_t)orow * C] : (R)0;
  R usum = obs_c[(int64_t)OBS_USUM * C];
  // log(alpha) of the last accepted proposal of this mcmc() call (inc/acceptance.jl:33-36), kept as the two products of
  // alpha = lag_num / lag_den until a literal evaluation or the spill needs the logarithm (cf. pstat_cluster_gm.hip)
  R lag = S.lag[c], lag_num = 1, lag_den = 1;
  bool lag_pending = false;
  const bool umb = umb_on != 0;
  const R wscale = umb ? (0.2 + 0.8 * exp(-(cc.Fx * cc.Fx + cc.Fz * cc.Fz) / cc.kT)) / cc.kT : 0.0;
  R uref = umb ? S.uref[c] : 0.0;
  bool regauged = false;
  double wnorm = umb ? S.wnorm[c] : 0.0;
  const R inv_nm1 = n > 1 ? 1.0 / (double)(n - 1) : 0.0;

  int64_t to_adj = A.adaptive ? A.steps_per_adjust - (A.step0 % A.steps_per_adjust) : 0;
  constexpr int FLUSH = 128;
  const int sign_a = (lane & 7) == 5 ? (int)0x80000000 : 0, sign_b = (lane & 7) == 7 ? (int)0x80000000 : 0;
  const int bond_mu = ((lane & 6) == 4) ? -1 : 0, bond_ml = ((lane & 6) == 6) ? -1 : 0;      // lanes of the upper / lower boundary bond
  const int bond_oa = ((lane & 7) < 2 || (lane & 7) >= 6) ? -1 : 0, bond_ob = ((lane & 7) >= 2 && (lane & 7) < 6) ? 1 : 0;
  const int bond_olda = (lane & 7) == 2 ? -1 : 0, bond_oldb = (lane & 7) == 0 ? -1 : 0;
  const int bond_sign = (lane & 1) ? 0 : (int)0x80000000;     // even lanes hold the bond BEFORE the proposal: subtracted
  int left = (int)A.nsteps;
  __builtin_amdgcn_wave_barrier();

  while (left > 0) {
    int chunk = left < FLUSH ? left : FLUSH;
    if (A.adaptive && to_adj < chunk) chunk = (int)to_adj;
    R a1v = 0, a2v = 0;      // lane q: sum of w x_q and of w x_q^2 over this block of steps
    R accw = 0;

    for (int s = 0; s < chunk; ++s) {
      // ---- this step's stretch of the stream: positions 0-3 = index, dphi, dtheta, skip draw (mcmc_clustering_eap_chain.jl:
      // 269-272, inc/eap_chain.jl:276); 4 ... = the link draws of cluster_flip!; then the acceptance draw
      {
        Mwc64x t;
        t.x = (uint32_t)base; t.c = (uint32_t)(base >> 32);
#pragma unroll
        for (int q = 0; q < WPL; ++q) draws[WPL * lane + q] = t.next();
      }
      __builtin_amdgcn_wave_barrier();
      int kz = 0;
      asm volatile("" : "+v"(kz));          // (the table is re-read every step, never held in registers across the loop)
      const double *const kt = &KT[kz];
      const uint4 fixed = *reinterpret_cast<const uint4 *>(&draws[0]);
      const uint32_t w0 = __builtin_amdgcn_readfirstlane(fixed.x);
      const uint32_t wphi = __builtin_amdgcn_readfirstlane(fixed.y);
      const uint32_t wth = __builtin_amdgcn_readfirstlane(fixed.z);
      const uint32_t wskip = __builtin_amdgcn_readfirstlane(fixed.w);
      const int idx = (int)__umulhi(w0, (uint32_t)n);
      const bool flipped = !(u01<R>(wskip) <= cprob);
      const bool hasL = idx > 0, hasR = idx + 1 < n;

      // ---- the single-monomer part
      const double2 a0 = cA[idx + 1], b0 = cB[idx + 1];
      const R ph0 = cP[idx + 1];
      const R th0 = b0.y;
      const T3 n0{a0.x, a0.y, b0.x};
      const R ct0 = n0.z;
      // (the trajectory itself: each product rounded before its sum, as the oracle and Julia round them -- through an opaque
      // register, so that no build flag can fuse them; cf. run_segment)
      auto rounded = [](R v) __attribute__((always_inline)) -> R { asm volatile("" : "+v"(v)); return v; };
      const R ph1 = AG::wrap(ph0 + rounded(phs * sym11<R>(wphi)));
      const R th1 = fmin(PI, fmax((R)0, th0 + rounded(ths * sym11<R>(wth))));
      // three sincos in one instruction stream: lane 0 theta', lane 1 phi', lane 2 theta (the cell carries no sin(theta))
      R st1, ct1, sp1, cp1, st0;
      {
        const R arg = sel_lanes(sel_lanes(th1, ph1, 2ull), th0, 4ull);
        R sv, cv;
        sincos_tab(arg, kt, &sv, &cv);
        st1 = lane_value<R>(sv, 0); ct1 = lane_value<R>(cv, 0);
        sp1 = lane_value<R>(sv, 1); cp1 = lane_value<R>(cv, 1);
        st0 = lane_value<R>(sv, 2);
      }
      const T3 n1{cp1 * st1, sp1 * st1, ct1};
      const T3 m0 = mu_of(n0), m1 = mu_of(n1);
      // the trial configuration goes into the chain now (growth, member sums and the boundary bonds read the moved monomer as
      // proposed, inc/eap_chain.jl:272-273); a rejection puts the old cell back
      if (lane == 0) {
        cA[idx + 1] = double2{n1.x, n1.y};
        cB[idx + 1] = double2{n1.z, th1};
        cA[NC + 2] = a0;
        cB[NC + 2] = b0;
      }
      __builtin_amdgcn_wave_barrier();

      // ---- cluster_flip!(trial, idx), inc/eap_chain.jl:269-333
      int upper = idx, lower = idx, ndraws = 0;
      R upper_p = 0, lower_p = 0;
      bool edge = false;
      R snz = 0, memx = 0, memy = 0;
      double2 ownB[M];
      bool mem[M];
#pragma unroll
      for (int j = 0; j < M; ++j) { ownB[j] = double2{0, 0}; mem[j] = false; }
      if (flipped) {
        R pk[M];
        double2 ownA[M];
        int rr[M], thr[M];
        bool link[M], up[M];
        bool f1[M];
        // pass 1: both ends alive -- round r tests link idx + r at stream position 4 + 2 r, link idx - 1 - r at 4 + 2 r + 1
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const int k = lane + 64 * j;                 // monomer k, and link (k, k + 1)
          const double2 a = cA[k + 1], bb = cB[k + 1], an = cA[k + 2], bn = cB[k + 2];
          ownA[j] = a; ownB[j] = bb;
          pk[j] = (1 + (a.x * an.x + a.y * an.y + bb.x * bn.x)) / 2;
          // u = (w >> 9) 2^-23 <= p  <=>  (w >> 9) <= floor(p 2^23): the draws are tested as integers (p < 0: never; p is never NaN)
          thr[j] = (int)(pk[j] * 8388608.0);
          thr[j] = pk[j] < 0 ? -1 : thr[j];
          link[j] = k < n - 1;
          up[j] = k >= idx;
          rr[j] = up[j] ? k - idx : idx - 1 - k;
          int pos = 4 + 2 * rr[j] + (up[j] ? 0 : 1);
          pos = pos < NPOS - 1 ? pos : NPOS - 1;       // (beyond what a step can consume: such a round is never reached)
          f1[j] = link[j] && !((int)(draws[pos] >> 9) <= thr[j]);
        }
        // first failing link of either end (-1: that end runs into the chain end, :282-284,299-301)
        int kfU = -1, kfL = -1;
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const uint64_t bu = __builtin_amdgcn_ballot_w64(f1[j] && up[j]);
          if (kfU < 0 && bu != 0) kfU = 64 * j + (int)__builtin_ctzll(bu);
        }
#pragma unroll
        for (int j = M - 1; j >= 0; --j) {
          const uint64_t bl = __builtin_amdgcn_ballot_w64(f1[j] && !up[j]);
          if (kfL < 0 && bl != 0) kfL = 64 * j + 63 - (int)__builtin_clzll(bl);
        }
        const int RU = n - 1 - idx, RL = idx;
        int dU = kfU >= 0 ? kfU - idx + 1 : RU;         // draws that end makes
        int dL = kfL >= 0 ? idx - kfL : RL;
        if (dU != dL) {
          // pass 2: the end that stopped first made m draws (exact: until then both were alive); from round m on the survivor
          // draws alone, at consecutive positions 4 + m + r
          const int m = dU < dL ? dU : dL;
          const bool contU = dU > dL;
          int kf = -1;
          if (contU) {
#pragma unroll
            for (int j = 0; j < M; ++j) {
              int pos = 4 + m + rr[j];
              pos = pos < NPOS - 1 ? pos : NPOS - 1;
              const bool f2 = link[j] && up[j] && rr[j] >= m && !((int)(draws[pos] >> 9) <= thr[j]);
              const uint64_t bu = __builtin_amdgcn_ballot_w64(f2);
              if (kf < 0 && bu != 0) kf = 64 * j + (int)__builtin_ctzll(bu);
            }
            kfU = kf;
            dU = kfU >= 0 ? kfU - idx + 1 : RU;
          } else {
#pragma unroll
            for (int j = M - 1; j >= 0; --j) {
              int pos = 4 + m + rr[j];
              pos = pos < NPOS - 1 ? pos : NPOS - 1;
              const bool f2 = link[j] && !up[j] && rr[j] >= m && !((int)(draws[pos] >> 9) <= thr[j]);
              const uint64_t bl = __builtin_amdgcn_ballot_w64(f2);
              if (kf < 0 && bl != 0) kf = 64 * j + 63 - (int)__builtin_clzll(bl);
            }
            kfL = kf;
            dL = kfL >= 0 ? idx - kfL : RL;
          }
        }
        ndraws = dU + dL;
        upper = kfU >= 0 ? kfU : n - 1;
        lower = kfL >= 0 ? kfL + 1 : 0;
        // the probability of the link each end stopped at (0 at a chain end)
#pragma unroll
        for (int j = 0; j < M; ++j) {
          if (kfU >= 0 && (kfU >> 6) == j) upper_p = lane_value<R>(pk[j], kfU & 63);
          if (kfL >= 0 && (kfL >> 6) == j) lower_p = lane_value<R>(pk[j], kfL & 63);
        }
        // members: sums of what the reflection flips; a member exactly on a clamp value makes the proposal unacceptable
        edge = is_edge(th1);
        R lz = 0, lx = 0, ly = 0;
        bool le = false;
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const int k = lane + 64 * j;
          mem[j] = k >= lower && k <= upper;
          le = le || (mem[j] && k != idx && is_edge(ownB[j].y));
          const R z = mem[j] ? ownB[j].x : (R)0;
          lz += z;
          const R q = a_or_mu * z;
          if constexpr (CT == PSTAT_DIELECTRIC) { lx += mem[j] ? q * ownA[j].x : (R)0; ly += mem[j] ? q * ownA[j].y : (R)0; }
        }
        edge = edge || __builtin_amdgcn_ballot_w64(le) != 0;
        snz = sum64(lz);          // enters dU through F_z r_z (and mu_z E for a polar chain)
        memx = lx; memy = ly;     // the dielectric's flipped dipole components are observables only: summed when accepted
      }
      const uint32_t weps = __builtin_amdgcn_readfirstlane(draws[4 + ndraws]);   // the acceptance draw comes after the cluster's draws
      const uint64_t skip = POW.v[5 + ndraws];

      // ---- the eight bonds of the proposal, one per lane of every group of eight: lanes 0-3 the single move's (L,0) (L,1)
      // (0,R) (1,R), lanes 4-7 the boundary bonds (cu,nu) (refl cu,nu) (nl,cl) (nl,refl cl) of :318-326; even lanes before,
      // odd lanes after.  All groups compute the same eight; group 0 then contributes its bond angles to one signed sum over
      // eight lanes, group 1 its bending energies, group 2 its pair energies: the three differences come out of ONE tree.
      const bool on_u = flipped && upper < n - 1, on_l = flipped && lower > 0;
      R dpsi_all, dbend_all, dpair_all, new_upper_p, new_lower_p, bond_dt;
      int bond_on;
      {
        const int q = lane & 7;
        // entries: (idx - 1, idx) (idx - 1, idx) (idx, idx + 1) (idx, idx + 1) (upper, upper + 1) x 2 (lower - 1, lower) x 2
        const int ebase = (idx + 1) + ((upper - idx) & bond_mu) + ((lower - idx) & bond_ml);
        int ia = ebase + bond_oa, ib = ebase + bond_ob;
        ia = (bond_olda & (NC + 2)) | (~bond_olda & ia);      // the bonds before the move see the old monomer (entry NC + 2)
        ib = (bond_oldb & (NC + 2)) | (~bond_oldb & ib);
        const double2 xa = cA[ia], za = cB[ia], xb = cA[ib], zb = cB[ib];
        T3 na{xa.x, xa.y, za.x}, nb{xb.x, xb.y, zb.x};
        na.z = __hiloint2double(__double2hiint(na.z) ^ sign_a, __double2loint(na.z));   // refl_n!: n_z -> -n_z (lanes 5: a, 7: b)
        nb.z = __hiloint2double(__double2hiint(nb.z) ^ sign_b, __double2loint(nb.z));
        const T3 ma = mu_of(na), mb = mu_of(nb);
        const R dt = dot3(na, nb);
        // psi_j (inc/eap_chain.jl:45-47) feeds the bending energy and the <psi> averager: without stiffness (kappa = 0, every
        // sweep of run/) only an accepted proposal needs it, and the commit takes the arc cosines then
        R psi = 0, ebend = 0;
        if (bend) {
          psi = acos_tab(fmin((R)1, fmax((R)-1, dt)), kt);
          ebend = khalf * (psi - psi0) * (psi - psi0);
        }
        R epair = 0;
        if constexpr (EN == PSTAT_ISING)
          epair = pair_term_tab(hb * (na.x + nb.x), hb * (na.y + nb.y), hb * (na.z + nb.z), ma.x, ma.y, ma.z, mb.x, mb.y, mb.z, kt);
        const R pnew = (1 + dt) / 2;
        // which bonds exist, as a per-lane 0 / -1 word out of a scalar bit mask (one v_bfe_i32); the sign (after minus before)
        // and the existence mask are applied to the bits of the value (a bond that does not exist may have computed anything)
        const uint32_t m8 = (hasL ? 3u : 0u) | (hasR ? 12u : 0u) | (on_u ? 48u : 0u) | (on_l ? 192u : 0u);
        const int onm = __builtin_amdgcn_sbfe((int)m8, (unsigned)q, 1u);
        bond_on = onm; bond_dt = dt;
        const R wsel = sel_lanes(sel_lanes(epair, ebend, 0xFF00ull), psi, 0xFFull);     // lanes 0-7: psi, 8-15: bending, 16-: pair
        const R w = __hiloint2double((__double2hiint(wsel) ^ bond_sign) & onm, __double2loint(wsel) & onm);
        const R tot = sum8(w);
        dpsi_all = lane_value<R>(tot, 0);
        dbend_all = lane_value<R>(tot, 8);
        dpair_all = EN == PSTAT_ISING ? lane_value<R>(tot, 16) : (R)0;
        new_upper_p = on_u ? lane_value<R>(pnew, 5) : (R)0;
        new_lower_p = on_l ? lane_value<R>(pnew, 7) : (R)0;
      }
      // alpha = anum / aden, :328-329 -- never divided on the common path (the filter below takes the two products)
      R anum = 1, aden = 1, drz_flip = 0, du_flip = 0;
      T3 dp_flip{0, 0, 0};
      if (flipped) {
        anum = (1 - new_upper_p) * (1 - new_lower_p);
        aden = (1 - upper_p) * (1 - lower_p);
        // members' own terms: n_z -> -n_z; dielectric mu -> (-mu_x, -mu_y, mu_z), polar mu_z -> -mu_z
        drz_flip = b * (-2 * snz);
        if constexpr (CT != PSTAT_DIELECTRIC) { dp_flip.z = -2 * (a_or_mu * snz); du_flip = mhalfE0 * dp_flip.z; }
      }

      // ---- energy difference of the whole proposal, inc/energy.jl:7-23
      const R du_field = mhalfE0 * (m1.z - m0.z);
      const R drx = b * (n1.x - n0.x), dry = b * (n1.y - n0.y), drz = b * (n1.z - n0.z) + drz_flip;
      const R dus = du_field + dbend_all + du_flip;        // change of sum(u), bending included
      const R dU = dus + dpair_all - (Fx * drx + Fz * drz);

      // ---- Metropolis-Hastings, inc/acceptance.jl:29-39 (the f32 filter of pstat_math.h, the literal expression in its margin)
      const R dw = umb ? dus * wscale : (R)0;
      bool ok = metropolis_filter(dU * ninv_kT + (dw - (lag_pending ? (R)0 : lag)), (st1 * anum) * (lag_pending ? lag_den : (R)1),
                                  (st0 * aden) * (lag_pending ? lag_num : (R)1), weps, [&]() -> bool {
        return literal_accept(dU, kT, st1, st0, dw, anum, aden, lag_pending, lag_num, lag_den, lag, A.wide_eps != 0, weps, w0, wphi, wth);
      });
      ok = ok && !edge;
      ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
      if constexpr (EN == PSTAT_ISING) nnan_seg += not_finite(dU) ? 1 : 0;

      // ---- commit (wave-uniform)
      if (ok) {
        if (flipped) {
#pragma unroll
          for (int j = 0; j < M; ++j)
            if (mem[j]) cB[lane + 64 * j + 1] = double2{-ownB[j].x, refl_theta(ownB[j].y)};
        }
        if (lane == 0) cP[idx + 1] = ph1;
        if constexpr (CT == PSTAT_DIELECTRIC) {
          if (flipped) { dp_flip.x = -2 * sum64(memx); dp_flip.y = -2 * sum64(memy); }
        }
        const R dpx = (m1.x - m0.x) + dp_flip.x, dpy = (m1.y - m0.y) + dp_flip.y, dpz = (m1.z - m0.z) + dp_flip.z;
        const R dc2 = ct1 * ct1 - ct0 * ct0;
        if (!bend) {                           // the bond angles of the accepted proposal, after minus before
          const R psi = acos_tab(fmin((R)1, fmax((R)-1, bond_dt)), kt);
          const R w = __hiloint2double((__double2hiint(psi) ^ bond_sign) & bond_on, __double2loint(psi) & bond_on);
          dpsi_all = lane_value<R>(sum8(w), 0);
        }
        R dv = dpsi_all;                       // lane 8 (lanes >= 9 stay 0 + whatever: never read)
        dv = sel_lanes(dv, dc2, 1ull << 7);
        dv = sel_lanes(dv, dU, 1ull << 6);
        dv = sel_lanes(dv, dpz, 1ull << 5);
        dv = sel_lanes(dv, dpy, 1ull << 4);
        dv = sel_lanes(dv, dpx, 1ull << 3);
        dv = sel_lanes(dv, drz, 1ull << 2);
        dv = sel_lanes(dv, dry, 1ull << 1);
        dv = sel_lanes(dv, drx, 1ull << 0);
        obsv += dv;
        usum += dus;
        lag_num = anum; lag_den = aden;
        lag_pending = true;
        ++nacc_seg;
      } else if (lane == 0) {
        cA[idx + 1] = a0;
        cB[idx + 1] = b0;
      }
      base = mwc_skip(base, skip);
      __builtin_amdgcn_wave_barrier();

      // ---- record! x 10
      R wgt = 1;
      if (umb) {
        bool raise;
        R wrel = umbrella_logw(usum, uref, wscale, raise);
        if (raise) {   // (wave-uniform) the gauge rises to this configuration (pstat_math.h); every lane scales its own rows
          const double f = exp_f64(-wrel);
          double *const sums_c = cold->sums;
          const int64_t Cc = cold->C;
          const int r1 = lane < 3 ? lane : (lane < 6 ? lane + 3 : (lane == 6 ? (int)S_U : lane + 7));
          const int r2 = lane < 3 ? lane + 3 : (lane < 6 ? lane + 6 : (int)S_USQ);
          if (lane < 9) sums_c[(int64_t)r1 * Cc] *= f;
          if (lane < 7) sums_c[(int64_t)r2 * Cc] *= f;
          a1v *= f; a2v *= f; accw *= f; wnorm *= f;
          uref = usum; regauged = true; wrel = 0;
        }
        wgt = exp_r(wrel);
      }
      accw += wgt;
      a1v = fma_r(wgt, obsv, a1v);
      a2v = fma_r(wgt * obsv, obsv, a2v);
    }

    // add the block to the averagers' sums (inc/average.jl:9): lane q its own rows; <psi> is sum psi / (n - 1)
    {
      const int r1 = lane < 3 ? lane : (lane < 6 ? lane + 3 : (lane == 6 ? (int)S_U : lane + 7));       // S_R*, S_P*, S_U, S_C2, S_PSI
      const int r2 = lane < 3 ? lane + 3 : (lane < 6 ? lane + 6 : (int)S_USQ);                         // S_R*SQ, S_P*SQ, S_USQ
      double *const sums_c = cold->sums;
      const int64_t Cc = cold->C;
      if (lane < 9) sums_c[(int64_t)r1 * Cc] += lane == 8 ? a1v * inv_nm1 : a1v;
      if (lane < 7) sums_c[(int64_t)r2 * Cc] += a2v;
      wnorm += accw;
    }
    left -= chunk;
    steps_seg += chunk;

    // ---- step-size adaptation, mcmc_clustering_eap_chain.jl:287-308
    if (A.adaptive) {
      to_adj -= chunk;
      if (to_adj == 0) {
        to_adj = cold->spa;
        const double adj_lb = cold->adj_lb, adj_ub = cold->adj_ub, adj_scale = cold->adj_scale;
        const int64_t nacc = nacc_off + nacc_seg, natt = natt_off + steps_seg;
        const double ratio = (double)nacc / (double)natt;
        if (ratio > adj_ub && phistep != K<double>::pi && thstep != K<double>::half_pi) {
          nacc_off = -nacc_seg; natt_off = -steps_seg;
          phistep = fmin(K<double>::pi, phistep * adj_scale);
          thstep = fmin(K<double>::half_pi, thstep * adj_scale);
        } else if (ratio < adj_lb) {
          nacc_off = -nacc_seg; natt_off = -steps_seg;
          phistep /= adj_scale;
          thstep /= adj_scale;
        }
        phs = phistep; ths = thstep;
      }
    }
  }

  // ---- spill: the angles back to the checkpoint planes
  __builtin_amdgcn_wave_barrier();
  {
    const int64_t Cc = cold->C;
    R *gth = cold->ang, *gph = gth + (int64_t)n * Cc;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int k = lane + 64 * j;
      if (k < n) {
        gth[(int64_t)k * Cc] = cB[k + 1].y;
        gph[(int64_t)k * Cc] = cP[k + 1];
      }
    }
    double *const ob = cold->obs;
    if (lane < 9) ob[(int64_t)orow * Cc] = obsv;
    if (lane == 0) {
      Mwc64x g;
      g.x = (uint32_t)base; g.c = (uint32_t)(base >> 32);
      g.store(cold->rng, Cc);
      double *const sz = cold->stepsz;
      sz[0] = phistep; sz[Cc] = thstep;
      int64_t *const wn = cold->win;
      wn[0] = nacc_off + nacc_seg; wn[Cc] = natt_off + steps_seg;
      *cold->nacc += nacc_seg;
      if constexpr (EN == PSTAT_ISING) *cold->nanrej += nnan_seg;
      ob[(int64_t)OBS_USUM * Cc] = usum;
      *cold->lag = lag_pending ? log_r(lag_num / lag_den) : lag;
      if (umb) *cold->wnorm = wnorm;
      if (regauged) S.uref[c] = uref;
    }
  }
}

using CwFn = void (*)(SweepArgs, DevState, const CaseConst *, int);

template <int CT, int EN>
CwFn pick_m(const int64_t n) {
  if (n <= 64) return cluster_cw_kernel<CT, EN, 1>;
  if (n <= 128) return cluster_cw_kernel<CT, EN, 2>;
  return cluster_cw_kernel<CT, EN, 4>;
}

CwFn pick(const LaunchCfg &cfg, const int64_t n) {
  const bool ising = cfg.energy_type == PSTAT_ISING;
  if (cfg.chain_type == PSTAT_DIELECTRIC)
    return ising ? pick_m<PSTAT_DIELECTRIC, PSTAT_ISING>(n) : pick_m<PSTAT_DIELECTRIC, PSTAT_NONINTERACTING>(n);
  return ising ? pick_m<PSTAT_POLAR, PSTAT_ISING>(n) : pick_m<PSTAT_POLAR, PSTAT_NONINTERACTING>(n);
}

}  // namespace

// The configurations this kernel covers, and the ensembles it is chosen for (measured at n = 100 on the (E0, kT) grid of
// run/K1_E0-kT-phase.jl, tools/time_cluster_cw.py -> profiles/r04/experiments/time_cluster_cw.txt).  One wave per chain costs the
// chip ~0.85 ns per chain-step whatever the clusters do (2 730 chains: 3.1 us per step, 43 680: 36 us).  The chain-per-lane
// kernel steps a whole ensemble in 4.5-5.5 us while its waves fit the chip once and every chain is disordered, but a wave runs at
// the pace of its longest cluster (25-33 us per step on an aligned chain) and a launch at the pace of its slowest wave: the
// same grids take it 29-45 us per step at 1 to 16 chains per case.  So: every ensemble of up to 4 096 chains, and sweeps of
// many small cases (<= 16 chains each) up to 49 152 chains; large ensembles of few cases keep the chain-per-lane kernels,
// whose full waves are 10 x cheaper per chain-step there.
bool cluster_chain_wave(const LaunchCfg &cfg, const int64_t n, const int64_t chains_per_case, const int64_t ncases) {
  if (cfg.move_set != PSTAT_MOVES_CLUSTER || cfg.precision != PSTAT_F64 || cfg.rng != PSTAT_RNG_MWC64X) return false;
  if (cfg.energy_type != PSTAT_NONINTERACTING && cfg.energy_type != PSTAT_ISING) return false;
  if (n < 1 || n > 256) return false;
  const char *e = getenv("PSTAT_F64_STATE");
  if (e && e[0] == 'w') return true;
  if (e && (e[0] == 'l' || e[0] == 'g')) return false;
  if (getenv("PSTAT_PACK")) return false;     // (a test or experiment about the block layout is about the chain-per-lane kernels)
  const int64_t total = chains_per_case * ncases;
  if (total <= 4096) return true;
  return ncases >= 8 && chains_per_case <= 16 && total <= (n <= 128 ? 49152 : 24576);
}

hipError_t cluster_cw_kernel_info(const LaunchCfg &cfg, const int64_t n, int *blocks_per_cu, const char **name) {
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)pick(cfg, n), 64, 0);
  if (e != hipSuccess) return e;
  if (blocks_per_cu) *blocks_per_cu = nb;
  if (name) *name = "cluster_chain_wave_kernel<double>";
  return hipSuccess;
}

hipError_t launch_cluster_cw(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s, const CaseConst *cases,
                             hipStream_t stream) {
  hipLaunchKernelGGL(pick(cfg, a.n), dim3((unsigned)s.C), dim3(64), 0, stream, a, s, cases, cfg.umbrella);
  return hipGetLastError();
}

}  // namespace pstat
