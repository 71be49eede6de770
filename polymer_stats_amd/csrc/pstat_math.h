// pstat_math.h -- device-side arithmetic shared by the sweep kernels (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/pstat.h"

namespace pstat {

// ------------------------------------------------------------------------------------------ math

template <typename R> struct Vec2;
template <> struct Vec2<float> { using type = float2; };
template <> struct Vec2<double> { using type = double2; };

template <typename R> struct K;  // constants of the adaptation logic (f64, like the reference)
template <> struct K<double> {
  static constexpr double pi = 3.14159265358979323846;
  static constexpr double half_pi = 1.57079632679489661923;
};

// exp and log of a double for the acceptance test of the f64 kernels, ~30 instructions each (the library
// versions cost several times that).  Classic forms: exp by x = k ln2 + r, |r| <= ln2/2, a degree-13
// polynomial and one v_ldexp; log by x = 2^k (1 + f), sqrt(1/2) < 1 + f < sqrt(2), s = f / (2 + f) and the
// fdlibm minimax polynomial in s^2 (the reciprocal is v_rcp_f64 + two Newton steps).  Both within 2 ulp
// (tools/mathcheck); the test they feed compares against a 23-bit uniform.  exp(-inf) = 0, log(0) = -inf,
// NaN in => NaN out, as the reject-on-NaN logic of the callers needs.
__device__ __forceinline__ double exp_f64(const double x) {
  const double k = rint(x * 1.44269504088896338700e+00);
  const double hi = __builtin_fma(-k, 6.93147180369123816490e-01, x);
  const double r = __builtin_fma(-k, 1.90821492927058770002e-10, hi);
  double p = 1.60590438368216145994e-10;                    // 1/13!
  p = __builtin_fma(p, r, 2.08767569878680989792e-09);      // 1/12!
  p = __builtin_fma(p, r, 2.50521083854417187751e-08);
  p = __builtin_fma(p, r, 2.75573192239858906526e-07);
  p = __builtin_fma(p, r, 2.75573192239858906526e-06);
  p = __builtin_fma(p, r, 2.48015873015873015873e-05);
  p = __builtin_fma(p, r, 1.98412698412698412698e-04);
  p = __builtin_fma(p, r, 1.38888888888888888889e-03);
  p = __builtin_fma(p, r, 8.33333333333333333333e-03);
  p = __builtin_fma(p, r, 4.16666666666666666667e-02);
  p = __builtin_fma(p, r, 1.66666666666666666667e-01);
  p = __builtin_fma(p, r, 5.00000000000000000000e-01);
  const double y = 1.0 + __builtin_fma(p, r * r, r);
  double v = ldexp(y, (int)k);
  v = x < -745.2 ? 0.0 : v;           // underflow (also x = -inf, where k and r are not finite)
  v = x > 709.8 ? __builtin_inf() : v;
  return v;                           // NaN: both comparisons false, the arithmetic already gave NaN
}
__device__ __forceinline__ double log_f64(const double x) {
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);                // x = m 2^e, 1/2 <= m < 1
  const bool small = m < 7.07106781186547524401e-01;
  m = small ? m + m : m;
  e = small ? e - 1 : e;
  const double f = m - 1.0, t = 2.0 + f;
  double rc = __builtin_amdgcn_rcp(t);
  rc = __builtin_fma(__builtin_fma(-t, rc, 1.0), rc, rc);
  rc = __builtin_fma(__builtin_fma(-t, rc, 1.0), rc, rc);
  const double s = f * rc, z = s * s, w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                                       2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t1 + t2, hfsq = 0.5 * f * f, dk = (double)e;
  double v = dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
  v = x == 0.0 ? -__builtin_inf() : v;
  v = x < 0.0 ? __builtin_nan("") : v;
  v = x == __builtin_inf() ? x : v;
  return v;                           // NaN in: frexp propagates it
}

// Umbrella sampling (inc/average.jl:52-124): records are weighted by 1 / e^w, w = sum(u) scale - log_gauge.  value / normaliser
// does not depend on the gauge; the reference fixes it a priori (-lead n / 3 kT + Omega_initial), the device gauges a chain on
// one of its own configurations (DevState::uref).  Neither is safe by itself: on a cold chain the reference's normaliser
// underflows to 0 (NaN averages; seen in tests/fuzz_packed.py), and a configuration e^700 heavier than the gauged one --
// one reflected cluster of a cold polar chain moves w by thousands -- overflows every sum (seen in tests/fuzz_cluster_wave.py).
// So the gauge only ever RISES: a step whose configuration weighs more than e^T times the gauged one re-gauges the chain on it
// before it is recorded -- every accumulated sum and the normaliser are multiplied by e^-w, its own weight becomes 1 --
// and lighter configurations simply underflow, which is what their share of the average is.  umbrella_logw returns the
// log-weight of the current configuration and whether the gauge must rise.
template <typename R>
__device__ __forceinline__ R umbrella_logw(const R usum, const R uref, const R wscale, bool &raise) {
  const R wrel = -(usum - uref) * wscale;
  constexpr R T = sizeof(R) == 4 ? (R)30 : (R)300;      // f32 weights and block sums live in floats (e^88 is their end)
  raise = wrel > T;                                     // (false for a NaN)
  return wrel;
}

__device__ __forceinline__ float exp_r(float x) { return __expf(x); }
__device__ __forceinline__ double exp_r(double x) { return exp_f64(x); }
__device__ __forceinline__ float log_r(float x) { return __logf(x); }
__device__ __forceinline__ double log_r(double x) { return log_f64(x); }
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_r(double a, double b, double c) { return a * b + c; }

// u(w) = (w >> 9) * 2^-23 in [0,1).  v_cvt_f32_u32 is a quarter-rate instruction on gfx950, so the
// f32 forms splice the 23 bits into the mantissa of a float in [1,2) (or [2,4) for 2u-1) instead:
// two full-rate integer ops and one add, and the result is exactly the same number.
template <typename R>
__device__ __forceinline__ R u01(uint32_t w) {
  if constexpr (sizeof(R) == 4) return __uint_as_float(0x3F800000u | (w >> 9)) - 1.0f;
  else return (R)(w >> 9) * (R)(1.0 / 8388608.0);
}
// the same two floats before the subtraction, built by one v_alignbit_b32: ({hi, w} >> 9) with
// hi = 0x7F gives 0x3F800000 | (w >> 9) (a float in [1,2)), hi = 0x80 gives one in [2,4)
__device__ __forceinline__ float bits12(uint32_t w) { return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, w, 9)); }
__device__ __forceinline__ float bits24(uint32_t w) { return __uint_as_float(__builtin_amdgcn_alignbit(0x80u, w, 9)); }
template <typename R>
__device__ __forceinline__ R sym11(uint32_t w) {  // 2u - 1 in [-1,1)
  if constexpr (sizeof(R) == 4) return __uint_as_float(0x40000000u | (w >> 9)) - 3.0f;
  else return __builtin_fma((R)(w >> 9), (R)(1.0 / 4194304.0), (R)-1);   // k 2^-22 - 1: every operation exact, = 2 u - 1
}

// The uniform eps of the Metropolis test (mcmc_eap_chain.jl:287: rand(), a Float64 with 53 random bits) under the two
// stream contracts of pstat_params.uniform_bits (include/pstat.h):
//   23 bits   eps = (w_eps >> 9) 2^-23 -- the f32 kernels' mantissa trick; acceptance probabilities have a floor of 2^-23
//   53 bits   eps = (w_eps 2^21 + lo) 2^-53 with lo = the bits of the step's OTHER draws that no proposal uses: the low 9 bits
//             of the dtheta word, the low 9 of the dphi word (a proposal takes the top 23 of each) and the low 3 of the index
//             word (idx = mulhi(w, n) does not see them for any n < 2^29).  No extra draw: the stream position of every
//             word is the same under both contracts, and since only the literal branch below ever forms eps in full, the
//             hot loop pays one f32 instruction for it (the filter's upper bound).
__device__ __forceinline__ double eps_uniform(const bool wide, const uint32_t weps, const uint32_t w0, const uint32_t wphi,
                                              const uint32_t wth) {
  const uint32_t lo = ((wth & 511u) << 12) | ((wphi & 511u) << 3) | (w0 & 7u);
  const double e53 = __builtin_fma((double)weps, 2097152.0, (double)lo) * 0x1p-53;   // exact: < 2^53
  const double e23 = (double)(weps >> 9) * (1.0 / 8388608.0);
#ifdef PSTAT_NARROW_EPS   // (timing experiment, tools/build_variant.sh: what the 53-bit contract costs the hot loop)
  return e23;
#else
  return wide ? e53 : e23;
#endif
}

// The Metropolis test of the f64 kernels (inc/acceptance.jl:29-39 with the cached log-density written as a
// difference):  ok = (delta >= 0) || (eps < exp(delta)),  delta = -dU/kT + log(st1/st0) + extra,  eps as above.
// The literal expression costs a log, an exp and two divisions in double (~95 instructions).  Since eps < 1, the test
// is  eps * st0 < st1 * exp(-dU/kT + extra)  wherever that product is finite, and an f32 evaluation of the two sides
// (v_exp_f32, three conversions) decides it whenever they differ by more than the f32 error bound m: only draws
// within a relative 1e-5 of the threshold -- one wave-step in ~1500 -- run the literal double expression, whose
// verdict is then taken unchanged.  The decision is therefore ALWAYS the literal one (the bit-parity tests against
// the oracle hold); the filter only spares its evaluation.  The filter sees the top 23 bits u of eps only, i.e. eps in
// [u, u + 2^-23) under either contract: it accepts on the upper end of that interval and rejects on the lower, so a draw
// whose low bits could matter (u * den within 2^-23 den of the threshold) goes to the literal branch too.  NaN, +-inf,
// underflow to 0 and eps = 0 all land in the literal branch or on the side the literal test takes.
// x = everything in delta except the logarithm of the ratio num / den (sweep: num / den = sin(theta') / sin(theta); clustering
// main: times the Hastings ratio alpha); literal() = the reference's expression, evaluated only in the rare branch.
template <typename Literal>
__device__ __forceinline__ bool metropolis_filter(const double x, const double num, const double den, const uint32_t weps,
                                                  Literal &&literal) {
  const float t = (float)x;
  const float e = __builtin_amdgcn_exp2f(t * 1.44269504f);
  const float fden = (float)den;
  const float lhs = (__uint_as_float(0x3F800000u | (weps >> 9)) - 1.0f) * fden, rhs = e * (float)num;
  const float lhs_hi = __builtin_fmaf(0x1p-23f, fden, lhs);     // (u + 2^-23) den: eps is below it whatever its low bits
  const float m = 2e-6f + 1e-6f * __builtin_fabsf(t);          // > 3x the f32 error of rhs / lhs
#ifdef PSTAT_NARROW_EPS
  const bool acc = __builtin_fmaf(lhs, m, lhs) < rhs, rej = lhs > __builtin_fmaf(rhs, m, rhs);
  (void)lhs_hi;
#else
  const bool acc = __builtin_fmaf(lhs_hi, m, lhs_hi) < rhs, rej = lhs > __builtin_fmaf(rhs, m, rhs);
#endif
  bool ok = acc;
  if (__builtin_amdgcn_ballot_w64(!(acc || rej)) != 0) {        // some lane is too close to call (or not finite)
    const bool lit = literal();
    ok = (acc || rej) ? acc : lit;
  }
  return ok;
}
__device__ __forceinline__ bool metropolis_f64(const double dU, const double kT, const double ninv_kT, const double st1,
                                               const double st0, const double extra, const bool wide, const uint32_t weps,
                                               const uint32_t w0, const uint32_t wphi, const uint32_t wth) {
  return metropolis_filter(dU * ninv_kT + extra, st1, st0, weps, [&]() -> bool {
    const double delta = -dU / kT + log_f64(st1 / st0) + extra;
    const double eps = eps_uniform(wide, weps, w0, wphi, wth);
    return (delta >= 0) || (eps < exp_f64(delta));
  });
}

// dipole of one monomer: inc/dipole_response.jl:7-11 (dielectric), :27-29 with M = mu*I (polar)
template <typename R, int CT>
__device__ __forceinline__ void dipole(R a_or_mu, R k2e, R nx, R ny, R nz, R &mx, R &my, R &mz) {
  if constexpr (CT == PSTAT_DIELECTRIC) {
    R a = a_or_mu * nz;  // (K1-K2) E0 cos(theta)
    mx = a * nx; my = a * ny; mz = a * nz + k2e;
  } else {
    mx = a_or_mu * nx; my = a_or_mu * ny; mz = a_or_mu * nz;
  }
}

// one dipole-dipole term (inc/eap_chain.jl:200-207) for a bond vector r = x_i - x_j
template <typename R>
__device__ __forceinline__ R pair_term(R rx, R ry, R rz, R mix, R miy, R miz, R mjx, R mjy, R mjz) {
  R r2 = rx * rx + ry * ry + rz * rz;
  R rmag = sqrt(r2);
  R hx = rx / rmag, hy = ry / rmag, hz = rz / rmag;
  R r3 = r2 * rmag;
  R mimj = mix * mjx + miy * mjy + miz * mjz;
  R mir = mix * hx + miy * hy + miz * hz;
  R mjr = mjx * hx + mjy * hy + mjz * hz;
  return (mimj - 3 * mir * mjr) / ((R)(4.0 * 3.14159265358979323846) * r3);
}

// Stored angle formats (the element type of DevState::ang):
//   double   : radians                                   (PSTAT_F64)
//   float    : turns, theta/2pi in [0,1/2], phi/2pi in [0,1)   (PSTAT_F32, see Ang<> below)
//   uint16_t : index on a 2^16-point MIDPOINT lattice,   (PSTAT_Q16)
//              theta_k = pi (k + 1/2) / 65536,  phi_j = 2 pi (j + 1/2) / 65536
// f32 form of pair_term for the hot loops: one v_rsq instead of sqrt + 4 divisions, explicit FMAs
__device__ __forceinline__ float pair_term_fast(float rx, float ry, float rz, float mix, float miy, float miz,
                                                float mjx, float mjy, float mjz) {
  const float r2 = __builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx));
  const float ir = __builtin_amdgcn_rsqf(r2);
  const float ir2 = ir * ir;
  const float mimj = __builtin_fmaf(miz, mjz, __builtin_fmaf(miy, mjy, mix * mjx));
  const float mir = __builtin_fmaf(miz, rz, __builtin_fmaf(miy, ry, mix * rx));
  const float mjr = __builtin_fmaf(mjz, rz, __builtin_fmaf(mjy, ry, mjx * rx));
  const float num = __builtin_fmaf(-3.0f * ir2, mir * mjr, mimj);
  return num * (ir2 * ir) * 0.0795774715459476679f;   // 1/(4 pi)
}
// 1 / sqrt(x) in full double precision from v_rsq_f64 (relative error 5.2e-8, measured) and ONE third-order correction:
// with e = 1 - x y^2,  1/sqrt(x) = y / sqrt(1 - e) = y (1 + e/2 + 3 e^2 / 8 + O(e^3)),  e^3 ~ 1e-22.  Five instructions
// where two Newton steps take eight (+3...5 % on the f64 all-pairs kernels); max error 2.6 ulp over 4 M arguments of
// 1e-12 ... 1e12 (tools/mathcheck; the two Newton steps gave 1.2 ulp: the rounding of x y enters e).  rsq(0) = inf gives
// x y = NaN and the result NaN: the callers' 1/r^3 singularity stays a NaN as in the reference.
__device__ __forceinline__ double rsqrt_f64(const double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-(x * y), y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);
}
__device__ __forceinline__ double pair_term_fast(double rx, double ry, double rz, double mix, double miy,
                                                 double miz, double mjx, double mjy, double mjz) {
  // f64 hot loops: the same algebraic form as the f32 one, 1/r from rsqrt_f64 (full double precision; ~27 instructions
  // against ~75 for the literal sqrt + divisions of pair_term<double>, which the initialisation kernels keep).
  // r = 0 gives NaN, as the literal form does.
  const double r2 = __builtin_fma(rz, rz, __builtin_fma(ry, ry, rx * rx));
  const double y = rsqrt_f64(r2);
  const double ir2 = y * y;
  const double mimj = __builtin_fma(miz, mjz, __builtin_fma(miy, mjy, mix * mjx));
  const double mir = __builtin_fma(miz, rz, __builtin_fma(miy, ry, mix * rx));
  const double mjr = __builtin_fma(mjz, rz, __builtin_fma(mjy, ry, mjx * rx));
  const double num = __builtin_fma(-3.0 * ir2, mir * mjr, mimj);
  return num * (ir2 * y) * 0.0795774715459476679;   // 1/(4 pi)
}

template <typename T> __device__ __forceinline__ T store_phi(double u) {   // phi ~ U(0, 2pi), u in [0,1)
  if constexpr (sizeof(T) == 2) return (T)(uint32_t)(u * 65536.0);
  else if constexpr (sizeof(T) == 4) return (T)u;
  else return (T)(6.28318530717958647692 * u);
}
template <typename T> __device__ __forceinline__ T store_theta(double u) { // theta ~ U(0, pi)
  if constexpr (sizeof(T) == 2) return (T)(uint32_t)(u * 65536.0);
  else if constexpr (sizeof(T) == 4) return (T)(0.5 * u);
  else return (T)(3.14159265358979323846 * u);
}
// an arbitrary angle in radians -> storage (the x0 start; f32/q16 keep phi mod 2 pi, theta as given)
template <typename T> __device__ __forceinline__ T store_phi_rad(double phi) {
  if constexpr (sizeof(T) == 8) return (T)phi;
  double t = phi / 6.28318530717958647692;
  t -= floor(t);
  if constexpr (sizeof(T) == 2) return (T)(uint32_t)fmin(65535.0, t * 65536.0);
  else return (T)t;
}
template <typename T> __device__ __forceinline__ T store_theta_rad(double theta) {
  if constexpr (sizeof(T) == 8) return (T)theta;
  else if constexpr (sizeof(T) == 2) return (T)(uint32_t)fmin(65535.0, fmax(0.0, theta / 3.14159265358979323846 * 65536.0));
  else return (T)(theta / 6.28318530717958647692);
}
template <typename T> __host__ __device__ inline double load_phi(T raw) {     // -> radians
  if constexpr (sizeof(T) == 2) return 6.28318530717958647692 * ((double)raw + 0.5) / 65536.0;
  else if constexpr (sizeof(T) == 4) return 6.28318530717958647692 * (double)raw;
  else return (double)raw;
}
template <typename T> __host__ __device__ inline double load_theta(T raw) {   // -> radians
  if constexpr (sizeof(T) == 2) return 3.14159265358979323846 * ((double)raw + 0.5) / 65536.0;
  else if constexpr (sizeof(T) == 4) return 6.28318530717958647692 * (double)raw;
  else return (double)raw;
}
// lattice index -> turns, exactly, without the quarter-rate v_cvt_f32_u32: 0x4B000000 | m is the
// float 2^23 + m
__device__ __forceinline__ float q16_theta_turns(uint32_t k) {  // (2k+1) 2^-18
  return __builtin_fmaf(__uint_as_float(0x4B000001u | (k << 1)), 0x1p-18f, -32.0f);
}
__device__ __forceinline__ float q16_phi_turns(uint32_t j) {    // (2j+1) 2^-17
  return __builtin_fmaf(__uint_as_float(0x4B000001u | (j << 1)), 0x1p-17f, -64.0f);
}
// round-to-nearest integer of (step * s) via the 1.5*2^23 magic constant (one fma, one integer sub)
__device__ __forceinline__ int q16_disp(float step, float s) {
  return (int)__float_as_uint(__builtin_fmaf(step, s, 12582912.0f)) - 0x4B400000;
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));
// {old, new} pairs: in f32 they map onto the packed VALU ops (v_pk_mul/add/fma_f32) with no shuffles
template <typename R> struct V2;
template <> struct V2<float> { using type = v2f; };
template <> struct V2<double> { using type = v2d; };
__device__ __forceinline__ v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2d pfma(v2d a, v2d b, v2d c) { return a * b + c; }  // unfused, like the oracle

// Angle representation per arithmetic type.
//   f64: radians, exactly the reference's variables (bit-reproduces the CPU oracle).
//   f32: TURNS (theta/2pi in [0, 1/2], phi/2pi in [0, 1)).  gfx950's v_sin_f32/v_cos_f32 take turns
//        and are accurate to 1.3e-7 absolute there (tools/ubench), phi wraps with one v_fract, and
//        no range reduction or 1/2pi pre-scale is ever needed.
// sin and cos of a double, both to < 1 ulp, in ~60 instructions (the library sincos spends ~280 on its
// double-double range reduction, and the f64 step needs four of them).  Reduction r = x - k pi/2 by
// three-part Cody-Waite with fused multiply-adds (pi/2 = HI + MID + LO, the k HI product is exact inside
// the fma; good far beyond any angle a chain reaches -- phi random-walks unwrapped, |phi| stays in the
// hundreds), then the classic minimax kernels on [-pi/4, pi/4] with the low part of r carried through
// (the fdlibm polynomials).  |x| >= 1e5 is first folded by whole turns (less accurate there; no chain
// gets near it).  sin(fl(pi)) = 1.2246e-16 and
// cos(fl(pi/2)) = 6.1e-17 come out as in glibc, which the clamp corner cases of the parity tests see.
__device__ __forceinline__ void sincos_f64(double x, double *s, double *c) {
  if (!(fabs(x) < 1.0e5)) {   // fold by whole turns: 2 pi = HI2 + LO2
    const double t = rint(x * 1.59154943091895345609e-01);
    x = __builtin_fma(-t, 2.44929359829470641435e-16, __builtin_fma(-t, 6.28318530717958623200e+00, x));
  }
  const double k = rint(x * 6.36619772367581382433e-01);                  // x * 2/pi
  const double r0 = __builtin_fma(-k, 1.57079632679489655800e+00, x);     // HI
  const double r = __builtin_fma(-k, 6.12323399573676603587e-17, r0);     // MID
  double y = __builtin_fma(-k, 6.12323399573676603587e-17, r0 - r);       // what the rounding of r dropped
  y = __builtin_fma(k, 1.49738490485916983e-33, y);                       // LO = -1.4974e-33
  const double z = r * r;
  // sin kernel
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double v = z * r;
  const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
  const double ks = r - ((z * (0.5 * y - v * ps) - y) - v * S1);
  // cos kernel
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double pc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double kc = w + (((1.0 - w) - hz) + (z * pc - r * y));
  const int q = (int)k & 3;
  const double sa = (q & 1) ? kc : ks, ca = (q & 1) ? ks : kc;
  *s = (q & 2) ? -sa : sa;
  *c = ((q + 1) & 2) ? -ca : ca;
}

// sin and cos of a double for the HOT LOOP of the f64 sweep: ~30 instructions instead of ~50, each result within
// 1.5 ulp (tools/mathcheck).  What is dropped against sincos_f64 is the low word of the reduced argument (r is carried
// as one double, itself correctly rounded: k HI is exact inside the first fma) and fdlibm's compensated final sums
// (plain Horner forms in fused multiply-adds).  Nothing here can move a chain off the oracle's trajectory: angles are
// stored and proposed without any trigonometry, and sin/cos only enter the energy difference and the Jacobian
// ratio of the Metropolis test, where a relative 1e-16 flips a decision with probability ~1e-16 per step; the
// running observables agree with the oracle's to ~1e-15 relative instead of bit for bit.
// BOUNDED: the caller guarantees 0 <= x <= pi (theta after the clamp), no huge-argument fold is compiled in.
// sin(0) = 0, sin(fl(pi)) = 1.2246e-16 and cos(fl(pi/2)) = 6.1e-17 exactly as sincos_f64 gives them.
// FOLD (unbounded arguments: phi random-walks unwrapped, inc/eap_chain.jl:232): arguments of |x| >= PSTAT_PHI_FOLD are first
// folded by whole turns.  The two-word reduction below is good far beyond that (k HI is exact inside its fma and the
// dropped third word contributes k * 1.5e-33: 1e-24 at |x| = 1e9; tools/mathcheck covers +-9e8), the bound only keeps k
// inside an int32.  The fold is seven instructions that no chain ever needs (|phi| ~ pi sqrt(steps / 3)); the f64 sweep
// therefore compiles its step loop twice and runs the FOLD = false copy whenever no chain of the wave can reach the
// bound within the segment (run_segment: max |phi| at fill + pi per remaining step).
#define PSTAT_PHI_FOLD 1.0e9
template <bool BOUNDED, bool FOLD = true>
__device__ __forceinline__ void sincos_fast_f64(double x, double *s, double *c) {
  if constexpr (!BOUNDED && FOLD) {
    if (!(fabs(x) < PSTAT_PHI_FOLD)) {   // fold by whole turns: 2 pi = HI2 + LO2
      const double t = rint(x * 1.59154943091895345609e-01);
      x = __builtin_fma(-t, 2.44929359829470641435e-16, __builtin_fma(-t, 6.28318530717958623200e+00, x));
    }
  }
  const double k = rint(x * 6.36619772367581382433e-01);                  // x * 2/pi
  const double r = __builtin_fma(-k, 6.12323399573676603587e-17, __builtin_fma(-k, 1.57079632679489655800e+00, x));
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2), S1);
  const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
#ifndef PSTAT_THETA_GENERIC_TAIL
  if constexpr (BOUNDED) {
    // theta in [0, pi]: k is 0, 1 or 2 and sin(theta) >= 0, so the quadrant logic shrinks to two compares on k itself, an
    // |.| and two negations (source modifiers) -- the same bits as the general form below gives on [0, pi]: there the sin
    // kernel's sign is applied to r before the polynomial, here after it (an odd function of r, evaluated in the same
    // operations), and for k = 0 the reduced r = theta is >= 0, for k = 2 it is theta - pi <= 0 (tools/mathcheck compares the two forms bit for bit).
    const double ks = __builtin_fma(r * z, ps, r);
    const double kc = __builtin_fma(z, __builtin_fma(z, pc, -0.5), 1.0);
    const bool one = k == 1.0, two = k == 2.0;
    const double cq = two ? -kc : kc;
    *s = one ? kc : __builtin_fabs(ks);
    *c = one ? -ks : cq;
    return;
  }
#endif
  // Quadrant q = k mod 4:  q  sin   cos     The sin kernel ks is odd in r, so its sign is applied to r BEFORE the
  //                        0  +ks   +kc     polynomial (one xor); kc's sign after it; then one swap.  Signs as
  //                        1  +kc   -ks     sign-bit masks straight from the bits of q: sign(kc) = bit 1 of q,
  //                        2  -ks   -kc     sign(ks) = bit 1 ^ bit 0.  11 integer/select instructions instead of the
  //                        3  -kc   +ks     ~15 that the obvious selects compile to, four times per step.
  const uint32_t q = (uint32_t)(int)k;
  const uint32_t odd = q << 31;                          // bit 0 of q in the sign position
  const uint32_t m_kc = (q >> 1) << 31;                  // bit 1 of q
  const uint32_t m_ks = m_kc ^ odd;
  const double rs = __hiloint2double((int)(__double2hiint(r) ^ m_ks), __double2loint(r));
  const double ks = __builtin_fma(rs * z, ps, rs);
  const double kc0 = __builtin_fma(z, __builtin_fma(z, pc, -0.5), 1.0);
  const double kc = __hiloint2double((int)(__double2hiint(kc0) ^ m_kc), __double2loint(kc0));
  const bool swap = (int)odd < 0;
  *s = swap ? kc : ks;
  *c = swap ? ks : kc;
}

// acos on [-1, 1] as sqrt(1 - |x|) * P7(|x|) (Abramowitz & Stegun 4.4.46, |error| <= 2e-8 -- below the
// f32 spacing of the result), mirrored for x < 0: a dozen full-rate ops and one v_sqrt
__device__ __forceinline__ float acos_r(float x) {
  const float a = fabsf(x);
  float p = -0.0012624911f;
  p = __builtin_fmaf(p, a, 0.0066700901f);
  p = __builtin_fmaf(p, a, -0.0170881256f);
  p = __builtin_fmaf(p, a, 0.0308918810f);
  p = __builtin_fmaf(p, a, -0.0501743046f);
  p = __builtin_fmaf(p, a, 0.0889789874f);
  p = __builtin_fmaf(p, a, -0.2145988016f);
  p = __builtin_fmaf(p, a, 1.5707963050f);
  const float r = __builtin_amdgcn_sqrtf(fmaxf(1.0f - a, 0.0f)) * p;
  return x < 0.0f ? 3.14159265358979f - r : r;
}
// f64: acos on [-1, 1] in ~35 instructions (the library's takes ~95 and the step needs up to eight).  With
// asin(t) = t + t z g(z), z = t^2 <= 1/4 and g a degree-12 polynomial (Chebyshev-node fit of (asin(sqrt z)/sqrt z - 1)/z
// on [0, 1/4], relative truncation error 3.6e-18):  |x| < 1/2: acos x = pi/2 - asin x;  |x| >= 1/2: with z = (1 - |x|)/2,
// acos |x| = 2 asin(sqrt z), mirrored for x < 0.  sqrt z by v_rsq_f64 + one Goldschmidt step.  Max error 1.2 ulp
// (tools/mathcheck); acos(1) = 0 and acos(-1) = fl(pi) exactly.  Bond angles feed the <psi> averager and, with
// --bend-mod, the bending energy kappa/2 (psi - psi0)^2, hence dU and the Metropolis decision of the clustering main:
// against the oracle's libm acos that perturbs dU by ~1 ulp of psi, the same 1e-16 class as the hot-loop sincos (a
// decision compares against a 23-bit uniform: it moves with probability ~1e-16).
__device__ __forceinline__ double acos_r(double x) {
  const double a = fabs(x);
  const bool big = a >= 0.5;
  const double z = big ? __builtin_fma(a, -0.5, 0.5) : a * a;
  double p = 2.87578513674215663354e-02;
  p = __builtin_fma(p, z, -1.48518870712472036977e-02);
  p = __builtin_fma(p, z, 1.74008794426940213707e-02);
  p = __builtin_fma(p, z, 5.45750671864035814818e-03);
  p = __builtin_fma(p, z, 1.03228143501857792808e-02);
  p = __builtin_fma(p, z, 1.14791774151849056834e-02);
  p = __builtin_fma(p, z, 1.39712129735529329289e-02);
  p = __builtin_fma(p, z, 1.73523927208699725588e-02);
  p = __builtin_fma(p, z, 2.23721729421498885526e-02);
  p = __builtin_fma(p, z, 3.03819441385312465076e-02);
  p = __builtin_fma(p, z, 4.46428571463554288434e-02);
  p = __builtin_fma(p, z, 7.49999999999843292020e-02);
  p = __builtin_fma(p, z, 1.66666666666666685170e-01);
  const double zp = z * p;
  // sqrt(z): y ~ 1/sqrt(z), then one coupled Goldschmidt step on (s, h) = (z y, y/2) and a final correction
  const double y = __builtin_amdgcn_rsq(z);
  const double s0 = z * y, h0 = 0.5 * y;
  const double r = __builtin_fma(-s0, h0, 0.5);
  const double s1 = __builtin_fma(s0, r, s0), h1 = __builtin_fma(h0, r, h0);
  double sq = __builtin_fma(__builtin_fma(-s1, s1, z), h1, s1);
  sq = z == 0.0 ? 0.0 : sq;                                  // (rsq(0) = inf)
  const double t = big ? sq : x;                             // asin(t) = t + t z g(z)
  const double as = __builtin_fma(t, zp, t);
  const double small = 1.57079632679489661923 - as;          // pi/2 - asin(x)
  const double two = as + as;                                // 2 asin(sqrt z) = acos |x|
  const double bigv = x < 0.0 ? 3.14159265358979323846 - two : two;
  return big ? bigv : small;
}

template <typename R> struct Ang;
template <> struct Ang<double> {
  static constexpr double theta_max = 3.14159265358979323846;
  static constexpr double unit = 1.0;  // radians per stored unit
  static __device__ __forceinline__ void sc(double x, double *s, double *c) { sincos_f64(x, s, c); }
  // the sweep's hot loop (see sincos_fast_f64): theta is clamped to [0, pi], phi random-walks
  static __device__ __forceinline__ void sc_theta(double x, double *s, double *c) { sincos_fast_f64<true>(x, s, c); }
  template <bool FOLD = true>
  static __device__ __forceinline__ void sc_phi(double x, double *s, double *c) { sincos_fast_f64<false, FOLD>(x, s, c); }
  static __device__ __forceinline__ double wrap(double x) { return x; }  // phi random-walks, eap_chain.jl:232
};
template <> struct Ang<float> {
  static constexpr float theta_max = 0.5f;
  static constexpr double unit = 6.28318530717958647692;
  static __device__ __forceinline__ void sc(float x, float *s, float *c) {
    *s = __builtin_amdgcn_sinf(x);   // v_sin_f32 / v_cos_f32 take turns
    *c = __builtin_amdgcn_cosf(x);
  }
  static __device__ __forceinline__ void sc_theta(float x, float *s, float *c) { sc(x, s, c); }
  template <bool FOLD = true>
  static __device__ __forceinline__ void sc_phi(float x, float *s, float *c) { sc(x, s, c); }
  static __device__ __forceinline__ float wrap(float x) { return __builtin_amdgcn_fractf(x); }
};


}  // namespace pstat
