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

__device__ __forceinline__ float exp_r(float x) { return __expf(x); }
__device__ __forceinline__ double exp_r(double x) { return exp(x); }
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
  else return (R)2 * ((R)(w >> 9) * (R)(1.0 / 8388608.0)) - (R)1;
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
__device__ __forceinline__ double pair_term_fast(double rx, double ry, double rz, double mix, double miy,
                                                 double miz, double mjx, double mjy, double mjz) {
  return pair_term<double>(rx, ry, rz, mix, miy, miz, mjx, mjy, mjz);   // f64: the literal form
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
template <typename R> struct Ang;
template <> struct Ang<double> {
  static constexpr double theta_max = 3.14159265358979323846;
  static constexpr double unit = 1.0;  // radians per stored unit
  static __device__ __forceinline__ void sc(double x, double *s, double *c) { sincos(x, s, c); }
  static __device__ __forceinline__ double wrap(double x) { return x; }  // phi random-walks, eap_chain.jl:232
};
template <> struct Ang<float> {
  static constexpr float theta_max = 0.5f;
  static constexpr double unit = 6.28318530717958647692;
  static __device__ __forceinline__ void sc(float x, float *s, float *c) {
    *s = __builtin_amdgcn_sinf(x);   // v_sin_f32 / v_cos_f32 take turns
    *c = __builtin_amdgcn_cosf(x);
  }
  static __device__ __forceinline__ float wrap(float x) { return __builtin_amdgcn_fractf(x); }
};


}  // namespace pstat
