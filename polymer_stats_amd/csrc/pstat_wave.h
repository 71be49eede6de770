// pstat_wave.h -- wave-level helpers of the one-chain-per-wavefront kernels (pstat_interacting.hip,
// pstat_cluster_wave.hip).  Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>

namespace pstat {

template <typename R>
__device__ __forceinline__ R wave_allsum(R v) {  // butterfly: every lane ends with the same bits
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <typename R>
__device__ __forceinline__ R wave_incl_scan(R v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const R t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

template <typename R>
__device__ __forceinline__ R lane_value(R v, int src);
template <>
__device__ __forceinline__ float lane_value<float>(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
template <>
__device__ __forceinline__ double lane_value<double>(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// pair term with the f32 fast form (one v_rsq instead of sqrt + 4 divisions); the f64 form is
// the literal expression of inc/eap_chain.jl:200-207
__device__ __forceinline__ float pair_fast(float rx, float ry, float rz, float mix, float miy, float miz,
                                           float mjx, float mjy, float mjz) {
  const float r2 = __builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx));
  const float ir = __builtin_amdgcn_rsqf(r2);
  const float ir2 = ir * ir;
  const float mimj = __builtin_fmaf(miz, mjz, __builtin_fmaf(miy, mjy, mix * mjx));
  const float mir = __builtin_fmaf(miz, rz, __builtin_fmaf(miy, ry, mix * rx));
  const float mjr = __builtin_fmaf(mjz, rz, __builtin_fmaf(mjy, ry, mjx * rx));
  const float num = __builtin_fmaf(-3.0f * ir2, mir * mjr, mimj);
  return num * (ir2 * ir);   // x 1/(4 pi), applied once per sum
}
__device__ __forceinline__ double pair_fast(double rx, double ry, double rz, double mix, double miy,
                                            double miz, double mjx, double mjy, double mjz) {
  // literal form of inc/eap_chain.jl:200-207 without its final 1/(4 pi)
  const double r2 = rx * rx + ry * ry + rz * rz;
  const double rmag = sqrt(r2);
  const double hx = rx / rmag, hy = ry / rmag, hz = rz / rmag;
  const double r3 = r2 * rmag;
  const double mimj = mix * mjx + miy * mjy + miz * mjz;
  const double mir = mix * hx + miy * hy + miz * hz;
  const double mjr = mjx * hx + mjy * hy + mjz * hz;
  return (mimj - 3 * mir * mjr) / r3;
}

template <typename R>
__device__ __forceinline__ R wave_excl_scan(R v, int lane) { return wave_incl_scan<R>(v, lane) - v; }


}  // namespace pstat
