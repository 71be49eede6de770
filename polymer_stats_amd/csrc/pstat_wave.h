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


// Sum over all pairs of one configuration held M consecutive monomers per lane (x, mu per monomer), the
// n(n-1)/2 terms of U_interaction (inc/eap_chain.jl:196-211); with CUT, UCutoff's r^2 > rc^2 => 0 (:171-192).
//
// The ring: L = ceil(n / M) lanes (made even) carry the chain; every monomer's (x, mu) is staged twice in LDS,
// at entries e and e + L*M, so "lane i meets lane i-k" is a read at offset (L - k) from the lane's own
// slot.  Rotations k = 1 .. L/2 - 1 meet every pair of distinct lanes once, rotation L/2 meets its pairs
// from both ends (weight 1/2), pairs inside a lane are taken directly.  Sizing the ring to the chain
// instead of to the wave matters for the reference's own sweeps: n = 100 (M = 2) needs 25 rotations
// instead of 32, n = 200 (M = 4) likewise.  Slots past n inside the ring hold zero dipoles at distinct
// far-away positions and contribute exactly 0; lanes outside the ring are masked out of the sum.
// ringA/ringB: 128*M entries each.  One wave per workgroup: LDS executes a wave's own ops in order.
template <typename R, int M, bool CUT, typename R4, typename R2>
__device__ __forceinline__ R ring_pair_sum(R4 *ringA, R2 *ringB, const int lane, const int n, const R crad2,
                                           const R (&tx)[M], const R (&ty)[M], const R (&tz)[M],
                                           const R (&tmx)[M], const R (&tmy)[M], const R (&tmz)[M]) {
  const int L = (((n + M - 1) / M) + 1) & ~1;      // lanes in the ring, even, <= 64
  const bool in_ring = lane < L;
  R4 va[M]; R2 vb[M];
  __builtin_amdgcn_wave_barrier();                 // the previous sum's reads are done
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const bool real = lane * M + j < n;
    const R far = (R)1e6 * (R)(lane * M + j + 1);  // parking position of an unused slot
    va[j].x = real ? tx[j] : far; va[j].y = real ? ty[j] : (R)0; va[j].z = real ? tz[j] : (R)0;
    va[j].w = tmx[j];                              // callers keep the dipoles of unused slots at zero
    vb[j].x = tmy[j]; vb[j].y = tmz[j];
    if (in_ring) {
      ringA[lane * M + j] = va[j]; ringA[(lane + L) * M + j] = va[j];
      ringB[lane * M + j] = vb[j]; ringB[(lane + L) * M + j] = vb[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
  auto term = [&](const R4 &a, const R2 &ab, const R4 &o, const R2 &ob) __attribute__((always_inline)) -> R {
    const R dx = a.x - o.x, dy = a.y - o.y, dz = a.z - o.z;
    const R t = pair_fast(dx, dy, dz, a.w, ab.x, ab.y, o.w, ob.x, ob.y);
    if constexpr (CUT) return dx * dx + dy * dy + dz * dz > crad2 ? (R)0 : t;
    else return t;
  };
  R e = 0;
#pragma unroll
  for (int j = 0; j < M; ++j)
#pragma unroll
    for (int jp = j + 1; jp < M; ++jp) e += term(va[j], vb[j], va[jp], vb[jp]);
  const int me = in_ring ? lane : 0;               // lanes outside the ring read valid entries, then drop the result
  const R4 *pa = ringA + me * M;                   // offsets stay non-negative: they fit the ds_read immediate
  const R2 *pb = ringB + me * M;
  auto rotation = [&](const int k) __attribute__((always_inline)) -> R {
    R t = 0;
#pragma unroll
    for (int jp = 0; jp < M; ++jp) {
      const R4 qa = pa[(L - k) * M + jp];
      const R2 qb = pb[(L - k) * M + jp];
#pragma unroll
      for (int j = 0; j < M; ++j) t += term(va[j], vb[j], qa, qb);
    }
    return t;
  };
  const int half = L >> 1;
#pragma unroll 4
  for (int k = 1; k < half; ++k) e += rotation(k);
  e += (R)0.5 * rotation(half);
  e = in_ring ? e * (R)0.0795774715459476679 : (R)0;   // 1/(4 pi)
  return wave_allsum<R>(e);
}

}  // namespace pstat
