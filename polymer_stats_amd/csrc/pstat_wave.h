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
// This form serves f64 (literal arithmetic of the reference) and f32 with one monomer per lane; f32 with
// M >= 2 uses ring_pair_sum_pk below (measured: +28 % at n = 100, +25 % at n = 200, but -10 % at n = 64).
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

// f32 form of ring_pair_sum for M >= 2: TWO partner monomers per instruction.  A v_pk_*_f32 occupies the
// SIMD twice as long as its scalar form (tools/ubench: 4.5 vs 2.1 cycles), so this buys no arithmetic
// throughput; what it buys is half the instruction stream and half the dependent accumulation chain per
// term, which is what limits these kernels at 1-2 waves per SIMD.  For the halves of a packed register to be two
// DIFFERENT partners without any shuffle, the ring is struct-of-arrays -- six float arrays x, y, z,
// mu_x, mu_y, mu_z in the same LDS bytes -- and the partners of a lane, rotations 1 .. L/2, are the
// half*M CONSECUTIVE entries below its own slot in the doubled ring: two neighbouring entries come out
// of one ds_read2_b32 per array as an adjacent register pair.  The lane's own monomer is broadcast to
// both halves by op_sel, which costs nothing.  The first M entries of the run belong to rotation L/2
// (weight 1/2) and go to their own accumulator; an odd leftover entry (M = 1 only) takes the scalar form.
typedef float pk2 __attribute__((ext_vector_type(2)));

template <int M, bool CUT>
__device__ __forceinline__ float ring_pair_sum_pk(float4 *ringA, float2 *ringB, const int lane, const int n,
                                                  const float crad2, const float (&tx)[M], const float (&ty)[M],
                                                  const float (&tz)[M], const float (&tmx)[M],
                                                  const float (&tmy)[M], const float (&tmz)[M]) {
  const int L = (((n + M - 1) / M) + 1) & ~1;      // lanes in the ring, even, <= 64
  const bool in_ring = lane < L;
  float *sx = reinterpret_cast<float *>(ringA), *sy = sx + 128 * M, *sz = sy + 128 * M, *smx = sz + 128 * M;
  float *smy = reinterpret_cast<float *>(ringB), *smz = smy + 128 * M;
  float ox[M], oy[M], oz[M];
  __builtin_amdgcn_wave_barrier();                 // the previous sum's reads are done
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const bool real = lane * M + j < n;
    ox[j] = real ? tx[j] : 1e6f * (float)(lane * M + j + 1);   // parking position of an unused slot
    oy[j] = real ? ty[j] : 0.0f; oz[j] = real ? tz[j] : 0.0f;
    if (in_ring) {
      const int e0 = lane * M + j, e1 = e0 + L * M;
      sx[e0] = ox[j]; sx[e1] = ox[j]; sy[e0] = oy[j]; sy[e1] = oy[j]; sz[e0] = oz[j]; sz[e1] = oz[j];
      smx[e0] = tmx[j]; smx[e1] = tmx[j]; smy[e0] = tmy[j]; smy[e1] = tmy[j]; smz[e0] = tmz[j]; smz[e1] = tmz[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
  // scalar form: own monomer j against one stored entry, or two monomers of this lane
  auto one = [&](const int j, const float px, const float py, const float pz, const float pmx, const float pmy,
                 const float pmz) __attribute__((always_inline)) -> float {
    const float dx = ox[j] - px, dy = oy[j] - py, dz = oz[j] - pz;
    const float t = pair_fast(dx, dy, dz, tmx[j], tmy[j], tmz[j], pmx, pmy, pmz);
    if constexpr (CUT) return dx * dx + dy * dy + dz * dz > crad2 ? 0.0f : t;
    else return t;
  };
  auto one_entry = [&](const int e) __attribute__((always_inline)) -> float {
    const float px = sx[e], py = sy[e], pz = sz[e], pmx = smx[e], pmy = smy[e], pmz = smz[e];
    float t = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) t += one(j, px, py, pz, pmx, pmy, pmz);
    return t;
  };
  // packed form: every own monomer against the two entries e, e + 1
  auto two_entries = [&](const int e) __attribute__((always_inline)) -> pk2 {
    const pk2 px = {sx[e], sx[e + 1]}, py = {sy[e], sy[e + 1]}, pz = {sz[e], sz[e + 1]};
    const pk2 qx = {smx[e], smx[e + 1]}, qy = {smy[e], smy[e + 1]}, qz = {smz[e], smz[e + 1]};
    pk2 t = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const pk2 ax = {ox[j], ox[j]}, ay = {oy[j], oy[j]}, az = {oz[j], oz[j]};
      const pk2 mx = {tmx[j], tmx[j]}, my = {tmy[j], tmy[j]}, mz = {tmz[j], tmz[j]};
      const pk2 dx = ax - px, dy = ay - py, dz = az - pz;
      const pk2 r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      const pk2 ir = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
      const pk2 ir2 = ir * ir;
      const pk2 mimj = __builtin_elementwise_fma(mz, qz, __builtin_elementwise_fma(my, qy, mx * qx));
      const pk2 mir = __builtin_elementwise_fma(mz, dz, __builtin_elementwise_fma(my, dy, mx * dx));
      const pk2 mjr = __builtin_elementwise_fma(qz, dz, __builtin_elementwise_fma(qy, dy, qx * dx));
      const pk2 m3 = {-3.0f, -3.0f};
      const pk2 num = __builtin_elementwise_fma(m3 * ir2, mir * mjr, mimj);
      pk2 v = num * (ir2 * ir);
      if constexpr (CUT) { v.x = r2.x > crad2 ? 0.0f : v.x; v.y = r2.y > crad2 ? 0.0f : v.y; }
      t += v;
    }
    return t;
  };
  float es = 0;                                    // scalar-form terms, full weight
#pragma unroll
  for (int j = 0; j < M; ++j)
#pragma unroll
    for (int jp = j + 1; jp < M; ++jp) es += one(j, ox[jp], oy[jp], oz[jp], tmx[jp], tmy[jp], tmz[jp]);
  const int half = L >> 1;
  const int me = in_ring ? lane : 0;               // lanes outside the ring read valid entries, then drop the result
  int e = (me + L - half) * M;                     // the run of partner entries: [e, (me + L) M)
  float eh = 0;                                    // rotation L/2: weight 1/2
  pk2 acc = {0.0f, 0.0f};
  if constexpr (M == 1) { eh = one_entry(e); e += 1; }
  else {
    pk2 h = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < M; q += 2) h += two_entries(e + q);
    eh = h.x + h.y;
    e += M;
  }
  const int cnt = (half - 1) * M;                  // full-weight entries
  constexpr int UNR = M >= 8 ? 1 : (M >= 4 ? 2 : 4);
#pragma unroll UNR
  for (int q = 0; q < (cnt >> 1); ++q) acc += two_entries(e + 2 * q);
  if constexpr (M == 1) {
    if (cnt & 1) es += one_entry(e + cnt - 1);
  }
  float tot = (es + (acc.x + acc.y)) + 0.5f * eh;
  tot = in_ring ? tot * 0.0795774715459476679f : 0.0f;   // 1/(4 pi)
  return wave_allsum<float>(tot);
}

}  // namespace pstat
