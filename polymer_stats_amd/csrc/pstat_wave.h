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
  // The term of inc/eap_chain.jl:200-207 without its final 1/(4 pi), in the algebraic form of the f32 version:
  // [mu_i.mu_j - 3 (mu_i.r)(mu_j.r) / r^2] / r^3 with 1/r from v_rsq_f64 and one third-order correction (rsqrt_f64,
  // pstat_math.h: full double precision) instead of the literal sqrt + four divisions: ~27 instructions against ~75.  Rounding differs from the literal
  // form in the last bits only, which cannot move a chain off the oracle's trajectory (see sincos_fast_f64 in
  // pstat_math.h); r = 0 still gives NaN (inf * 0) and the proposal is rejected as in the reference.
  const double r2 = __builtin_fma(rz, rz, __builtin_fma(ry, ry, rx * rx));
  const double y = rsqrt_f64(r2);
  const double ir2 = y * y;
  const double mimj = __builtin_fma(miz, mjz, __builtin_fma(miy, mjy, mix * mjx));
  const double mir = __builtin_fma(miz, rz, __builtin_fma(miy, ry, mix * rx));
  const double mjr = __builtin_fma(mjz, rz, __builtin_fma(mjy, ry, mjx * rx));
  const double num = __builtin_fma(-3.0 * ir2, mir * mjr, mimj);
  return num * (ir2 * y);
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
// ringA: 128*M entries; ringB: 128*M entries of 16 bytes.  One wave per workgroup: LDS executes a wave's own ops in order.
// This form serves f64 (literal arithmetic of the reference) and f32 with one monomer per lane; f32 with
// M >= 2 uses ring_pair_sum_pk below (measured: +28 % at n = 100, +25 % at n = 200, but -10 % at n = 64).
template <typename R, int M, bool CUT, typename R4, typename R2>
__device__ __forceinline__ R ring_pair_sum(R4 *ringA, R2 *ringB_, const int lane, const int n, const R crad2,
                                           const R (&tx)[M], const R (&ty)[M], const R (&tz)[M],
                                           const R (&tmx)[M], const R (&tmy)[M], const R (&tmz)[M]) {
  const int L = (((n + M - 1) / M) + 1) & ~1;      // lanes in the ring, even, <= 64
  const bool in_ring = lane < L;
  // f32: (mu_y, mu_z) sit in 16-byte entries as well.  Two neighbouring 8-byte entries would be fetched by
  // one ds_read2_b64, which occupies the LDS array for 16 cycles per wave where a ds_read_b128 takes 4
  // (MI355X_MICROARCH.md, LDS table): with the padding a partner costs 8 LDS cycles instead of 12, and at
  // 3-4 waves per SIMD the array was busy more than half of the time.
  using RB = typename std::conditional<sizeof(R) == 4, R4, R2>::type;
  RB *ringB = reinterpret_cast<RB *>(ringB_);
  R4 va[M]; RB vb[M];
  __builtin_amdgcn_wave_barrier();                 // the previous sum's reads are done
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const bool real = lane * M + j < n;
    const R far = (R)1e6 * (R)(lane * M + j + 1);  // parking position of an unused slot
    va[j].x = real ? tx[j] : far; va[j].y = real ? ty[j] : (R)0; va[j].z = real ? tz[j] : (R)0;
    va[j].w = tmx[j];                              // callers keep the dipoles of unused slots at zero
    vb[j] = RB{};
    vb[j].x = tmy[j]; vb[j].y = tmz[j];
    if constexpr (sizeof(R) == 4) { vb[j].z = tmy[j]; vb[j].w = tmz[j]; }   // both halves are read: see accum
    if (in_ring) {
      if constexpr (sizeof(R) == 8) {
        // f64: a double4 entry would be read as two ds_read_b128 at a 32-byte lane stride -- two-way bank conflicts on
        // every read (measured: 38 % of the LDS-active cycles of the f64 interacting kernel).  (x, y) and (z, mu_x)
        // therefore live in two separate arrays of 16-byte entries, like (mu_y, mu_z).
        // With M >= 2 monomers per lane the entries are laid out MONOMER-MAJOR, [j][slot]: consecutive lanes then read
        // consecutive 16-byte entries (lane-major, a lane stride of 16 M bytes, gave two-way conflicts at M = 2 -- measured on
        // the clustering main's all-pairs kernel at n = 100: SQ_LDS_BANK_CONFLICT 37 % of SQ_LDS_IDX_ACTIVE).
        R2 *rxy = reinterpret_cast<R2 *>(ringA), *rzm = rxy + 128 * M;
        const R2 xy = {va[j].x, va[j].y}, zm = {va[j].z, va[j].w};
        rxy[j * 128 + lane] = xy; rxy[j * 128 + lane + L] = xy;
        rzm[j * 128 + lane] = zm; rzm[j * 128 + lane + L] = zm;
        ringB[j * 128 + lane] = vb[j]; ringB[j * 128 + lane + L] = vb[j];
      } else {
        ringA[lane * M + j] = va[j]; ringA[(lane + L) * M + j] = va[j];
        ringB[lane * M + j] = vb[j]; ringB[(lane + L) * M + j] = vb[j];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  R n3x[M], n3y[M], n3z[M];                        // f32: -3 mu of the lane's own monomers
#pragma unroll
  for (int j = 0; j < M; ++j) { n3x[j] = (R)-3 * va[j].w; n3y[j] = (R)-3 * vb[j].x; n3z[j] = (R)-3 * vb[j].y; }
  // t += the pair term of own monomer j with the entry (o, ob)
  auto accum = [&](R &t, const int j, const R4 &o, const RB &ob) __attribute__((always_inline)) {
    const R4 &a = va[j];
    const RB &ab = vb[j];
    const R dx = a.x - o.x, dy = a.y - o.y, dz = a.z - o.z;
    if constexpr (sizeof(R) == 4) {
      // 21 instructions and one v_rsq.  mu_j.r reads the entry's SECOND copy of (mu_y, mu_z): with all four
      // floats in use the compiler cannot narrow the read back to 8 bytes (and pair two of them into a
      // ds_read2_b64).
      const R r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const R ir = __builtin_amdgcn_rsqf(r2);
      const R ir2 = ir * ir;
      const R mimj = __builtin_fmaf(ab.y, ob.y, __builtin_fmaf(ab.x, ob.x, a.w * o.w));
      const R mir3 = __builtin_fmaf(n3z[j], dz, __builtin_fmaf(n3y[j], dy, n3x[j] * dx));
      const R mjr = __builtin_fmaf(ob.w, dz, __builtin_fmaf(ob.z, dy, o.w * dx));
      const R num = __builtin_fmaf(ir2 * mir3, mjr, mimj);
      const R tn = __builtin_fmaf(num, ir2 * ir, t);
      if constexpr (CUT) t = r2 > crad2 ? t : tn;
      else t = tn;
    } else {
      const R v = pair_fast(dx, dy, dz, a.w, ab.x, ab.y, o.w, ob.x, ob.y);
      if constexpr (CUT) t += dx * dx + dy * dy + dz * dz > crad2 ? (R)0 : v;
      else t += v;
    }
  };
  R e = 0;
#pragma unroll
  for (int j = 0; j < M; ++j)
#pragma unroll
    for (int jp = j + 1; jp < M; ++jp) accum(e, j, va[jp], vb[jp]);
  const int me = in_ring ? lane : 0;               // lanes outside the ring read valid entries, then drop the result
  const R4 *pa = ringA + me * M;                   // offsets stay non-negative: they fit the ds_read immediate
  const RB *pb = ringB + me * M;
  auto rotation = [&](const int k) __attribute__((always_inline)) -> R {
    R t = 0;
#pragma unroll
    for (int jp = 0; jp < M; ++jp) {
      R4 qa;
      RB qb;
      if constexpr (sizeof(R) == 8) {
        const R2 *rxy = reinterpret_cast<const R2 *>(ringA) + me, *rzm = rxy + 128 * M;
        const R2 xy = rxy[jp * 128 + (L - k)], zm = rzm[jp * 128 + (L - k)];
        qa.x = xy.x; qa.y = xy.y; qa.z = zm.x; qa.w = zm.y;
        qb = (ringB + me)[jp * 128 + (L - k)];
      } else {
        qa = pa[(L - k) * M + jp];
        qb = pb[(L - k) * M + jp];
      }
#pragma unroll
      for (int j = 0; j < M; ++j) accum(t, j, qa, qb);
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
// SIMD about twice as long as its scalar form (tools/ubench: 5.2 vs 2.4-3.4 cycles), so this buys little arithmetic
// throughput; what it buys is half the instruction stream and half the dependent accumulation chain per
// term, which is what limits these kernels at 1-2 waves per SIMD.  For the halves of a packed register to be two
// DIFFERENT partners without any shuffle, the ring stores the entries in PAIRS -- three float4 per two
// consecutive entries e, e + 1 (e even): (x_e, x_e+1, y_e, y_e+1), (z, z', mu_x, mu_x'), (mu_y, mu_y', mu_z,
// mu_z') -- and the partners of a lane, rotations 1 .. L/2, are the half*M CONSECUTIVE entries below its
// own slot in the doubled ring.  A pair of partners is then three ds_read_b128, each delivering two adjacent
// register pairs, i.e. 12 LDS-array cycles per wave for 2*M terms (MI355X_MICROARCH.md, LDS table; the
// struct-of-arrays form this replaced was read with ds_read2_b32/ds_read2_b64, which take 1 cycle per byte
// and lane: 48 cycles for the same data, and kept the LDS array busy ~60 % of the time at n = 100).  The
// lane's own monomer is broadcast to both halves by op_sel, which costs nothing.  The first M entries of
// the run belong to rotation L/2 (weight 1/2) and go to their own accumulator.
typedef float pk2 __attribute__((ext_vector_type(2)));

template <int M, bool CUT>
__device__ __forceinline__ float ring_pair_sum_pk(float4 *ringA, float2 *ringB, const int lane, const int n,
                                                  const float crad2, const float (&tx)[M], const float (&ty)[M],
                                                  const float (&tz)[M], const float (&tmx)[M],
                                                  const float (&tmy)[M], const float (&tmz)[M]) {
  static_assert(M % 2 == 0, "entries are staged in pairs");
  const int L = (((n + M - 1) / M) + 1) & ~1;      // lanes in the ring, even, <= 64
  const bool in_ring = lane < L;
  constexpr int NP = 64 * M;                       // entry pairs of the doubled ring (128*M entries)
  float4 *q0 = ringA, *q1 = ringA + NP, *q2 = reinterpret_cast<float4 *>(ringB);
  float ox[M], oy[M], oz[M];
  __builtin_amdgcn_wave_barrier();                 // the previous sum's reads are done
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const bool real = lane * M + j < n;
    ox[j] = real ? tx[j] : 1e6f * (float)(lane * M + j + 1);   // parking position of an unused slot
    oy[j] = real ? ty[j] : 0.0f; oz[j] = real ? tz[j] : 0.0f;
  }
  if (in_ring) {
#pragma unroll
    for (int j = 0; j < M; j += 2) {
      const int p0 = (lane * M + j) >> 1, p1 = p0 + ((L * M) >> 1);
      const float4 a = {ox[j], ox[j + 1], oy[j], oy[j + 1]};
      const float4 b = {oz[j], oz[j + 1], tmx[j], tmx[j + 1]};
      const float4 c = {tmy[j], tmy[j + 1], tmz[j], tmz[j + 1]};
      q0[p0] = a; q0[p1] = a; q1[p0] = b; q1[p1] = b; q2[p0] = c; q2[p1] = c;
    }
  }
  __builtin_amdgcn_wave_barrier();
  // scalar form: two monomers of this lane
  auto one = [&](const int j, const float px, const float py, const float pz, const float pmx, const float pmy,
                 const float pmz) __attribute__((always_inline)) -> float {
    const float dx = ox[j] - px, dy = oy[j] - py, dz = oz[j] - pz;
    const float t = pair_fast(dx, dy, dz, tmx[j], tmy[j], tmz[j], pmx, pmy, pmz);
    if constexpr (CUT) return dx * dx + dy * dy + dz * dz > crad2 ? 0.0f : t;
    else return t;
  };
  float n3x[M], n3y[M], n3z[M];                    // -3 mu of the lane's own monomers
#pragma unroll
  for (int j = 0; j < M; ++j) { n3x[j] = -3.0f * tmx[j]; n3y[j] = -3.0f * tmy[j]; n3z[j] = -3.0f * tmz[j]; }
  // packed form: every own monomer against the two entries of pair p
  auto two_entries = [&](const int p) __attribute__((always_inline)) -> pk2 {
    const float4 A = q0[p], B = q1[p], Cq = q2[p];
    const pk2 px = {A.x, A.y}, py = {A.z, A.w}, pz = {B.x, B.y};
    const pk2 qx = {B.z, B.w}, qy = {Cq.x, Cq.y}, qz = {Cq.z, Cq.w};
    pk2 t = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const pk2 ax = {ox[j], ox[j]}, ay = {oy[j], oy[j]}, az = {oz[j], oz[j]};
      const pk2 mx = {tmx[j], tmx[j]}, my = {tmy[j], tmy[j]}, mz = {tmz[j], tmz[j]};
      const pk2 kx = {n3x[j], n3x[j]}, ky = {n3y[j], n3y[j]}, kz = {n3z[j], n3z[j]};
      const pk2 dx = ax - px, dy = ay - py, dz = az - pz;
      const pk2 r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      const pk2 ir = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
      const pk2 ir2 = ir * ir;
      const pk2 mimj = __builtin_elementwise_fma(mz, qz, __builtin_elementwise_fma(my, qy, mx * qx));
      const pk2 mir3 = __builtin_elementwise_fma(kz, dz, __builtin_elementwise_fma(ky, dy, kx * dx));
      const pk2 mjr = __builtin_elementwise_fma(qz, dz, __builtin_elementwise_fma(qy, dy, qx * dx));
      const pk2 num = __builtin_elementwise_fma(ir2 * mir3, mjr, mimj);
      const pk2 tn = __builtin_elementwise_fma(num, ir2 * ir, t);
      if constexpr (CUT) { t.x = r2.x > crad2 ? t.x : tn.x; t.y = r2.y > crad2 ? t.y : tn.y; }
      else t = tn;
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
  int p = ((me + L - half) * M) >> 1;              // the run of partner pairs: [p, (me + L) M / 2)
  pk2 h = {0.0f, 0.0f};                            // rotation L/2: weight 1/2
#pragma unroll
  for (int q = 0; q < M / 2; ++q) h += two_entries(p + q);
  p += M / 2;
  pk2 acc = {0.0f, 0.0f};
  const int cnt = (half - 1) * (M / 2);            // full-weight pairs
  constexpr int UNR = M >= 8 ? 1 : (M >= 4 ? 2 : 4);
#pragma unroll UNR
  for (int q = 0; q < cnt; ++q) acc += two_entries(p + q);
  float tot = (es + (acc.x + acc.y)) + 0.5f * (h.x + h.y);
  tot = in_ring ? tot * 0.0795774715459476679f : 0.0f;   // 1/(4 pi)
  return wave_allsum<float>(tot);
}

}  // namespace pstat
