// pstat_kernels.hip -- gfx950 (CDNA4) kernels for the fixed-force MCMC hot path.
//
// Mapping: ONE INDEPENDENT MARKOV CHAIN PER WAVEFRONT LANE.  A workgroup is
// a single wave64; lane l of workgroup g owns chain g*lanes + l for the whole launch.
//   * the chain's orientation angles (theta_i, phi_i), i < n, are "cells" [monomer][lane]: lane l only ever touches
//     column l, so no barrier or cross-lane traffic is needed.  f32 / q16 and short f64 chains keep every cell in
//     LDS (a per-lane random monomer index is bank-conflict-free); the f64 sweep of longer chains (n > 40) keeps the
//     first 39 rows in LDS and the rest in a global working buffer that it reads two steps ahead through L2 /
//     Infinity Cache (run_segment, ST = 2);
//   * the generator (MWC64X or xoshiro128++), step sizes, adaptation counters, end-to-end vector r, dipole p,
//     energy U and the running sums stay in registers;
//   * HBM holds the checkpoint layout (struct-of-arrays over the chain index); it is touched only to fill the
//     cells/registers at launch start and to spill them at the end.
// There is no dense contraction anywhere on this path, hence no MFMA.
//
// Reference semantics implemented (file:line relative to the reference tree):
//   proposal draw ............ mcmc_eap_chain.jl:277-280
//   move! (force ensemble) ... inc/eap_chain.jl:230-257   (O(1) energy difference instead of the
//                               deep copy :137-163 + full recompute :252-254)
//   dipole response .......... inc/dipole_response.jl:7-29
//   energies ................. inc/energy.jl:7-23, inc/eap_chain.jl:53,215-228
//   Metropolis ............... inc/acceptance.jl:18-39
//   adaptation ............... mcmc_eap_chain.jl:301-322
//   averagers ................ inc/average.jl:38-48,63-67,99-124; mcmc_eap_chain.jl:242-255,327-328
#include "pstat_device.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>

#include "../../include/pstat.h"
#include "pstat_math.h"

#ifndef PSTAT_GM_DEPTH
#define PSTAT_GM_DEPTH 2   // prefetch depth (steps) of the f64 non-interacting sweep with its cells in memory: 2 or 3
#endif
#ifndef PSTAT_GI_DEPTH
#define PSTAT_GI_DEPTH 1   // prefetch depth (steps) of the f64 Ising sweep with its cells in memory
#endif
#ifndef PSTAT_GI_CC
#define PSTAT_GI_CC 1      // f64 Ising sweep with its cells in memory: chain-contiguous working buffer
#endif
#ifndef PSTAT_UNROLL_ISING
#define PSTAT_UNROLL_ISING 16  // the Ising step is ~3x as long
#endif
#ifndef PSTAT_UNROLL
#define PSTAT_UNROLL 16  // steps per basic block of the sweep loop (even; measured: 8 -> 16 +1.2 %, 32 -4 %)
#endif

namespace pstat {

#ifndef PSTAT_PART   // (the per-precision sweep objects are built from this same file with -DPSTAT_PART=1|2|3)
// ------------------------------------------------------------------------------------------ init

// EAPChain(pargs), inc/eap_chain.jl:60-135: all phi draws, then all theta draws; then r, p, U.
// One thread per chain; angles are rounded to the storage type R before anything is derived.
template <typename R, typename G>
__global__ void init_kernel(SweepArgs A, DevState S, const CaseConst *__restrict__ cases,
                            int chain_type, int energy_type, double phi_step, double theta_step,
                            InitOpts io) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S.C) return;
  const int64_t icase = c / A.chains_per_case, local = c % A.chains_per_case;
  const CaseConst cc = cases[icase];
  R *th = (R *)S.ang, *ph = (R *)S.ang + A.n * S.C;
  G g;
  g.seed(cc.seed, cc.chain_id0 + (uint64_t)local);
  if (io.use_x0 == 2) {  // per-monomer x0 (interleaved) plus Uniform(0, dx0), inc/eap_chain.jl:73-75
    for (int64_t i = 0; i < A.n; ++i)
      ph[i * S.C + c] = store_phi_rad<R>(io.x0_vec[2 * i] + io.dx0_phi * u01<double>(g.next()));
    for (int64_t i = 0; i < A.n; ++i)
      th[i * S.C + c] = store_theta_rad<R>(io.x0_vec[2 * i + 1] + io.dx0_theta * u01<double>(g.next()));
  } else if (io.use_x0) {  // x0 = [phi; theta] plus Uniform(0, dx0), inc/eap_chain.jl:69-72
    for (int64_t i = 0; i < A.n; ++i)
      ph[i * S.C + c] = store_phi_rad<R>(io.x0_phi + io.dx0_phi * u01<double>(g.next()));
    for (int64_t i = 0; i < A.n; ++i)
      th[i * S.C + c] = store_theta_rad<R>(io.x0_theta + io.dx0_theta * u01<double>(g.next()));
  } else {
    for (int64_t i = 0; i < A.n; ++i) ph[i * S.C + c] = store_phi<R>(u01<double>(g.next()));
    for (int64_t i = 0; i < A.n; ++i) th[i * S.C + c] = store_theta<R>(u01<double>(g.next()));
  }

  double r[3] = {0, 0, 0}, p[3] = {0, 0, 0}, usum = 0, upair = 0, omega = 0, c2sum = 0, psisum = 0;
  double pnx = 0, pny = 0, pnz = 0, pmx = 0, pmy = 0, pmz = 0;
  for (int64_t i = 0; i < A.n; ++i) {
    double t = load_theta<R>(th[i * S.C + c]), f = load_phi<R>(ph[i * S.C + c]);
    double st = sin(t), ct = cos(t), sp = sin(f), cp = cos(f);
    double nx = cp * st, ny = sp * st, nz = ct, mx, my, mz;
    if (chain_type == PSTAT_DIELECTRIC)
      dipole<double, PSTAT_DIELECTRIC>((cc.K1 - cc.K2) * cc.E0, cc.K2 * cc.E0, nx, ny, nz, mx, my, mz);
    else
      dipole<double, PSTAT_POLAR>(cc.mu, 0.0, nx, ny, nz, mx, my, mz);
    r[0] += cc.b * nx; r[1] += cc.b * ny; r[2] += cc.b * nz;
    p[0] += mx; p[1] += my; p[2] += mz;
    usum += -0.5 * cc.E0 * mz;
    omega += log(st);
    c2sum += ct * ct;
    if (i > 0) {  // bond (i-1, i): angle psi and its bending energy, inc/eap_chain.jl:45-47,54-58
      const double psi = acos(fmin(1.0, fmax(-1.0, pnx * nx + pny * ny + pnz * nz)));
      psisum += psi;
      usum += cc.kappa / 2 * (psi - cc.psi0) * (psi - cc.psi0);
    }
    if (energy_type == PSTAT_ISING && i > 0) {
      double h = -cc.b / 2;
      upair += pair_term<double>(h * (pnx + nx), h * (pny + ny), h * (pnz + nz), pmx, pmy, pmz, mx, my, mz);
    }
    pnx = nx; pny = ny; pnz = nz; pmx = mx; pmy = my; pmz = mz;
  }
  double U = usum + upair - (r[0] * cc.Fx + r[2] * cc.Fz);
  S.obs[OBS_R1 * S.C + c] = r[0]; S.obs[OBS_R2 * S.C + c] = r[1]; S.obs[OBS_R3 * S.C + c] = r[2];
  S.obs[OBS_P1 * S.C + c] = p[0]; S.obs[OBS_P2 * S.C + c] = p[1]; S.obs[OBS_P3 * S.C + c] = p[2];
  S.obs[OBS_U * S.C + c] = U; S.obs[OBS_USUM * S.C + c] = usum;
  S.obs[OBS_C2 * S.C + c] = c2sum; S.obs[OBS_PSI * S.C + c] = psisum;
  g.store(S.rng + c, S.C);
  S.stepsz[0 * S.C + c] = phi_step; S.stepsz[1 * S.C + c] = theta_step;
  S.win[0 * S.C + c] = 0; S.win[1 * S.C + c] = 0;
  S.nacc_total[c] = 0;
  for (int q = 0; q < NSUMS; ++q) S.sums[q * S.C + c] = 0.0;
  S.wnorm[c] = 0.0;
  S.lag[c] = 0.0;
  S.uref[c] = usum;
  S.nanrej[c] = 0;
  (void)omega;
}

#endif  // !PSTAT_PART

// ------------------------------------------------------------------------------------------ sweep

struct Draw {  // raw words of one step's proposal, mcmc_eap_chain.jl:277-280,287
  uint32_t idx, cell, wphi, wth, weps, wflip;   // cell = byte offset of the monomer's LDS slot [idx][lane]
  uint32_t w0;                                  // the index draw's raw word (its low bits feed the 53-bit eps, pstat_math.h)
};

template <bool RARE, typename G>
__device__ __forceinline__ Draw draw_step(G &g, uint32_t n, bool flips, uint32_t row_bytes, uint32_t lane_bytes) {
  Draw d;
  d.w0 = g.next();
  d.idx = __umulhi(d.w0, n);
  d.cell = __umul24(d.idx, row_bytes) + lane_bytes;   // one v_mad_u32_u24 (idx < 2^24, a row < 2^24 bytes)
  d.wphi = g.next();
  d.wflip = 0;
  if constexpr (RARE) {
    if (flips) d.wflip = g.next();
  }
  d.wth = g.next();
  d.weps = g.next();
  return d;
}

struct SweepRare {  // wave-uniform switches of the rarely used options (RARE instantiations only)
  int flips;        // --do-flips
  int lag;          // a re-init happened: the acceptor's cached log-pi may be offset (see reinit_kernel)
  int umb;          // --umbrella-sampling: AntiDipoleWeightFunction + UmbrellaAverager
};

// One time-segment of one chain block: fill LDS/registers from HBM, run `nsteps` steps, spill.
template <typename R, typename G, int CT, int EN, bool FX, bool RARE, int ST>
__device__ __forceinline__ void run_segment(const SweepArgs &A, const DevState &S, const CaseConst &cc,
                                            const SweepRare rare, unsigned char *smem, const int lane,
                                            const int64_t c, int64_t step, int64_t remaining, const int blk) {
  using R2 = typename Vec2<R>::type;
  using AG = Ang<R>;
  // ST = 0: the LDS cell is the (theta, phi) pair in R.  ST = 1 (PSTAT_Q16, R = float): the cell is
  // one 32-bit word, theta lattice index in the low half and phi index in the high half.
  // ST = 2 (R = double): the cells live in GLOBAL memory -- DevState::work, [chain block][n][64] double2 -- and are
  // random-accessed through L2 / the Infinity Cache.  An f64 cell is 16 bytes: at n = 100 LDS seats 102 chains per CU
  // (two 51-lane waves on two of the four SIMDs); with the cells in memory every SIMD carries a full 64-lane wave.
  // The f64 step is ~470 instructions (~2300 cycles) long, so the one-step-ahead prefetch of the pipeline below
  // covers an Infinity-Cache hit (~550 cycles) several times over.
  constexpr bool Q = ST == 1;
  constexpr bool GM = ST == 2;
  static_assert(!Q || sizeof(R) == 4, "the lattice state runs on f32 arithmetic");
  static_assert(!GM || sizeof(R) == 8, "global-memory state: f64");
  // GM pipeline depth: 2 steps for the non-interacting step; the Ising step is ~2.5x as long and fetches three rows (the
  // monomer and its two neighbours), so one step ahead gives its loads the same time and a third of the registers
  constexpr bool GI = GM && EN == PSTAT_ISING;
  constexpr int DEPTH = GI ? PSTAT_GI_DEPTH : PSTAT_GM_DEPTH;
  using Cell = typename std::conditional<Q, uint32_t, R2>::type;
  const int lanes = GM ? 64 : A.lanes;
  unsigned char *const cells = GM ? reinterpret_cast<unsigned char *>(S.work) + (size_t)blk * (size_t)A.n * 64 * sizeof(Cell) : smem;
  Cell *ang = reinterpret_cast<Cell *>(cells);  // [n][lanes]
  const int64_t C = S.C;
  const int n = (int)A.n;

  // ---- per-case scalars (wave-uniform => SGPRs).  Step sizes are kept in radians (f64) for the
  // adaptation logic and converted to the storage unit when they change.
  const R Fz = (R)cc.Fz, Fx = (R)cc.Fx, b = (R)cc.b, kT = (R)cc.kT;
  const R a_or_mu = (CT == PSTAT_DIELECTRIC) ? (R)((cc.K1 - cc.K2) * cc.E0) : (R)cc.mu;
  const R k2e = (R)(cc.K2 * cc.E0);
  const R mhalfE0 = (R)(-0.5 * cc.E0);
  const R hb = (R)(-cc.b / 2);
  // f32 Ising: see the step.  (--mlen 0 puts every monomer on one point: the reference's pair term is 0 * inf = NaN there, not
  // the +-inf that scaling a finite sum by 1 / 0 would give -- and -inf would be accepted.)
  const R ising_scale = cc.b == 0.0 ? (R)__builtin_nan("") : (R)(0.0795774715459476679 / fabs(cc.b * cc.b * cc.b / 8));
  const R nbeta_log2e = (R)(-1.4426950408889634 / cc.kT);  // exp(-dU/kT) = exp2(dU * this)
  const double ninv_kT = -1.0 / cc.kT;
  (void)ninv_kT;
  (void)Fx; (void)kT; (void)hb; (void)nbeta_log2e; (void)ising_scale;

  // ---- fill
  R maxphi = 0;      // f64: the largest |phi| of the lane's chain as filled (chooses the step loop's sincos form, see below)
  (void)maxphi;
  if constexpr (Q) {
    const uint16_t *gth = (const uint16_t *)S.ang, *gph = (const uint16_t *)S.ang + (int64_t)n * C;
#pragma unroll 8
    for (int i = 0; i < n; ++i)
      ang[i * lanes + lane] = (uint32_t)gth[(int64_t)i * C + c] | ((uint32_t)gph[(int64_t)i * C + c] << 16);
  } else {
    const R *gth = (const R *)S.ang, *gph = (const R *)S.ang + (int64_t)n * C;
#pragma unroll 8   // 16 independent loads in flight per batch
    for (int i = 0; i < n; ++i) {
      R2 v;
      v.x = gth[(int64_t)i * C + c];
      v.y = gph[(int64_t)i * C + c];
      if constexpr (sizeof(R) == 8) maxphi = fmax(maxphi, fabs(v.y));
      if (GM && i < A.lds_rows) reinterpret_cast<Cell *>(smem)[i * lanes + lane] = v;
      else if constexpr (GM) *reinterpret_cast<Cell *>(cells + ((EN == PSTAT_ISING && PSTAT_GI_CC) ? (uint32_t)lane * (uint32_t)n * (uint32_t)sizeof(Cell) + (uint32_t)i * (uint32_t)sizeof(Cell)
                                                                                                          : (uint32_t)(i * lanes + lane) * (uint32_t)sizeof(Cell))) = v;
      else ang[i * lanes + lane] = v;
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
  int nnan_seg = 0;   // proposals with a non-finite energy difference (only the pair energies can produce one)
  // observables O = (rx, ry | rz, px | py, pz | U, unused) as four 2-vectors (packed f32 math)
  R O[8];
  O[0] = (R)S.obs[OBS_R1 * C + c]; O[1] = (R)S.obs[OBS_R2 * C + c]; O[2] = (R)S.obs[OBS_R3 * C + c];
  O[3] = (R)S.obs[OBS_P1 * C + c]; O[4] = (R)S.obs[OBS_P2 * C + c]; O[5] = (R)S.obs[OBS_P3 * C + c];
  O[6] = (R)S.obs[OBS_U * C + c]; O[7] = 0;
  R usum = (R)S.obs[OBS_USUM * C + c];
  R lag = RARE ? (R)S.lag[c] : (R)0;
  // umbrella sampling (inc/average.jl:104-124): w = sum(u) * wscale - log_gauge enters log pi, and
  // every record is weighted by 1/e^w.  Only w - w(first configuration) is ever needed here.
  const bool umb = RARE && rare.umb;
  const R wscale = umb ? (R)((0.2 + 0.8 * exp(-(cc.Fx * cc.Fx + cc.Fz * cc.Fz) / cc.kT)) / cc.kT) : (R)0;
  R uref = umb ? (R)S.uref[c] : (R)0;
  bool regauged = false;
  double wnorm = umb ? S.wnorm[c] : 0.0;
  double sums[NSUMS_BASE];
#pragma unroll
  for (int q = 0; q < NSUMS_BASE; ++q) sums[q] = S.sums[q * C + c];

  const int64_t spa = A.steps_per_adjust;
  int64_t to_adj = A.adaptive ? spa - (step % spa) : 0;
  constexpr int FLUSH = 128;  // f32 partial sums are folded into the f64 sums this often
  const bool flips = RARE && rare.flips;

  // software pipeline: the draws and the LDS row of the NEXT step are fetched while the current
  // step computes; if both steps hit the same monomer and the current one is accepted, the
  // prefetched row is replaced by the freshly accepted angles.
  using P = typename V2<R>::type;   // a 2-vector: (x, y), {n_z, mu_z}, (theta, phi) or {old, new}
  const uint32_t row_bytes = (uint32_t)lanes * (uint32_t)sizeof(Cell), lane_bytes = (uint32_t)lane * (uint32_t)sizeof(Cell);
  auto slot = [&](uint32_t off) __attribute__((always_inline)) -> Cell & { return *reinterpret_cast<Cell *>(cells + off); };
  // GM: where monomer idx of this lane's chain lives in the global working buffer.  [n][64] (a 1 KiB row per monomer)
  // for the non-interacting step, which touches one cell; CHAIN-CONTIGUOUS [lane][n] for the Ising step (PSTAT_GI_CC),
  // whose three cells -- the monomer and its two neighbours -- then share one or two 128-byte lines instead of
  // lying in three rows (measured at n = 200: fabric traffic 267 -> ~120 bytes per update; the kernel was bound by it).
  constexpr bool CC = GI && PSTAT_GI_CC;
  const uint32_t chain_g = (uint32_t)lane * (uint32_t)n * (uint32_t)sizeof(Cell);
  auto gofs = [&](const uint32_t idx) __attribute__((always_inline)) -> uint32_t {
    return CC ? chain_g + idx * (uint32_t)sizeof(Cell) : idx * row_bytes + lane_bytes;
  };
  // GM: rows [0, nL) of the wave's cells sit in LDS (as many as four resident waves per CU can share), the rest in
  // memory.  One CU sustains only about three waves' worth of divergent 16-byte accesses per step (measured: 3 waves per
  // CU run at full speed, the 4th stretches every step by 50 %), so every access kept on chip counts.  Which home a
  // lane's monomer has differs from lane to lane; to keep the step ONE basic block (branches would cost the exact
  // vmcnt/lgkmcnt counts that the prefetch depends on) every step issues both a ds_read and a buffer_load, both a
  // ds_write and a buffer_store, and steers each lane by its ADDRESS: the memory side goes through a buffer resource
  // whose bounds check drops an out-of-range offset (no traffic, a load returns 0), the LDS side is pointed at a trash
  // row.  A rejected step is steered away on both sides, so it stores nothing.
  typedef int v4i __attribute__((ext_vector_type(4)));
  struct RowG { Cell l; v4i g; };
  struct RowG3 { RowG c, lo, hi; };     // Ising: the monomer, its lower and its upper neighbour
  using Row = typename std::conditional<GI, RowG3, typename std::conditional<GM, RowG, Cell>::type>::type;
  const uint32_t nL = GM ? (uint32_t)A.lds_rows : 0u;
  const uint32_t trash = nL * row_bytes + lane_bytes;        // LDS row nL: never read for its contents
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(GM ? (void *)cells : (void *)nullptr, 0,
                                                                  GM ? (int)((uint32_t)n * row_bytes) : 0, 0x00020000);
  auto rdcell = [&](const uint32_t idx, const uint32_t cell) __attribute__((always_inline)) -> RowG {
    const bool inL = idx < nL;
    RowG r;
    r.l = *reinterpret_cast<Cell *>(smem + (inL ? cell : trash));
    r.g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, inL ? 0xFFFFFFFFu : gofs(idx), 0, 0);
    return r;
  };
  // a fetched cell at its first use: its home, then the commits made after its load was issued (fw2 older, fw1 newer)
  auto resolve = [&](const RowG &r, const uint32_t idx, const uint32_t cell, const Cell &f1, const uint32_t at1,
                     const Cell &f2, const uint32_t at2, const Cell &f3, const uint32_t at3) __attribute__((always_inline)) -> Cell {
    if constexpr (GM) {
      typedef double v2dd __attribute__((ext_vector_type(2)));
      const v2dd gv = __builtin_bit_cast(v2dd, r.g);
      const bool inL = idx < nL;
      Cell a;
      a.x = inL ? r.l.x : gv.x; a.y = inL ? r.l.y : gv.y;
      if constexpr (DEPTH == 3) {
        const bool s3 = at3 == cell;
        a.x = s3 ? f3.x : a.x; a.y = s3 ? f3.y : a.y;
      }
      if constexpr (DEPTH >= 2) {
        const bool s2 = at2 == cell;
        a.x = s2 ? f2.x : a.x; a.y = s2 ? f2.y : a.y;
      }
      const bool s1 = at1 == cell;
      a.x = s1 ? f1.x : a.x; a.y = s1 ? f1.y : a.y;
      return a;
    } else {
      return r.l;
    }
  };
  auto rdrow = [&](const Draw &d) __attribute__((always_inline)) -> Row {
    if constexpr (GI) {
      // at a chain end the missing neighbour's slot re-reads the monomer itself; its bond is masked out in the step
      const bool hasLo = d.idx > 0, hasHi = d.idx + 1 < (uint32_t)n;
      RowG3 r;
      r.c = rdcell(d.idx, d.cell);
      r.lo = rdcell(hasLo ? d.idx - 1 : d.idx, hasLo ? d.cell - row_bytes : d.cell);
      r.hi = rdcell(hasHi ? d.idx + 1 : d.idx, hasHi ? d.cell + row_bytes : d.cell);
      return r;
    } else if constexpr (GM) {
      return rdcell(d.idx, d.cell);
    } else {
      return slot(d.cell);
    }
  };
  auto wr = [&](const Draw &d, const Cell v, const bool ok) __attribute__((always_inline)) {
    if constexpr (GM) {
      const bool inL = d.idx < nL;
      *reinterpret_cast<Cell *>(smem + ((ok && inL) ? d.cell : trash)) = v;
      typedef double v2dd __attribute__((ext_vector_type(2)));
      const v2dd vv = {v.x, v.y};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, vv), rsrc, (ok && !inL) ? gofs(d.idx) : 0xFFFFFFFFu, 0, 0);
    } else {
      slot(d.cell) = v;
    }
  };
  Draw dA = draw_step<RARE>(g, (uint32_t)n, flips, row_bytes, lane_bytes), dB = dA;
  Row aA = rdrow(dA), aB = aA;
  // Cells in memory (GM): the pipeline is TWO steps deep -- three (draw, row) register sets, step s issues the load of
  // step s + 2 -- because a load that misses L2 comes back from the Infinity Cache in more than one step's time once
  // every CU is issuing them (measured with a one-step pipeline: 27 % of the wave's cycles spent waiting).  A row is
  // therefore loaded BEFORE the stores of the two steps that precede its use; their committed cells are forwarded
  // into it at its first use (fw1 = previous step, fw2 = the one before; ~0 = no such cell).
  Draw dC = dA;
  Row aC = aA;
  Draw dD = dA;             // (a third step of depth, PSTAT_GM_DEPTH = 3: four sets, three forwarded commits)
  Row aD = aA;
  Cell fw1_cell{}, fw2_cell{}, fw3_cell{};
  uint32_t fw1_at = ~0u, fw2_at = ~0u, fw3_at = ~0u;
  (void)dC; (void)aC; (void)dD; (void)aD; (void)fw1_cell; (void)fw2_cell; (void)fw3_cell; (void)fw1_at; (void)fw2_at; (void)fw3_at;
  if constexpr (GM && DEPTH >= 2) {
    if (remaining > 1) {
      dB = draw_step<RARE>(g, (uint32_t)n, flips, row_bytes, lane_bytes);
      aB = rdrow(dB);
    }
    if constexpr (DEPTH == 3) {
      if (remaining > 2) {
        dC = draw_step<RARE>(g, (uint32_t)n, flips, row_bytes, lane_bytes);
        aC = rdrow(dC);
      }
    }
  }
  R phistep3 = 3 * phistep, thstep3 = 3 * thstep;
  (void)phistep3; (void)thstep3;
  P stepv = {thstep, phistep}, step3v = {thstep3, phistep3};   // f32: the (theta, phi) proposal as one 2-vector
  (void)stepv; (void)step3v;
  int left = (int)remaining;        // steps still to run in this segment (<= 2^30)
  // running observables: (rx, ry), (px, py) and {rz, pz} as pairs, U as a scalar; f32 carries r in units of b
  P Orxy = {O[0], O[1]}, Opxy = {O[3], O[4]};
  P Oz = {O[2], O[5]};              // {r_z, p_z}
  R OU = O[6];
  const P bb = {b, b};
  (void)bb;

  // f32 arithmetic keeps r, p, U as running totals of accepted differences; their rounding errors
  // random-walk (~3e-6 per accepted move at n = 100).  At every segment start the totals are therefore
  // re-derived from the angles just filled into LDS, as the reference derives them every step, and
  // pstat_advance keeps f32 segments at most 32 768 steps long: an O(n) pass costing < 0.1 %.
  auto refresh_totals = [&]() {
    double tx = 0, ty = 0, tz = 0, qx = 0, qy = 0, qz = 0, tu = 0, tp = 0;
    R pnx = 0, pny = 0, pnz = 0, pmx = 0, pmy = 0, pmz = 0;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
      const Cell a = ang[i * lanes + lane];
      R thi, phi;
      if constexpr (Q) { thi = q16_theta_turns(a & 0xFFFFu); phi = q16_phi_turns(a >> 16); }
      else { thi = a.x; phi = a.y; }
      R si, ci, spi, cpi, mix, miy, miz;
      AG::sc(thi, &si, &ci);
      AG::sc(phi, &spi, &cpi);
      const R nix = cpi * si, niy = spi * si, niz = ci;
      dipole<R, CT>(a_or_mu, k2e, nix, niy, niz, mix, miy, miz);
      tx += (double)nix; ty += (double)niy; tz += (double)niz;
      qx += (double)mix; qy += (double)miy; qz += (double)miz;
      tu += (double)(mhalfE0 * miz);
      if constexpr (EN == PSTAT_ISING) {
        if (i > 0)
          tp += (double)pair_term_fast(hb * (pnx + nix), hb * (pny + niy), hb * (pnz + niz), pmx, pmy, pmz, mix, miy, miz);
        pnx = nix; pny = niy; pnz = niz; pmx = mix; pmy = miy; pmz = miz;
      }
    }
    // (each total goes through an opaque register: seen as a group, these assignments make the SLP
    // vectoriser re-pair the accumulators of the hot loop, which costs it ~10 moves per step)
    auto opaque = [](R v) __attribute__((always_inline)) -> R { asm volatile("" : "+v"(v)); return v; };
    const double bd = (double)b;
    // f32 carries r in units of b (the hot loop then needs no m*b): r_x, r_y, r_z here are sums of n
    Orxy.x = opaque((R)tx); Orxy.y = opaque((R)ty); Oz.x = opaque((R)tz);
    Opxy.x = opaque((R)qx); Opxy.y = opaque((R)qy); Oz.y = opaque((R)qz);
    usum = opaque((R)tu);
    OU = opaque((R)(tu + tp - ((double)Fx * bd * tx + (double)Fz * bd * tz)));
  };

  if constexpr (sizeof(R) == 4) refresh_totals();

  // The step loop, compiled twice for f64: with and without the huge-argument fold of the phi sincos (pstat_math.h,
  // PSTAT_PHI_FOLD).  phi random-walks unwrapped by at most pi per step, so a wave none of whose chains can reach the bound
  // within this segment runs the copy without it -- always, in practice; the other copy keeps any start (--x0) and any
  // run length correct.
  auto run_steps = [&](auto fold_tag) __attribute__((always_inline)) {
  constexpr bool FOLD = decltype(fold_tag)::value;
  while (left > 0) {
    int chunk = left < FLUSH ? left : FLUSH;
    if (A.adaptive && to_adj < chunk) chunk = (int)to_adj;
    P a1rxy = {0, 0}, a1pxy = {0, 0}, a2rxy = {0, 0}, a2pxy = {0, 0};
    P a1z = {0, 0}, a2z = {0, 0};
    R a1U = 0, a2U = 0;
    R accw = 0;

    // one Monte-Carlo step on (d, a0); fetches the next step's draws and LDS row into (dn, an)
    auto one_step = [&](const Draw &d, const Row &a0_, Draw &dn, Row &an, const bool more)
        __attribute__((always_inline)) {
      Cell a0, aLo{}, aHi{};      // (aLo, aHi: Ising with cells in memory -- the neighbours, fetched with the row)
      (void)aLo; (void)aHi;
      if constexpr (GI) {
        const bool hasLo = d.idx > 0, hasHi = d.idx + 1 < (uint32_t)n;
        a0 = resolve(a0_.c, d.idx, d.cell, fw1_cell, fw1_at, fw2_cell, fw2_at, fw3_cell, fw3_at);
        aLo = resolve(a0_.lo, hasLo ? d.idx - 1 : d.idx, hasLo ? d.cell - row_bytes : d.cell, fw1_cell, fw1_at, fw2_cell, fw2_at, fw3_cell, fw3_at);
        aHi = resolve(a0_.hi, hasHi ? d.idx + 1 : d.idx, hasHi ? d.cell + row_bytes : d.cell, fw1_cell, fw1_at, fw2_cell, fw2_at, fw3_cell, fw3_at);
      } else if constexpr (GM) {
        a0 = resolve(a0_, d.idx, d.cell, fw1_cell, fw1_at, fw2_cell, fw2_at, fw3_cell, fw3_at);
      } else {
        a0 = a0_;
      }
      if (more) {
        dn = draw_step<RARE>(g, (uint32_t)n, flips, row_bytes, lane_bytes);
        an = rdrow(dn);
        // keep the next row's read up here, a whole step ahead of its use: left alone, the scheduler sinks it
        // to ~12 instructions before the forwarding select, and a lone wave then waits for LDS (+1 % measured)
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- proposal, mcmc_eap_chain.jl:277-280, and trial angles, inc/eap_chain.jl:232-236
      R eps;                        // u in [0,1) (f64) or 1 + u (f32: the -1 is folded into the test)
      if constexpr (sizeof(R) == 8) eps = 0;   // (metropolis_f64 derives it from the raw word)
      else eps = bits12(d.weps);
      (void)eps;
      R th0, ph0, th1, ph1;
      bool inside = true;           // q16: the trial theta stayed on the lattice (else: clamped => rejected)
      uint32_t cell1 = 0;           // q16: packed trial state
      if constexpr (Q) {
        uint32_t k0 = a0 & 0xFFFFu;
        const uint32_t j0 = a0 >> 16;
        th0 = q16_theta_turns(k0);
        ph0 = q16_phi_turns(j0);
        if constexpr (RARE) {
          if (flips && (d.wflip >> 31)) k0 = 65535u - k0;   // theta -> pi - theta, exact on the lattice
        }
        const int k1 = (int)k0 + q16_disp(thstep, sym11<R>(d.wth));
        const uint32_t j1 = (j0 + (uint32_t)q16_disp(phistep, sym11<R>(d.wphi))) & 0xFFFFu;
        inside = (uint32_t)k1 < 65536u;   // theta' clamped to 0 or pi has sin = 0: never accepted
        const uint32_t k1c = (uint32_t)min(max(k1, 0), 65535);
        th1 = q16_theta_turns(k1c);
        ph1 = q16_phi_turns(j1);
        cell1 = k1c | (j1 << 16);
      } else {
        th0 = a0.x; ph0 = a0.y;
        R flip = 0;
        if constexpr (RARE) {
          if (flips && (d.wflip >> 31)) flip = AG::theta_max - 2 * th0;
        }
        if constexpr (sizeof(R) == 8) {
          // The trajectory itself: every product and sum rounded separately, as the oracle (and Julia) round them.  The
          // f64 sweep objects are otherwise built with -ffp-contract=fast: nothing else in the step can change which
          // angles a chain visits (see sincos_fast_f64), and the fused forms save ~50 instructions per step.
          // (each product passes through an opaque register, so no pass can fuse it into the sum that follows)
          auto rounded = [](R v) __attribute__((always_inline)) -> R { asm volatile("" : "+v"(v)); return v; };
          const R dphi = rounded(phistep * rounded(sym11<R>(d.wphi)));
          R dth = rounded(thstep * rounded(sym11<R>(d.wth)));
          if constexpr (RARE) dth = rounded(flip + dth);     // (x + 0 is not foldable under IEEE rules: -0 + 0 = +0; th0 + -0 = th0 + 0)
          ph1 = AG::wrap(ph0 + dphi);
          th1 = fmin(AG::theta_max, fmax((R)0, th0 + dth));
        } else {
          // angle + step * (f - 3) with f in [2,4): the -3*step rides on the base angle
          R base = th0;
          if constexpr (RARE) base = th0 + flip;
          const P basev = {base, ph0}, fv = {bits24(d.wth), bits24(d.wphi)};
          const P raw = pfma(stepv, fv, basev - step3v);
          th1 = fmin(AG::theta_max, fmax((R)0, raw.x));
          ph1 = AG::wrap(raw.y);
        }
      }
      R st0, ct0, sp0, cp0, st1, ct1, sp1, cp1;
      AG::sc_theta(th0, &st0, &ct0);
      AG::sc_theta(th1, &st1, &ct1);
      AG::template sc_phi<FOLD>(ph0, &sp0, &cp0);
      AG::template sc_phi<FOLD>(ph1, &sp1, &cp1);
      P dNxy, dMxy, D;   // old -> new differences of {n_x, n_y}, {mu_x, mu_y}, {n_z, mu_z}
      R dpair = 0;
      if constexpr (EN == PSTAT_ISING && sizeof(R) == 4) {
        // f32 Ising: every quantity of the touched monomer travels as an {old, new} pair, so that the four
        // neighbour terms (two neighbours x old/new) are two packed evaluations -- a lone wave pays per
        // instruction, and a packed one costs little more than a scalar one
        const P S01 = {st0, st1}, X01 = P{cp0, cp1} * S01, Y01 = P{sp0, sp1} * S01, C01 = {ct0, ct1};
        P MX01, MY01, MZ01;                               // dipole, inc/dipole_response.jl:7-29
        if constexpr (CT == PSTAT_DIELECTRIC) {
          const P q01 = C01 * P{a_or_mu, a_or_mu};        // (K1-K2) E0 cos(theta)
          MX01 = q01 * X01; MY01 = q01 * Y01;
          MZ01 = pfma(q01, C01, P{k2e, k2e});
        } else {
          const P mus = {a_or_mu, a_or_mu};
          MX01 = mus * X01; MY01 = mus * Y01; MZ01 = mus * C01;
        }
        dNxy = P{X01.y - X01.x, Y01.y - Y01.x};
        dMxy = P{MX01.y - MX01.x, MY01.y - MY01.x};
        D = P{C01.y - C01.x, MZ01.y - MZ01.x};
        P e01 = {0, 0};
        const P m3 = {-3.0f, -3.0f};
#pragma unroll
        for (int side = -1; side <= 1; side += 2) {
          const int j = (int)d.idx + side;
          if (j >= 0 && j < n) {
            const Cell aj = ang[j * lanes + lane];
            R thj, phj;
            if constexpr (Q) { thj = q16_theta_turns(aj & 0xFFFFu); phj = q16_phi_turns(aj >> 16); }
            else { thj = aj.x; phj = aj.y; }
            R sj, cj, spj, cpj, mjx, mjy, mjz;
            AG::sc(thj, &sj, &cj);
            AG::sc(phj, &spj, &cpj);
            const R njx = cpj * sj, njy = spj * sj, njz = cj;
            dipole<R, CT>(a_or_mu, k2e, njx, njy, njz, mjx, mjy, mjz);
            // bond vector r = -b/2 (n_i + n_j), inc/eap_chain.jl:215-228; the term of :200-207.  Only r's
            // direction (twice: the sign cancels) and |r|^3 enter, so the sum runs on s = n_i + n_j and the
            // factor 1/|b/2|^3 is applied once at the end
            const P rx = X01 + P{njx, njx}, ry = Y01 + P{njy, njy}, rz = C01 + P{njz, njz};
            const P jx = {mjx, mjx}, jy = {mjy, mjy}, jz = {mjz, mjz};
            const P r2 = pfma(rz, rz, pfma(ry, ry, rx * rx));
            const P ir = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
            const P ir2 = ir * ir;
            const P mimj = pfma(MZ01, jz, pfma(MY01, jy, MX01 * jx));
            const P mir = pfma(MZ01, rz, pfma(MY01, ry, MX01 * rx));
            const P mjr = pfma(jz, rz, pfma(jy, ry, jx * rx));
            const P num = pfma(m3 * ir2, mir * mjr, mimj);
            e01 = pfma(num, ir2 * ir, e01);
          }
        }
        dpair = (e01.y - e01.x) * ising_scale;              // 1/(4 pi |b/2|^3)
      } else {
        // (x, y) components travel as one 2-vector: {n_x, n_y} = sin(theta) * {cos(phi), sin(phi)}
        // (inc/eap_chain.jl:40), so the old->new differences come out packed with no shuffles
        const P cs0 = {cp0, sp0}, cs1 = {cp1, sp1};
        const P s0s = {st0, st0}, s1s = {st1, st1};
        const P Nxy0 = cs0 * s0s, Nxy1 = cs1 * s1s;
        P Mxy0, Mxy1;
        R mz0, mz1;                                         // dipole, inc/dipole_response.jl:7-29
        if constexpr (CT == PSTAT_DIELECTRIC) {
          const R q0 = a_or_mu * ct0, q1 = a_or_mu * ct1;  // (K1-K2) E0 cos(theta)
          const P q0s = {q0, q0}, q1s = {q1, q1};
          Mxy0 = q0s * Nxy0; Mxy1 = q1s * Nxy1;
          mz0 = q0 * ct0 + k2e; mz1 = q1 * ct1 + k2e;
        } else {
          const P mus = {a_or_mu, a_or_mu};
          Mxy0 = mus * Nxy0; Mxy1 = mus * Nxy1;
          mz0 = a_or_mu * ct0; mz1 = a_or_mu * ct1;
        }
        dNxy = Nxy1 - Nxy0; dMxy = Mxy1 - Mxy0;
        // the z components travel as {n_z, mu_z} pairs: D = {dn_z, dmu_z}
        const P Z0 = {ct0, mz0}, Z1 = {ct1, mz1};
        D = Z1 - Z0;

        // ---- energy difference, inc/energy.jl:7-9,20-23; inc/eap_chain.jl:53
        if constexpr (GI) {
          // cells in memory: the neighbours came with the row; branch-free, a missing neighbour's bond is masked out
          R e0 = 0, e1 = 0;
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const Cell aj = side ? aHi : aLo;
            const bool has = side ? d.idx + 1 < (uint32_t)n : d.idx > 0;
            R sj, cj, spj, cpj, mjx, mjy, mjz;
            AG::sc_theta(aj.x, &sj, &cj);
            AG::template sc_phi<FOLD>(aj.y, &spj, &cpj);
            const R njx = cpj * sj, njy = spj * sj, njz = cj;
            dipole<R, CT>(a_or_mu, k2e, njx, njy, njz, mjx, mjy, mjz);
            const R t0 = pair_term_fast(hb * (Nxy0.x + njx), hb * (Nxy0.y + njy), hb * (ct0 + njz),
                                        Mxy0.x, Mxy0.y, mz0, mjx, mjy, mjz);
            const R t1 = pair_term_fast(hb * (Nxy1.x + njx), hb * (Nxy1.y + njy), hb * (ct1 + njz),
                                        Mxy1.x, Mxy1.y, mz1, mjx, mjy, mjz);
            e0 += has ? t0 : (R)0; e1 += has ? t1 : (R)0;
          }
          dpair = e1 - e0;
        } else if constexpr (EN == PSTAT_ISING) {
          R e0 = 0, e1 = 0;
  #pragma unroll
          for (int side = -1; side <= 1; side += 2) {
            const int j = (int)d.idx + side;
            if (j >= 0 && j < n) {
              const Cell aj = ang[j * lanes + lane];
              R thj, phj;
              if constexpr (Q) { thj = q16_theta_turns(aj & 0xFFFFu); phj = q16_phi_turns(aj >> 16); }
              else { thj = aj.x; phj = aj.y; }
              R sj, cj, spj, cpj, mjx, mjy, mjz;
              AG::sc(thj, &sj, &cj);
              AG::sc(phj, &spj, &cpj);
              const R njx = cpj * sj, njy = spj * sj, njz = cj;
              dipole<R, CT>(a_or_mu, k2e, njx, njy, njz, mjx, mjy, mjz);
              e0 += pair_term_fast(hb * (Nxy0.x + njx), hb * (Nxy0.y + njy), hb * (ct0 + njz),
                                   Mxy0.x, Mxy0.y, mz0, mjx, mjy, mjz);
              e1 += pair_term_fast(hb * (Nxy1.x + njx), hb * (Nxy1.y + njy), hb * (ct1 + njz),
                                   Mxy1.x, Mxy1.y, mz1, mjx, mjy, mjz);
            }
          }
          dpair = e1 - e0;
        }
      }
      const R dmz = D.y;
      (void)dmz;

      // ---- energy difference, inc/energy.jl:7-9,20-23; inc/eap_chain.jl:53
      const P dd = D * P{b, mhalfE0};
      const R drz = dd.x, du = dd.y;
      R dUi = du;   // (x + 0 is not folded under IEEE rules: keep the zero terms out of the arithmetic)
      if constexpr (EN == PSTAT_ISING) dUi = du + dpair;
      R dU;
      if constexpr (FX) dU = dUi - (Fx * (b * dNxy.x) + Fz * drz);
      else              dU = dUi - (Fz * drz);

      // ---- Metropolis, inc/acceptance.jl:18-39.  pi ~ exp(-U/kT) * prod sin(theta)
      bool ok;
      R dw = 0;
      if constexpr (RARE) dw = du * wscale;   // change of the umbrella weight function (0 if off)
      if constexpr (sizeof(R) == 8) {
        ok = metropolis_f64(dU, kT, ninv_kT, st1, st0, dw - lag, A.wide_eps != 0, d.weps, d.w0, d.wphi, d.wth);
      } else {
        // same test with the logarithm folded away: eps * sin(th0) < sin(th1) * exp(-dU/kT + dw - lag)
        R e;
        if constexpr (RARE) e = __builtin_amdgcn_exp2f((R)1.44269504f * (dw - lag) + dU * nbeta_log2e);
        else                e = __builtin_amdgcn_exp2f(dU * nbeta_log2e);
        ok = eps * st0 < fma_r(st1, e, st0);   // (1 + u) sin0 < sin1 e + sin0
        if constexpr (Q) ok = ok && inside;
      }

      // ---- commit (branch-free): angles, observables, counters
      Cell a1;
      if constexpr (Q) a1 = ok ? cell1 : a0;
      else { a1.x = ok ? th1 : th0; a1.y = ok ? ph1 : ph0; }
      if constexpr (GM) { Cell t1; t1.x = th1; t1.y = ph1; wr(d, t1, ok); }   // (steered away unless accepted)
      else wr(d, a1, ok);
      const R m = ok ? (R)1 : (R)0;
      const P mm = {m, m};
      if constexpr (sizeof(R) == 8) {  // the oracle's update order: r += b*dn, p += dm, U += dU
        Orxy = pfma(mm, bb * dNxy, Orxy);
        Oz = pfma(mm, P{drz, dmz}, Oz);
      } else {                         // f32: r is carried in units of b (see unit_r)
        Orxy = pfma(mm, dNxy, Orxy);
        Oz = pfma(mm, D, Oz);
      }
      Opxy = pfma(mm, dMxy, Opxy);
      if constexpr (EN == PSTAT_ISING) OU = ok ? OU + dU : OU;  // dU may be inf/NaN (1/r^3): 0*NaN would poison U
      else                             OU = fma_r(m, dU, OU);
      if constexpr (RARE) {
        usum = fma_r(m, du, usum);
        lag = ok ? (R)0 : lag;
      }
      nacc_seg += ok ? 1 : 0;
      if constexpr (EN == PSTAT_ISING) nnan_seg += not_finite(dU) ? 1 : 0;
      if constexpr (GM) {
        // (a rejected step changed nothing in memory: it forwards nothing, so the trial angles need no select here)
        fw3_cell = fw2_cell; fw3_at = fw2_at;
        fw2_cell = fw1_cell; fw2_at = fw1_at;
        fw1_cell.x = th1; fw1_cell.y = ph1; fw1_at = ok ? d.cell : ~0u;
      } else if (more) {  // forward the accepted angles into the prefetched row if it is the same monomer
        const bool same = dn.cell == d.cell;
        if constexpr (Q) an = same ? a1 : an;
        else if constexpr (!GM) { an.x = same ? a1.x : an.x; an.y = same ? a1.y : an.y; }
      }

      // ---- record! x 8, mcmc_eap_chain.jl:327-328 (every step, accepted or not)
      if (umb) {  // UmbrellaAverager: value += v / e^w, normalizer += 1 / e^w
        bool raise;
        R wrel = umbrella_logw(usum, uref, wscale, raise);
        if constexpr (RARE) {   // (umbrella sampling lives in the rare-options instantiation: nothing of this exists in the default one)
          if (__builtin_amdgcn_ballot_w64(raise) != 0) {   // the gauge rises to this configuration (pstat_math.h)
            if (raise) {
              const double f = exp_f64(-(double)wrel);
              const R fr = (R)f;
              const P ff = {fr, fr};
              a1rxy *= ff; a1pxy *= ff; a2rxy *= ff; a2pxy *= ff; a1z *= ff; a2z *= ff; a1U *= fr; a2U *= fr; accw *= fr;
#pragma unroll
              for (int q = 0; q < NSUMS_BASE; ++q) sums[q] *= f;
              wnorm *= f;
              uref = usum; regauged = true; wrel = 0;
            }
          }
        }
        const R wgt = exp_r(wrel);
        const P ww = {wgt, wgt};
        accw += wgt;
        a1rxy = pfma(ww, Orxy, a1rxy); a1pxy = pfma(ww, Opxy, a1pxy);
        a2rxy = pfma(ww * Orxy, Orxy, a2rxy); a2pxy = pfma(ww * Opxy, Opxy, a2pxy);
        a1z = pfma(ww, Oz, a1z); a1U = fma_r(wgt, OU, a1U);
        a2z = pfma(ww * Oz, Oz, a2z); a2U = fma_r(wgt * OU, OU, a2U);
      } else {
        a1rxy += Orxy; a1pxy += Opxy;
        a2rxy = pfma(Orxy, Orxy, a2rxy); a2pxy = pfma(Opxy, Opxy, a2pxy);
        a1z += Oz; a1U += OU;
        a2z = pfma(Oz, Oz, a2z); a2U = fma_r(OU, OU, a2U);
      }
    };
    // ping-pong between two (draw, row) register sets: no copies in the steady state
    // ping-pong between two (draw, row) register sets -- no copies in the steady state -- and unroll
    // PSTAT_UNROLL steps into one basic block: a lone wave issues independent instructions ~1.5x
    // faster than dependent ones, and consecutive steps overlap (next step's generator, prefetch and
    // old-state trig against the current step's tail)
    int k = 0;
    // (f64: two steps per block -- its step is ~550 instructions, eight of them overflow the 64 KiB
    // instruction cache)
    constexpr int UNROLL = sizeof(R) == 8 ? 2 : (EN == PSTAT_ISING ? PSTAT_UNROLL_ISING : PSTAT_UNROLL);
    if constexpr (GM) {
      // Cells in memory: loads and stores share one in-order counter (vmcnt), so a wait for a prefetched row must know
      // exactly how many younger operations are in flight, or it ends up waiting for the previous step's STORE as well.
      // The main loop therefore prefetches unconditionally and the segment's last steps run in the tail.
      // (`more` of a step = a step two after it exists in this segment; the main loop covers only such steps)
      if constexpr (DEPTH == 3) {
        const int lim = chunk < left - 3 ? chunk : left - 3;
        for (; k + 4 <= lim; k += 4) {
          one_step(dA, aA, dD, aD, true);
          one_step(dB, aB, dA, aA, true);
          one_step(dC, aC, dB, aB, true);
          one_step(dD, aD, dC, aC, true);
        }
        for (; k < chunk; ++k) {
          one_step(dA, aA, dD, aD, left - k > 3);
          dA = dB; aA = aB;
          dB = dC; aB = aC;
          dC = dD; aC = aD;
        }
      } else if constexpr (DEPTH == 2) {
        const int lim = chunk < left - 2 ? chunk : left - 2;
        for (; k + 3 <= lim; k += 3) {
          one_step(dA, aA, dC, aC, true);
          one_step(dB, aB, dA, aA, true);
          one_step(dC, aC, dB, aB, true);
        }
        for (; k < chunk; ++k) {
          one_step(dA, aA, dC, aC, left - k > 2);
          dA = dB; aA = aB;
          dB = dC; aB = aC;
        }
      } else {   // one step deep (Ising): two register sets
        const int lim = chunk < left - 1 ? chunk : left - 1;
        for (; k + 2 <= lim; k += 2) {
          one_step(dA, aA, dB, aB, true);
          one_step(dB, aB, dA, aA, true);
        }
        for (; k < chunk; ++k) {
          one_step(dA, aA, dB, aB, left - k > 1);
          dA = dB; aA = aB;
        }
      }
    } else {
    for (; k + (UNROLL - 1) < chunk; k += UNROLL) {
#pragma unroll
      for (int u = 0; u < UNROLL; u += 2) {
        one_step(dA, aA, dB, aB, true);
        one_step(dB, aB, dA, aA, u + 2 < UNROLL || left - k > UNROLL);
      }
    }
    for (; k + 1 < chunk; k += 2) {
      one_step(dA, aA, dB, aB, true);
      one_step(dB, aB, dA, aA, left - k > 2);
    }
    if (k < chunk) {
      one_step(dA, aA, dB, aB, left - k > 1);
      dA = dB; aA = aB;
    }
    }

    R acc1[7] = {a1rxy.x, a1rxy.y, a1z.x, a1pxy.x, a1pxy.y, a1z.y, a1U};
    R acc2[7] = {a2rxy.x, a2rxy.y, a2z.x, a2pxy.x, a2pxy.y, a2z.y, a2U};
    if constexpr (sizeof(R) == 4) {   // r was carried in units of b
#pragma unroll
      for (int q = 0; q < 3; ++q) { acc1[q] *= b; acc2[q] *= b * b; }
    }
    sums[S_R1] += (double)acc1[0]; sums[S_R2] += (double)acc1[1]; sums[S_R3] += (double)acc1[2];
    sums[S_P1] += (double)acc1[3]; sums[S_P2] += (double)acc1[4]; sums[S_P3] += (double)acc1[5];
    sums[S_U] += (double)acc1[6];
    sums[S_R1SQ] += (double)acc2[0]; sums[S_R2SQ] += (double)acc2[1]; sums[S_R3SQ] += (double)acc2[2];
    sums[S_P1SQ] += (double)acc2[3]; sums[S_P2SQ] += (double)acc2[4]; sums[S_P3SQ] += (double)acc2[5];
    sums[S_USQ] += (double)acc2[6];
    wnorm += (double)accw;
    step += chunk;
    left -= chunk;
    steps_seg += chunk;

    // ---- step-size adaptation, mcmc_eap_chain.jl:301-322 (per chain, in f64 like the reference)
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
        phistep3 = 3 * phistep; thstep3 = 3 * thstep;
        stepv = P{thstep, phistep}; step3v = P{thstep3, phistep3};
      }
    }
  }
  };
  if constexpr (sizeof(R) == 8) {
    const bool can_reach = !((double)maxphi + 3.1416 * (double)remaining < PSTAT_PHI_FOLD);   // (also true for a NaN angle)
    if (__builtin_amdgcn_ballot_w64(can_reach) != 0) run_steps(std::true_type{});
    else run_steps(std::false_type{});
  } else {
    run_steps(std::false_type{});
  }

  // ---- spill
  if constexpr (Q) {
    uint16_t *gth = (uint16_t *)S.ang, *gph = (uint16_t *)S.ang + (int64_t)n * C;
    for (int i = 0; i < n; ++i) {
      const uint32_t v = ang[i * lanes + lane];
      gth[(int64_t)i * C + c] = (uint16_t)(v & 0xFFFFu);
      gph[(int64_t)i * C + c] = (uint16_t)(v >> 16);
    }
  } else {
    R *gth = (R *)S.ang, *gph = (R *)S.ang + (int64_t)n * C;
    for (int i = 0; i < n; ++i) {
      R2 v;
      if (GM && i < A.lds_rows) v = reinterpret_cast<Cell *>(smem)[i * lanes + lane];
      else if constexpr (GM) v = *reinterpret_cast<Cell *>(cells + gofs((uint32_t)i));
      else v = ang[i * lanes + lane];
      gth[(int64_t)i * C + c] = v.x;
      gph[(int64_t)i * C + c] = v.y;
    }
  }
  g.store(S.rng + c, C);
  S.stepsz[0 * C + c] = phistep_d; S.stepsz[1 * C + c] = thstep_d;
  S.win[0 * C + c] = nacc_off + nacc_seg; S.win[1 * C + c] = natt_off + steps_seg;
  S.nacc_total[c] += nacc_seg;
  if constexpr (EN == PSTAT_ISING) S.nanrej[c] += nnan_seg;
  const R unit_r = sizeof(R) == 4 ? b : (R)1;   // f32 carries r in units of b
  S.obs[OBS_R1 * C + c] = unit_r * Orxy.x; S.obs[OBS_R2 * C + c] = unit_r * Orxy.y; S.obs[OBS_R3 * C + c] = unit_r * Oz.x;
  S.obs[OBS_P1 * C + c] = Opxy.x; S.obs[OBS_P2 * C + c] = Opxy.y; S.obs[OBS_P3 * C + c] = Oz.y;
  S.obs[OBS_U * C + c] = OU;
  if constexpr (RARE) S.obs[OBS_USUM * C + c] = usum;   // only the umbrella weights ever read it
  if constexpr (RARE) {
    S.lag[c] = lag;
    if (umb) S.wnorm[c] = wnorm;
    if (regauged) S.uref[c] = (double)uref;
  }
#pragma unroll
  for (int q = 0; q < NSUMS_BASE; ++q) S.sums[q * C + c] = sums[q];
}

// Persistent sweep kernel: run_segment under the (block, segment) job loop of pstat_device.h.
// PACKED: chain blocks straddle cases, the case's scalars are per-lane values (run_job_queue, pstat_device.h).
template <typename R, typename G, int CT, int EN, bool FX, bool RARE, int ST, bool PACKED = false>
__global__ __launch_bounds__(64) void sweep_kernel(SweepArgs A, DevState S,
                                                   const CaseConst *__restrict__ cases,
                                                   SweepRare rare, int *__restrict__ queue) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  run_job_queue<PACKED>(A, queue, lane, [&](const CaseConst &cc, int64_t chain, int64_t first, int64_t len, int blk) {
    run_segment<R, G, CT, EN, FX, RARE, ST>(A, S, cc, rare, smem, lane, chain, first, len, blk);
  }, cases);
}

#ifndef PSTAT_PART
// ------------------------------------------------------------------------------------------ reduce

constexpr int RED_BLOCKS = 256;
constexpr int RED_THREADS = 256;
constexpr int NQ = 19;  // 16 observables + acceptance ratio + the clustering main's two extras
constexpr int NX = 2;   // plain sums: non-finite-energy rejections, collapsed chains (pstat.h)
constexpr int NP = 2 * NQ + NX;   // entries of one block's partial

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// stage 1: every block folds a strided slice of chains into partial[block][2*NQ] (deterministic)
__global__ __launch_bounds__(RED_THREADS) void reduce_stage1(DevState S, int64_t c0, int64_t c1,
                                                             int64_t steps, int umbrella,
                                                             const CaseConst *__restrict__ cases,
                                                             int64_t chains_per_case, int64_t n,
                                                             double *__restrict__ partial) {
  __shared__ double red[RED_THREADS / 64][NP];
  double m1[NQ], m2[NQ], mx[NX] = {0, 0};
#pragma unroll
  for (int q = 0; q < NQ; ++q) { m1[q] = 0; m2[q] = 0; }
  const int64_t C = S.C;
  for (int64_t c = c0 + (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; c < c1;
       c += (int64_t)RED_BLOCKS * RED_THREADS) {
    const double norm = umbrella ? S.wnorm[c] : (double)steps;
    const double inv = norm != 0.0 ? 1.0 / norm : 0.0;
    double v[NQ];
    v[PSTAT_R1] = S.sums[S_R1 * C + c]; v[PSTAT_R2] = S.sums[S_R2 * C + c]; v[PSTAT_R3] = S.sums[S_R3 * C + c];
    v[PSTAT_R1SQ] = S.sums[S_R1SQ * C + c]; v[PSTAT_R2SQ] = S.sums[S_R2SQ * C + c];
    v[PSTAT_R3SQ] = S.sums[S_R3SQ * C + c];
    v[PSTAT_RSQ] = v[PSTAT_R1SQ] + v[PSTAT_R2SQ] + v[PSTAT_R3SQ];
    v[PSTAT_P1] = S.sums[S_P1 * C + c]; v[PSTAT_P2] = S.sums[S_P2 * C + c]; v[PSTAT_P3] = S.sums[S_P3 * C + c];
    v[PSTAT_P1SQ] = S.sums[S_P1SQ * C + c]; v[PSTAT_P2SQ] = S.sums[S_P2SQ * C + c];
    v[PSTAT_P3SQ] = S.sums[S_P3SQ * C + c];
    v[PSTAT_PSQ] = v[PSTAT_P1SQ] + v[PSTAT_P2SQ] + v[PSTAT_P3SQ];
    v[PSTAT_U] = S.sums[S_U * C + c]; v[PSTAT_USQ] = S.sums[S_USQ * C + c];
#pragma unroll
    for (int q = 0; q < PSTAT_NOBS; ++q) v[q] *= inv;
    v[16] = steps > 0 ? (double)S.nacc_total[c] / (double)steps : 0.0;
    v[17] = S.sums[S_C2 * C + c] * inv;    // sum cos^2(theta)
    v[18] = S.sums[S_PSI * C + c] * inv;   // mean bond angle
#pragma unroll
    for (int q = 0; q < NQ; ++q) { m1[q] += v[q]; m2[q] = fma(v[q], v[q], m2[q]); }
    mx[0] += (double)S.nanrej[c];
    // collapsed: |U| of the current configuration is 1e3 times beyond what n separated monomers can hold in field,
    // force and thermal energy -- only a 1/r^3 contact gets there (pstat.h, pstat_summary.chains_collapsed)
    const CaseConst &cc = cases[c / chains_per_case];
    const double mu_max = fmax(fmax(fabs(cc.K1), fabs(cc.K2)) * fabs(cc.E0), fabs(cc.mu));
    const double per_monomer = cc.kT + 0.5 * fabs(cc.E0) * mu_max + fabs(cc.b) * (fabs(cc.Fx) + fabs(cc.Fz));
    mx[1] += !(fabs(S.obs[OBS_U * C + c]) <= 1e3 * (double)n * per_monomer) ? 1.0 : 0.0;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double a = wave_sum(m1[q]), b = wave_sum(m2[q]);
    if (lane == 0) { red[wave][q] = a; red[wave][NQ + q] = b; }
  }
#pragma unroll
  for (int q = 0; q < NX; ++q) {
    double a = wave_sum(mx[q]);
    if (lane == 0) red[wave][2 * NQ + q] = a;
  }
  __syncthreads();
  if (threadIdx.x < NP) {
    double t = 0;
#pragma unroll
    for (int w = 0; w < RED_THREADS / 64; ++w) t += red[w][threadIdx.x];
    partial[blockIdx.x * NP + threadIdx.x] = t;
  }
}

// stage 2: one wave per output folds the RED_BLOCKS partials in a fixed order
__global__ __launch_bounds__(64) void reduce_stage2(const double *__restrict__ partial,
                                                    int64_t nchains, double *__restrict__ out) {
  const int q = blockIdx.x;  // 0 .. NP-1
  double t = 0;
  for (int blk = threadIdx.x; blk < RED_BLOCKS; blk += 64) t += partial[blk * NP + q];
  t = wave_sum(t);
  if (threadIdx.x == 0) {
    out[1 + q] = t;
    if (q == 0) out[0] = (double)nchains;
  }
}

// ------------------------------------------------------------------------------------------ re-init

// What a fresh mcmc(nsteps, pargs, chain) call starts from (mcmc_clustering_eap_chain.jl:172-181):
// default step sizes, zeroed adaptation window, an acceptor with no history, a weight function
// gauged on the current chain.  The averagers are reset separately (pstat_reset_averages).
__global__ void reset_sampler_kernel(DevState S, double phi_step, double theta_step) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S.C) return;
  S.stepsz[0 * S.C + c] = phi_step; S.stepsz[1 * S.C + c] = theta_step;
  S.win[0 * S.C + c] = 0; S.win[1 * S.C + c] = 0;
  S.lag[c] = 0.0;
  S.uref[c] = S.obs[OBS_USUM * S.C + c];
}

// mcmc_eap_chain.jl:352-361: draw a fresh configuration; adopt it if forced or by
// metropolis_acc (inc/acceptance.jl:1-3).  One thread per chain.  The reference's acceptor keeps
// the log-density it cached at the last acceptance, so after an adoption its comparisons are offset
// by `lag` until the next accepted move -- reproduced here.
template <typename R, typename G>
__global__ void reinit_kernel(SweepArgs A, DevState S, const CaseConst *__restrict__ cases,
                              int chain_type, int energy_type, int force_init, int umbrella) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S.C) return;
  const int64_t C = S.C, n = A.n;
  const CaseConst cc = cases[c / A.chains_per_case];
  R *th = (R *)S.ang, *ph = (R *)S.ang + n * C;
  R *nth = (R *)S.ang_tmp, *nph = (R *)S.ang_tmp + n * C;
  G g;
  g.load(S.rng + c, C);
  for (int64_t i = 0; i < n; ++i) nph[i * C + c] = store_phi<R>(u01<double>(g.next()));
  for (int64_t i = 0; i < n; ++i) nth[i * C + c] = store_theta<R>(u01<double>(g.next()));

  double r[3] = {0, 0, 0}, p[3] = {0, 0, 0}, usum = 0, upair = 0, prod_new = 1.0, prod_old = 1.0;
  double pnx = 0, pny = 0, pnz = 0, pmx = 0, pmy = 0, pmz = 0;
  for (int64_t i = 0; i < n; ++i) {
    double t = load_theta<R>(nth[i * C + c]), f = load_phi<R>(nph[i * C + c]);
    double st = sin(t), ct = cos(t), sp = sin(f), cp = cos(f);
    double nx = cp * st, ny = sp * st, nz = ct, mx, my, mz;
    if (chain_type == PSTAT_DIELECTRIC)
      dipole<double, PSTAT_DIELECTRIC>((cc.K1 - cc.K2) * cc.E0, cc.K2 * cc.E0, nx, ny, nz, mx, my, mz);
    else
      dipole<double, PSTAT_POLAR>(cc.mu, 0.0, nx, ny, nz, mx, my, mz);
    r[0] += cc.b * nx; r[1] += cc.b * ny; r[2] += cc.b * nz;
    p[0] += mx; p[1] += my; p[2] += mz;
    usum += -0.5 * cc.E0 * mz;
    prod_new *= st;
    prod_old *= sin(load_theta<R>(th[i * C + c]));
    if (energy_type == PSTAT_ISING && i > 0) {
      double h = -cc.b / 2;
      upair += pair_term<double>(h * (pnx + nx), h * (pny + ny), h * (pnz + nz), pmx, pmy, pmz, mx, my, mz);
    }
    pnx = nx; pny = ny; pnz = nz; pmx = mx; pmy = my; pmz = mz;
  }
  const double U_new = usum + upair - (r[0] * cc.Fx + r[2] * cc.Fz);
  const double U_old = S.obs[OBS_U * C + c];
  bool adopt = force_init != 0;
  if (!adopt) {
    const double eps = u01<double>(g.next());
    adopt = eps <= (exp(-(U_new - U_old) / cc.kT) * prod_new / prod_old);
  }
  if (adopt) {
    // the cached log-density includes the umbrella weight function w = sum(u) * wscale (inc/acceptance.jl:13-16,
    // inc/average.jl:104-124); wscale as in run_segment
    const double ws = umbrella ? (0.2 + 0.8 * exp(-(cc.Fx * cc.Fx + cc.Fz * cc.Fz) / cc.kT)) / cc.kT : 0.0;
    const double lp_old = -U_old / cc.kT + log(prod_old) + S.obs[OBS_USUM * C + c] * ws;
    const double lp_new = -U_new / cc.kT + log(prod_new) + usum * ws;
    S.lag[c] = (lp_old + S.lag[c]) - lp_new;
    for (int64_t i = 0; i < n; ++i) { th[i * C + c] = nth[i * C + c]; ph[i * C + c] = nph[i * C + c]; }
    S.obs[OBS_R1 * C + c] = r[0]; S.obs[OBS_R2 * C + c] = r[1]; S.obs[OBS_R3 * C + c] = r[2];
    S.obs[OBS_P1 * C + c] = p[0]; S.obs[OBS_P2 * C + c] = p[1]; S.obs[OBS_P3 * C + c] = p[2];
    S.obs[OBS_U * C + c] = U_new; S.obs[OBS_USUM * C + c] = usum;
  }
  g.store(S.rng + c, C);
}

#endif  // !PSTAT_PART

// ------------------------------------------------------------------------------------------ dispatch

using SweepFn = void (*)(SweepArgs, DevState, const CaseConst *, SweepRare, int *);

#ifdef PSTAT_PART
// One object per state format: 1 = f32 (turns), 2 = q16 (lattice, f32 arithmetic), 3 = f64 with the cells in LDS,
// 4 = f64 with the cells in global memory.
template <typename G, int CT, int EN, bool FX, bool RARE, bool PACKED = false>
static SweepFn pick_state() {
#if PSTAT_PART == 4
  return sweep_kernel<double, G, CT, EN, FX, RARE, 2, PACKED>;
#elif PSTAT_PART == 3
  return sweep_kernel<double, G, CT, EN, FX, RARE, 0, PACKED>;
#elif PSTAT_PART == 2
  return sweep_kernel<float, G, CT, EN, FX, RARE, 1, PACKED>;
#else
  return sweep_kernel<float, G, CT, EN, FX, RARE, 0, PACKED>;
#endif
}
template <typename G, int CT, int EN>
static SweepFn pick_flags(const LaunchCfg &cfg) {
  // packed cases: ONE instantiation per (generator, chain, energy) -- the general one (Fx term and the rare options compiled
  // in, switched by their wave-uniform flags), which makes the same decisions as the specialised ones
  if (cfg.packed) return pick_state<G, CT, EN, true, true, true>();
  const bool rare = cfg.do_flips || cfg.lag || cfg.umbrella;
  if (cfg.has_fx) return rare ? pick_state<G, CT, EN, true, true>() : pick_state<G, CT, EN, true, false>();
  return rare ? pick_state<G, CT, EN, false, true>() : pick_state<G, CT, EN, false, false>();
}
template <typename G>
static SweepFn pick_model(const LaunchCfg &cfg) {
  const bool ising = cfg.energy_type == PSTAT_ISING;
  if (cfg.chain_type == PSTAT_DIELECTRIC)
    return ising ? pick_flags<G, PSTAT_DIELECTRIC, PSTAT_ISING>(cfg)
                 : pick_flags<G, PSTAT_DIELECTRIC, PSTAT_NONINTERACTING>(cfg);
  return ising ? pick_flags<G, PSTAT_POLAR, PSTAT_ISING>(cfg)
               : pick_flags<G, PSTAT_POLAR, PSTAT_NONINTERACTING>(cfg);
}
#if PSTAT_PART == 4
SweepFn pick_sweep_f64g(const LaunchCfg &cfg) {
#elif PSTAT_PART == 3
SweepFn pick_sweep_f64(const LaunchCfg &cfg) {
#elif PSTAT_PART == 2
SweepFn pick_sweep_q16(const LaunchCfg &cfg) {
#else
SweepFn pick_sweep_f32(const LaunchCfg &cfg) {
#endif
  return cfg.rng == PSTAT_RNG_XOSHIRO128PP ? pick_model<Xoshiro128pp>(cfg) : pick_model<Mwc64x>(cfg);
}

#else  // common object

SweepFn pick_sweep_f32(const LaunchCfg &cfg);
SweepFn pick_sweep_q16(const LaunchCfg &cfg);
SweepFn pick_sweep_f64(const LaunchCfg &cfg);
SweepFn pick_sweep_f64g(const LaunchCfg &cfg);

// f64 cells are 16 bytes: LDS seats 160 KiB / (16 n) chains per CU.  Once that is fewer than four full waves
// (n > 40) the non-interacting f64 sweep keeps its cells in global memory instead and runs 64 lanes on every
// SIMD (run_segment, ST = 2).  PSTAT_F64_STATE=lds|global overrides the choice (experiments, tests).
// The clustering main's chain-per-lane kernel always keeps its chains in memory (pstat_cluster_gm.hip).
bool f64_state_global(const LaunchCfg &cfg, int64_t n, int64_t total_chains) {
  if (cfg.energy_type != PSTAT_NONINTERACTING && cfg.energy_type != PSTAT_ISING) return false;
  if (cfg.precision == PSTAT_F32 && cfg.move_set == PSTAT_MOVES_CLUSTER) {
    // The f32 clustering main has the in-memory kernel too (20-byte cells, pstat_cluster_gm.hip).  Its LDS kernel is the
    // faster one while the ensemble is resident or nearly so (measured, 65 536 chains: n <= 80 2.2-2.5e10 proposals/s
    // against 1.8-2.4e10; n = 100 a tie; the 546 x 64-chain n = 200 phase scan 1.88 s against 2.18 s); an ensemble of more
    // than twice what LDS seats (160 KiB / 8 n chains per CU) runs in memory (n = 200, 65 536 chains: 1.5e10 against 7.7e9).
    // PSTAT_F32_STATE=lds|global overrides (tests).
    const char *e = getenv("PSTAT_F32_STATE");
    if (e && e[0] == 'l') return false;
    if (e && e[0] == 'g') return true;
    const int64_t seats = (160 * 1024 / (8 * (n > 0 ? n : 1))) * 256;
    return total_chains > 2 * seats;
  }
  if (cfg.precision != PSTAT_F64) return false;
  const char *e = getenv("PSTAT_F64_STATE");
  if (e && e[0] == 'l') return false;
  if (e && e[0] == 'g') return true;
  // the clustering main's kernel in memory carries the trigonometric cache and is faster at every chain length
  // (measured n = 10 / 20 / 40: 1.5e10 / 1.4e10 / 1.3e10 proposals/s against 1.0e10 with the cells in LDS)
  if (cfg.move_set == PSTAT_MOVES_CLUSTER) return true;
  return n * 16 * 256 > 160 * 1024;
}

static int cell_bytes(int precision) { return precision == PSTAT_F64 ? 16 : (precision == PSTAT_Q16 ? 4 : 8); }

bool supports_packed_cases(const LaunchCfg &cfg) {
  // every chain-per-lane kernel has a packed instantiation; the all-pairs energies run a chain per wavefront: nothing to pack
  return cfg.energy_type == PSTAT_NONINTERACTING || cfg.energy_type == PSTAT_ISING;
}

int choose_lanes(int precision, int64_t n, int energy_type) {
  (void)energy_type;
  const int64_t per_lane = n * cell_bytes(precision);
  const int64_t budget = 160 * 1024;
  for (int lanes = 64; lanes >= 8; lanes >>= 1)
    if (per_lane * lanes <= budget) return lanes;
  return 0;
}

static SweepFn pick_sweep(const LaunchCfg &cfg) {
  if (cfg.precision == PSTAT_F64 && cfg.state_global) return pick_sweep_f64g(cfg);
  return cfg.precision == PSTAT_F64 ? pick_sweep_f64(cfg)
       : (cfg.precision == PSTAT_Q16 ? pick_sweep_q16(cfg) : pick_sweep_f32(cfg));
}

static int sweep_lds_bytes(const LaunchCfg &cfg, const SweepArgs &a) {
  if (cfg.state_global) return (a.lds_rows + 1) * 64 * 16;   // + the trash row
  return (int)(a.n * a.lanes * cell_bytes(cfg.precision));
}

hipError_t sweep_kernel_info(const LaunchCfg &cfg, const SweepArgs &a, int *lds_bytes,
                             int *blocks_per_cu, const char **name) {
  SweepFn fn = pick_sweep(cfg);
  const int lds = sweep_lds_bytes(cfg, a);
  hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  int nb = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)fn, 64, lds);
  if (e != hipSuccess) return e;
  if (lds_bytes) *lds_bytes = lds;
  if (blocks_per_cu) *blocks_per_cu = nb;
  if (name) *name = cfg.precision == PSTAT_F64 ? (cfg.state_global ? (cfg.packed ? "sweep_kernel<double, state in L2> [packed cases]" : "sweep_kernel<double, state in L2>")
                                                                   : (cfg.packed ? "sweep_kernel<double> [packed cases]" : "sweep_kernel<double>"))
                 : (cfg.precision == PSTAT_Q16 ? (cfg.packed ? "sweep_kernel<float, q16 state> [packed cases]" : "sweep_kernel<float, q16 state>")
                                               : (cfg.packed ? "sweep_kernel<float> [packed cases]" : "sweep_kernel<float>"));
  return hipSuccess;
}

// queue layout: [0] error flag (sticky: never cleared by a launch), [1] job counter, [2 ..] per-block "segments done"
size_t sweep_queue_ints(const SweepArgs &a) { return 2 + (size_t)a.nblocks; }

hipError_t launch_sweep(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                        const CaseConst *cases, int *queue, unsigned grid, hipStream_t stream) {
  SweepFn fn = pick_sweep(cfg);
  const int lds = sweep_lds_bytes(cfg, a);
  hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(queue + 1, 0, sizeof(int) * (sweep_queue_ints(a) - 1), stream);
  if (e != hipSuccess) return e;
  SweepRare rare{cfg.do_flips, cfg.lag, cfg.umbrella};
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, stream, a, s, cases, rare, queue);
  return hipGetLastError();
}

template <typename G>
static void launch_init_g(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s, const CaseConst *cases,
                          double phi_step, double theta_step, const InitOpts &io, unsigned grid, hipStream_t stream) {
  if (cfg.precision == PSTAT_F64)
    hipLaunchKernelGGL((init_kernel<double, G>), dim3(grid), dim3(256), 0, stream, a, s, cases,
                       cfg.chain_type, cfg.energy_type, phi_step, theta_step, io);
  else if (cfg.precision == PSTAT_Q16)
    hipLaunchKernelGGL((init_kernel<uint16_t, G>), dim3(grid), dim3(256), 0, stream, a, s, cases,
                       cfg.chain_type, cfg.energy_type, phi_step, theta_step, io);
  else
    hipLaunchKernelGGL((init_kernel<float, G>), dim3(grid), dim3(256), 0, stream, a, s, cases,
                       cfg.chain_type, cfg.energy_type, phi_step, theta_step, io);
}

hipError_t launch_reset_sampler(const DevState &s, double phi_step, double theta_step, hipStream_t stream) {
  hipLaunchKernelGGL(reset_sampler_kernel, dim3((unsigned)((s.C + 255) / 256)), dim3(256), 0, stream, s,
                     phi_step, theta_step);
  return hipGetLastError();
}

hipError_t launch_init(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                       const CaseConst *cases, double phi_step, double theta_step,
                       const InitOpts &io, hipStream_t stream) {
  const unsigned grid = (unsigned)((s.C + 255) / 256);
  if (cfg.rng == PSTAT_RNG_XOSHIRO128PP) launch_init_g<Xoshiro128pp>(cfg, a, s, cases, phi_step, theta_step, io, grid, stream);
  else launch_init_g<Mwc64x>(cfg, a, s, cases, phi_step, theta_step, io, grid, stream);
  return hipGetLastError();
}

template <typename G>
static void launch_reinit_g(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s, const CaseConst *cases,
                            int force_init, unsigned grid, hipStream_t stream) {
  if (cfg.precision == PSTAT_F64)
    hipLaunchKernelGGL((reinit_kernel<double, G>), dim3(grid), dim3(256), 0, stream, a, s, cases,
                       cfg.chain_type, cfg.energy_type, force_init, cfg.umbrella);
  else if (cfg.precision == PSTAT_Q16)
    hipLaunchKernelGGL((reinit_kernel<uint16_t, G>), dim3(grid), dim3(256), 0, stream, a, s, cases,
                       cfg.chain_type, cfg.energy_type, force_init, cfg.umbrella);
  else
    hipLaunchKernelGGL((reinit_kernel<float, G>), dim3(grid), dim3(256), 0, stream, a, s, cases,
                       cfg.chain_type, cfg.energy_type, force_init, cfg.umbrella);
}

hipError_t launch_reinit(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                         const CaseConst *cases, int force_init, hipStream_t stream) {
  const unsigned grid = (unsigned)((s.C + 255) / 256);
  if (cfg.rng == PSTAT_RNG_XOSHIRO128PP) launch_reinit_g<Xoshiro128pp>(cfg, a, s, cases, force_init, grid, stream);
  else launch_reinit_g<Mwc64x>(cfg, a, s, cases, force_init, grid, stream);
  return hipGetLastError();
}

size_t reduce_scratch_doubles() { return (size_t)RED_BLOCKS * NP; }
static_assert(NQ == PSTAT_NQ && NX == PSTAT_NX && 1 + NP == PSTAT_NRED, "reduction layout of include/pstat.h");

hipError_t launch_reduce(const DevState &s, int64_t c0, int64_t c1, int64_t steps_recorded,
                         int umbrella, const CaseConst *cases, int64_t chains_per_case, int64_t n,
                         double *partial, double *out, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_stage1, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, stream, s, c0, c1,
                     steps_recorded, umbrella, cases, chains_per_case, n, partial);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(reduce_stage2, dim3(NP), dim3(64), 0, stream, partial, c1 - c0, out);
  return hipGetLastError();
}

#endif  // PSTAT_PART

}  // namespace pstat
