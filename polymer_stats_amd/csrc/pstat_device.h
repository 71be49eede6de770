// pstat_device.h -- internal layout shared by the HIP kernels and the C-ABI host code.
// Not part of the public interface (that is include/pstat.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pstat {

// ---------------------------------------------------------------------------------------------
// Device-resident state of one handle.  Everything is struct-of-arrays over the chain index c
// (fastest-varying), so that the 64 lanes of a wave -- 64 consecutive chains -- touch 64
// consecutive elements: every spill/fill of chain state is a fully coalesced 256/512-byte access.
// C = ncases * chains_per_case.
// ---------------------------------------------------------------------------------------------
enum { OBS_R1 = 0, OBS_R2, OBS_R3, OBS_P1, OBS_P2, OBS_P3, OBS_U, OBS_USUM,
       OBS_C2,    // sum_i cos^2(theta_i)            (clustering main, mcmc_clustering_eap_chain.jl:243)
       OBS_PSI,   // sum of the n-1 bond angles psi   (:244)
       NOBS_STATE };
// per-chain running sums kept on the device; r.r and p.p are the sums of their components.  The
// first NSUMS_BASE are what mcmc_eap_chain.jl records; the clustering main adds two more.
enum { S_R1 = 0, S_R2, S_R3, S_R1SQ, S_R2SQ, S_R3SQ, S_P1, S_P2, S_P3, S_P1SQ, S_P2SQ, S_P3SQ,
       S_U, S_USQ, NSUMS_BASE, S_C2 = NSUMS_BASE, S_PSI, NSUMS };

struct DevState {
  void *ang;            // R  [2][n][C]   plane 0 = theta, plane 1 = phi (radians)
  void *ang_tmp;        // R  [2][n][C]   scratch for re-initialisation
  uint32_t *rng;        // u32[4][C]      xoshiro128++ state
  double *stepsz;       // f64[2][C]      phi_step, theta_step (mcmc_eap_chain.jl:172)
  int64_t *win;         // i64[2][C]      nacc, natt since the last adaptation (:263,265; Int in the reference:
                        //                a window is never reset while the ratio stays inside the band)
  int64_t *nacc_total;  // i64[C]         (:264)
  double *obs;          // f64[NOBS_STATE][C]  r, p, U, sum(u) of the current microstate
  double *sums;         // f64[NSUMS][C]  averager .value fields (inc/average.jl:9)
  double *wnorm;        // f64[C]         averager .normalizer under umbrella sampling
  double *lag;          // f64[C]         acceptor's cached log-pi minus the chain's own (re-init)
  double *uref;         // f64[C]         sum(u) of the chain's first configuration: the umbrella weights
                        //                are taken relative to it (a per-chain constant factor cancels
                        //                in value/normalizer, inc/average.jl:38,63-67)
  void *work;           // f64 "state in memory" kernels only: the waves' working copy of their chains while a segment
                        //                runs (filled from / spilled to `ang` like LDS is); not checkpointed.  Sweep:
                        //                [chain block][n][64] double2 (theta, phi), chain-contiguous for the Ising energy.
                        //                Clustering main: [chain block][lane][n] 40-byte (f32: 20-byte) cells (pstat_cluster_gm.hip)
  int64_t *nanrej;      // i64[C]         proposals whose trial energy was NaN or +-Inf (1/r^3 singularities of the
                        //                pair energies; the reference rejects them silently, inc/acceptance.jl:29-39)
  int64_t C;
};

struct CaseConst {      // physics scalars of one case (inc/eap_chain.jl:89-108)
  double E0, K1, K2, mu, kT, Fz, Fx, b;
  uint64_t seed, chain_id0;
  double kappa, psi0;     // --bend-mod, --bend-angle (clustering main; 0 in mcmc_eap_chain.jl)
  double cluster_prob;    // --cluster-prob: probability of NOT attempting a cluster flip
  double cutoff_radius;   // --cutoff-radius in monomer lengths (energy-type cutoff)
};

struct InitOpts {       // how EAPChain(pargs) draws the first configuration (inc/eap_chain.jl:61-79)
  int use_x0;           // 0: uniform angles; 1: (x0_phi, x0_theta) for every monomer; 2: per monomer from x0_vec
  double x0_phi, x0_theta, dx0_phi, dx0_theta;
  const double *x0_vec; // device, [phi1, theta1, phi2, theta2, ...] (use_x0 == 2)
};

struct SweepArgs {
  int64_t n;
  int64_t chains_per_case;
  int64_t blocks_per_case;
  int64_t nsteps;            // steps to run in this launch
  int64_t step0;             // steps already done in the current init
  int64_t steps_per_adjust;
  double adj_lb, adj_ub, adj_scale;
  int32_t lanes;             // chains per workgroup (64, or fewer when n is too long for LDS)
  int32_t adaptive;          // adj_scale != 1 && steps_per_adjust > 0
  int64_t ncases;
  int64_t seg_len;           // steps per time segment (job) of this launch
  int32_t nseg;              // segments per chain block in this launch
  int32_t max_spins;         // bound on the predecessor wait (each spin sleeps ~2 us)
  int32_t lds_rows;          // f64 state-in-memory sweep: monomers [0, lds_rows) keep their cells in LDS
  int32_t wide_eps;          // f64 kernels: the Metropolis eps carries 53 random bits (pstat_params.uniform_bits; eps_uniform, pstat_math.h)
  int64_t nblocks;           // chain blocks (workgroup-sized groups of `lanes` chains) of the whole handle
  int32_t packed;            // 1: a block holds `lanes` CONSECUTIVE GLOBAL chains, i.e. several cases when a case has fewer
                             // chains than a wave has lanes (run_job_queue<true>); 0: blocks never straddle a case
  int32_t pad_;
};

// ---------------------------------------------------------------------------------------------
// Random stream contract (restated, not shared, in oracle/eap_oracle.c).  Two generators:
//
//  PSTAT_RNG_MWC64X (default): D. Thomas' MWC64X multiply-with-carry, state (x, c) 32+32 bits,
//      out = x ^ c;  (c:x) <- A * x + c,  A = 4294883355,  period ~2^63, passes TestU01 BigCrush.
//      One v_mad_u64_u32 + one v_xor per output: measured 125 cycles per MC step (4 outputs) for a
//      lone wave against 226 for xoshiro128++ (tools/ubench).  As an LCG it is s <- A*s mod M with
//      s = c*2^32 + x, M = A*2^32 - 1, so streams are split by SKIP-AHEAD: all chains of a run walk
//      ONE sequence, chain id k starting k * 2^40 outputs after the seed-selected base state
//      (s_k = s_base * G^k mod M, G = A^(2^40) mod M): disjoint by construction for 2^40 outputs
//      (~2.7e11 MC steps) per chain.  The base state comes from Philox4x32-10(key = seed).
//  PSTAT_RNG_XOSHIRO128PP: xoshiro128++ seeded per chain by Philox4x32-10(key = seed,
//      ctr = (chain_lo, chain_hi, 0x5eed, 0)).
//
//  u(w) = (w >> 9) * 2^-23;  idx = mulhi32(w, n).  The f64 kernels' Metropolis eps has 53 bits by default, made of its own
//  word and the unused low bits of the step's other draws (eps_uniform, pstat_math.h; pstat_params.uniform_bits).
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Xoshiro128pp {
  uint32_t s0, s1, s2, s3;
  __host__ __device__ inline void seed(uint64_t seed, uint64_t chain_id) {
    uint32_t o[4];
    philox4x32_10((uint32_t)chain_id, (uint32_t)(chain_id >> 32), 0x5eedu, 0u,
                  (uint32_t)seed, (uint32_t)(seed >> 32), o);
    if ((o[0] | o[1] | o[2] | o[3]) == 0u) o[0] = 1u;
    s0 = o[0]; s1 = o[1]; s2 = o[2]; s3 = o[3];
  }
  __host__ __device__ inline uint32_t next() {
    uint32_t a = s0 + s3;
    uint32_t result = ((a << 7) | (a >> 25)) + s0;
    uint32_t t = s1 << 9;
    s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
    s2 ^= t;
    s3 = (s3 << 11) | (s3 >> 21);
    return result;
  }
  // adopt the advanced copy `o` where `take` holds (branch-free conditional draw)
  __host__ __device__ inline void pick(bool take, const Xoshiro128pp &o) {
    s0 = take ? o.s0 : s0; s1 = take ? o.s1 : s1; s2 = take ? o.s2 : s2; s3 = take ? o.s3 : s3;
  }
  __host__ __device__ inline void load(const uint32_t *p, int64_t stride) {
    s0 = p[0]; s1 = p[stride]; s2 = p[2 * stride]; s3 = p[3 * stride];
  }
  __host__ __device__ inline void store(uint32_t *p, int64_t stride) const {
    p[0] = s0; p[stride] = s1; p[2 * stride] = s2; p[3 * stride] = s3;
  }
};

struct Mwc64x {
  static constexpr uint32_t A = 4294883355u;
  static constexpr uint64_t M = 0xFFFEB81AFFFFFFFFull;   // A * 2^32 - 1
  static constexpr uint64_t G40 = 0x82A211110E454078ull; // A^(2^40) mod M
  uint32_t x, c;
  __host__ __device__ static inline uint64_t addmod(uint64_t a, uint64_t b) {
    uint64_t s = a + b;                       // a, b < M < 2^64
    if (s < a || s >= M) s -= M;              // on wrap-around the true sum is s + 2^64
    return s;
  }
  __host__ __device__ static inline uint64_t mulmod(uint64_t a, uint64_t b) {
    uint64_t r = 0;
    while (b) {
      if (b & 1) r = addmod(r, a);
      a = addmod(a, a);
      b >>= 1;
    }
    return r;
  }
  __host__ __device__ static inline uint64_t powmod(uint64_t g, uint64_t e) {
    uint64_t r = 1;
    while (e) {
      if (e & 1) r = mulmod(r, g);
      g = mulmod(g, g);
      e >>= 1;
    }
    return r;
  }
  __host__ __device__ inline void seed(uint64_t seed, uint64_t chain_id) {
    uint32_t o[4];
    philox4x32_10(0u, 0u, 0x5eedu, 1u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const uint64_t v = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
    const uint64_t base = 1 + v % (M - 2);                 // in [1, M-2]
    const uint64_t s = mulmod(base, powmod(G40, chain_id));
    x = (uint32_t)s; c = (uint32_t)(s >> 32);
  }
  __host__ __device__ inline uint32_t next() {
    const uint32_t r = x ^ c;
    const uint64_t t = (uint64_t)x * A + c;
    x = (uint32_t)t; c = (uint32_t)(t >> 32);
    return r;
  }
  __host__ __device__ inline void pick(bool take, const Mwc64x &o) { x = take ? o.x : x; c = take ? o.c : c; }
  __host__ __device__ inline void load(const uint32_t *p, int64_t stride) { x = p[0]; c = p[stride]; }
  __host__ __device__ inline void store(uint32_t *p, int64_t stride) const {
    p[0] = x; p[stride] = c; p[2 * stride] = 0u; p[3 * stride] = 0u;
  }
};


#if defined(__HIPCC__)
// Persistent job loop of the chain-per-lane kernels (sweep_kernel, cluster_kernel).  A job = (chain
// block, time segment); jobs are handed out in segment-major order from one atomic counter, so all
// LDS-limited workgroup slots of the chip stay busy even when the number of chain blocks is not a
// multiple of the slots (e.g. 1286 blocks on 1024 slots at n = 100).  Segment s of a block may start
// only after segment s-1 of the same block has been spilled: a per-block counter, published with an
// agent-scope release and awaited with a relaxed poll + one agent-scope acquire
// (placement-independent; cdna_hip_programming.md Guideline 16).  Deadlock-free for any residency: a
// job's predecessor was handed out earlier, to a workgroup that is running and that itself only ever
// waits on still earlier jobs; the wait is bounded (max_spins) and a timeout is reported through the
// queue's error word.  `body(case constants, global chain, first step, number of steps, chain block)` runs one
// segment of one lane's chain.
//
// Which chains a block holds.  PACKED = false: blocks never straddle a case -- block (case, j) holds chains j * lanes ... of
// that case, so the case's physics scalars are wave-uniform and live in SGPRs; a case of 16 chains lights 16 lanes.
// PACKED = true (pstat_create picks it when it shortens the launch: the reference's own sweeps run 1-25 chains per case,
// run/K1_E0-kT-phase.jl:19-45): block b holds the `lanes` consecutive GLOBAL chains b * lanes ..., whichever cases they
// belong to; lane -> (case, chain) by one division per job, and `cases[icase]` is then a per-lane load, so the same body
// compiles with the case's scalars in VGPRs.  A chain's trajectory does not depend on which lanes share its wave.
template <bool PACKED = false, typename Body>
__device__ __forceinline__ void run_job_queue(const SweepArgs &A, int *__restrict__ queue, const int lane, Body &&body,
                                              const CaseConst *__restrict__ cases) {
  const int nblocks = (int)A.nblocks;
  const int njobs = nblocks * A.nseg;
  int *error = queue, *head = queue + 1, *done = queue + 2;   // the error word is sticky: launches clear queue[1..]
  bool failed = false;
  while (!failed) {
    int job = 0;
    if (lane == 0) job = atomicAdd(head, 1);
    job = __builtin_amdgcn_readfirstlane(job);
    if (job >= njobs) break;
    const int blk = job % nblocks, seg = job / nblocks;
    if (seg > 0) {
      int spins = 0;
      for (;;) {
        const int have = __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(&done[blk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (have >= seg) break;
        if (++spins > A.max_spins) { failed = true; break; }
        __builtin_amdgcn_s_sleep(64);
      }
      if (failed) {  // a predecessor never finished: flag it and stop taking jobs
        if (lane == 0) atomicExch(error, 1 + job);
        break;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const int64_t first = (int64_t)seg * A.seg_len;
    const int64_t left = A.nsteps - first;
    const int64_t len = left < A.seg_len ? left : A.seg_len;
    // lanes own disjoint LDS columns and never exchange data: idle lanes just skip the body
    if constexpr (PACKED) {
      const int64_t chain = (int64_t)blk * A.lanes + lane;
      if (len > 0 && lane < A.lanes && chain < A.ncases * A.chains_per_case)
        body(cases[chain / A.chains_per_case], chain, A.step0 + first, len, blk);
    } else {
      const int64_t icase = blk / A.blocks_per_case;
      const int64_t local = (int64_t)(blk % A.blocks_per_case) * A.lanes + lane;
      if (len > 0 && lane < A.lanes && local < A.chains_per_case)
        body(cases[icase], icase * A.chains_per_case + local, A.step0 + first, len, blk);
    }
    if (A.nseg > 1) {
      // publish: this wave's spill stores are complete and written back before the counter moves
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&done[blk], seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
#endif  // __HIPCC__

// host-callable launchers implemented in pstat_kernels.hip; all asynchronous on `stream`
struct LaunchCfg {
  int precision, chain_type, energy_type, do_flips, umbrella, has_fx;
  int lag;  // a re-init has happened on this handle
  int rng;  // PSTAT_RNG_MWC64X | PSTAT_RNG_XOSHIRO128PP
  int move_set;  // PSTAT_MOVES_SINGLE (mcmc_eap_chain.jl) | PSTAT_MOVES_CLUSTER (mcmc_clustering_eap_chain.jl)
  int state_global;  // f64 chain-per-lane kernels: state cells in the global working buffer (DevState::work) instead of LDS
  int packed;        // chain blocks straddle cases (SweepArgs::packed): the kernel instantiation with per-lane case scalars
  int chain_wave;    // clustering main, small f64 ensembles: one chain per wavefront (pstat_cluster_cw.hip, cluster_chain_wave())
};
// the kernel of this configuration has a packed-cases instantiation (every chain-per-lane kernel; not the all-pairs ones)
bool supports_packed_cases(const LaunchCfg &cfg);
hipError_t launch_init(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                       const CaseConst *cases, double phi_step, double theta_step,
                       const InitOpts &io, hipStream_t stream);
hipError_t launch_reset_sampler(const DevState &s, double phi_step, double theta_step, hipStream_t stream);
// clustering main (pstat_cluster.hip): single-monomer move + cluster_flip! per step
hipError_t launch_cluster(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                          const CaseConst *cases, int *queue, unsigned grid, hipStream_t stream);
hipError_t cluster_kernel_info(const LaunchCfg &cfg, const SweepArgs &a, int *lds_bytes,
                               int *blocks_per_cu, const char **name);
// the cluster kernel with its chains in DevState::work (pstat_cluster_gm.hip: f64, and f32 for large ensembles); chosen by f64_state_global()
hipError_t launch_cluster_gm(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                             const CaseConst *cases, int *queue, unsigned grid, hipStream_t stream);
hipError_t cluster_gm_kernel_info(const LaunchCfg &cfg, const SweepArgs &a, int *lds_bytes,
                                  int *blocks_per_cu, const char **name);
size_t cluster_gm_work_bytes(const LaunchCfg &cfg, const SweepArgs &a);
hipError_t launch_sweep(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                        const CaseConst *cases, int *queue, unsigned grid, hipStream_t stream);
size_t sweep_queue_ints(const SweepArgs &a);
hipError_t launch_reinit(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                         const CaseConst *cases, int force_init, hipStream_t stream);
// true iff x is NaN or +-Inf (one v_cmp_class)
template <typename R> __host__ __device__ inline bool not_finite(R x) { return !__builtin_isfinite(x); }
// reduction of chains [c0, c1) into out[PSTAT_NRED]; partial = scratch of reduce_scratch_doubles()
hipError_t launch_reduce(const DevState &s, int64_t c0, int64_t c1, int64_t steps_recorded,
                         int umbrella, const CaseConst *cases, int64_t chains_per_case, int64_t n,
                         double *partial, double *out, hipStream_t stream);
size_t reduce_scratch_doubles();
// LDS bytes and kernel attributes of the sweep kernel chosen for cfg
hipError_t sweep_kernel_info(const LaunchCfg &cfg, const SweepArgs &a, int *lds_bytes,
                             int *blocks_per_cu, const char **name);
int choose_lanes(int precision, int64_t n, int energy_type);
bool f64_state_global(const LaunchCfg &cfg, int64_t n, int64_t total_chains);   // the f64 chain-per-lane kernel of this configuration keeps its state in DevState::work
// --energy-type interacting: one chain per wavefront (pstat_interacting.hip), n <= 512
hipError_t launch_interacting(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s,
                              const CaseConst *cases, int reinit_mode, hipStream_t stream);
// clustering main with the all-pairs energies (interacting, cutoff): pstat_cluster_wave.hip
hipError_t launch_cluster_wave(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s, const CaseConst *cases,
                               hipStream_t stream);
hipError_t cluster_wave_kernel_info(const LaunchCfg &cfg, int64_t n, int *blocks_per_cu, const char **name);
// clustering main, non-interacting / Ising, f64, small ensembles: one chain per wavefront (pstat_cluster_cw.hip)
bool cluster_chain_wave(const LaunchCfg &cfg, int64_t n, int64_t chains_per_case, int64_t ncases);
hipError_t launch_cluster_cw(const LaunchCfg &cfg, const SweepArgs &a, const DevState &s, const CaseConst *cases,
                             hipStream_t stream);
hipError_t cluster_cw_kernel_info(const LaunchCfg &cfg, int64_t n, int *blocks_per_cu, const char **name);
hipError_t interacting_kernel_info(const LaunchCfg &cfg, int64_t n, int *blocks_per_cu, const char **name);

}  // namespace pstat
