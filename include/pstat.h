/*
 * pstat.h -- C ABI of libpstat.so: the MI355X (gfx950) implementation of the fixed-force-ensemble
 * MCMC hot path of grasingerm/polymer-stats.
 *
 * The reference exposes no FFI for this path: the whole of it lives inside one Julia function,
 * mcmc(nsteps, pargs) (mcmc_eap_chain.jl:171-376), reached only through the command line
 * (mcmc_eap_chain.jl:19-155) -- and, for the clustering main, mcmc(nsteps, pargs, chain)
 * (mcmc_clustering_eap_chain.jl:172-352) under its annealing driver (:354-387).  This header is
 * therefore the boundary a maintainer would bind with `ccall` when moving the step loop of those
 * functions onto the GPU; each entry point names the reference code it stands in for.
 * INTEGRATION.md shows the Julia-side binding.
 *
 * Conventions: plain C, caller-allocated output buffers, no callbacks, no exceptions across the ABI.
 * Every function returns PSTAT_OK (0) or a negative pstat_status; pstat_strerror() names it and
 * pstat_last_error() returns a thread-local detail string.  A handle is confined to one host thread
 * at a time; distinct handles may be used concurrently.  There is no CPU fallback: without a HIP
 * device pstat_create() fails with PSTAT_ERR_NO_DEVICE.
 *
 * One handle = `num_chains` independent Markov chains per case (chain-per-lane on the device), each
 * one statistically identical to one reference run with `--num-inits 1`; results are pooled.
 *
 * RESOLUTION OF THE UNIFORM DRAWS (ours; the reference draws Float64 uniforms with 52-53 random bits from an unseeded
 * generator, mcmc_eap_chain.jl:277-280,287).  Every uniform here is made from 32-bit generator words:
 *   * proposals  dphi = phi_step (2u - 1), dtheta = theta_step (2u - 1), u = (w >> 9) 2^-23: a lattice of step / 2^22
 *     (symmetric, so detailed balance holds on it; the adapted step sizes are incommensurate, so chains are not confined
 *     to one lattice), in every precision;
 *   * the Metropolis eps (mcmc_eap_chain.jl:287, inc/acceptance.jl:29-39): `uniform_bits` below.  With 23 bits eps = 0
 *     comes up once per 2^23 = 8.4e6 proposals and then ANY proposal of finite energy with exp(delta) > 0 is accepted, so
 *     acceptance probabilities have a floor of 2^-23 = 1.2e-7 (tests/test_gpu_parity.py pins this on a cold, strongly
 *     coupled chain).  With 53 bits -- the default of the f64 kernels -- the floor is 2^-53 = 1.1e-16, the reference's own;
 *   * the clustering main's link and skip draws (inc/eap_chain.jl:276,286,303) and the re-initialisation draw
 *     (mcmc_eap_chain.jl:357): 23 bits, `u <= p`.
 */
#ifndef PSTAT_H
#define PSTAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSTAT_ABI_VERSION 6

typedef enum pstat_status {
  PSTAT_OK = 0,
  PSTAT_ERR_INVALID_ARG = -1,   /* bad parameter value (message in pstat_last_error)             */
  PSTAT_ERR_NO_DEVICE = -2,     /* no HIP device / device index out of range                     */
  PSTAT_ERR_HIP = -3,           /* a HIP runtime call failed                                     */
  PSTAT_ERR_UNSUPPORTED = -4,   /* valid reference option that has no device implementation      */
  PSTAT_ERR_NOMEM = -5,
  PSTAT_ERR_BAD_CHECKPOINT = -6,
  PSTAT_ERR_TOO_SMALL = -7      /* caller buffer too small; required size has been written back  */
} pstat_status;

/* --chain-type (mcmc_eap_chain.jl:25-28; inc/eap_chain.jl:81-87) */
enum { PSTAT_DIELECTRIC = 0, PSTAT_POLAR = 1 };
/* --energy-type (mcmc_eap_chain.jl:41-44; inc/eap_chain.jl:95-105) */
enum { PSTAT_NONINTERACTING = 0, PSTAT_INTERACTING = 1, PSTAT_ISING = 2,
       /* clustering main only (mcmc_clustering_eap_chain.jl:44-51; UCutoff, inc/eap_chain.jl:165-192):
        * dipole-dipole terms of pairs within cutoff_radius monomer lengths.  NB the reference's UCutoff
        * functor returns that sum ALONE -- with this energy neither the field nor the force enters U. */
       PSTAT_CUTOFF = 3 };
/* arithmetic of the device path:
 *   PSTAT_F64  f64 throughout: the reference's Float64 (inc/types.jl:1-4), and the default.  Reproduces the CPU oracle's
 *              (phi, theta) trajectory, generator state and counters bit for bit; observables to ~1e-15 relative.
 *   PSTAT_F32  opt-in fast path: f32 state and transcendentals, f64 running sums.  Bias of the pooled averages of the
 *              non-interacting energies <= 5e-6 relative.  NOT equivalent to f64 once chains
 *              have collapsed into 1/r^3 contacts (|U| >~ 1e4 kT; interacting/Ising/cutoff energies at strong coupling):
 *              kT-level fidelity there needs position differences good to ~1e-10 b (profiles/r02/config4_f32_vs_f64.json).
 *   PSTAT_Q16  opt-in: both angles live on a 2^16-point midpoint lattice (4 bytes of state per
 *              monomer, twice the chains resident per CU), f32 arithmetic, f64 running sums.
 *              Discretisation bias of ensemble averages: O(h^2) ~ 1e-10. */
enum { PSTAT_F32 = 0, PSTAT_F64 = 1, PSTAT_Q16 = 2 };
/* per-chain generator (the reference uses Julia's unseeded default RNG; ours are seeded):
 *   PSTAT_RNG_MWC64X        multiply-with-carry MWC64X, streams split by 2^40-output skip-ahead (default)
 *   PSTAT_RNG_XOSHIRO128PP  xoshiro128++ seeded per chain through Philox4x32-10 */
enum { PSTAT_RNG_MWC64X = 0, PSTAT_RNG_XOSHIRO128PP = 1 };
#define PSTAT_MWC64X_MAX_CHAINS (1ull << 22)   /* global chain ids with pairwise disjoint MWC64X streams */
/* which main's step is run:
 *   PSTAT_MOVES_SINGLE   mcmc_eap_chain.jl:276-291 -- one single-monomer trial move per step (default)
 *   PSTAT_MOVES_CLUSTER  mcmc_clustering_eap_chain.jl:268-279 -- the same move followed, on the trial
 *                        chain, by cluster_flip! (inc/eap_chain.jl:269-333); bending energy; two more
 *                        averagers (sum cos^2 theta, mean bond angle).  All four energies; the all-pairs
 *                        ones (interacting, cutoff) run one chain per wavefront, n <= 512.  Non-interacting and
 *                        Ising: one chain per lane, or -- f64 handles of up to 4 096 chains and sweeps of many cases
 *                        of <= 16 chains each, n <= 256, MWC64X -- one chain per wavefront as well (a phase scan of
 *                        the reference is 2 730 single-chain cases: 2.8 us per step instead of 29; pstat_launch_info.kernel
 *                        names the choice).  Trajectories do not depend on it. */
enum { PSTAT_MOVES_SINGLE = 0, PSTAT_MOVES_CLUSTER = 1 };

/* Flattened pargs::Dict (mcmc_eap_chain.jl:155) -- the keys the force-ensemble step loop reads. */
typedef struct pstat_params {
  /* physics: inc/eap_chain.jl:89-108 */
  double E0, K1, K2, mu, kT, Fz, Fx, b;
  /* proposal + adaptation: mcmc_eap_chain.jl:88-118,172-174,301-322 */
  double phi_step, theta_step;
  double adj_lb, adj_ub, adj_scale;
  int64_t steps_per_adjust;
  int64_t n;             /* --num-monomers                                                    */
  int64_t num_chains;    /* chains per case on this handle (ours)                             */
  uint64_t seed;         /* ours: the reference never seeds its RNG                           */
  uint64_t chain_id0;    /* global id of this handle's first chain: shards over GPUs.  With
                          * PSTAT_RNG_MWC64X chain_id0 + num_chains must stay <= 2^22 (PSTAT_MWC64X_MAX_CHAINS):
                          * chain k starts k * 2^40 outputs down ONE sequence of period ~2^63, so ids beyond
                          * 2^23 would wrap onto earlier streams (the bound keeps a factor 2 in hand); xoshiro128++ has no such bound */
  int32_t chain_type;    /* PSTAT_DIELECTRIC | PSTAT_POLAR                                    */
  int32_t energy_type;   /* PSTAT_NONINTERACTING | PSTAT_INTERACTING | PSTAT_ISING | PSTAT_CUTOFF */
  int32_t do_flips;      /* --do-flips                                                        */
  int32_t umbrella;      /* --umbrella-sampling                                               */
  int32_t precision;     /* PSTAT_F32 | PSTAT_F64 | PSTAT_Q16                                 */
  int32_t device;        /* HIP device ordinal                                                */
  int32_t rng;           /* PSTAT_RNG_MWC64X | PSTAT_RNG_XOSHIRO128PP                         */
  int32_t move_set;      /* PSTAT_MOVES_SINGLE | PSTAT_MOVES_CLUSTER                          */
  /* options of mcmc_clustering_eap_chain.jl (:36-43,87-90,142-148); all 0 in mcmc_eap_chain.jl */
  double bend_mod, bend_angle;   /* --bend-mod, --bend-angle (per case)                       */
  double cluster_prob;           /* --cluster-prob (per case)                                 */
  double x0_phi, x0_theta;       /* --x0 "[phi; theta]"                                       */
  double dx0_phi, dx0_theta;     /* --dx0                                                     */
  int32_t use_x0;                /* start from x0 + Uniform(0, dx0) instead of uniform angles */
  int32_t uniform_bits;          /* random bits of the Metropolis eps: 0 = the precision's default (53 for PSTAT_F64, 23
                                  * for PSTAT_F32 / PSTAT_Q16, whose comparison runs in a 24-bit mantissa), 23, or 53 (f64
                                  * only).  Under 53 eps = (w_eps 2^21 + lo) 2^-53, lo = the low 9 bits of the step's dtheta
                                  * word, the low 9 of its dphi word and the low 3 of its index word -- bits no proposal
                                  * uses; no extra draw, so the two settings consume the same stream and differ only in
                                  * decisions that fall inside [u, u + 2^-23).  23 reproduces ABI 5's trajectories.      */
  double cutoff_radius;          /* --cutoff-radius, monomer lengths (PSTAT_CUTOFF; per case)  */
} pstat_params;

/* Order of every 16-vector below = the columns of <prefix>_rolling.csv after "step"
 * (mcmc_eap_chain.jl:259). */
enum {
  PSTAT_R1, PSTAT_R2, PSTAT_R3, PSTAT_R1SQ, PSTAT_R2SQ, PSTAT_R3SQ, PSTAT_RSQ,
  PSTAT_P1, PSTAT_P2, PSTAT_P3, PSTAT_P1SQ, PSTAT_P2SQ, PSTAT_P3SQ, PSTAT_PSQ,
  PSTAT_U, PSTAT_USQ, PSTAT_NOBS
};

/* Length (in doubles) of the device-side reduction vector of pstat_reduce_device():
 *   [0]        number of chains reduced
 *   [1..19]    sum over chains of the per-chain running mean of: the 16 observables, the per-chain
 *              acceptance ratio, sum_i cos^2(theta_i), the mean bond angle (the last two are recorded
 *              by the clustering main only, mcmc_clustering_eap_chain.jl:243-244)
 *   [20..38]   sum over chains of the squares of those per-chain means
 *   [39]       proposals rejected because their trial energy was not finite (see pstat_summary.nan_rejects)
 *   [40]       chains whose current configuration has collapsed (see pstat_summary.chains_collapsed)
 * Every entry is additive across handles/GPUs, so one all-reduce(SUM) merges ensembles. */
#define PSTAT_NQ 19
#define PSTAT_NX 2
#define PSTAT_NRED (1 + 2 * PSTAT_NQ + PSTAT_NX)

/* The ten stdout quantities of mcmc_eap_chain.jl:386-395 plus bookkeeping. */
typedef struct pstat_summary {
  double avg[PSTAT_NOBS];      /* pooled running averages, rolling.csv order                    */
  double stderr_[PSTAT_NOBS];  /* across-chain standard error of each (0 if one chain)          */
  double acceptance_ratio;     /* "AR": accepted / attempted over all chains and steps          */
  double ar_stderr;
  int64_t num_chains;
  int64_t steps_per_chain;     /* steps recorded so far by every chain                          */
  double attempted_updates;    /* num_chains * steps_per_chain                                  */
  double extra_avg[2];         /* <sum cos^2 theta>, <psi> (clustering main: "<cos2(theta)>", "<psi>") */
  double extra_stderr[2];
  /* Failure surfacing.  The reference rejects a proposal whose energy is NaN silently
   * (inc/acceptance.jl:29-39: every comparison with NaN is false) and has no excluded volume, so with the
   * pair energies (inc/eap_chain.jl:200-207,215-228: 1/r^3) a chain can fall into r -> 0 and stay there.
   *   nan_rejects       proposals, over all chains and recorded steps, whose trial energy AS THE DEVICE EVALUATES IT was
   *                     NaN or +-Inf (identically 0 for the non-interacting energy, whose dU is always finite).  This is
   *                     the device's arithmetic, not a replay of the reference's: whether a contact is r = 0 exactly
   *                     (0 * inf = NaN) or r ~ 1e-17 (a finite 1e50) depends on the order positions are summed in.  The
   *                     reference's cumsum (inc/eap_chain.jl:49-51) absorbs the 1e-16 components of a pole-clamped
   *                     monomer into its running sum and lands two such monomers on exactly the same point; the kernels
   *                     (bond vector b/2 (n_i + n_j) for neighbours, prefix scan / incremental shifts for all pairs) keep
   *                     them 1e-17 apart.  Measured on one seeded case (23 monomers, --do-flips, 5 x 1500 steps): the
   *                     CPU restatement of the reference's arithmetic counts 17, the device 0 -- and every one of those
   *                     proposals is rejected on both sides anyway (the clamp gives sin(theta') = 0), which is why the
   *                     trajectories agree bit for bit.  Where the non-finite value does not hinge on rounding (every
   *                     pair at r = 0 with --mlen 0: tests/test_gpu_validation.py) the counts are equal.
   *   chains_collapsed  chains whose CURRENT energy has |U| > 1e3 * n * (kT + |E0| mu_max / 2 + b (|Fx| + |Fz|)),
   *                     mu_max = max(|K1|, |K2|) |E0| or |mu|: a thousand times what n separated monomers can hold
   *                     in field, force and thermal energy; only a 1/r^3 contact gets there (also counts NaN). */
  int64_t nan_rejects;
  int64_t chains_collapsed;
} pstat_summary;

typedef struct pstat_handle pstat_handle;

int pstat_abi_version(void);
const char *pstat_strerror(int status);
const char *pstat_last_error(void);
int pstat_device_count(void);

/* Fills *p with the reference's option defaults (mcmc_eap_chain.jl:19-153). */
void pstat_default_params(pstat_params *p);

/* Replaces `chain = EAPChain(pargs); chain.U = U(chain)` and the averager construction
 * (mcmc_eap_chain.jl:175-176,242-255; inc/eap_chain.jl:60-135): draws phi~U(0,2pi), theta~U(0,pi)
 * for every chain on the device, derives r, p, U, zeroes the running sums.
 * `cases`/`ncases`: ncases >= 1 parameter sets that differ only in the physics scalars
 * (E0,K1,K2,mu,kT,Fz,Fx,b; bend_mod, bend_angle, cluster_prob, cutoff_radius; seed, chain_id0) -- a sweep grid run in
 * one launch; every case gets num_chains chains.  Chains of several cases share a wavefront when a case has fewer
 * chains than a wave has lanes and that shortens the launch (pstat_launch_info.packed_cases); a chain's trajectory
 * depends on its (seed, chain id) and its case's options only, never on what else is in the handle.
 * `stream`: a hipStream_t to launch on (e.g. torch's current stream) or NULL for the handle's own. */
int pstat_create(const pstat_params *cases, int32_t ncases, void *stream, pstat_handle **out);
void pstat_destroy(pstat_handle *h);

/* Replaces `nsteps` iterations of the step loop, mcmc_eap_chain.jl:276-328, for every chain:
 * proposal draw (:277-280), move! (inc/eap_chain.jl:230-257), energy (inc/energy.jl:7-23),
 * Metropolis (inc/acceptance.jl:29-39), step-size adaptation (:301-322), record! x 8 (:327-328).
 * With move_set = PSTAT_MOVES_CLUSTER: the step loop of mcmc_clustering_eap_chain.jl:268-311 instead
 * (the same move, then cluster_flip! on the trial chain, Metropolis-Hastings with alpha, record! x 10).
 * Asynchronous on the handle's stream. */
int pstat_advance(pstat_handle *h, int64_t nsteps);
int pstat_sync(pstat_handle *h);

/* Replaces the re-initialisation between inits, mcmc_eap_chain.jl:352-361: every chain draws a
 * fresh random configuration and adopts it if `force_init` or by metropolis_acc
 * (inc/acceptance.jl:1-3); the within-init step counter restarts at 1. */
int pstat_reinit(pstat_handle *h, int32_t force_init);

/* Burn-in support (the reference's clustering main, mcmc_clustering_eap_chain.jl:134-141,365-386,
 * discards a burn-in run made on a temperature ladder; mcmc_eap_chain.jl itself records from step 1).
 * pstat_reset_averages: zero the running sums, the acceptance totals and the recorded-step count,
 * keep the chains, generators and adapted step sizes.  pstat_set_kT: change the temperature of case
 * `icase` (all cases if < 0) for subsequent launches; the microstate is unaffected (U does not
 * depend on kT). */
int pstat_reset_averages(pstat_handle *h);
int pstat_set_kT(pstat_handle *h, int32_t icase, double kT);
/* kT of every case <- (the kT it was created with) * mult: one rung of the burn-in ladder for a whole
 * sweep grid (mcmc_clustering_eap_chain.jl:368,379: burnargs["kT"] = kT_base * kT_mult). */
int pstat_scale_kT(pstat_handle *h, double mult);
/* What a fresh call of the reference's mcmc(nsteps, pargs, chain) resets besides the averagers
 * (mcmc_clustering_eap_chain.jl:172-181,263-266): step sizes back to --phi-step/--theta-step, the
 * adaptation counters, the acceptor's cache, the in-run step counter.  The chains are kept. */
int pstat_reset_sampler(pstat_handle *h);

/* Device-side reduction over the chains of case `icase` (or over all cases if icase < 0) into
 * `dev_out`, a DEVICE pointer to PSTAT_NRED doubles owned by the caller (e.g. a torch tensor that
 * is then all-reduced with RCCL).  Asynchronous on the handle's stream. */
int pstat_reduce_device(pstat_handle *h, int32_t icase, double *dev_out);
/* The same vector copied back to host memory (synchronises): shards held by several handles in one
 * process (one per device) are merged by adding their vectors. */
int pstat_reduce_host(pstat_handle *h, int32_t icase, double red_out[PSTAT_NRED]);

/* Replaces get_avg() over the 8 averagers as written to rolling.csv (mcmc_eap_chain.jl:334-346;
 * inc/average.jl:38): pooled running averages of case `icase`, plus across-chain standard errors.
 * Either output pointer may be NULL.  Synchronises. */
int pstat_rolling(pstat_handle *h, int32_t icase, double avg_out[PSTAT_NOBS],
                  double stderr_out[PSTAT_NOBS]);

/* Replaces the trajectory.csv row source, mcmc_eap_chain.jl:330-333: r(3), p(3), U of one chain
 * (`chain` counts over all cases: case = chain / num_chains).  Synchronises. */
int pstat_microstate(pstat_handle *h, int64_t chain, double out[7]);

/* The quantities printed at mcmc_eap_chain.jl:365,386-395.  Synchronises.  Like every accessor that
 * synchronises (pstat_sync, pstat_reduce_host, pstat_rolling, pstat_microstate, pstat_chain_state,
 * pstat_chain_extras, pstat_checkpoint) it fails with PSTAT_ERR_HIP if a launch since the last successful
 * call did not run to completion (a job of the persistent kernels timed out waiting for its predecessor):
 * the handle's averages are then not the averages of the steps it was asked for. */
int pstat_summary_get(pstat_handle *h, int32_t icase, pstat_summary *out);

/* Turns already-merged reduction vectors (host memory, PSTAT_NRED doubles, e.g. after an
 * all-reduce over GPUs) into a summary.  Pure host arithmetic. */
int pstat_summary_from_reduction(const double red[PSTAT_NRED], int64_t steps_per_chain,
                                 pstat_summary *out);

/* Per-chain accessors for tests and tooling (host buffers).  angles: theta[n] then phi[n] as
 * doubles, radians; sums: the 16 per-chain running sums in rolling.csv order;
 * counters: {accepted_total, steps_recorded, nacc_window, natt_window};
 * steps: {phi_step, theta_step, normalizer}, normalizer = the averagers' denominator for this chain
 * (steps recorded, or the sum of 1/e^w under umbrella sampling up to a per-chain factor that the sums share: the gauge
 * of w rises with the heaviest configuration a chain has visited, so that neither overflows -- DESIGN.md 3.5). */
int pstat_chain_state(pstat_handle *h, int64_t chain, double *angles /* [2n] */,
                      double sums[PSTAT_NOBS], int64_t counters[4], double steps[3],
                      uint32_t rng[4]);

/* Per-chain running means of case `icase` (all cases if < 0), host memory: out[q * nchains + k], q < PSTAT_NQ in the
 * order of the reduction vector (16 observables, acceptance ratio, the clustering main's two extras), k counting
 * the chains of that case.  This is what the device reduction folds; a host that honours --numeric-type
 * (mcmc_eap_chain.jl:186-197: Float128 / Dec128 / BigFloat averagers) merges these in its own wide type -- the
 * per-chain sums themselves are Float64 on the device, like the reference's default.  Synchronises. */
int pstat_chain_means(pstat_handle *h, int32_t icase, double *out /* [PSTAT_NQ][nchains] */);

/* Start every chain over from a given configuration, as EAPChain(pargs) does with --x0/--dx0
 * (inc/eap_chain.jl:61-79): x0 holds [phi; theta] (len 2: every monomer) or the interleaved
 * [phi1, theta1, phi2, theta2, ...] (len 2n); each angle gets + Uniform(0, dx0).  Generators are
 * re-seeded, so the result is what pstat_create would have produced with that start; averagers,
 * step sizes and counters are reset.  x0 is host memory, copied before the call returns. */
int pstat_restart_from_x0(pstat_handle *h, const double *x0, int64_t len, double dx0_phi, double dx0_theta);
/* The clustering main's two extra averagers for one chain: their running sums and their value in
 * the current configuration (sum cos^2 theta; mean bond angle). */
int pstat_chain_extras(pstat_handle *h, int64_t chain, double extra_sums[2], double extra_now[2]);

/* Checkpoint / resume of the full device state (angles, generators, step sizes, counters, running
 * sums) and of every case's CURRENT kT (a rung of the burn-in ladder set by pstat_scale_kT / pstat_set_kT is
 * restored with the image).  pstat_checkpoint: call with buf == NULL to get the size in *bytes; a buffer that is too
 * small fails with PSTAT_ERR_TOO_SMALL and the required size written back.  pstat_restore continues exactly the run
 * the image was taken from, on a handle created with the same options: the image's header (format 4) names the ABI
 * version, n, the chain and case counts, precision, chain / energy type, generator, move set, umbrella, do-flips,
 * uniform_bits, case 0's seed and first chain id, and a fingerprint of all cases' physics scalars (other than the
 * current kT), seeds, chain ids, num_chains and the proposal / adaptation options; any mismatch, a foreign or
 * truncated buffer fails with PSTAT_ERR_BAD_CHECKPOINT and leaves the handle untouched.  The reference has no
 * equivalent. */
int pstat_checkpoint(pstat_handle *h, void *buf, size_t *bytes);
int pstat_restore(pstat_handle *h, const void *buf, size_t bytes);

/* Introspection for benchmarks: kernel name, LDS bytes per workgroup, workgroups, resident
 * workgroups per CU as given by the occupancy API for the sweep kernel of this handle. */
typedef struct pstat_launch_info {
  char kernel[64];
  int32_t lds_bytes;
  int32_t threads_per_block;
  int32_t lanes_per_block;
  int64_t blocks;
  int32_t blocks_per_cu;
  int32_t num_cus;
  int32_t packed_cases;     /* 1: a workgroup holds `lanes_per_block` consecutive chains whichever cases they belong to (picked
                             * by pstat_create when cases have few chains each: the reference's sweeps run 1-25 per case,
                             * run/K1_E0-kT-phase.jl:19-45); 0: workgroups never straddle a case                          */
  int32_t reserved;
} pstat_launch_info;
int pstat_launch_info_get(pstat_handle *h, pstat_launch_info *out);

#ifdef __cplusplus
}
#endif
#endif
