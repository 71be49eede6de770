/*
 * eap_oracle.h -- CPU restatement of the fixed-force-ensemble MCMC hot path of
 * grasingerm/polymer-stats (mcmc_eap_chain.jl + inc/{eap_chain,dipole_response,
 * energy,acceptance,average}.jl).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The product
 * (libpstat.so, polymer_stats_amd/) never links, imports or calls anything here.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this
 * path, never seeds its RNG, and is Julia (no interpreter in this image), so no
 * reference output exists to pin this restatement against.  It is pinned instead by
 * closed-form single-monomer integrals (tests/golden/ni_closed_form.json), by
 * hand-computable pair energies and by faithful-vs-incremental agreement.
 *
 * Citations "file:line" are relative to the reference tree.
 */
#ifndef EAP_ORACLE_H
#define EAP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { EAP_DIELECTRIC = 0, EAP_POLAR = 1 };              /* --chain-type  */
enum { EAP_NONINTERACTING = 0, EAP_INTERACTING = 1, EAP_ISING = 2,
       EAP_CUTOFF = 3 /* clustering main only: UCutoff, inc/eap_chain.jl:165-192 */ }; /* --energy-type */
enum { EAP_RNG_MWC64X = 0, EAP_RNG_XOSHIRO128PP = 1 };               /* per-chain generator */

/* Flat mirror of the option table mcmc_eap_chain.jl:19-153 (hot-path subset). */
typedef struct eap_params {
  double E0, K1, K2, mu, kT, Fz, Fx, b;
  double phi_step, theta_step;          /* mcmc_eap_chain.jl:88-98  */
  double adj_lb, adj_ub, adj_scale;     /* :103-114 */
  int64_t n;                            /* --num-monomers */
  int64_t num_steps;                    /* --num-steps    */
  int64_t num_inits;                    /* --num-inits    */
  int64_t steps_per_adjust;             /* :115-118 */
  int64_t stepout;                      /* :142-145 (rows only if buffers given) */
  uint64_t seed;                        /* ours: the reference never seeds */
  int32_t chain_type;
  int32_t energy_type;
  int32_t do_flips;
  int32_t force_init;
  int32_t umbrella;
  int32_t rng;          /* EAP_RNG_MWC64X | EAP_RNG_XOSHIRO128PP */
  /* --- options that only mcmc_clustering_eap_chain.jl has (eap_run_cluster); zero = absent --- */
  double bend_mod, bend_angle;          /* --bend-mod, --bend-angle (:36-43; inc/eap_chain.jl:54-58,91-92) */
  double cluster_prob;                  /* --cluster-prob (:87-90): probability of NOT attempting a cluster flip */
  double x0_phi, x0_theta;              /* --x0 "[phi; theta]" (:142-144; inc/eap_chain.jl:61-72) */
  double dx0_phi, dx0_theta;            /* --dx0 (:145-148) */
  double burn_sched[8];                 /* --burn-schedule kT multipliers (:138-141) */
  int64_t burn_in;                      /* --burn-in steps per rung (:134-137) */
  int32_t burn_nsched;                  /* number of rungs used */
  int32_t use_x0;
  double cutoff_radius;                 /* --cutoff-radius, in monomer lengths (:48-51; x mlen at inc/eap_chain.jl:102) */
  const double *x0_vec;                 /* --x0 of length 2n, [phi1, theta1, phi2, theta2, ...] (inc/eap_chain.jl:73-75); */
  int64_t x0_len;                       /*   used when use_x0 != 0 and x0_len == 2n, else x0_phi/x0_theta */
  int32_t uniform_bits;                 /* random bits of the Metropolis eps: 0 or 53 = 53 (the reference's rand() is a Float64
                                           with 52-53 random bits, mcmc_eap_chain.jl:287), 23 = (w >> 9) 2^-23; see eap_eps() */
  int32_t pad_;
} eap_params;

/* Index order = the rolling.csv columns after "step" (mcmc_eap_chain.jl:259). */
enum {
  EAP_R1, EAP_R2, EAP_R3, EAP_R1SQ, EAP_R2SQ, EAP_R3SQ, EAP_RSQ,
  EAP_P1, EAP_P2, EAP_P3, EAP_P1SQ, EAP_P2SQ, EAP_P3SQ, EAP_PSQ,
  EAP_U, EAP_USQ, EAP_NOBS
};

typedef struct eap_result {
  double sum[EAP_NOBS];   /* averager .value fields (inc/average.jl:9)       */
  double norm;            /* averager .normalizer (count, or sum of 1/e^w)   */
  int64_t nacc_total;     /* mcmc_eap_chain.jl:264,290                      */
  int64_t nsteps_total;   /* num_inits * num_steps                          */
  double phi_step, theta_step; /* step sizes after the last adaptation      */
  double r[3], p[3], U;   /* final microstate (trajectory.csv columns)      */
  uint32_t rng[4];        /* final generator state                          */
  double extra_sum[2];    /* clustering main only: sum cos^2(theta_i) and mean bond angle psi
                             (mcmc_clustering_eap_chain.jl:243-244) */
  int64_t nan_rejects;    /* proposals whose trial energy was NaN or +-Inf: the reference rejects them
                             silently (every comparison in inc/acceptance.jl:29-39 is false) */
} eap_result;

/* Optional per-run outputs; any pointer may be NULL. */
typedef struct eap_trace {
  double *final_phi;      /* [n]  */
  double *final_theta;    /* [n]  */
  uint8_t *accepted;      /* [num_inits*num_steps] 1 = accepted              */
  double *rolling_rows;   /* [rows][17]: step + 16 running averages          */
  double *traj_rows;      /* [rows][8]:  step,r1,r2,r3,p1,p2,p3,U            */
  int64_t max_rows;
  int64_t rows_written;
} eap_trace;

/* --- random stream contract (shared with the HIP path by specification) --- */
void eap_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void eap_rng_seed(uint64_t seed, uint64_t chain_id, uint32_t s[4]);   /* xoshiro128++ state */
uint32_t eap_xoshiro128pp_next(uint32_t s[4]);
void eap_mwc64x_seed(uint64_t seed, uint64_t chain_id, uint32_t s[4]); /* s[0] = x, s[1] = c */
uint32_t eap_mwc64x_next(uint32_t s[4]);
uint64_t eap_mwc64x_skip(uint64_t state, uint64_t nsteps);             /* state = c*2^32 + x */
double eap_u01(uint32_t w);                       /* (w>>9) * 2^-23 in [0,1) */
/* the Metropolis eps of one step from that step's words: w_eps, and -- under 53 bits -- the bits of the index, dphi and
 * dtheta words that nothing else uses: (w_eps 2^21 + (w_th & 511) 2^12 + (w_phi & 511) 2^3 + (w_idx & 7)) 2^-53 */
double eap_eps(int uniform_bits, uint32_t w_eps, uint32_t w_idx, uint32_t w_phi, uint32_t w_th);
/* Stream scan for tests: the 0-based steps s < nsteps of chain `chain_id` (fixed-force main, no --do-flips, one init: word
 * 2n + 4s + 3 of the chain's stream is that step's w_eps) whose eps has all 23 leading bits zero; returns how many were found
 * (at most `max_hits` are written). */
int64_t eap_find_eps23_zero(const eap_params *P, uint64_t chain_id, int64_t nsteps, int64_t *hits, int64_t max_hits);

/* --- the two restatements --- */
/* Literal algorithm: trial = deep copy, full prefix sum, full energy recompute,
 * cached log-density in the acceptor (mcmc_eap_chain.jl:171-376). */
int eap_run_faithful(const eap_params *P, uint64_t chain_id, eap_result *out, eap_trace *tr);
/* Same Markov chain from the same stream with O(1) incremental energy for
 * non-interacting/Ising and in-place update + pair recompute for interacting. */
int eap_run_fast(const eap_params *P, uint64_t chain_id, eap_result *out, eap_trace *tr);

/* The literal algorithm of mcmc_clustering_eap_chain.jl (single-monomer move followed by
 * cluster_flip!, inc/eap_chain.jl:269-333; bending energy; burn-in on a temperature ladder,
 * :365-386).  Energy types: noninteracting, Ising, interacting. */
int eap_run_cluster(const eap_params *P, uint64_t chain_id, eap_result *out, eap_trace *tr);

/* Many independent chains (chain ids id0 .. id0+nchains-1), one per worker
 * thread at a time, mirroring the reference's pmap process farm. mode: 0 faithful, 1 fast. */
int eap_run_many(const eap_params *P, uint64_t id0, int64_t nchains, int nthreads,
                 int mode /* 0 faithful, 1 fast, 2 clustering main */, eap_result *out /* [nchains] */);

/* Building blocks exported for hand-computable tests. */
void eap_dipole(const eap_params *P, double cphi, double sphi, double cth, double sth,
                double mu_out[3]);                                  /* dipole_response.jl:7-29 */
/* UCutoff: pairs with |r|^2 > rc^2 contribute 0 (inc/eap_chain.jl:171-192) */
double eap_pair_energy_cutoff(int64_t n, const double *xs, const double *mus, double rc);
double eap_pair_energy(int64_t n, const double *xs /*3xn col-major*/,
                       const double *mus /*3xn*/, int ising);       /* eap_chain.jl:196-228 */
double eap_chain_energy(const eap_params *P, const double *phi, const double *theta,
                        double r_out[3], double p_out[3]);          /* energy.jl:7-23 */

#ifdef __cplusplus
}
#endif
#endif
