/*
 * eap_oracle.c -- CPU restatement (plain C, fp64) of the reference's fixed-force MCMC path.
 * TEST INFRASTRUCTURE ONLY -- see eap_oracle.h.  PARITY UNPINNED (no reference fixtures exist).
 *
 * Two modes are provided:
 *   eap_run_faithful : the literal algorithm of mcmc_eap_chain.jl:171-376 -- every step makes a
 *                      deep copy of the chain, applies the move, recomputes all positions and the
 *                      full energy, and asks a Metropolis functor that caches log pi of the last
 *                      accepted state.
 *   eap_run_fast     : the same Markov chain (same stream, same decisions up to fp rounding)
 *                      with an O(1) energy difference; this is the form the HIP kernels use.
 *   eap_run_cluster  : mcmc_clustering_eap_chain.jl, literally: deep copy, move!, cluster_flip! with a
 *                      full recomputation per reflected member, bending energy, the acceptor caching
 *                      log(pi) + log(alpha), the burn-in ladder of fresh mcmc() calls.
 *
 * Random stream contract (ours; the reference is unseeded, mcmc_eap_chain.jl has no seed):
 *   generator, one per chain (eap_params.rng):
 *     EAP_RNG_MWC64X       MWC64X (D. Thomas): out = x ^ c; (c:x) <- A*x + c, A = 4294883355.  All chains
 *                          of a run walk one sequence; chain k starts k * 2^40 outputs after the base
 *                          state 1 + Philox4x32-10(key = seed, ctr = (0,0,0x5eed,1))[0:1] mod (M - 2),
 *                          M = A*2^32 - 1 (skip-ahead: state * A^(k 2^40) mod M)
 *     EAP_RNG_XOSHIRO128PP xoshiro128++ seeded with Philox4x32-10(key = seed, ctr = (chain_lo, chain_hi, 0x5eed, 0))
 *   u(w)    = (w >> 9) * 2^-23   (23 bits: exactly the f32 mantissa trick the kernels use)
 *   eps     : the Metropolis draw.  uniform_bits = 23: u(w_eps).  uniform_bits = 0 | 53 (default): 53 random bits like the
 *             reference's rand(), from w_eps and the bits of the step's other words that nothing else uses -- eap_eps()
 *   init    : phi_i = 2pi*u  for i = 1..n, then theta_i = pi*u for i = 1..n   (eap_chain.jl:6-7,61-62)
 *   step    : idx = mulhi32(w, n); dphi = phi_step*(2u-1); [flip bit = w>>31 if --do-flips];
 *             dtheta = theta_step*(2u-1); eps = u                              (mcmc_eap_chain.jl:277-287)
 *   re-init : 2n init draws, then eps = u unless --force-init                  (:353-358)
 *   clustering main (eap_run_cluster), per step: idx, dphi, dtheta as above; then the cluster's draws --
 *             skip = u (no cluster if skip <= cluster_prob), then per round one draw for the link
 *             above the cluster while that end grows and one for the link below while that end grows
 *             (the reference finishes the upper loop before the lower one; same law, see cluster_flip)
 *             -- and eps = u last.  --x0 start: phi_i = x0_phi + dx0_phi*u (all i), then theta likewise.
 */
#include "eap_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ RNG */

void eap_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  /* Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11). */
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
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

void eap_rng_seed(uint64_t seed, uint64_t chain_id, uint32_t s[4]) {
  uint32_t ctr[4] = {(uint32_t)chain_id, (uint32_t)(chain_id >> 32), 0x5eedu, 0u};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  eap_philox4x32_10(ctr, key, s);
  if ((s[0] | s[1] | s[2] | s[3]) == 0u) s[0] = 1u; /* xoshiro must not start at zero */
}

static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

uint32_t eap_xoshiro128pp_next(uint32_t s[4]) {
  /* Blackman & Vigna, xoshiro128++ 1.0 */
  uint32_t result = rotl32(s[0] + s[3], 7) + s[0];
  uint32_t t = s[1] << 9;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl32(s[3], 11);
  return result;
}

/* MWC64X, David B. Thomas, "The MWC64X random number generator" (2011).  As an LCG:
 * s' = A s mod M with s = c 2^32 + x and M = A 2^32 - 1, which gives exact skip-ahead. */
#define MWC_A 4294883355u
#define MWC_M 0xFFFEB81AFFFFFFFFull

static uint64_t mwc_mulmod(uint64_t a, uint64_t b) {
  return (uint64_t)(((unsigned __int128)a * b) % MWC_M);
}
static uint64_t mwc_powmod(uint64_t g, uint64_t e) {
  uint64_t r = 1;
  for (; e; e >>= 1, g = mwc_mulmod(g, g))
    if (e & 1) r = mwc_mulmod(r, g);
  return r;
}
uint64_t eap_mwc64x_skip(uint64_t state, uint64_t nsteps) {
  return mwc_mulmod(state, mwc_powmod(MWC_A, nsteps));
}
void eap_mwc64x_seed(uint64_t seed, uint64_t chain_id, uint32_t s[4]) {
  uint32_t ctr[4] = {0u, 0u, 0x5eedu, 1u};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t o[4];
  eap_philox4x32_10(ctr, key, o);
  uint64_t v = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
  uint64_t base = 1 + v % (MWC_M - 2);
  /* chain k starts k * 2^40 outputs down the sequence: A^(k 2^40) = (A^(2^40))^k */
  uint64_t g40 = mwc_powmod(MWC_A, 1ull << 40);
  uint64_t st = mwc_mulmod(base, mwc_powmod(g40, chain_id));
  s[0] = (uint32_t)st; s[1] = (uint32_t)(st >> 32); s[2] = 0u; s[3] = 0u;
}
uint32_t eap_mwc64x_next(uint32_t s[4]) {
  uint32_t r = s[0] ^ s[1];
  uint64_t t = (uint64_t)s[0] * MWC_A + s[1];
  s[0] = (uint32_t)t; s[1] = (uint32_t)(t >> 32);
  return r;
}

double eap_u01(uint32_t w) { return (double)(w >> 9) * (1.0 / 8388608.0); }

double eap_eps(int uniform_bits, uint32_t w_eps, uint32_t w_idx, uint32_t w_phi, uint32_t w_th) {
  if (uniform_bits == 23) return eap_u01(w_eps);
  const uint64_t lo = ((uint64_t)(w_th & 511u) << 12) | ((uint64_t)(w_phi & 511u) << 3) | (uint64_t)(w_idx & 7u);
  return (double)(((uint64_t)w_eps << 21) | lo) * 0x1p-53;      /* < 2^53: exact */
}

/* generator state: s[0..3] + the kind in s[4] (kept beside the state so every draw site stays a
 * one-liner) */
static inline uint32_t draw_w(uint32_t s[5]) {
  return s[4] == EAP_RNG_XOSHIRO128PP ? eap_xoshiro128pp_next(s) : eap_mwc64x_next(s);
}
static inline double draw_u(uint32_t s[5]) { return eap_u01(draw_w(s)); }
static inline int64_t idx_of(uint32_t w, int64_t n) { return (int64_t)(((uint64_t)w * (uint64_t)n) >> 32); }
static inline double sym_of(uint32_t w) { return 2.0 * eap_u01(w) - 1.0; }
static void seed_chain(const eap_params *P, uint64_t chain_id, uint32_t s[5]) {
  s[4] = (uint32_t)P->rng;
  if (P->rng == EAP_RNG_XOSHIRO128PP) eap_rng_seed(P->seed, chain_id, s);
  else eap_mwc64x_seed(P->seed, chain_id, s);
}

/* ------------------------------------------------------------------ physics pieces */

/* dipole_response.jl:7-11 (dielectric) and :27-29 (polar, M = mu*I from eap_chain.jl:84) */
void eap_dipole(const eap_params *P, double cphi, double sphi, double cth, double sth,
                double m[3]) {
  double nx = cphi * sth, ny = sphi * sth, nz = cth; /* eap_chain.jl:40 */
  if (P->chain_type == EAP_DIELECTRIC) {
    double a = (P->K1 - P->K2) * P->E0 * cth;
    m[0] = a * nx;
    m[1] = a * ny;
    m[2] = a * nz + P->K2 * P->E0;
  } else {
    m[0] = P->mu * nx;
    m[1] = P->mu * ny;
    m[2] = P->mu * nz;
  }
}

/* one dipole-dipole term, eap_chain.jl:200-207 */
static inline double pair_term(const double *xi, const double *xj, const double *mi,
                               const double *mj) {
  double rx = xi[0] - xj[0], ry = xi[1] - xj[1], rz = xi[2] - xj[2];
  double r2 = rx * rx + ry * ry + rz * rz;
  double rmag = sqrt(r2);
  double hx = rx / rmag, hy = ry / rmag, hz = rz / rmag;
  double r3 = r2 * rmag;
  double mimj = mi[0] * mj[0] + mi[1] * mj[1] + mi[2] * mj[2];
  double mir = mi[0] * hx + mi[1] * hy + mi[2] * hz;
  double mjr = mj[0] * hx + mj[1] * hy + mj[2] * hz;
  return (mimj - 3 * mir * mjr) / (4 * M_PI * r3);
}

/* U_interaction (eap_chain.jl:196-211) or U_Ising (:215-228) */
double eap_pair_energy(int64_t n, const double *xs, const double *mus, int ising) {
  double U = 0.0;
  if (ising) {
    for (int64_t i = 0; i + 1 < n; ++i)
      U += pair_term(xs + 3 * i, xs + 3 * (i + 1), mus + 3 * i, mus + 3 * (i + 1));
  } else {
    for (int64_t i = 0; i < n; ++i)
      for (int64_t j = i + 1; j < n; ++j)
        U += pair_term(xs + 3 * i, xs + 3 * j, mus + 3 * i, mus + 3 * j);
  }
  return U;
}

double eap_pair_energy_cutoff(int64_t n, const double *xs, const double *mus, double rc) {
  const double crad2 = rc * rc;
  double U = 0.0;
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = i + 1; j < n; ++j) {
      const double *xi = xs + 3 * i, *xj = xs + 3 * j;
      double rx = xi[0] - xj[0], ry = xi[1] - xj[1], rz = xi[2] - xj[2];
      double r2 = rx * rx + ry * ry + rz * rz;
      U += (r2 > crad2) ? 0.0 : pair_term(xi, xj, mus + 3 * i, mus + 3 * j);
    }
  return U;
}

/* ------------------------------------------------------------------ literal chain object */

typedef struct chain_t { /* inc/eap_chain.jl:12-36 */
  int64_t n;
  double *phi, *cphi, *sphi, *th, *cth, *sth; /* n each      */
  double *nh, *mus, *xs;                      /* 3 x n, column-major */
  double *us;                                 /* n: u_i + bending energy of bond (i, i+1), :53-58 */
  double *psis;                               /* n (n-1 used): bond angles, :45-47 */
  double r[3];
  double Omega;
  double U;
  double *block; /* owns everything above */
} chain_t;

static int chain_alloc(chain_t *c, int64_t n) {
  c->n = n;
  c->block = (double *)malloc(sizeof(double) * (size_t)(17 * n));
  if (!c->block) return -1;
  double *q = c->block;
  c->phi = q; q += n;  c->cphi = q; q += n;  c->sphi = q; q += n;
  c->th = q;  q += n;  c->cth = q;  q += n;  c->sth = q;  q += n;
  c->us = q;  q += n;  c->psis = q; q += n;
  c->nh = q;  q += 3 * n;  c->mus = q; q += 3 * n;  c->xs = q;
  return 0;
}
static void chain_free(chain_t *c) { free(c->block); c->block = NULL; }

/* EAPChain(chain::EAPChain) deep copy, eap_chain.jl:137-163 */
static void chain_copy(chain_t *dst, const chain_t *src) {
  memcpy(dst->block, src->block, sizeof(double) * (size_t)(17 * src->n));
  dst->r[0] = src->r[0]; dst->r[1] = src->r[1]; dst->r[2] = src->r[2];
  dst->Omega = src->Omega;
  dst->U = src->U;
}

/* update_xs!, eap_chain.jl:49-51: x_i = b (cumsum(n)_i - n_i/2) */
static void update_xs(const eap_params *P, chain_t *c) {
  double sx = 0, sy = 0, sz = 0;
  for (int64_t i = 0; i < c->n; ++i) {
    sx += c->nh[3 * i]; sy += c->nh[3 * i + 1]; sz += c->nh[3 * i + 2];
    c->xs[3 * i]     = P->b * (sx - 0.5 * c->nh[3 * i]);
    c->xs[3 * i + 1] = P->b * (sy - 0.5 * c->nh[3 * i + 1]);
    c->xs[3 * i + 2] = P->b * (sz - 0.5 * c->nh[3 * i + 2]);
  }
}

/* end_to_end, eap_chain.jl:405-406 */
static void end_to_end(const eap_params *P, const chain_t *c, double r[3]) {
  int64_t l = c->n - 1;
  for (int k = 0; k < 3; ++k) r[k] = c->xs[3 * l + k] + P->b / 2.0 * c->nh[3 * l + k];
}

/* chain_mu, eap_chain.jl:408 */
static void chain_mu(const chain_t *c, double p[3]) {
  p[0] = p[1] = p[2] = 0.0;
  for (int64_t i = 0; i < c->n; ++i) {
    p[0] += c->mus[3 * i]; p[1] += c->mus[3 * i + 1]; p[2] += c->mus[3 * i + 2];
  }
}

static double sum_us(const chain_t *c) {
  double s = 0.0;
  for (int64_t i = 0; i < c->n; ++i) s += c->us[i];
  return s;
}

/* energy.jl:7-23 */
static double chain_U(const eap_params *P, const chain_t *c) {
  /* UCutoff's functor returns the truncated pair sum ALONE -- no sum(us), no -F.r (inc/eap_chain.jl:171-192
   * vs inc/energy.jl:13-16): with --energy-type cutoff neither the field nor the force enters U. */
  if (P->energy_type == EAP_CUTOFF)
    return eap_pair_energy_cutoff(c->n, c->xs, c->mus, P->cutoff_radius * P->b);
  double r[3];
  end_to_end(P, c, r);
  double U = sum_us(c);
  if (P->energy_type == EAP_INTERACTING) U += eap_pair_energy(c->n, c->xs, c->mus, 0);
  else if (P->energy_type == EAP_ISING)  U += eap_pair_energy(c->n, c->xs, c->mus, 1);
  return U - (r[0] * P->Fx + r[1] * 0.0 + r[2] * P->Fz);
}

static void set_nhat_mu(const eap_params *P, chain_t *c, int64_t i) {
  c->nh[3 * i]     = c->cphi[i] * c->sth[i];
  c->nh[3 * i + 1] = c->sphi[i] * c->sth[i];
  c->nh[3 * i + 2] = c->cth[i];
  eap_dipole(P, c->cphi[i], c->sphi[i], c->cth[i], c->sth[i], c->mus + 3 * i);
}
/* psi_j, eap_chain.jl:45-47: angle between monomers j and j+1 */
static double psi_j(const chain_t *c, int64_t j) {
  double d = c->nh[3 * j] * c->nh[3 * j + 3] + c->nh[3 * j + 1] * c->nh[3 * j + 4] + c->nh[3 * j + 2] * c->nh[3 * j + 5];
  return acos(fmin(1, fmax(-1, d)));
}
/* u() + ubend(), eap_chain.jl:53-58: kappa = 0 in mcmc_eap_chain.jl (no --bend-mod there, :91) */
static void set_u(const eap_params *P, chain_t *c, int64_t i) {
  double ubend = (i != c->n - 1) ? P->bend_mod / 2 * (c->psis[i] - P->bend_angle) * (c->psis[i] - P->bend_angle) : 0.0;
  c->us[i] = -1.0 / 2.0 * P->E0 * c->mus[3 * i + 2] + ubend;
}

/* derive every cached field from (phi, theta): the tail of EAPChain(pargs), eap_chain.jl:109-134 */
static void chain_derive(const eap_params *P, chain_t *c) {
  double prod = 1.0;
  for (int64_t i = 0; i < c->n; ++i) {
    c->cphi[i] = cos(c->phi[i]); c->sphi[i] = sin(c->phi[i]);
    c->cth[i] = cos(c->th[i]);   c->sth[i] = sin(c->th[i]);
    prod *= c->sth[i];
  }
  c->Omega = log(prod); /* eap_chain.jl:117 */
  for (int64_t i = 0; i < c->n; ++i) set_nhat_mu(P, c, i);          /* :124-129 */
  for (int64_t i = 0; i + 1 < c->n; ++i) c->psis[i] = psi_j(c, i);  /* :125 */
  if (c->n > 0) c->psis[c->n - 1] = 0.0;
  for (int64_t i = 0; i < c->n; ++i) set_u(P, c, i);                /* :130 */
  update_xs(P, c);
  end_to_end(P, c, c->r);
  c->U = chain_U(P, c);
}

/* EAPChain(pargs), eap_chain.jl:60-135: all phi draws, then all theta draws */
static void chain_random(const eap_params *P, uint32_t rng[5], chain_t *c) {
  if (P->use_x0 && P->x0_vec && P->x0_len == 2 * c->n) { /* eap_chain.jl:73-75: per-monomer start, interleaved */
    for (int64_t i = 0; i < c->n; ++i) c->phi[i] = P->x0_vec[2 * i] + P->dx0_phi * draw_u(rng);
    for (int64_t i = 0; i < c->n; ++i) c->th[i] = P->x0_vec[2 * i + 1] + P->dx0_theta * draw_u(rng);
  } else if (P->use_x0) { /* eap_chain.jl:69-72: x0 = [phi; theta] plus Uniform(0, dx0) */
    for (int64_t i = 0; i < c->n; ++i) c->phi[i] = P->x0_phi + P->dx0_phi * draw_u(rng);
    for (int64_t i = 0; i < c->n; ++i) c->th[i] = P->x0_theta + P->dx0_theta * draw_u(rng);
  } else {
    for (int64_t i = 0; i < c->n; ++i) c->phi[i] = (2.0 * M_PI) * draw_u(rng);
    for (int64_t i = 0; i < c->n; ++i) c->th[i] = M_PI * draw_u(rng);
  }
  chain_derive(P, c);
}

/* move!(chain, idx, dphi, dtheta), eap_chain.jl:230-257 */
static void chain_move(const eap_params *P, chain_t *c, int64_t idx, double dphi, double dth) {
  c->phi[idx] += dphi;
  c->cphi[idx] = cos(c->phi[idx]);
  c->sphi[idx] = sin(c->phi[idx]);
  c->th[idx] = fmin(M_PI, fmax(0.0, c->th[idx] + dth));
  double sth = sin(c->th[idx]);
  c->Omega += log(sth / c->sth[idx]);
  c->cth[idx] = cos(c->th[idx]);
  c->sth[idx] = sth;
  set_nhat_mu(P, c, idx);                                    /* :243-245, "the order ... is important" */
  if (idx < c->n - 1) c->psis[idx] = psi_j(c, idx);          /* :246 */
  if (idx > 0) { c->psis[idx - 1] = psi_j(c, idx - 1); set_u(P, c, idx - 1); }   /* :247-250 */
  set_u(P, c, idx);                                          /* :251 */
  update_xs(P, c);
  end_to_end(P, c, c->r);
  c->U = chain_U(P, c);
}

/* AntiDipoleWeightFunction, average.jl:104-124 */
typedef struct { int on; double log_gauge; double scale; } weight_t;

static weight_t weight_make(const eap_params *P, double Omega_initial) {
  weight_t w;
  w.on = P->umbrella;
  double lead = (P->chain_type == EAP_DIELECTRIC) ? (P->K1 + 2 * P->K2) * P->E0 * P->E0
                                                  : P->mu * P->E0; /* eigvals(mu*I)[end] = mu */
  w.log_gauge = -lead * (double)P->n / (3 * P->kT) + Omega_initial;
  w.scale = (0.2 + 0.8 * exp(-(P->Fx * P->Fx + P->Fz * P->Fz) / P->kT)) / P->kT;
  return w;
}
static double weight_eval(const weight_t *w, double usum) {
  return w->on ? usum * w->scale - w->log_gauge : 1.0; /* WeightlessFunction == 1.0, average.jl:102 */
}

typedef struct { double sum[EAP_NOBS]; double norm; } averagers_t;

/* record! x 8, mcmc_eap_chain.jl:242-255,327-328; average.jl:40-48,63-67 */
static void record(averagers_t *A, const double r[3], const double p[3], double U, int umbrella,
                   double w) {
  double expw = umbrella ? exp(w) : 1.0;
  double v[EAP_NOBS];
  v[EAP_R1] = r[0]; v[EAP_R2] = r[1]; v[EAP_R3] = r[2];
  v[EAP_R1SQ] = r[0] * r[0]; v[EAP_R2SQ] = r[1] * r[1]; v[EAP_R3SQ] = r[2] * r[2];
  v[EAP_RSQ] = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  v[EAP_P1] = p[0]; v[EAP_P2] = p[1]; v[EAP_P3] = p[2];
  v[EAP_P1SQ] = p[0] * p[0]; v[EAP_P2SQ] = p[1] * p[1]; v[EAP_P3SQ] = p[2] * p[2];
  v[EAP_PSQ] = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
  v[EAP_U] = U; v[EAP_USQ] = U * U;
  if (umbrella) {
    for (int k = 0; k < EAP_NOBS; ++k) A->sum[k] += v[k] / expw;
    A->norm += 1.0 / expw;
  } else {
    for (int k = 0; k < EAP_NOBS; ++k) A->sum[k] += v[k];
    A->norm += 1;
  }
}

static void emit_rows(eap_trace *tr, int64_t step, const averagers_t *A, const double r[3],
                      const double p[3], double U) {
  if (!tr || tr->rows_written >= tr->max_rows) return;
  int64_t k = tr->rows_written;
  if (tr->rolling_rows) { /* mcmc_eap_chain.jl:334-346 */
    double *row = tr->rolling_rows + 17 * k;
    row[0] = (double)step;
    for (int q = 0; q < EAP_NOBS; ++q) row[1 + q] = A->sum[q] / A->norm;
  }
  if (tr->traj_rows) { /* :330-333 */
    double *row = tr->traj_rows + 8 * k;
    row[0] = (double)step;
    row[1] = r[0]; row[2] = r[1]; row[3] = r[2];
    row[4] = p[0]; row[5] = p[1]; row[6] = p[2];
    row[7] = U;
  }
  if (tr->rolling_rows || tr->traj_rows) tr->rows_written = k + 1;
}

/* step-size adaptation, mcmc_eap_chain.jl:301-322 */
static void adapt(const eap_params *P, int64_t step, double *phistep, double *thstep,
                  int64_t *nacc, int64_t *natt) {
  if (P->adj_scale != 1.0 && P->steps_per_adjust > 0 && step % P->steps_per_adjust == 0) {
    double ratio = (double)*nacc / (double)*natt;
    if (ratio > P->adj_ub && *phistep != M_PI && *thstep != M_PI / 2) {
      *nacc = 0; *natt = 0;
      *phistep = fmin(M_PI, *phistep * P->adj_scale);
      *thstep = fmin(M_PI / 2, *thstep * P->adj_scale);
    } else if (ratio < P->adj_lb) {
      *nacc = 0; *natt = 0;
      *phistep /= P->adj_scale;
      *thstep /= P->adj_scale;
    }
  }
}

static int check_params(const eap_params *P) {
  if (P->n < 1 || P->num_steps < 0 || P->num_inits < 1) return -1;
  if (P->uniform_bits != 0 && P->uniform_bits != 23 && P->uniform_bits != 53) return -1;
  if (P->chain_type != EAP_DIELECTRIC && P->chain_type != EAP_POLAR) return -1;
  if (P->energy_type < 0 || P->energy_type > EAP_CUTOFF) return -1;
  if (P->rng != EAP_RNG_MWC64X && P->rng != EAP_RNG_XOSHIRO128PP) return -1;
  return 0;
}

/* ------------------------------------------------------------------ faithful run */

int eap_run_faithful(const eap_params *P, uint64_t chain_id, eap_result *out, eap_trace *tr) {
  if (check_params(P)) return -1;
  if (P->energy_type == EAP_CUTOFF) return -1; /* mcmc_eap_chain.jl has no --cutoff-radius: KeyError there */
  uint32_t rng[5];
  seed_chain(P, chain_id, rng);
  chain_t cur, trial, fresh;
  if (chain_alloc(&cur, P->n) || chain_alloc(&trial, P->n) || chain_alloc(&fresh, P->n)) return -2;
  if (tr) tr->rows_written = 0;

  double phistep = P->phi_step, thstep = P->theta_step;   /* :172 */
  chain_random(P, rng, &cur);                             /* :175-176 */
  weight_t wf = weight_make(P, cur.Omega);                /* :177 */
  double logpi_prev = -cur.U / P->kT + cur.Omega + (wf.on ? weight_eval(&wf, sum_us(&cur)) : 1.0); /* :178-182 */

  averagers_t A;
  memset(&A, 0, sizeof A);
  int64_t nacc = 0, nacc_total = 0, natt = 0, nan_rejects = 0;
  int64_t t = 0;

  for (int64_t init = 1; init <= P->num_inits; ++init) {        /* :266 */
    for (int64_t step = 1; step <= P->num_steps; ++step, ++t) { /* :276 */
      const uint32_t w_idx = draw_w(rng), w_phi = draw_w(rng);
      int64_t idx = idx_of(w_idx, P->n);                        /* :277 */
      double dphi = phistep * sym_of(w_phi);                    /* :278 */
      double flip = 0.0;
      if (P->do_flips && (draw_w(rng) >> 31))                   /* :279 */
        flip = M_PI - 2 * cur.th[idx];
      const uint32_t w_th = draw_w(rng);
      double dth = flip + thstep * sym_of(w_th);                /* :280 */
      chain_copy(&trial, &cur);                                 /* :281 */
      chain_move(P, &trial, idx, dphi, dth);                    /* :283 */
      double eps = eap_eps(P->uniform_bits, draw_w(rng), w_idx, w_phi, w_th);   /* :287 */
      /* Metropolis functor, acceptance.jl:29-39 */
      double logpi = -trial.U / P->kT + trial.Omega + (wf.on ? weight_eval(&wf, sum_us(&trial)) : 1.0);
      int ok = (logpi >= logpi_prev) || (eps < exp(logpi - logpi_prev));
      if (!isfinite(trial.U)) ++nan_rejects;
      if (ok) {
        logpi_prev = logpi;
        chain_t tmp = cur; cur = trial; trial = tmp;            /* :288 */
        ++nacc; ++nacc_total;
      }
      ++natt;
      if (tr && tr->accepted) tr->accepted[t] = (uint8_t)ok;
      adapt(P, step, &phistep, &thstep, &nacc, &natt);          /* :301-322 */
      double p[3];
      chain_mu(&cur, p);
      double w = wf.on ? weight_eval(&wf, sum_us(&cur)) : 1.0;
      record(&A, cur.r, p, cur.U, P->umbrella, w);              /* :327-328 */
      if (P->stepout > 0 && step % P->stepout == 0) emit_rows(tr, step, &A, cur.r, p, cur.U);
    }
    /* re-initialisation, :352-361.  The reference also does it after the last init, where nothing
     * can observe it any more; skipped there so that `out` reports the last sampled microstate. */
    if (init == P->num_inits) break;
    chain_random(P, rng, &fresh);
    int adopt = P->force_init;
    if (!adopt) {
      double pa = 1.0, pb = 1.0;
      for (int64_t i = 0; i < P->n; ++i) { pa *= cur.sth[i]; pb *= fresh.sth[i]; }
      double eps = draw_u(rng);
      adopt = eps <= (exp(-(fresh.U - cur.U) / P->kT) * pb / pa); /* acceptance.jl:1-3 */
    }
    if (adopt) { chain_t tmp = cur; cur = fresh; fresh = tmp; }  /* acceptor keeps its stale cache */
  }

  memcpy(out->sum, A.sum, sizeof A.sum);
  out->norm = A.norm;
  out->extra_sum[0] = out->extra_sum[1] = 0.0;
  out->nacc_total = nacc_total;
  out->nan_rejects = nan_rejects;
  out->nsteps_total = P->num_inits * P->num_steps;
  out->phi_step = phistep;
  out->theta_step = thstep;
  memcpy(out->r, cur.r, sizeof cur.r);
  chain_mu(&cur, out->p);
  out->U = cur.U;
  memcpy(out->rng, rng, sizeof out->rng);  /* the four state words */
  if (tr && tr->final_phi) memcpy(tr->final_phi, cur.phi, sizeof(double) * (size_t)P->n);
  if (tr && tr->final_theta) memcpy(tr->final_theta, cur.th, sizeof(double) * (size_t)P->n);
  chain_free(&cur); chain_free(&trial); chain_free(&fresh);
  return 0;
}

/* ------------------------------------------------------------------ clustering main */

/* cluster_flip!, inc/eap_chain.jl:269-333 (flip_f! = refl_n!, pflip = pflip_linear) */
static double cluster_flip(const eap_params *P, uint32_t rng[5], chain_t *c, int64_t idx) {
  if (draw_u(rng) <= P->cluster_prob) return 1.0;                                  /* :276 */
  const int64_t n = c->n;
  double upper_p = 0.0, lower_p = 0.0;
  int64_t upper = idx, lower = idx;
  /* :281-309.  The reference grows the upper end to completion, then the lower end.  Here the two
   * loops are interleaved -- per round one link test above (if that end is still growing), then one
   * below -- which is the stream contract shared with the device kernel.  The two ends read disjoint
   * links and every test has its own iid draw, so the law of (lower, upper) is unchanged. */
  int gu = upper < n - 1, gl = lower > 0;     /* at a chain end: p = 0, no draw (:282-284,299-301) */
  while (gu || gl) {
    if (gu) {
      double d = c->nh[3 * upper] * c->nh[3 * upper + 3] + c->nh[3 * upper + 1] * c->nh[3 * upper + 4] +
                 c->nh[3 * upper + 2] * c->nh[3 * upper + 5];
      upper_p = (1 + d) / 2;
      if (draw_u(rng) <= upper_p) { ++upper; if (upper >= n - 1) { upper_p = 0.0; gu = 0; } }
      else gu = 0;
    }
    if (gl) {
      double d = c->nh[3 * lower] * c->nh[3 * lower - 3] + c->nh[3 * lower + 1] * c->nh[3 * lower - 2] +
                 c->nh[3 * lower + 2] * c->nh[3 * lower - 1];
      lower_p = (1 + d) / 2;
      if (draw_u(rng) <= lower_p) { --lower; if (lower <= 0) { lower_p = 0.0; gl = 0; } }
      else gl = 0;
    }
  }
  for (int64_t i = lower; i <= upper; ++i) chain_move(P, c, i, 0.0, M_PI - 2 * c->th[i]);  /* refl_n!, :263-265,314-316 */
  double new_upper_p = 0.0, new_lower_p = 0.0;                                      /* :318-327 */
  if (upper < n - 1)
    new_upper_p = (1 + (c->nh[3 * upper] * c->nh[3 * upper + 3] + c->nh[3 * upper + 1] * c->nh[3 * upper + 4] +
                        c->nh[3 * upper + 2] * c->nh[3 * upper + 5])) / 2;
  if (lower > 0)
    new_lower_p = (1 + (c->nh[3 * lower] * c->nh[3 * lower - 3] + c->nh[3 * lower + 1] * c->nh[3 * lower - 2] +
                        c->nh[3 * lower + 2] * c->nh[3 * lower - 1])) / 2;
  return ((1 - new_upper_p) * (1 - new_lower_p)) / ((1 - upper_p) * (1 - lower_p));
}

static void record_extra(double ex[2], const chain_t *c, int umbrella, double w) {
  double c2 = 0.0, ps = 0.0;
  for (int64_t i = 0; i < c->n; ++i) c2 += c->cth[i] * c->cth[i];   /* :243 */
  for (int64_t i = 0; i + 1 < c->n; ++i) ps += c->psis[i];
  ps /= (double)(c->n - 1);                                           /* :244 */
  double expw = umbrella ? exp(w) : 1.0;
  ex[0] += umbrella ? c2 / expw : c2;
  ex[1] += umbrella ? ps / expw : ps;
}

/* one call of mcmc(nsteps, pargs, chain), mcmc_clustering_eap_chain.jl:172-352 */
static void cluster_stage(const eap_params *P, int64_t nsteps, uint32_t rng[5], chain_t *cur, chain_t *trial,
                          averagers_t *A, double extra[2], int64_t *nacc_total_out, double steps_out[2],
                          eap_trace *tr, int64_t *t, int64_t *nan_rejects) {
  *nan_rejects = 0;
  double phistep = P->phi_step, thstep = P->theta_step;           /* :174 */
  cur->U = chain_U(P, cur);                                       /* :177 */
  weight_t wf = weight_make(P, cur->Omega);                       /* :178 */
  double logpi_prev = -cur->U / P->kT + cur->Omega + (wf.on ? weight_eval(&wf, sum_us(cur)) : 1.0);
  memset(A, 0, sizeof *A);
  extra[0] = extra[1] = 0.0;
  int64_t nacc = 0, natt = 0, nacc_total = 0;
  for (int64_t step = 1; step <= nsteps; ++step, ++*t) {          /* :268 */
    const uint32_t w_idx = draw_w(rng), w_phi = draw_w(rng), w_th = draw_w(rng);
    int64_t idx = idx_of(w_idx, P->n);
    double dphi = phistep * sym_of(w_phi);
    double dth = thstep * sym_of(w_th);
    chain_copy(trial, cur);
    chain_move(P, trial, idx, dphi, dth);                         /* :272 */
    double alpha = cluster_flip(P, rng, trial, idx);              /* :273 */
    double eps = eap_eps(P->uniform_bits, draw_w(rng), w_idx, w_phi, w_th);
    /* acceptance.jl:29-39 with alpha; the cached value keeps the log(alpha) of the accepted move */
    double logpi = -trial->U / P->kT + trial->Omega + (wf.on ? weight_eval(&wf, sum_us(trial)) : 1.0) + log(alpha);
    int ok = (logpi >= logpi_prev) || (eps < exp(logpi - logpi_prev));
    if (!isfinite(trial->U)) ++*nan_rejects;
    if (ok) {
      logpi_prev = logpi;
      chain_t tmp = *cur; *cur = *trial; *trial = tmp;
      ++nacc; ++nacc_total;
    }
    ++natt;
    if (tr && tr->accepted) tr->accepted[*t] = (uint8_t)ok;
    adapt(P, step, &phistep, &thstep, &nacc, &natt);              /* :287-308 */
    double p[3];
    chain_mu(cur, p);
    double w = wf.on ? weight_eval(&wf, sum_us(cur)) : 1.0;
    record(A, cur->r, p, cur->U, P->umbrella, w);                 /* :310-311 */
    record_extra(extra, cur, P->umbrella, w);
    if (P->stepout > 0 && step % P->stepout == 0) emit_rows(tr, step, A, cur->r, p, cur->U);
  }
  *nacc_total_out = nacc_total;
  steps_out[0] = phistep; steps_out[1] = thstep;
}

int eap_run_cluster(const eap_params *P0, uint64_t chain_id, eap_result *out, eap_trace *tr) {
  if (check_params(P0)) return -1;
  if (P0->burn_nsched < 0 || P0->burn_nsched > 8) return -1;
  uint32_t rng[5];
  seed_chain(P0, chain_id, rng);
  chain_t cur, trial;
  if (chain_alloc(&cur, P0->n) || chain_alloc(&trial, P0->n)) return -2;
  if (tr) tr->rows_written = 0;
  eap_params P = *P0;
  averagers_t A;
  double extra[2], steps[2] = {P0->phi_step, P0->theta_step};
  int64_t nacc_total = 0, t = 0, nan_rejects = 0;
  chain_random(&P, rng, &cur);                                    /* mcmc(nsteps, pargs): EAPChain(pargs), :167-170 */
  /* burn-in ladder, :365-386: every rung is a fresh mcmc() call (fresh acceptor, averagers, step sizes) */
  for (int s = 0; s < P0->burn_nsched; ++s) {
    P.kT = P0->kT * P0->burn_sched[s];
    if (tr) tr->rows_written = 0;                                 /* each call rewrites the CSV files */
    cluster_stage(&P, P0->burn_in, rng, &cur, &trial, &A, extra, &nacc_total, steps, tr, &t, &nan_rejects);
  }
  P.kT = P0->kT;
  if (tr) tr->rows_written = 0;
  cluster_stage(&P, P0->num_steps, rng, &cur, &trial, &A, extra, &nacc_total, steps, tr, &t, &nan_rejects);

  memcpy(out->sum, A.sum, sizeof A.sum);
  out->norm = A.norm;
  out->extra_sum[0] = extra[0]; out->extra_sum[1] = extra[1];
  out->nacc_total = nacc_total;
  out->nan_rejects = nan_rejects;
  out->nsteps_total = P0->num_steps;
  out->phi_step = steps[0]; out->theta_step = steps[1];
  memcpy(out->r, cur.r, sizeof cur.r);
  chain_mu(&cur, out->p);
  out->U = cur.U;
  memcpy(out->rng, rng, sizeof out->rng);
  if (tr && tr->final_phi) memcpy(tr->final_phi, cur.phi, sizeof(double) * (size_t)P0->n);
  if (tr && tr->final_theta) memcpy(tr->final_theta, cur.th, sizeof(double) * (size_t)P0->n);
  chain_free(&cur); chain_free(&trial);
  return 0;
}

/* ------------------------------------------------------------------ incremental run */

typedef struct fast_t {
  int64_t n;
  double *phi, *th, *sth; /* n   */
  double *nh, *mus, *xs;  /* 3n  */
  double *cs;             /* 3n running sum of n-hat (cumsum of eap_chain.jl:50) */
  double r[3], p[3];
  double usum;            /* sum of u_i                         */
  double upair;           /* interaction / Ising part           */
  double U;               /* usum + upair - F.r                 */
  double *block;
} fast_t;

static int fast_alloc(fast_t *c, int64_t n) {
  c->n = n;
  c->block = (double *)malloc(sizeof(double) * (size_t)(15 * n));
  if (!c->block) return -1;
  double *q = c->block;
  c->phi = q; q += n; c->th = q; q += n; c->sth = q; q += n;
  c->nh = q; q += 3 * n; c->mus = q; q += 3 * n;
  c->xs = q; q += 3 * n; c->cs = q; /* xs and cs adjacent: one backup copy covers both */
  return 0;
}

static void fast_positions(const eap_params *P, fast_t *c, int64_t from) {
  /* x_i = b (sum_{k<=i} n_k - n_i/2); the running sum restarts from the stored prefix at `from`,
   * which gives bit-for-bit what a full sequential cumsum would */
  double sx = 0, sy = 0, sz = 0;
  if (from > 0) {
    int64_t q = from - 1;
    sx = c->cs[3 * q]; sy = c->cs[3 * q + 1]; sz = c->cs[3 * q + 2];
  }
  for (int64_t i = from; i < c->n; ++i) {
    sx += c->nh[3 * i]; sy += c->nh[3 * i + 1]; sz += c->nh[3 * i + 2];
    c->cs[3 * i] = sx; c->cs[3 * i + 1] = sy; c->cs[3 * i + 2] = sz;
    c->xs[3 * i]     = P->b * (sx - 0.5 * c->nh[3 * i]);
    c->xs[3 * i + 1] = P->b * (sy - 0.5 * c->nh[3 * i + 1]);
    c->xs[3 * i + 2] = P->b * (sz - 0.5 * c->nh[3 * i + 2]);
  }
}

static void fast_derive(const eap_params *P, fast_t *c, double *Omega) {
  double prod = 1.0;
  c->r[0] = c->r[1] = c->r[2] = 0; c->p[0] = c->p[1] = c->p[2] = 0; c->usum = 0;
  for (int64_t i = 0; i < c->n; ++i) {
    double cp = cos(c->phi[i]), sp = sin(c->phi[i]), ct = cos(c->th[i]), st = sin(c->th[i]);
    c->sth[i] = st; prod *= st;
    c->nh[3 * i] = cp * st; c->nh[3 * i + 1] = sp * st; c->nh[3 * i + 2] = ct;
    eap_dipole(P, cp, sp, ct, st, c->mus + 3 * i);
    for (int k = 0; k < 3; ++k) { c->r[k] += P->b * c->nh[3 * i + k]; c->p[k] += c->mus[3 * i + k]; }
    c->usum += -0.5 * P->E0 * c->mus[3 * i + 2];
  }
  fast_positions(P, c, 0);
  c->upair = 0.0;
  if (P->energy_type == EAP_INTERACTING) c->upair = eap_pair_energy(c->n, c->xs, c->mus, 0);
  else if (P->energy_type == EAP_ISING)  c->upair = eap_pair_energy(c->n, c->xs, c->mus, 1);
  c->U = c->usum + c->upair - (c->r[0] * P->Fx + c->r[2] * P->Fz);
  if (Omega) *Omega = log(prod);
}

static void fast_random(const eap_params *P, uint32_t rng[5], fast_t *c, double *Omega) {
  for (int64_t i = 0; i < c->n; ++i) c->phi[i] = (2.0 * M_PI) * draw_u(rng);
  for (int64_t i = 0; i < c->n; ++i) c->th[i] = M_PI * draw_u(rng);
  fast_derive(P, c, Omega);
}

/* energy of the (at most two) nearest-neighbour bonds touching idx, with monomer idx replaced
 * by (nh, mu); bond vector x_i - x_{i+1} = -b/2 (n_i + n_{i+1}) */
static double ising_bonds(const eap_params *P, const fast_t *c, int64_t idx, const double nh[3],
                          const double mu[3]) {
  double e = 0.0;
  for (int side = -1; side <= 1; side += 2) {
    int64_t j = idx + side;
    if (j < 0 || j >= c->n) continue;
    const double *nj = c->nh + 3 * j, *mj = c->mus + 3 * j;
    double xi[3], xj[3] = {0, 0, 0};
    /* place the lower-index monomer first: r = x_lo - x_hi */
    for (int k = 0; k < 3; ++k) xi[k] = -P->b / 2 * (nh[k] + nj[k]);
    e += (side > 0) ? pair_term(xi, xj, mu, mj) : pair_term(xi, xj, mj, mu);
  }
  return e;
}

int eap_run_fast(const eap_params *P, uint64_t chain_id, eap_result *out, eap_trace *tr) {
  if (check_params(P)) return -1;
  if (P->energy_type == EAP_CUTOFF) return -1;
  uint32_t rng[5];
  seed_chain(P, chain_id, rng);
  fast_t c, fresh;
  if (fast_alloc(&c, P->n) || fast_alloc(&fresh, P->n)) return -2;
  double *xs_backup = (double *)malloc(sizeof(double) * (size_t)(6 * P->n)); /* xs then cs */
  if (!xs_backup) return -2;
  if (tr) tr->rows_written = 0;

  double phistep = P->phi_step, thstep = P->theta_step;
  double Omega0;
  fast_random(P, rng, &c, &Omega0);
  weight_t wf = weight_make(P, Omega0);
  double lag = 0.0; /* (acceptor's cached log pi) - (log pi of the current chain); see re-init */

  averagers_t A;
  memset(&A, 0, sizeof A);
  int64_t nacc = 0, nacc_total = 0, natt = 0, t = 0, nan_rejects = 0;

  for (int64_t init = 1; init <= P->num_inits; ++init) {
    for (int64_t step = 1; step <= P->num_steps; ++step, ++t) {
      const uint32_t w_idx = draw_w(rng), w_phi = draw_w(rng);
      int64_t idx = idx_of(w_idx, P->n);
      double dphi = phistep * sym_of(w_phi);
      double flip = 0.0;
      if (P->do_flips && (draw_w(rng) >> 31)) flip = M_PI - 2 * c.th[idx];
      const uint32_t w_th = draw_w(rng);
      double dth = flip + thstep * sym_of(w_th);
      double eps = eap_eps(P->uniform_bits, draw_w(rng), w_idx, w_phi, w_th);

      double phi1 = c.phi[idx] + dphi;
      double th1 = fmin(M_PI, fmax(0.0, c.th[idx] + dth));
      double cp = cos(phi1), sp = sin(phi1), ct = cos(th1), st = sin(th1);
      double nh1[3] = {cp * st, sp * st, ct}, mu1[3];
      eap_dipole(P, cp, sp, ct, st, mu1);
      const double *nh0 = c.nh + 3 * idx, *mu0 = c.mus + 3 * idx;
      double du = -0.5 * P->E0 * (mu1[2] - mu0[2]);
      double dr[3] = {P->b * (nh1[0] - nh0[0]), P->b * (nh1[1] - nh0[1]), P->b * (nh1[2] - nh0[2])};
      double dpair = 0.0, upair1 = c.upair;
      double nh_old[3] = {nh0[0], nh0[1], nh0[2]}, mu_old[3] = {mu0[0], mu0[1], mu0[2]};
      if (P->energy_type == EAP_ISING) {
        dpair = ising_bonds(P, &c, idx, nh1, mu1) - ising_bonds(P, &c, idx, nh_old, mu_old);
        upair1 = c.upair + dpair;
      } else if (P->energy_type == EAP_INTERACTING) {
        memcpy(xs_backup, c.xs, sizeof(double) * (size_t)(6 * P->n));
        for (int k = 0; k < 3; ++k) { c.nh[3 * idx + k] = nh1[k]; c.mus[3 * idx + k] = mu1[k]; }
        fast_positions(P, &c, idx);
        upair1 = eap_pair_energy(c.n, c.xs, c.mus, 0);
        dpair = upair1 - c.upair;
      }
      double dU = du + dpair - (P->Fx * dr[0] + P->Fz * dr[2]);
      double dw = wf.on ? du * wf.scale : 0.0;
      double delta = -dU / P->kT + log(st / c.sth[idx]) + dw - lag;
      int ok = (delta >= 0.0) || (eps < exp(delta));
      if (!isfinite(dU)) ++nan_rejects;
      if (ok) {
        c.phi[idx] = phi1; c.th[idx] = th1; c.sth[idx] = st;
        for (int k = 0; k < 3; ++k) {
          c.r[k] += dr[k];
          c.p[k] += mu1[k] - mu_old[k];
          c.nh[3 * idx + k] = nh1[k]; c.mus[3 * idx + k] = mu1[k];
        }
        c.usum += du; c.upair = upair1; c.U += dU;
        lag = 0.0;
        ++nacc; ++nacc_total;
      } else if (P->energy_type == EAP_INTERACTING) { /* undo the in-place trial */
        for (int k = 0; k < 3; ++k) { c.nh[3 * idx + k] = nh_old[k]; c.mus[3 * idx + k] = mu_old[k]; }
        memcpy(c.xs, xs_backup, sizeof(double) * (size_t)(6 * P->n));
      }
      ++natt;
      if (tr && tr->accepted) tr->accepted[t] = (uint8_t)ok;
      adapt(P, step, &phistep, &thstep, &nacc, &natt);
      record(&A, c.r, c.p, c.U, P->umbrella, weight_eval(&wf, c.usum));
      if (P->stepout > 0 && step % P->stepout == 0) emit_rows(tr, step, &A, c.r, c.p, c.U);
    }
    /* re-init: the reference's acceptor keeps the log pi it cached at the last acceptance, so after
     * adopting a new chain its comparisons are offset by `lag` until the next accepted move. */
    if (init == P->num_inits) break;
    double Om_new, Om_old, pa = 1.0, pb = 1.0;
    fast_random(P, rng, &fresh, &Om_new);
    for (int64_t i = 0; i < P->n; ++i) { pa *= c.sth[i]; pb *= fresh.sth[i]; }
    Om_old = log(pa);
    int adopt = P->force_init;
    if (!adopt) {
      double eps = draw_u(rng);
      adopt = eps <= (exp(-(fresh.U - c.U) / P->kT) * pb / pa);
    }
    if (adopt) {
      double lp_old = -c.U / P->kT + Om_old + weight_eval(&wf, c.usum);
      double lp_new = -fresh.U / P->kT + Om_new + weight_eval(&wf, fresh.usum);
      lag = (lp_old + lag) - lp_new;
      fast_t tmp = c; c = fresh; fresh = tmp;
    }
  }

  memcpy(out->sum, A.sum, sizeof A.sum);
  out->norm = A.norm;
  out->extra_sum[0] = out->extra_sum[1] = 0.0;
  out->nacc_total = nacc_total;
  out->nan_rejects = nan_rejects;
  out->nsteps_total = P->num_inits * P->num_steps;
  out->phi_step = phistep; out->theta_step = thstep;
  memcpy(out->r, c.r, sizeof c.r); memcpy(out->p, c.p, sizeof c.p);
  out->U = c.U;
  memcpy(out->rng, rng, sizeof out->rng);  /* the four state words */
  if (tr && tr->final_phi) memcpy(tr->final_phi, c.phi, sizeof(double) * (size_t)P->n);
  if (tr && tr->final_theta) memcpy(tr->final_theta, c.th, sizeof(double) * (size_t)P->n);
  free(c.block); free(fresh.block); free(xs_backup);
  return 0;
}

/* ------------------------------------------------------------------ helpers */

double eap_chain_energy(const eap_params *P, const double *phi, const double *theta,
                        double r_out[3], double p_out[3]) {
  chain_t c;
  if (chain_alloc(&c, P->n)) return NAN;
  memcpy(c.phi, phi, sizeof(double) * (size_t)P->n);
  memcpy(c.th, theta, sizeof(double) * (size_t)P->n);
  chain_derive(P, &c);
  if (r_out) memcpy(r_out, c.r, sizeof c.r);
  if (p_out) chain_mu(&c, p_out);
  double U = c.U;
  chain_free(&c);
  return U;
}

int64_t eap_find_eps23_zero(const eap_params *P, uint64_t chain_id, int64_t nsteps, int64_t *hits, int64_t max_hits) {
  uint32_t rng[5];
  seed_chain(P, chain_id, rng);
  for (int64_t i = 0; i < 2 * P->n; ++i) (void)draw_w(rng);       /* EAPChain(pargs): n phi draws, n theta draws */
  int64_t found = 0;
  for (int64_t s = 0; s < nsteps; ++s) {
    (void)draw_w(rng); (void)draw_w(rng); (void)draw_w(rng);      /* idx, dphi, dtheta */
    if ((draw_w(rng) >> 9) == 0u) {
      if (found < max_hits) hits[found] = s;
      ++found;
    }
  }
  return found;
}

typedef struct {
  const eap_params *P; uint64_t id0; int64_t nchains; int mode; eap_result *out;
  int64_t next; int err; pthread_mutex_t mu;
} farm_t;

static void *farm_worker(void *arg) {
  farm_t *F = (farm_t *)arg;
  for (;;) {
    pthread_mutex_lock(&F->mu);
    int64_t k = F->next++;
    pthread_mutex_unlock(&F->mu);
    if (k >= F->nchains) break;
    int rc = F->mode == 2 ? eap_run_cluster(F->P, F->id0 + (uint64_t)k, F->out + k, NULL)
           : F->mode      ? eap_run_fast(F->P, F->id0 + (uint64_t)k, F->out + k, NULL)
                          : eap_run_faithful(F->P, F->id0 + (uint64_t)k, F->out + k, NULL);
    if (rc) F->err = rc;
  }
  return NULL;
}

int eap_run_many(const eap_params *P, uint64_t id0, int64_t nchains, int nthreads, int mode,
                 eap_result *out) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  farm_t F = {P, id0, nchains, mode, out, 0, 0, PTHREAD_MUTEX_INITIALIZER};
  pthread_t th[256];
  for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, farm_worker, &F);
  for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  return F.err;
}
