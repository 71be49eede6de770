// pstat_api.hip -- the C ABI of include/pstat.h on top of the kernels in pstat_kernels.hip.
// Host-side only: owns device memory, validates options the way the reference's constructors do
// (inc/eap_chain.jl:81-105 error() branches), sequences launches on one HIP stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../../include/pstat.h"
#include "pstat_cluster_common.h"
#include "pstat_device.h"

using namespace pstat;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return fail(PSTAT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));          \
  } while (0)

constexpr uint64_t CKPT_MAGIC = 0x5053544154434b34ull;  // "PSTATCK4" (v4: the header identifies the handle it was taken from)

struct Buffer {
  void *ptr = nullptr;
  size_t bytes = 0;
};

}  // namespace

struct pstat_handle {
  pstat_params base{};              // case 0's parameters (shared, non-physics fields)
  std::vector<CaseConst> cases;     // host copy
  std::vector<double> kT0;          // kT each case was created with (pstat_scale_kT)
  CaseConst *d_cases = nullptr;
  int ncases = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  LaunchCfg cfg{};
  SweepArgs args{};
  DevState S{};
  std::vector<Buffer> bufs;         // every device allocation, in checkpoint order
  int *d_queue = nullptr;           // sweep job queue: counter, error flag, per-block progress
  int slots = 0;                    // resident sweep workgroups on the whole device
  double *d_partial = nullptr;      // reduction scratch
  double *d_red = nullptr;          // PSTAT_NRED doubles
  int64_t steps_recorded = 0;       // steps every chain has recorded so far (all inits)
  int64_t step_in_init = 0;         // the reference's loop variable `step` (mcmc_eap_chain.jl:276)
  size_t elem = 4;                  // sizeof(R)
  int failed_job = 0;               // sticky: 1 + the job of a persistent launch that timed out (0 = none)
};

namespace {

// energies whose every step needs all n(n-1)/2 pairs: one chain per wavefront
bool all_pairs(int energy_type) { return energy_type == PSTAT_INTERACTING || energy_type == PSTAT_CUTOFF; }
// handles whose steps run one chain per wavefront (no chain blocks, no job queue)
bool chain_per_wave(const pstat_handle *h) { return all_pairs(h->base.energy_type) || h->cfg.chain_wave != 0; }

int alloc(pstat_handle *h, void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) return fail(PSTAT_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  h->bufs.push_back({*p, bytes});
  return PSTAT_OK;
}

int validate(const pstat_params *c, int ncases) {
  if (ncases < 1) return fail(PSTAT_ERR_INVALID_ARG, "ncases must be >= 1");
  const pstat_params &b = c[0];
  if (b.n < 1) return fail(PSTAT_ERR_INVALID_ARG, "num-monomers must be >= 1");
  if (b.n > 0x7fffffff / 64) return fail(PSTAT_ERR_INVALID_ARG, "num-monomers too large");
  if (b.num_chains < 1) return fail(PSTAT_ERR_INVALID_ARG, "num-chains must be >= 1");
  if (b.chain_type != PSTAT_DIELECTRIC && b.chain_type != PSTAT_POLAR)
    return fail(PSTAT_ERR_INVALID_ARG, "chain-type is not understood.");       // eap_chain.jl:86
  if (b.energy_type < PSTAT_NONINTERACTING || b.energy_type > PSTAT_CUTOFF)
    return fail(PSTAT_ERR_INVALID_ARG, "energy-type is not understood.");      // eap_chain.jl:104
  if (b.energy_type == PSTAT_CUTOFF && b.move_set != PSTAT_MOVES_CLUSTER)
    return fail(PSTAT_ERR_INVALID_ARG, "energy-type 'cutoff' belongs to the clustering main (mcmc_eap_chain.jl has "
                "no --cutoff-radius)");
  if (all_pairs(b.energy_type) && b.n > 512)
    return fail(PSTAT_ERR_UNSUPPORTED, "the all-pairs energies run one chain per 64-lane wavefront with up to 8 "
                "monomers per lane: num-monomers must be <= 512, got %lld", (long long)b.n);
  if (b.precision != PSTAT_F32 && b.precision != PSTAT_F64 && b.precision != PSTAT_Q16)
    return fail(PSTAT_ERR_INVALID_ARG, "precision must be PSTAT_F32, PSTAT_F64 or PSTAT_Q16");
  if (b.precision == PSTAT_Q16 && all_pairs(b.energy_type))
    return fail(PSTAT_ERR_UNSUPPORTED, "the lattice state (PSTAT_Q16) is not implemented for energy-type 'interacting'");
  if (b.rng != PSTAT_RNG_MWC64X && b.rng != PSTAT_RNG_XOSHIRO128PP)
    return fail(PSTAT_ERR_INVALID_ARG, "rng must be PSTAT_RNG_MWC64X or PSTAT_RNG_XOSHIRO128PP");
  if (b.uniform_bits != 0 && b.uniform_bits != 23 && b.uniform_bits != 53)
    return fail(PSTAT_ERR_INVALID_ARG, "uniform_bits must be 0 (the precision's default), 23 or 53");
  if (b.uniform_bits == 53 && b.precision != PSTAT_F64)
    return fail(PSTAT_ERR_INVALID_ARG, "uniform_bits = 53 needs PSTAT_F64: the f32 / q16 arithmetic compares eps in a 24-bit mantissa");
  if (b.rng == PSTAT_RNG_MWC64X)
    for (int i = 0; i < ncases; ++i)
      if (c[i].chain_id0 > PSTAT_MWC64X_MAX_CHAINS || (uint64_t)b.num_chains > PSTAT_MWC64X_MAX_CHAINS - c[i].chain_id0)
        return fail(PSTAT_ERR_INVALID_ARG, "MWC64X streams are disjoint only for global chain ids < 2^22 (chain k starts "
                    "k * 2^40 outputs down one sequence of period ~2^63): chain_id0 + num_chains = %llu + %lld exceeds "
                    "that (case %d); use PSTAT_RNG_XOSHIRO128PP for larger ids",
                    (unsigned long long)c[i].chain_id0, (long long)b.num_chains, i);
  if (b.move_set != PSTAT_MOVES_SINGLE && b.move_set != PSTAT_MOVES_CLUSTER)
    return fail(PSTAT_ERR_INVALID_ARG, "move_set must be PSTAT_MOVES_SINGLE or PSTAT_MOVES_CLUSTER");
  if (b.move_set == PSTAT_MOVES_CLUSTER) {
    if (b.do_flips) return fail(PSTAT_ERR_INVALID_ARG, "mcmc_clustering_eap_chain.jl has no --do-flips");
    if (b.n < 2) return fail(PSTAT_ERR_INVALID_ARG, "cluster moves need num-monomers >= 2 (the mean bond angle)");
  }
  if (b.use_x0 != 0 && b.use_x0 != 1) return fail(PSTAT_ERR_INVALID_ARG, "use_x0 must be 0 or 1");
  if (b.use_x0 && (!std::isfinite(b.x0_phi) || !std::isfinite(b.x0_theta) || !std::isfinite(b.dx0_phi) ||
                   !std::isfinite(b.dx0_theta)))
    return fail(PSTAT_ERR_INVALID_ARG, "non-finite x0 / dx0");
  if (!(b.phi_step > 0) || !(b.theta_step > 0))
    return fail(PSTAT_ERR_INVALID_ARG, "phi-step and theta-step must be > 0");
  if (!(b.adj_scale > 0)) return fail(PSTAT_ERR_INVALID_ARG, "step-adjust-scale must be > 0");
  for (int i = 0; i < ncases; ++i) {
    const pstat_params &p = c[i];
    if (!(p.kT > 0)) return fail(PSTAT_ERR_INVALID_ARG, "kT must be > 0 (case %d)", i);
    if (!std::isfinite(p.E0) || !std::isfinite(p.K1) || !std::isfinite(p.K2) || !std::isfinite(p.mu) ||
        !std::isfinite(p.Fz) || !std::isfinite(p.Fx) || !std::isfinite(p.b) || !std::isfinite(p.bend_mod) ||
        !std::isfinite(p.bend_angle))
      return fail(PSTAT_ERR_INVALID_ARG, "non-finite physics parameter (case %d)", i);
    if (b.energy_type == PSTAT_CUTOFF && !(p.cutoff_radius > 0.0))
      return fail(PSTAT_ERR_INVALID_ARG, "cutoff-radius must be > 0 (case %d)", i);
    if (!(p.cluster_prob >= 0.0 && p.cluster_prob <= 1.0) && b.move_set == PSTAT_MOVES_CLUSTER)
      return fail(PSTAT_ERR_INVALID_ARG, "cluster-prob must be in [0, 1] (case %d)", i);
    if (b.move_set == PSTAT_MOVES_SINGLE && p.bend_mod != 0.0)
      return fail(PSTAT_ERR_INVALID_ARG, "bend-mod belongs to the clustering main (move_set = PSTAT_MOVES_CLUSTER)");
    if (p.n != b.n || p.num_chains != b.num_chains || p.chain_type != b.chain_type ||
        p.energy_type != b.energy_type || p.do_flips != b.do_flips || p.umbrella != b.umbrella ||
        p.precision != b.precision || p.device != b.device || p.rng != b.rng || p.phi_step != b.phi_step ||
        p.theta_step != b.theta_step || p.adj_lb != b.adj_lb || p.adj_ub != b.adj_ub ||
        p.adj_scale != b.adj_scale || p.steps_per_adjust != b.steps_per_adjust || p.move_set != b.move_set ||
        p.use_x0 != b.use_x0 || p.x0_phi != b.x0_phi || p.x0_theta != b.x0_theta || p.dx0_phi != b.dx0_phi ||
        p.dx0_theta != b.dx0_theta || p.uniform_bits != b.uniform_bits)
      return fail(PSTAT_ERR_INVALID_ARG, "case %d differs from case 0 in a non-physics field", i);
  }
  return PSTAT_OK;
}

// attributes of the chain-per-lane kernel that runs this handle's steps
hipError_t kernel_info(const LaunchCfg &cfg, const SweepArgs &a, int *lds, int *bpc, const char **name) {
  return cfg.move_set == PSTAT_MOVES_CLUSTER ? cluster_kernel_info(cfg, a, lds, bpc, name)
                                             : sweep_kernel_info(cfg, a, lds, bpc, name);
}

int set_device(pstat_handle *h) {
  HIP_TRY(hipSetDevice(h->device));
  return PSTAT_OK;
}

int report_failed_job(pstat_handle *h) {
  int q[10] = {0};
  (void)hipMemcpy(q, h->d_queue, sizeof q, hipMemcpyDeviceToHost);
  return fail(PSTAT_ERR_HIP, "a persistent launch did not complete: job %d waited too long for its predecessor "
              "(queue head %d, done[0..7] = %d %d %d %d %d %d %d %d); the handle's averages are not those of the "
              "steps asked for", h->failed_job - 1, q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9]);
}

// Waits for the handle's stream and then looks at the error word of the persistent kernels' job queue
// (run_job_queue, pstat_device.h).  The word is never cleared by a launch and the failure is sticky on
// the handle: every entry point that hands results to the caller goes through here.
int sync_checked(pstat_handle *h) {
  if (h->failed_job) return report_failed_job(h);
  int flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, h->d_queue, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (flag) {
    h->failed_job = flag;
    return report_failed_job(h);
  }
  return PSTAT_OK;
}

int reduce_to_host(pstat_handle *h, int icase, double red[PSTAT_NRED]) {
  int rc = pstat_reduce_device(h, icase, h->d_red);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(red, h->d_red, sizeof(double) * PSTAT_NRED, hipMemcpyDeviceToHost, h->stream));
  return sync_checked(h);
}

}  // namespace

extern "C" {

int pstat_abi_version(void) { return PSTAT_ABI_VERSION; }

const char *pstat_strerror(int status) {
  switch (status) {
    case PSTAT_OK: return "ok";
    case PSTAT_ERR_INVALID_ARG: return "invalid argument";
    case PSTAT_ERR_NO_DEVICE: return "no HIP device";
    case PSTAT_ERR_HIP: return "HIP runtime error";
    case PSTAT_ERR_UNSUPPORTED: return "option not supported on the device path";
    case PSTAT_ERR_NOMEM: return "out of device memory";
    case PSTAT_ERR_BAD_CHECKPOINT: return "checkpoint does not match this handle";
    case PSTAT_ERR_TOO_SMALL: return "buffer too small";
    default: return "unknown status";
  }
}

const char *pstat_last_error(void) { return g_err; }

int pstat_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void pstat_default_params(pstat_params *p) {
  // mcmc_eap_chain.jl:19-153
  std::memset(p, 0, sizeof *p);
  p->E0 = 0.0; p->K1 = 1.0; p->K2 = 0.0; p->mu = 1e-2; p->kT = 1.0;
  p->Fz = 0.0; p->Fx = 0.0; p->b = 1.0;
  p->phi_step = 3 * M_PI / 8; p->theta_step = 3 * M_PI / 16;
  p->adj_lb = 0.15; p->adj_ub = 0.55; p->adj_scale = 1.1;
  p->steps_per_adjust = 2500;
  p->n = 100;
  p->num_chains = 1;
  p->seed = 0; p->chain_id0 = 0;
  p->chain_type = PSTAT_DIELECTRIC; p->energy_type = PSTAT_NONINTERACTING;
  p->do_flips = 0; p->umbrella = 0;
  p->precision = PSTAT_F64; p->device = 0;   // the reference's Float64 (inc/types.jl); PSTAT_F32 is the opt-in fast path
  p->rng = PSTAT_RNG_MWC64X; p->uniform_bits = 0;   // 0 = the precision's default: 53 random bits in eps for f64, 23 for f32 / q16
  // mcmc_clustering_eap_chain.jl:36-43,87-90,142-148 (only read when move_set = PSTAT_MOVES_CLUSTER / use_x0)
  p->move_set = PSTAT_MOVES_SINGLE;
  p->bend_mod = 0.0; p->bend_angle = 0.0; p->cluster_prob = 0.5;
  p->use_x0 = 0; p->x0_phi = 0.0; p->x0_theta = 0.0; p->dx0_phi = 2 * M_PI; p->dx0_theta = 1e-1;
  p->cutoff_radius = 7.5;
}

int pstat_create(const pstat_params *cases, int32_t ncases, void *stream, pstat_handle **out) {
  if (!cases || !out) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  int rc = validate(cases, ncases);
  if (rc) return rc;
  int ndev = pstat_device_count();
  if (ndev < 1) return fail(PSTAT_ERR_NO_DEVICE, "no HIP device is visible (this library has no CPU path)");
  if (cases[0].device < 0 || cases[0].device >= ndev)
    return fail(PSTAT_ERR_NO_DEVICE, "device %d out of range (have %d)", cases[0].device, ndev);

  pstat_handle *h = new (std::nothrow) pstat_handle;
  if (!h) return fail(PSTAT_ERR_NOMEM, "host allocation failed");
  try {   // std::vector growth may throw: nothing propagates through the C ABI
  h->base = cases[0];
  h->ncases = ncases;
  h->device = cases[0].device;
  h->elem = cases[0].precision == PSTAT_F64 ? 8 : (cases[0].precision == PSTAT_Q16 ? 2 : 4);
  for (int i = 0; i < ncases; ++i) {
    const pstat_params &p = cases[i];
    h->cases.push_back({p.E0, p.K1, p.K2, p.mu, p.kT, p.Fz, p.Fx, p.b, p.seed, p.chain_id0,
                        p.bend_mod, p.bend_angle, p.cluster_prob, p.cutoff_radius});
    h->kT0.push_back(p.kT);
  }
  bool any_fx = false;
  for (auto &c : h->cases) any_fx = any_fx || c.Fx != 0.0;
  h->cfg = {h->base.precision, h->base.chain_type, h->base.energy_type, h->base.do_flips ? 1 : 0,
            h->base.umbrella ? 1 : 0, any_fx ? 1 : 0, 0, h->base.rng, h->base.move_set, 0};
  h->cfg.chain_wave = cluster_chain_wave(h->cfg, h->base.n, h->base.num_chains, ncases) ? 1 : 0;
  h->cfg.state_global = (!h->cfg.chain_wave && f64_state_global(h->cfg, h->base.n, (int64_t)ncases * h->base.num_chains)) ? 1 : 0;

  // one chain per wavefront: the all-pairs energies, and the clustering main's small f64 ensembles (pstat_cluster_cw.hip)
  const bool inter = chain_per_wave(h);
  int lanes = (inter || h->cfg.state_global) ? 64 : choose_lanes(h->base.precision, h->base.n, h->base.energy_type);
  if (lanes == 0) {
    delete h;
    return fail(PSTAT_ERR_UNSUPPORTED, "num-monomers = %lld does not fit the 160 KiB LDS of a CU",
                (long long)cases[0].n);
  }
  SweepArgs &A = h->args;
  A.n = h->base.n;
  A.chains_per_case = h->base.num_chains;
  A.blocks_per_case = (h->base.num_chains + lanes - 1) / lanes;
  A.nblocks = A.blocks_per_case * ncases;
  A.nsteps = 0; A.step0 = 0;
  A.steps_per_adjust = h->base.steps_per_adjust;
  A.adj_lb = h->base.adj_lb; A.adj_ub = h->base.adj_ub; A.adj_scale = h->base.adj_scale;
  A.lanes = lanes;
  A.adaptive = (h->base.adj_scale != 1.0 && h->base.steps_per_adjust > 0) ? 1 : 0;  // mcmc_eap_chain.jl:302
  A.ncases = ncases; A.seg_len = 0; A.nseg = 1; A.max_spins = 1 << 22;
  A.lds_rows = 0; A.packed = 0; A.pad_ = 0;
  A.wide_eps = (h->base.precision == PSTAT_F64 && h->base.uniform_bits != 23) ? 1 : 0;
  const bool cluster_gm = h->cfg.state_global && h->cfg.move_set == PSTAT_MOVES_CLUSTER;   // pstat_cluster_gm.hip
  if (h->cfg.state_global && !cluster_gm) {   // a quarter of a CU's LDS per wave: four resident waves, 64 lanes x 16 B per row
    int rows = 160 * 1024 / 4 / (64 * 16) - 1;   // one row of the quarter is the trash row of run_segment
    const char *e = getenv("PSTAT_F64_LDS_ROWS");
    if (e && atoi(e) >= 0 && atoi(e) <= rows) rows = atoi(e);
    A.lds_rows = (int32_t)(h->base.n < rows ? h->base.n : rows);
  }

#define CREATE_TRY(expr)            \
  do {                              \
    int _rc = (expr);               \
    if (_rc) { pstat_destroy(h); return _rc; } \
  } while (0)
#define CREATE_HIP(expr)                                                                   \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      pstat_destroy(h);                                                                    \
      return fail(PSTAT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));           \
    }                                                                                      \
  } while (0)

  CREATE_HIP(hipSetDevice(h->device));
  if (!inter) {
    // ---- launch shape of the chain-per-lane kernels: active lanes per workgroup (one wave) and what a workgroup holds.
    // `shape(packed)` prices the best lane count of one block layout with the makespan model of its kernel family;
    // packed blocks (run_job_queue<true>: a block holds `lanes` consecutive global chains, whichever cases they belong
    // to) are taken when they shorten the launch by more than 5 % -- an ensemble of 2 730 cases x 16 chains is 683 full
    // waves instead of 2 730 quarter-filled ones: measured 2.3 x (non-interacting) and 2.5 x (Ising) on the f64 sweep.
    // Otherwise blocks stay inside a case and its scalars in SGPRs.
    // The clustering main packs into waves no larger than a case's own chains would fill (16 lanes for cases of up to 16
    // chains) unless the unpacked launch is at least four rounds of the resident slots deep.  Its step time is not
    // uniform: a wave runs at the pace of its longest cluster, and across a phase grid that is 4.5 us per step for a
    // disordered chain against 25-33 us for an aligned one (n = 100, tools/phase_latency.py).  A sweep that mixes them is
    // paced by the sequential step time of its cold cases, not by throughput, and there a 64-lane wave of four cases is a
    // little slower than four 16-lane waves (run/K1_E0-kT-phase.jl's grid, 2 730 x 16 chains, 3e5 steps: 11.5 s unpacked,
    // 13.0 s packed four to a wave), while filling the idle lanes of a 16-lane wave with further cases is a gain throughout
    // (2 730 x 5 chains: 1 012 -> 865 ms per 2e4 steps on the whole grid, 747 -> 353 on its cold part; 5 760 x 1 chain:
    // 1.6-2.4 x); only when workgroups queue several deep does the throughput of full waves win (all-cold 2 730 x 16:
    // 676 -> 366 ms per 1e4 steps).  profiles/r04/experiments/time_packed*.txt, twin.txt.
    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, h->device));
    const int64_t per_case = h->base.num_chains, total = per_case * ncases;
    const char *le = getenv("PSTAT_LANES");
    struct Shape { int lanes; int64_t nblocks; double cost; };
    auto shape = [&](const bool packed, const bool deep) -> Shape {
      LaunchCfg cfg = h->cfg;
      cfg.packed = packed ? 1 : 0;
      auto wgs_of = [&](const int cand) -> int64_t {
        return packed ? (total + cand - 1) / cand : (int64_t)ncases * ((per_case + cand - 1) / cand);
      };
      Shape best{lanes, wgs_of(lanes), 1e300};
      if (h->cfg.state_global && !cluster_gm) {   // f64 sweep with its cells in memory: 64 lanes on every SIMD
        int lds0 = 0, bpc = 0;
        if (kernel_info(cfg, h->args, &lds0, &bpc, nullptr) != hipSuccess || bpc < 1) bpc = 4;
        double cost = (double)best.nblocks / ((double)bpc * prop.multiProcessorCount);
        best.cost = cost < 1.0 ? 1.0 : cost;
        return best;
      }
      if (cluster_gm) {
        // Chains in device memory: nothing limits a wave to fewer than 64 lanes, but an ensemble of fewer waves than the
        // chip has SIMDs (a phase scan: 546 grid points x 64 chains) runs faster as more, emptier waves -- they fill the
        // idle SIMDs, and a wave's step lasts as long as its LONGEST cluster, which grows like the logarithm of its lanes.
        int lds0 = 0, bpc = 0;
        if (kernel_info(cfg, h->args, &lds0, &bpc, nullptr) != hipSuccess || bpc < 1) bpc = 1;
        const double slots = (double)bpc * prop.multiProcessorCount;
        // (packed, and the unpacked launch not many rounds deep: no wave larger than a case's chains rounded up to 16 / 32 / 64
        // -- the clustering main's packing rule, above)
        int cmax = 64;
        if (packed && !deep) cmax = per_case <= 16 ? 16 : (per_case <= 32 ? 32 : 64);
        for (int cand = cmax; cand >= 16; cand >>= 1) {
          if (le && atoi(le) >= 1 && atoi(le) <= 64 && cand != atoi(le)) continue;
          double cost = (double)wgs_of(cand) / slots;
          if (cost < 1.0) cost = 1.0;
          cost *= 1.0 + 0.1 * std::log2(cand / 16.0);
          if (cost < best.cost) best = Shape{cand, wgs_of(cand), cost};
        }
        return best;
      }
      // State in LDS: a CU holds at most 160 KiB / (bytes per chain) chains; pick the lane count that minimises the
      // makespan max(1, workgroups / resident slots) of one launch -- e.g. f32, n = 100: 51 lanes x 4 workgroups per CU
      // (204 chains, all four SIMDs) instead of 64 x 3 (192 chains, three SIMDs).
      bool lds_starved = false;     // full waves: fewer than one per SIMD fit a CU's LDS
      {
        int lds0 = 0, bpc0 = 0;
        lds_starved = kernel_info(cfg, h->args, &lds0, &bpc0, nullptr) == hipSuccess && bpc0 < 4;
      }
      for (int cand = lanes; cand >= 8; --cand) {
        if (le && atoi(le) >= 1 && atoi(le) <= lanes && cand != atoi(le)) continue;
        SweepArgs probe = h->args;
        probe.lanes = cand;
        int lds = 0, bpc = 0;
        if (kernel_info(cfg, probe, &lds, &bpc, nullptr) != hipSuccess || bpc < 1) continue;
        const double slots = (double)bpc * prop.multiProcessorCount;
        double cost = (double)wgs_of(cand) / slots;
        if (cost < 1.0) cost = 1.0;
        cost *= 1.0 + 1e-4 * (64 - cand);   // ties: prefer fuller waves
        if (h->cfg.move_set == PSTAT_MOVES_CLUSTER && lds_starved) {
          // (only when LDS seats fewer than four FULL waves per CU.)  The cluster step runs a wave for as long as its
          // LONGEST cluster, so a wave of fewer lanes finishes its steps sooner (~ log of the lane count), and more,
          // emptier waves also put the idle SIMDs to work.  Measured, n = 100, f64 (LDS seats 102 chains per CU): 4 x 25 lanes
          // 4.08e9 proposals/s against 2 x 51 lanes 3.61e9 (non-interacting; Ising 5.70e9 / 5.57e9).  Sharing a SIMD
          // between waves costs more than it gains here (f32: 4 x 51 lanes 1.77e10, 8 x 25 lanes 1.52e10).
          const double waves_per_simd = bpc / 4.0;
          cost *= 1.0 + 0.1 * std::log2(cand / 16.0);
          if (waves_per_simd > 1.0) cost *= 1.0 + 0.25 * (waves_per_simd - 1.0);
        }
        if (cost < best.cost) best = Shape{cand, wgs_of(cand), cost};
      }
      return best;
    };
    Shape pick = shape(false, false);
    if (ncases > 1 && supports_packed_cases(h->cfg)) {
      const char *pe = getenv("PSTAT_PACK");     // 0 | 1: tests and experiments
      const Shape pk = shape(true, pick.cost >= 4.0 || (pe && atoi(pe) != 0));
      if (pe ? atoi(pe) != 0 : pk.cost < 0.95 * pick.cost) {
        pick = pk;
        h->cfg.packed = 1;
        A.packed = 1;
      }
    }
    lanes = pick.lanes;
    A.lanes = lanes;
    A.blocks_per_case = (per_case + lanes - 1) / lanes;
    A.nblocks = pick.nblocks;
    if (cluster_gm && (uint64_t)lanes * (uint64_t)h->base.n * (h->base.precision == PSTAT_F64 ? PSTAT_CLUSTER_GM_CELL : 20u) >= 0x80000000ull) {
      pstat_destroy(h);
      return fail(PSTAT_ERR_UNSUPPORTED, "num-monomers = %lld: a wave's working buffer must stay below 2 GiB",
                  (long long)cases[0].n);
    }
  }
  // (checked on the FINAL lane count: the job queue counts blocks and (block, segment) jobs in 32 bits)
  if (A.nblocks > 0x3fffffffLL) {
    pstat_destroy(h);
    return fail(PSTAT_ERR_INVALID_ARG, "too many chains for one handle: %lld chain blocks", (long long)A.nblocks);
  }
  if (stream) {
    h->stream = (hipStream_t)stream;
  } else {
    CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  DevState &S = h->S;
  const int64_t C = (int64_t)ncases * h->base.num_chains;
  S.C = C;
  const size_t n = (size_t)h->base.n, Cz = (size_t)C;
  // checkpoint order = allocation order
  CREATE_TRY(alloc(h, &S.ang, 2 * n * Cz * h->elem));
  CREATE_TRY(alloc(h, (void **)&S.rng, 4 * Cz * sizeof(uint32_t)));
  CREATE_TRY(alloc(h, (void **)&S.stepsz, 2 * Cz * sizeof(double)));
  CREATE_TRY(alloc(h, (void **)&S.win, 2 * Cz * sizeof(int64_t)));
  CREATE_TRY(alloc(h, (void **)&S.nacc_total, Cz * sizeof(int64_t)));
  CREATE_TRY(alloc(h, (void **)&S.obs, NOBS_STATE * Cz * sizeof(double)));
  CREATE_TRY(alloc(h, (void **)&S.sums, NSUMS * Cz * sizeof(double)));
  CREATE_TRY(alloc(h, (void **)&S.wnorm, Cz * sizeof(double)));
  CREATE_TRY(alloc(h, (void **)&S.lag, Cz * sizeof(double)));
  CREATE_TRY(alloc(h, (void **)&S.uref, Cz * sizeof(double)));
  CREATE_TRY(alloc(h, (void **)&S.nanrej, Cz * sizeof(int64_t)));
  const size_t nstate = h->bufs.size();
  CREATE_TRY(alloc(h, &S.ang_tmp, 2 * n * Cz * h->elem));
  if (cluster_gm)            // working copy of the chains, [chain block][lane][n] cells of 40 (f64) / 20 (f32) bytes (pstat_cluster_gm.hip)
    CREATE_TRY(alloc(h, &S.work, cluster_gm_work_bytes(h->cfg, A)));
  else if (h->cfg.state_global)   // working copy of the cells, [chain block][n][64] double2 (run_segment, ST = 2)
    CREATE_TRY(alloc(h, &S.work, (size_t)A.nblocks * n * 64 * 16));
  CREATE_TRY(alloc(h, (void **)&h->d_cases, sizeof(CaseConst) * (size_t)ncases));
  CREATE_TRY(alloc(h, (void **)&h->d_queue, sizeof(int) * sweep_queue_ints(h->args)));
  CREATE_TRY(alloc(h, (void **)&h->d_partial, sizeof(double) * reduce_scratch_doubles()));
  CREATE_TRY(alloc(h, (void **)&h->d_red, sizeof(double) * PSTAT_NRED));
  (void)nstate;
  CREATE_HIP(hipMemsetAsync(h->d_queue, 0, sizeof(int) * sweep_queue_ints(h->args), h->stream));
  CREATE_HIP(hipMemcpyAsync(h->d_cases, h->cases.data(), sizeof(CaseConst) * (size_t)ncases,
                            hipMemcpyHostToDevice, h->stream));
  const InitOpts io{h->base.use_x0, h->base.x0_phi, h->base.x0_theta, h->base.dx0_phi, h->base.dx0_theta, nullptr};
  CREATE_HIP(launch_init(h->cfg, h->args, h->S, h->d_cases, h->base.phi_step, h->base.theta_step, io, h->stream));
  if (all_pairs(h->base.energy_type)) {  // a zero-step launch derives r, p, U (with the pair energy) from the fresh angles
    h->args.nsteps = 0; h->args.step0 = 0;
    if (h->cfg.move_set == PSTAT_MOVES_CLUSTER)
      CREATE_HIP(launch_cluster_wave(h->cfg, h->args, h->S, h->d_cases, h->stream));
    else
      CREATE_HIP(launch_interacting(h->cfg, h->args, h->S, h->d_cases, 0, h->stream));
  }
  CREATE_HIP(hipStreamSynchronize(h->stream));  // h->cases must outlive the copy; also surfaces faults here
  if (!inter) {
    int lds = 0, bpc = 0;
    hipDeviceProp_t prop;
    CREATE_HIP(kernel_info(h->cfg, h->args, &lds, &bpc, nullptr));
    CREATE_HIP(hipGetDeviceProperties(&prop, h->device));
    h->slots = (bpc > 0 ? bpc : 1) * prop.multiProcessorCount;
  }
#undef CREATE_TRY
#undef CREATE_HIP
  } catch (const std::bad_alloc &) {
    pstat_destroy(h);
    return fail(PSTAT_ERR_NOMEM, "host allocation failed");
  }
  *out = h;
  return PSTAT_OK;
}

void pstat_destroy(pstat_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto &b : h->bufs) (void)hipFree(b.ptr);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// Splits a launch of `nsteps` steps into time segments so that blocks*segments fills the resident
// workgroup slots evenly (see sweep_kernel).  Returns segments per block.
static int choose_segments(int64_t blocks, int64_t slots, int64_t nsteps, int64_t min_seg) {
  if (blocks <= slots || nsteps < 2 * min_seg) return 1;
  int best = 1;
  double best_eff = 0;
  for (int s = 1; s <= 12; ++s) {
    if (nsteps / s < min_seg) break;              // keep fill/spill amortised
    const double jobs = (double)blocks * s;
    const double eff = jobs / (std::ceil(jobs / slots) * slots);   // busy fraction of the slots
    if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
  }
  return best;
}

int pstat_advance(pstat_handle *h, int64_t nsteps) {
  if (!h) return fail(PSTAT_ERR_INVALID_ARG, "null handle");
  if (nsteps < 0) return fail(PSTAT_ERR_INVALID_ARG, "nsteps must be >= 0");
  if (nsteps == 0) return PSTAT_OK;
  int rc = set_device(h);
  if (rc) return rc;
  if (h->failed_job) return report_failed_job(h);
  const int64_t max_launch = 1ll << 30;  // per-launch step counters are 32-bit
  if (chain_per_wave(h)) {
    while (nsteps > 0) {
      const int64_t len = nsteps < max_launch ? nsteps : max_launch;
      h->args.nsteps = len;
      h->args.step0 = h->step_in_init;
      if (h->cfg.chain_wave)
        HIP_TRY(launch_cluster_cw(h->cfg, h->args, h->S, h->d_cases, h->stream));
      else if (h->cfg.move_set == PSTAT_MOVES_CLUSTER)
        HIP_TRY(launch_cluster_wave(h->cfg, h->args, h->S, h->d_cases, h->stream));
      else
        HIP_TRY(launch_interacting(h->cfg, h->args, h->S, h->d_cases, 0, h->stream));
      h->step_in_init += len;
      h->steps_recorded += len;
      nsteps -= len;
    }
    return PSTAT_OK;
  }
  const int64_t blocks = h->args.nblocks;
  const char *env = getenv("PSTAT_SEGMENTS");
  const char *ms = getenv("PSTAT_MAX_SPINS");
  while (nsteps > 0) {
    const int64_t len = nsteps < max_launch ? nsteps : max_launch;
    // (a fill + spill of the f64 cluster kernel's working buffer costs about ten of its steps, the LDS kernels' a few
    // hundred of theirs)
    const bool cluster_gm = h->cfg.state_global && h->cfg.move_set == PSTAT_MOVES_CLUSTER;
    int nseg = env ? atoi(env) : choose_segments(blocks, h->slots, len, cluster_gm ? 400 : 2000);
    if (nseg < 1) nseg = 1;
    // f32/q16 running totals are re-derived from the angles at every segment start: bound the stretch
    // over which their rounding errors can random-walk
    if (h->base.precision != PSTAT_F64 && !env) {
      const int64_t need = (len + 32767) / 32768;
      if (need > nseg) nseg = (int)(need < 0x3fffffffLL ? need : 0x3fffffffLL);
    }
    if (nseg > len) nseg = (int)len;
    if (blocks * nseg > 0x3fffffffLL) nseg = 1;
    h->args.nsteps = len;
    h->args.step0 = h->step_in_init;
    h->args.nseg = nseg;
    h->args.seg_len = (len + nseg - 1) / nseg;
    // A job waits at most for one segment of its predecessor.  Bound the wait by a generous multiple of the
    // longest plausible segment (a spin sleeps ~2 us; a step of the cluster kernel on a long, aligned chain
    // can take tens of us), so that a long launch is never mistaken for a lost predecessor.
    {
      const int64_t want = (1ll << 22) + h->args.seg_len * 256;
      h->args.max_spins = (int32_t)(want < 0x7fffffffLL ? want : 0x7fffffffLL);
      if (ms && atoi(ms) > 0) h->args.max_spins = atoi(ms);
    }
    // Segments of one block run one after the other, so at most `blocks` jobs are runnable at any time:
    // more workgroups than that would only sit in the predecessor wait -- and, worse, leave the working
    // ones unevenly spread over the SIMDs.
    const unsigned grid = (unsigned)(blocks < h->slots ? blocks : h->slots);
    if (h->cfg.move_set == PSTAT_MOVES_CLUSTER)
      HIP_TRY(launch_cluster(h->cfg, h->args, h->S, h->d_cases, h->d_queue, grid, h->stream));
    else
      HIP_TRY(launch_sweep(h->cfg, h->args, h->S, h->d_cases, h->d_queue, grid, h->stream));
    h->step_in_init += len;
    h->steps_recorded += len;
    nsteps -= len;
  }
  return PSTAT_OK;
}

int pstat_sync(pstat_handle *h) {
  if (!h) return fail(PSTAT_ERR_INVALID_ARG, "null handle");
  int rc = set_device(h);
  if (rc) return rc;
  return sync_checked(h);
}

int pstat_reinit(pstat_handle *h, int32_t force_init) {
  if (!h) return fail(PSTAT_ERR_INVALID_ARG, "null handle");
  if (h->cfg.move_set == PSTAT_MOVES_CLUSTER)
    return fail(PSTAT_ERR_UNSUPPORTED, "mcmc_clustering_eap_chain.jl has no --num-inits: nothing to re-initialise");
  int rc = set_device(h);
  if (rc) return rc;
  if (h->base.energy_type == PSTAT_INTERACTING) {   // done inside the one-chain-per-wave kernel
    h->args.nsteps = 0; h->args.step0 = 0;
    HIP_TRY(launch_interacting(h->cfg, h->args, h->S, h->d_cases, force_init ? 2 : 1, h->stream));
    h->step_in_init = 0;
    h->cfg.lag = 1;
    return PSTAT_OK;
  }
  HIP_TRY(launch_reinit(h->cfg, h->args, h->S, h->d_cases, force_init, h->stream));
  h->step_in_init = 0;
  h->cfg.lag = 1;  // from now on the sweep tracks the acceptor's stale-cache offset
  return PSTAT_OK;
}

int pstat_reset_averages(pstat_handle *h) {
  if (!h) return fail(PSTAT_ERR_INVALID_ARG, "null handle");
  int rc = set_device(h);
  if (rc) return rc;
  const size_t C = (size_t)h->S.C;
  HIP_TRY(hipMemsetAsync(h->S.sums, 0, NSUMS * C * sizeof(double), h->stream));
  HIP_TRY(hipMemsetAsync(h->S.wnorm, 0, C * sizeof(double), h->stream));
  HIP_TRY(hipMemsetAsync(h->S.nacc_total, 0, C * sizeof(int64_t), h->stream));
  HIP_TRY(hipMemsetAsync(h->S.nanrej, 0, C * sizeof(int64_t), h->stream));
  h->steps_recorded = 0;
  return PSTAT_OK;
}

int pstat_reset_sampler(pstat_handle *h) {
  if (!h) return fail(PSTAT_ERR_INVALID_ARG, "null handle");
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(launch_reset_sampler(h->S, h->base.phi_step, h->base.theta_step, h->stream));
  h->step_in_init = 0;
  return PSTAT_OK;
}

int pstat_set_kT(pstat_handle *h, int32_t icase, double kT) {
  if (!h) return fail(PSTAT_ERR_INVALID_ARG, "null handle");
  if (!(kT > 0) || !std::isfinite(kT)) return fail(PSTAT_ERR_INVALID_ARG, "kT must be > 0");
  if (icase >= h->ncases) return fail(PSTAT_ERR_INVALID_ARG, "case %d out of range", icase);
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));   // the host copy of the constants is re-uploaded
  for (int i = 0; i < h->ncases; ++i)
    if (icase < 0 || icase == i) h->cases[(size_t)i].kT = kT;
  HIP_TRY(hipMemcpy(h->d_cases, h->cases.data(), sizeof(CaseConst) * (size_t)h->ncases, hipMemcpyHostToDevice));
  return PSTAT_OK;
}

int pstat_restart_from_x0(pstat_handle *h, const double *x0, int64_t len, double dx0_phi, double dx0_theta) {
  if (!h || !x0) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  if (len != 2 && len != 2 * h->base.n)
    return fail(PSTAT_ERR_INVALID_ARG, "Invalid input for 'x0': length %lld is neither 2 nor 2 * num-monomers",
                (long long)len);                                                      // inc/eap_chain.jl:77
  for (int64_t i = 0; i < len; ++i)
    if (!std::isfinite(x0[i])) return fail(PSTAT_ERR_INVALID_ARG, "non-finite x0[%lld]", (long long)i);
  if (!std::isfinite(dx0_phi) || !std::isfinite(dx0_theta)) return fail(PSTAT_ERR_INVALID_ARG, "non-finite dx0");
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  double *d_x0 = nullptr;
  InitOpts io{1, x0[0], x0[1], dx0_phi, dx0_theta, nullptr};
  if (len != 2) {
    HIP_TRY(hipMalloc((void **)&d_x0, sizeof(double) * (size_t)len));
    hipError_t e = hipMemcpy(d_x0, x0, sizeof(double) * (size_t)len, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d_x0); return fail(PSTAT_ERR_HIP, "copy of x0 failed: %s", hipGetErrorString(e)); }
    io.use_x0 = 2;
    io.x0_vec = d_x0;
  }
  hipError_t e = launch_init(h->cfg, h->args, h->S, h->d_cases, h->base.phi_step, h->base.theta_step, io, h->stream);
  if (e == hipSuccess && all_pairs(h->base.energy_type)) {
    h->args.nsteps = 0; h->args.step0 = 0;
    e = h->cfg.move_set == PSTAT_MOVES_CLUSTER ? launch_cluster_wave(h->cfg, h->args, h->S, h->d_cases, h->stream)
                                               : launch_interacting(h->cfg, h->args, h->S, h->d_cases, 0, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (d_x0) (void)hipFree(d_x0);
  if (e != hipSuccess) return fail(PSTAT_ERR_HIP, "re-initialisation from x0 failed: %s", hipGetErrorString(e));
  h->steps_recorded = 0;
  h->step_in_init = 0;
  h->cfg.lag = 0;
  return PSTAT_OK;
}

int pstat_scale_kT(pstat_handle *h, double mult) {
  if (!h) return fail(PSTAT_ERR_INVALID_ARG, "null handle");
  if (!(mult > 0) || !std::isfinite(mult)) return fail(PSTAT_ERR_INVALID_ARG, "kT multiplier must be > 0");
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int i = 0; i < h->ncases; ++i) h->cases[(size_t)i].kT = h->kT0[(size_t)i] * mult;
  HIP_TRY(hipMemcpy(h->d_cases, h->cases.data(), sizeof(CaseConst) * (size_t)h->ncases, hipMemcpyHostToDevice));
  return PSTAT_OK;
}

int pstat_reduce_device(pstat_handle *h, int32_t icase, double *dev_out) {
  if (!h || !dev_out) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  if (icase >= h->ncases) return fail(PSTAT_ERR_INVALID_ARG, "case %d out of range", icase);
  int rc = set_device(h);
  if (rc) return rc;
  const int64_t per = h->base.num_chains;
  const int64_t c0 = icase < 0 ? 0 : icase * per;
  const int64_t c1 = icase < 0 ? h->S.C : c0 + per;
  HIP_TRY(launch_reduce(h->S, c0, c1, h->steps_recorded, h->cfg.umbrella, h->d_cases, h->base.num_chains, h->base.n,
                        h->d_partial, dev_out, h->stream));
  return PSTAT_OK;
}

int pstat_reduce_host(pstat_handle *h, int32_t icase, double red_out[PSTAT_NRED]) {
  if (!h || !red_out) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  return reduce_to_host(h, icase, red_out);
}

int pstat_summary_from_reduction(const double red[PSTAT_NRED], int64_t steps_per_chain,
                                 pstat_summary *out) {
  if (!red || !out) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  std::memset(out, 0, sizeof *out);
  const double C = red[0];
  out->num_chains = (int64_t)std::llround(C);
  out->steps_per_chain = steps_per_chain;
  out->attempted_updates = C * (double)steps_per_chain;
  if (C < 1) return PSTAT_OK;
  for (int q = 0; q < PSTAT_NQ; ++q) {
    const double mean = red[1 + q] / C;
    double se = 0.0;
    if (C > 1) {
      double var = (red[1 + PSTAT_NQ + q] / C - mean * mean) * C / (C - 1);  // unbiased across-chain variance
      se = var > 0 ? std::sqrt(var / C) : 0.0;
    }
    if (q < PSTAT_NOBS) { out->avg[q] = mean; out->stderr_[q] = se; }
    else if (q == PSTAT_NOBS) { out->acceptance_ratio = mean; out->ar_stderr = se; }
    else { out->extra_avg[q - PSTAT_NOBS - 1] = mean; out->extra_stderr[q - PSTAT_NOBS - 1] = se; }
  }
  out->nan_rejects = (int64_t)std::llround(red[1 + 2 * PSTAT_NQ]);
  out->chains_collapsed = (int64_t)std::llround(red[2 + 2 * PSTAT_NQ]);
  return PSTAT_OK;
}

int pstat_summary_get(pstat_handle *h, int32_t icase, pstat_summary *out) {
  if (!h || !out) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  double red[PSTAT_NRED];
  int rc = reduce_to_host(h, icase, red);
  if (rc) return rc;
  return pstat_summary_from_reduction(red, h->steps_recorded, out);
}

int pstat_rolling(pstat_handle *h, int32_t icase, double avg_out[PSTAT_NOBS],
                  double stderr_out[PSTAT_NOBS]) {
  pstat_summary s;
  int rc = pstat_summary_get(h, icase, &s);
  if (rc) return rc;
  if (avg_out) std::memcpy(avg_out, s.avg, sizeof s.avg);
  if (stderr_out) std::memcpy(stderr_out, s.stderr_, sizeof s.stderr_);
  return PSTAT_OK;
}

int pstat_microstate(pstat_handle *h, int64_t chain, double out[7]) {
  if (!h || !out) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  if (chain < 0 || chain >= h->S.C) return fail(PSTAT_ERR_INVALID_ARG, "chain out of range");
  int rc = set_device(h);
  if (rc) return rc;
  // obs is [NOBS_STATE][C]: a strided gather of 7 doubles
  HIP_TRY(hipMemcpy2DAsync(out, sizeof(double), h->S.obs + chain, sizeof(double) * (size_t)h->S.C,
                           sizeof(double), 7, hipMemcpyDeviceToHost, h->stream));
  return sync_checked(h);
}

int pstat_chain_state(pstat_handle *h, int64_t chain, double *angles, double sums[PSTAT_NOBS],
                      int64_t counters[4], double steps[3], uint32_t rng[4]) {
  if (!h) return fail(PSTAT_ERR_INVALID_ARG, "null handle");
  if (chain < 0 || chain >= h->S.C) return fail(PSTAT_ERR_INVALID_ARG, "chain out of range");
  int rc = set_device(h);
  if (rc) return rc;
  const size_t C = (size_t)h->S.C, n = (size_t)h->base.n;
  rc = sync_checked(h);
  if (rc) return rc;
  if (angles) {
    std::unique_ptr<unsigned char[]> tmp(new (std::nothrow) unsigned char[2 * n * h->elem]);
    if (!tmp) return fail(PSTAT_ERR_NOMEM, "host allocation failed");
    HIP_TRY(hipMemcpy2D(tmp.get(), h->elem, (char *)h->S.ang + (size_t)chain * h->elem, C * h->elem,
                        h->elem, 2 * n, hipMemcpyDeviceToHost));
    // storage formats: pstat_math.h (radians | turns | lattice index); the ABI speaks radians
    for (size_t i = 0; i < 2 * n; ++i) {
      const bool is_theta = i < n;
      if (h->elem == 8) angles[i] = ((double *)tmp.get())[i];
      else if (h->elem == 4) angles[i] = (double)((float *)tmp.get())[i] * 6.28318530717958647692;
      else angles[i] = (is_theta ? 3.14159265358979323846 : 6.28318530717958647692) *
                       ((double)((uint16_t *)tmp.get())[i] + 0.5) / 65536.0;
    }
  }
  if (sums) {
    double s[NSUMS];
    HIP_TRY(hipMemcpy2D(s, sizeof(double), h->S.sums + chain, C * sizeof(double), sizeof(double), NSUMS,
                        hipMemcpyDeviceToHost));
    sums[PSTAT_R1] = s[S_R1]; sums[PSTAT_R2] = s[S_R2]; sums[PSTAT_R3] = s[S_R3];
    sums[PSTAT_R1SQ] = s[S_R1SQ]; sums[PSTAT_R2SQ] = s[S_R2SQ]; sums[PSTAT_R3SQ] = s[S_R3SQ];
    sums[PSTAT_RSQ] = s[S_R1SQ] + s[S_R2SQ] + s[S_R3SQ];
    sums[PSTAT_P1] = s[S_P1]; sums[PSTAT_P2] = s[S_P2]; sums[PSTAT_P3] = s[S_P3];
    sums[PSTAT_P1SQ] = s[S_P1SQ]; sums[PSTAT_P2SQ] = s[S_P2SQ]; sums[PSTAT_P3SQ] = s[S_P3SQ];
    sums[PSTAT_PSQ] = s[S_P1SQ] + s[S_P2SQ] + s[S_P3SQ];
    sums[PSTAT_U] = s[S_U]; sums[PSTAT_USQ] = s[S_USQ];
  }
  if (counters) {
    int64_t w[2];
    int64_t tot;
    HIP_TRY(hipMemcpy2D(w, sizeof(int64_t), h->S.win + chain, C * sizeof(int64_t), sizeof(int64_t), 2,
                        hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&tot, h->S.nacc_total + chain, sizeof tot, hipMemcpyDeviceToHost));
    counters[0] = tot; counters[1] = h->steps_recorded; counters[2] = w[0]; counters[3] = w[1];
  }
  if (steps) {
    HIP_TRY(hipMemcpy2D(steps, sizeof(double), h->S.stepsz + chain, C * sizeof(double), sizeof(double), 2,
                        hipMemcpyDeviceToHost));
    if (h->cfg.umbrella) HIP_TRY(hipMemcpy(&steps[2], h->S.wnorm + chain, sizeof(double), hipMemcpyDeviceToHost));
    else steps[2] = (double)h->steps_recorded;
  }
  if (rng)
    HIP_TRY(hipMemcpy2D(rng, sizeof(uint32_t), h->S.rng + chain, C * sizeof(uint32_t), sizeof(uint32_t), 4,
                        hipMemcpyDeviceToHost));
  return PSTAT_OK;
}

int pstat_chain_means(pstat_handle *h, int32_t icase, double *out) {
  if (!h || !out) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  if (icase >= h->ncases) return fail(PSTAT_ERR_INVALID_ARG, "case %d out of range", icase);
  int rc = set_device(h);
  if (rc) return rc;
  rc = sync_checked(h);
  if (rc) return rc;
  const size_t C = (size_t)h->S.C, per = (size_t)h->base.num_chains;
  const size_t c0 = icase < 0 ? 0 : (size_t)icase * per, m = icase < 0 ? C : per;
  try {
    std::vector<double> sums((size_t)NSUMS * m), wn(m);
    std::vector<int64_t> nacc(m);
    HIP_TRY(hipMemcpy2D(sums.data(), m * sizeof(double), h->S.sums + c0, C * sizeof(double), m * sizeof(double), NSUMS,
                        hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nacc.data(), h->S.nacc_total + c0, m * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (h->cfg.umbrella) HIP_TRY(hipMemcpy(wn.data(), h->S.wnorm + c0, m * sizeof(double), hipMemcpyDeviceToHost));
    const double steps = (double)h->steps_recorded;
    // same arithmetic as reduce_stage1 (pstat_kernels.hip)
    static const int src[PSTAT_NOBS] = {S_R1, S_R2, S_R3, S_R1SQ, S_R2SQ, S_R3SQ, -1, S_P1, S_P2, S_P3,
                                        S_P1SQ, S_P2SQ, S_P3SQ, -2, S_U, S_USQ};
    for (size_t k = 0; k < m; ++k) {
      const double norm = h->cfg.umbrella ? wn[k] : steps;
      const double inv = norm != 0.0 ? 1.0 / norm : 0.0;
      for (int q = 0; q < PSTAT_NOBS; ++q) {
        double v;
        if (src[q] == -1) v = sums[S_R1SQ * m + k] + sums[S_R2SQ * m + k] + sums[S_R3SQ * m + k];
        else if (src[q] == -2) v = sums[S_P1SQ * m + k] + sums[S_P2SQ * m + k] + sums[S_P3SQ * m + k];
        else v = sums[(size_t)src[q] * m + k];
        out[(size_t)q * m + k] = v * inv;
      }
      out[16 * m + k] = steps > 0 ? (double)nacc[k] / steps : 0.0;
      out[17 * m + k] = sums[S_C2 * m + k] * inv;
      out[18 * m + k] = sums[S_PSI * m + k] * inv;
    }
  } catch (const std::bad_alloc &) {
    return fail(PSTAT_ERR_NOMEM, "host allocation failed");
  }
  return PSTAT_OK;
}

int pstat_chain_extras(pstat_handle *h, int64_t chain, double extra_sums[2], double extra_now[2]) {
  if (!h) return fail(PSTAT_ERR_INVALID_ARG, "null handle");
  if (chain < 0 || chain >= h->S.C) return fail(PSTAT_ERR_INVALID_ARG, "chain out of range");
  int rc = set_device(h);
  if (rc) return rc;
  const size_t C = (size_t)h->S.C;
  rc = sync_checked(h);
  if (rc) return rc;
  if (extra_sums)
    HIP_TRY(hipMemcpy2D(extra_sums, sizeof(double), h->S.sums + (size_t)S_C2 * C + chain, C * sizeof(double),
                        sizeof(double), 2, hipMemcpyDeviceToHost));
  if (extra_now) {
    HIP_TRY(hipMemcpy2D(extra_now, sizeof(double), h->S.obs + (size_t)OBS_C2 * C + chain, C * sizeof(double),
                        sizeof(double), 2, hipMemcpyDeviceToHost));
    extra_now[1] /= (double)(h->base.n > 1 ? h->base.n - 1 : 1);
  }
  return PSTAT_OK;
}

// checkpoint image: header, the cases' current kT, then the eleven state buffers in allocation order
struct CkptHeader {
  uint64_t magic;
  int32_t abi_version, header_bytes;
  int64_t n, C, ncases, steps_recorded, step_in_init;
  int32_t precision, chain_type, energy_type, rng, move_set, umbrella, do_flips, uniform_bits, lag, reserved;
  uint64_t seed, chain_id0;     // case 0's
  uint64_t params_fnv;          // fingerprint of everything else a chain's continuation depends on (params_fingerprint)
};
static const int kStateBuffers = 11;

// FNV-1a over the options that are not spelled out in the header: every case's physics scalars except its current kT
// (which the image carries and restore re-instates), its seed and first chain id, the kT it was created with, and the
// proposal / adaptation options.  A checkpoint continues exactly the run it was taken from, on a handle created with
// the same options -- anything else would silently be a different Markov chain wearing this one's averages.
static uint64_t params_fingerprint(const pstat_handle *h) {
  uint64_t f = 0xcbf29ce484222325ull;
  auto mix = [&](const void *p, size_t nbytes) {
    const unsigned char *q = (const unsigned char *)p;
    for (size_t i = 0; i < nbytes; ++i) { f ^= q[i]; f *= 0x100000001b3ull; }
  };
  for (size_t i = 0; i < h->cases.size(); ++i) {
    const CaseConst &c = h->cases[i];
    const double v[12] = {c.E0, c.K1, c.K2, c.mu, c.Fz, c.Fx, c.b, c.kappa, c.psi0, c.cluster_prob, c.cutoff_radius, h->kT0[i]};
    mix(v, sizeof v);
    mix(&c.seed, sizeof c.seed);
    mix(&c.chain_id0, sizeof c.chain_id0);
  }
  const double a[5] = {h->base.phi_step, h->base.theta_step, h->base.adj_lb, h->base.adj_ub, h->base.adj_scale};
  mix(a, sizeof a);
  mix(&h->base.steps_per_adjust, sizeof h->base.steps_per_adjust);
  mix(&h->base.num_chains, sizeof h->base.num_chains);
  return f;
}

static size_t checkpoint_bytes(const pstat_handle *h) {
  size_t need = sizeof(CkptHeader) + sizeof(double) * (size_t)h->ncases;
  for (int i = 0; i < kStateBuffers; ++i) need += h->bufs[i].bytes;
  return need;
}

int pstat_checkpoint(pstat_handle *h, void *buf, size_t *bytes) {
  if (!h || !bytes) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  const size_t need = checkpoint_bytes(h);
  if (!buf) { *bytes = need; return PSTAT_OK; }
  if (*bytes < need) { *bytes = need; return fail(PSTAT_ERR_TOO_SMALL, "checkpoint needs %zu bytes", need); }
  int rc = set_device(h);
  if (rc) return rc;
  rc = sync_checked(h);
  if (rc) return rc;
  CkptHeader hd{};
  hd.magic = CKPT_MAGIC; hd.abi_version = PSTAT_ABI_VERSION; hd.header_bytes = (int32_t)sizeof(CkptHeader);
  hd.n = h->base.n; hd.C = h->S.C; hd.ncases = h->ncases;
  hd.steps_recorded = h->steps_recorded; hd.step_in_init = h->step_in_init;
  hd.precision = h->base.precision; hd.chain_type = h->base.chain_type; hd.energy_type = h->base.energy_type;
  hd.rng = h->base.rng; hd.move_set = h->base.move_set; hd.umbrella = h->base.umbrella ? 1 : 0;
  hd.do_flips = h->base.do_flips ? 1 : 0; hd.uniform_bits = h->args.wide_eps ? 53 : 23; hd.lag = h->cfg.lag;
  hd.seed = h->cases[0].seed; hd.chain_id0 = h->cases[0].chain_id0;
  hd.params_fnv = params_fingerprint(h);
  char *q = (char *)buf;
  std::memcpy(q, &hd, sizeof hd);
  q += sizeof hd;
  for (int i = 0; i < h->ncases; ++i, q += sizeof(double)) std::memcpy(q, &h->cases[(size_t)i].kT, sizeof(double));
  for (int i = 0; i < kStateBuffers; ++i) {
    HIP_TRY(hipMemcpy(q, h->bufs[i].ptr, h->bufs[i].bytes, hipMemcpyDeviceToHost));
    q += h->bufs[i].bytes;
  }
  *bytes = need;
  return PSTAT_OK;
}

int pstat_restore(pstat_handle *h, const void *buf, size_t bytes) {
  if (!h || !buf) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  if (bytes < sizeof(CkptHeader)) return fail(PSTAT_ERR_BAD_CHECKPOINT, "checkpoint is %zu bytes: not even a header", bytes);
  CkptHeader hd;
  std::memcpy(&hd, buf, sizeof hd);
  if (hd.magic != CKPT_MAGIC)
    return fail(PSTAT_ERR_BAD_CHECKPOINT, "not a checkpoint of this library version (magic %016llx)", (unsigned long long)hd.magic);
  if (hd.abi_version != PSTAT_ABI_VERSION || hd.header_bytes != (int32_t)sizeof(CkptHeader))
    return fail(PSTAT_ERR_BAD_CHECKPOINT, "checkpoint written under ABI version %d, this library is version %d", hd.abi_version,
                PSTAT_ABI_VERSION);
#define CKPT_SAME(field, mine, what)                                                                              \
  if ((long long)(hd.field) != (long long)(mine))                                                                 \
    return fail(PSTAT_ERR_BAD_CHECKPOINT, "checkpoint does not match this handle: %s is %lld in the checkpoint, " \
                "%lld here", what, (long long)(hd.field), (long long)(mine))
  CKPT_SAME(n, h->base.n, "num-monomers");
  CKPT_SAME(C, h->S.C, "the number of chains");
  CKPT_SAME(ncases, h->ncases, "the number of cases");
  CKPT_SAME(precision, h->base.precision, "precision");
  CKPT_SAME(chain_type, h->base.chain_type, "chain-type");
  CKPT_SAME(energy_type, h->base.energy_type, "energy-type");
  CKPT_SAME(rng, h->base.rng, "the generator (0 MWC64X, 1 xoshiro128++: the state words mean different things)");
  CKPT_SAME(move_set, h->base.move_set, "move_set (0 mcmc_eap_chain.jl, 1 mcmc_clustering_eap_chain.jl)");
  CKPT_SAME(umbrella, h->base.umbrella ? 1 : 0, "umbrella-sampling");
  CKPT_SAME(do_flips, h->base.do_flips ? 1 : 0, "do-flips");
  CKPT_SAME(uniform_bits, h->args.wide_eps ? 53 : 23, "uniform_bits");
#undef CKPT_SAME
  if (hd.seed != h->cases[0].seed || hd.chain_id0 != h->cases[0].chain_id0)
    return fail(PSTAT_ERR_BAD_CHECKPOINT, "checkpoint does not match this handle: seed / chain_id0 %llu / %llu in the "
                "checkpoint, %llu / %llu here", (unsigned long long)hd.seed, (unsigned long long)hd.chain_id0,
                (unsigned long long)h->cases[0].seed, (unsigned long long)h->cases[0].chain_id0);
  if (hd.params_fnv != params_fingerprint(h))
    return fail(PSTAT_ERR_BAD_CHECKPOINT, "checkpoint does not match this handle: a physics scalar, a case's seed or chain "
                "ids, num-chains, or a proposal / adaptation option differs from the run it was taken from");
  const size_t need = checkpoint_bytes(h);
  if (bytes < need) return fail(PSTAT_ERR_BAD_CHECKPOINT, "checkpoint is truncated: %zu bytes, this handle's image has %zu", bytes, need);
  const char *q = (const char *)buf + sizeof hd;
  std::unique_ptr<double[]> kT(new (std::nothrow) double[(size_t)h->ncases]);
  if (!kT) return fail(PSTAT_ERR_NOMEM, "host allocation failed");
  for (int i = 0; i < h->ncases; ++i, q += sizeof(double)) {
    std::memcpy(&kT[(size_t)i], q, sizeof(double));
    if (!(kT[(size_t)i] > 0) || !std::isfinite(kT[(size_t)i]))
      return fail(PSTAT_ERR_BAD_CHECKPOINT, "checkpoint holds kT = %g for case %d", kT[(size_t)i], i);
  }
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  // the temperature of every case as it was when the image was taken (a rung of the burn-in ladder: pstat_scale_kT / pstat_set_kT)
  for (int i = 0; i < h->ncases; ++i) h->cases[(size_t)i].kT = kT[(size_t)i];
  HIP_TRY(hipMemcpy(h->d_cases, h->cases.data(), sizeof(CaseConst) * (size_t)h->ncases, hipMemcpyHostToDevice));
  for (int i = 0; i < kStateBuffers; ++i) {
    HIP_TRY(hipMemcpy(h->bufs[i].ptr, q, h->bufs[i].bytes, hipMemcpyHostToDevice));
    q += h->bufs[i].bytes;
  }
  h->steps_recorded = hd.steps_recorded;
  h->step_in_init = hd.step_in_init;
  h->cfg.lag = hd.lag;
  return PSTAT_OK;
}

int pstat_launch_info_get(pstat_handle *h, pstat_launch_info *out) {
  if (!h || !out) return fail(PSTAT_ERR_INVALID_ARG, "null argument");
  int rc = set_device(h);
  if (rc) return rc;
  std::memset(out, 0, sizeof *out);
  int lds = 0, bpc = 0;
  const char *name = "";
  if (chain_per_wave(h)) {
    if (h->cfg.chain_wave) HIP_TRY(cluster_cw_kernel_info(h->cfg, h->base.n, &bpc, &name));
    else if (h->cfg.move_set == PSTAT_MOVES_CLUSTER) HIP_TRY(cluster_wave_kernel_info(h->cfg, h->base.n, &bpc, &name));
    else HIP_TRY(interacting_kernel_info(h->cfg, h->base.n, &bpc, &name));
  } else HIP_TRY(kernel_info(h->cfg, h->args, &lds, &bpc, &name));
  std::snprintf(out->kernel, sizeof out->kernel, "%s", name);
  out->lds_bytes = lds;
  out->threads_per_block = 64;
  out->lanes_per_block = h->args.lanes;
  out->blocks = chain_per_wave(h) ? h->S.C : h->args.nblocks;
  out->packed_cases = h->args.packed;
  out->blocks_per_cu = bpc;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, h->device));
  out->num_cus = prop.multiProcessorCount;
  return PSTAT_OK;
}

}  // extern "C"
