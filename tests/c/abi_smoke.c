/* Plain-C client of include/pstat.h: what a C caller (or a cgo / JNI / ccall stub) does, without any Python in
 * between.  Built and run by tests/test_abi.py (CPU: expects PSTAT_ERR_NO_DEVICE -- the library has no CPU path) and by
 * tests/test_gpu_host.py (GPU: one small f64 ensemble, checked against the freely-jointed-chain closed form). */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "pstat.h"

int main(void) {
  if (pstat_abi_version() != PSTAT_ABI_VERSION) {
    fprintf(stderr, "ABI mismatch: library %d, header %d\n", pstat_abi_version(), PSTAT_ABI_VERSION);
    return 3;
  }
  pstat_params p;
  pstat_default_params(&p);
  p.n = 20; p.E0 = 0.0; p.Fz = 1.0; p.kT = 1.0;      /* BASELINE configs[0] */
  p.num_chains = 2048; p.seed = 7;
  pstat_handle *h = NULL;
  int rc = pstat_create(&p, 1, NULL, &h);
  if (rc != PSTAT_OK) {
    printf("create: %d (%s): %s\n", rc, pstat_strerror(rc), pstat_last_error());
    return rc == PSTAT_ERR_NO_DEVICE ? 2 : 1;
  }
  rc = pstat_advance(h, 20000);                      /* burn-in */
  if (rc == PSTAT_OK) rc = pstat_reset_averages(h);
  if (rc == PSTAT_OK) rc = pstat_advance(h, 30000);
  pstat_summary s;
  memset(&s, 0, sizeof s);
  if (rc == PSTAT_OK) rc = pstat_summary_get(h, -1, &s);
  if (rc != PSTAT_OK) {
    printf("run: %d (%s): %s\n", rc, pstat_strerror(rc), pstat_last_error());
    pstat_destroy(h);
    return 1;
  }
  const double want = 20.0 * (1.0 / tanh(1.0) - 1.0);   /* n b L(F b / kT) */
  printf("<r_z> = %.6f +- %.6f (closed form %.6f)  AR = %.4f  chains = %lld  steps = %lld  nan_rejects = %lld\n",
         s.avg[PSTAT_R3], s.stderr_[PSTAT_R3], want, s.acceptance_ratio, (long long)s.num_chains,
         (long long)s.steps_per_chain, (long long)s.nan_rejects);
  pstat_destroy(h);
  return fabs(s.avg[PSTAT_R3] - want) < 5 * s.stderr_[PSTAT_R3] + 1e-3 ? 0 : 4;
}
