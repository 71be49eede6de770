// Accuracy of the f64 device math of polymer_stats_amd/csrc/pstat_math.h (sincos_f64, exp_f64, log_f64) against
// the host's long-double functions:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/mathcheck tools/mathcheck.hip && ./tools/mathcheck
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../polymer_stats_amd/csrc/pstat_math.h"

// argument of the acos check: 2^24 points of [-1, 1], the same double on device and host; the first few are special
__host__ __device__ inline double acos_arg(int i) {
  if (i == 0) return 1.0;
  if (i == 1) return -1.0;
  if (i == 2) return 0.5;
  if (i == 3) return -0.5;
  if (i == 4) return 0.0;
  return (double)(((unsigned)i * 2654435761u) >> 8) / 8388608.0 - 1.0;
}

// argument of the rsqrt check: r^2 from 1e-12 to 1e+12, log-spaced with a per-point mantissa jitter
__host__ __device__ inline double rsq_arg(int i) { return exp(-27.6 + 55.2 * ((double)(i & 0xFFFFF) + 0.5) / 1048576.0) * (1.0 + 1e-3 * (double)(i % 997)); }

__device__ int g_form_mismatch = 0;   // arguments of [0, pi] on which the bounded hot-loop form and the general one differ in any bit

__global__ void run(const double *x, double *s, double *c, double *ex, double *lg, double *sf, double *cf, double *ac, double *rq, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (x[i] >= 0.0 && x[i] <= M_PI) {
    double s1, c1, s2, c2;
    pstat::sincos_fast_f64<true>(x[i], &s1, &c1);
    pstat::sincos_fast_f64<false>(x[i], &s2, &c2);
    if (__double_as_longlong(s1) != __double_as_longlong(s2) || __double_as_longlong(c1) != __double_as_longlong(c2)) atomicAdd(&g_form_mismatch, 1);
  }
  rq[i] = pstat::rsqrt_f64(rsq_arg(i));             // 1/r of the f64 pair terms
  pstat::sincos_f64(x[i], &s[i], &c[i]);
  if (x[i] >= 0.0 && x[i] <= M_PI) pstat::sincos_fast_f64<true>(x[i], &sf[i], &cf[i]);   // the sweep's hot-loop form
  else pstat::sincos_fast_f64<false>(x[i], &sf[i], &cf[i]);
  ac[i] = pstat::acos_r(acos_arg(i));               // the clustering main's bond angles (f64 form)
  ex[i] = pstat::exp_f64(-fabs(x[i]) * 0.007);     // the acceptance test only ever needs exp of negatives
  lg[i] = pstat::log_f64(fabs(x[i]));
}

static double ulps(double got, long double want) {
  if (want == 0.0L) return got == 0.0 ? 0.0 : 1e9;
  int e;
  frexpl(want, &e);
  return (double)(fabsl((long double)got - want) / ldexpl(1.0L, e - 53));
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  std::mt19937_64 g(7);
  // (beyond 1e5 the fold-by-turns path of sincos_f64 is only ~1e-11 accurate: not part of the 1-ulp claim)
  // (huge: up to the fold bound of the hot-loop form, PSTAT_PHI_FOLD = 1e9 -- its two-word reduction runs unfolded there)
  std::uniform_real_distribution<double> th(0.0, M_PI), ph(-2000.0, 2000.0), big(-9.9e4, 9.9e4), huge(-9.0e8, 9.0e8);
  for (int i = 0; i < n; ++i)
    x[i] = i % 5 == 0 ? th(g) : (i % 5 == 1 ? ph(g) : (i % 5 == 2 ? big(g) : (i % 5 == 3 ? std::ldexp(th(g), -(i % 60)) : huge(g))));
  const double special[] = {0.0, M_PI, M_PI / 2, M_PI / 4, 3 * M_PI / 4, -M_PI, 2 * M_PI, 1e-300, 9e4, -7e4, std::nextafter(M_PI, 0.0), 1.0};
  const int nsp = (int)(sizeof special / sizeof *special);
  for (int i = 0; i < nsp; ++i) x[i] = special[i];
  double *d[9];
  for (auto &p : d) if (hipMalloc(&p, n * 8) != hipSuccess) return 2;
  if (hipMemcpy(d[0], x.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess) return 2;
  hipLaunchKernelGGL(run, dim3(n / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], n);
  std::vector<double> s(n), c(n), ex(n), lg(n), sf(n), cf(n), ac(n), rq(n);
  if (hipMemcpy(rq.data(), d[8], n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  if (hipMemcpy(ac.data(), d[7], n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  if (hipMemcpy(sf.data(), d[5], n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  if (hipMemcpy(cf.data(), d[6], n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  if (hipMemcpy(s.data(), d[1], n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  if (hipMemcpy(c.data(), d[2], n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  if (hipMemcpy(ex.data(), d[3], n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  if (hipMemcpy(lg.data(), d[4], n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  double ms = 0, mc = 0, me = 0, ml = 0, msf = 0, mcf = 0, mac = 0, mrq = 0;
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const double es = ulps(s[i], sinl((long double)x[i])), ec = ulps(c[i], cosl((long double)x[i]));
    if (fabs(x[i]) < 1.0e5) {     // (sincos_f64 folds by whole turns beyond that: outside its 1-ulp claim)
      ms = fmax(ms, es); mc = fmax(mc, ec);
      if (es > 1.0 || ec > 1.0) ++bad;
    }
    // fast form: judged in ulps of the result, except next to a zero of the function where its absolute error is what
    // matters (|r| rounding: 2^-53 |r|), i.e. in units of 2^-53 there
    const long double ws = sinl((long double)x[i]), wc = cosl((long double)x[i]);
    const double efs = fmin(ulps(sf[i], ws), (double)(fabsl((long double)sf[i] - ws) / 1.1102230246251565e-16L));
    const double efc = fmin(ulps(cf[i], wc), (double)(fabsl((long double)cf[i] - wc) / 1.1102230246251565e-16L));
    msf = fmax(msf, efs); mcf = fmax(mcf, efc);
    if (efs > 1.5 || efc > 1.5) ++bad;
    {
      const long double wa = acosl((long double)acos_arg(i));
      const double ea = wa == 0.0L ? (ac[i] == 0.0 ? 0.0 : 1e9) : ulps(ac[i], wa);
      mac = fmax(mac, ea);
      if (ea > 1.5) ++bad;
    }
    {
      const double er = ulps(rq[i], 1.0L / sqrtl((long double)rsq_arg(i)));
      mrq = fmax(mrq, er);
      if (er > 3.0) ++bad;
    }
    const double a = fabs(x[i]);
    const double arg = -a * 0.007;                  // the same double the device saw
    const double ee = ulps(ex[i], expl((long double)arg));
    if (a < 1.0e5) me = fmax(me, ee);
    if (a > 0) { const double el = ulps(lg[i], logl((long double)a)); ml = fmax(ml, el); if (el > 2.0) ++bad; }
    if (ee > 2.0 && a < 1.0e5) ++bad;     // (beyond: exp underflows in double, not in the long-double reference)
  }
  printf("over %d arguments: max error sin %.3f ulp, cos %.3f ulp, exp %.3f ulp, log %.3f ulp; %d out of bounds\n", n, ms, mc, me, ml, bad);
  printf("rsqrt_f64 (v_rsq_f64 + one third-order correction): max error %.3f ulp\n", mrq);
  printf("acos_r (f64): max error %.3f ulp; acos(1) = %g, acos(-1) = %.17g\n", mac, ac[0], ac[1]);
  printf("fast hot-loop form: max error sin %.3f, cos %.3f (ulp of the result, or units of 2^-53 next to a zero); sin(fl(pi)) = %.17g, "
         "cos(fl(pi/2)) = %.17g, sin(0) = %g\n", msf, mcf, sf[1], cf[2], sf[0]);
  int mism = -1;
  if (hipMemcpyFromSymbol(&mism, HIP_SYMBOL(g_form_mismatch), sizeof mism) != hipSuccess) return 2;
  printf("bounded hot-loop form against the general one on [0, pi]: %d arguments differ in some bit\n", mism);
  if (mism != 0) bad += 1;
  printf("sin(fl(pi)) = %.17g (glibc %.17g)  cos(fl(pi/2)) = %.17g (glibc %.17g)  sin(0) = %g  log(0) = %g  log(1) = %g\n", s[1],
         std::sin(M_PI), c[2], std::cos(M_PI / 2), s[0], lg[0], lg[nsp - 1]);
  return bad ? 1 : 0;
}
