// ubench.hip -- instruction-rate and trig-accuracy microbenchmarks that size the sweep kernel's
// design choices on gfx950.  Build: hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 32768;
constexpr int ACC = 8;

enum Op { FMA_3V, FMAC_3V, MUL_2V, PKFMA_3V, FMA_1V1S, PKFMA_IND, PKADD_IND, FMA32, MULLO, MULHI, SIN, EXP2, RCP, FMA64, ADD64, XORSHIFT, CVT, PKFMA, LDSB64, MAD64, ALIGNBIT, XOR32, ADDU32, XOSHIRO, MWC64X, DPP_WAVE_ROR, DPP_ROW_ROR, BPERMUTE, DPP_WAVE_ROR_IND };

template <int OP>
__global__ __launch_bounds__(64) void rate_kernel(float *out, int nrows) {
  extern __shared__ float2 lds[];
  float a[ACC]; uint32_t u[ACC]; double d[ACC];
  typedef float v2q __attribute__((ext_vector_type(2)));
  v2q pk[ACC];
  for (int i = 0; i < ACC; ++i) pk[i] = (v2q){0.001f * (threadIdx.x + i + 1), 0.002f * (threadIdx.x + i + 1)};
  for (int i = 0; i < ACC; ++i) { a[i] = 0.001f * (threadIdx.x + i + 1); u[i] = threadIdx.x * 2654435761u + i; d[i] = a[i]; }
  if (OP == LDSB64) for (int r = 0; r < nrows; ++r) lds[r * 64 + threadIdx.x] = make_float2(r, threadIdx.x);
  // operands that live in VGPRs (per-lane values the compiler cannot fold)
  float bq[ACC], cq[ACC]; v2q pb[ACC], pc[ACC];
  for (int i = 0; i < ACC; ++i) {
    bq[i] = 1.0f + 1e-6f * (threadIdx.x + i); cq[i] = 1e-3f * (threadIdx.x + 2 * i + 1);
    pb[i] = (v2q){bq[i], bq[i] + 1e-7f}; pc[i] = (v2q){cq[i], cq[i] * 0.5f};
  }
  const float sconst = 1.0f + 1e-6f * (float)nrows;   // wave-uniform: an SGPR operand
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) {
      // how many DISTINCT VGPR sources an instruction reads
      if (OP == FMA_3V) a[i] = __builtin_fmaf(a[i], bq[i], cq[i]);            // v_fma_f32 d, v, v, v
      if (OP == FMAC_3V) a[i] = __builtin_fmaf(bq[i], cq[(i + 1) % ACC], a[i]);   // v_fmac_f32 acc, v, v
      if (OP == MUL_2V) a[i] = a[i] * bq[i];                                   // v_mul_f32 d, v, v
      if (OP == FMA_1V1S) a[i] = __builtin_fmaf(a[i], sconst, 0.5f);           // v_fma_f32 d, v, s, const
      if (OP == PKFMA_3V) pk[i] = __builtin_elementwise_fma(pk[i], pb[i], pc[i]);
      if (OP == FMA32) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
      // packed ops on ACC independent register pairs: the issue cost of v_pk_fma_f32 / v_pk_add_f32 themselves
      if (OP == PKFMA_IND) pk[i] = __builtin_elementwise_fma(pk[i], (v2q){1.0001f, 1.0002f}, (v2q){0.5f, 0.25f});
      if (OP == PKADD_IND) pk[i] = pk[i] + (v2q){0.5f, 0.25f};
      if (OP == MULLO) u[i] = u[i] * 0xD2511F53u + 1u;
      if (OP == MULHI) u[i] = __umulhi(u[i], 0xCD9E8D57u) + 12345u;
      if (OP == SIN) a[i] = __builtin_amdgcn_sinf(a[i]);
      if (OP == EXP2) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if (OP == RCP) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (OP == FMA64) d[i] = d[i] * 1.0000001 + 0.5;
      if (OP == ADD64) d[i] = d[i] + 0.5;
      if (OP == XORSHIFT) u[i] = (u[i] ^ (u[i] << 9)) + __builtin_rotateleft32(u[i], 7);
      if (OP == CVT) a[i] += (float)(u[i] >> 8), u[i] += 77u;
      if (OP == PKFMA) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 x = {a[i], a[(i + 1) % ACC]};
        x = __builtin_elementwise_fma(x, (v2){1.0001f, 1.0002f}, (v2){0.5f, 0.25f});
        a[i] = x.x;
      }
      if (OP == MAD64) { unsigned long long t = (unsigned long long)u[i] * 4294883355ull + (unsigned long long)(uint32_t)a[i]; u[i] = (uint32_t)t; a[i] = __uint_as_float((uint32_t)(t >> 32)); }
      if (OP == ALIGNBIT) u[i] = __builtin_amdgcn_alignbit(u[i], u[i], 25 - i);
      if (OP == XOR32) u[i] = u[i] ^ (0x9E3779B9u + it);
      if (OP == ADDU32) u[i] = u[i] + (0x9E3779B9u ^ it);
      if (OP == DPP_WAVE_ROR) u[i] = __builtin_amdgcn_update_dpp(0, (int)u[i], 0x13C, 0xF, 0xF, false);   // dependent chain per accumulator
      if (OP == DPP_ROW_ROR) u[i] = __builtin_amdgcn_update_dpp(0, (int)u[i], 0x121, 0xF, 0xF, false);    // row_ror:1
      if (OP == BPERMUTE) u[i] = __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x + 1) & 63) << 2), (int)u[i]);
      if (OP == DPP_WAVE_ROR_IND) u[i] = u[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(threadIdx.x + it + i), 0x13C, 0xF, 0xF, false);
      if (OP == LDSB64) {
        uint32_t row = __umulhi(u[i], (uint32_t)nrows);
        float2 v = lds[row * 64 + threadIdx.x];
        u[i] = u[i] * 1664525u + 1013904223u + (uint32_t)v.x;
      }
    }
  }
  float s = 0; for (int i = 0; i < ACC; ++i) s += a[i] + (float)u[i] + (float)d[i] + pk[i].x + pk[i].y + bq[i] + cq[i] + pb[i].y + pc[i].y;
  if (s == 123.456f) out[0] = s;
}

template <int OP>
int run(const char *name, int waves_per_cu, int nrows = 0) {
  float *out; CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int grid = 256 * waves_per_cu;
  size_t lds = OP == LDSB64 ? (size_t)nrows * 64 * 8 : 0;
  if (lds) CHECK(hipFuncSetAttribute((const void *)rate_kernel<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(64), lds, 0, out, nrows);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(64), lds, 0, out, nrows);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double instr_per_wave = (double)ITERS * ACC;
  double ns_per_instr = ms * 1e6 / instr_per_wave;            // per wave-instruction, as seen by one wave
  // waves per SIMD = waves_per_cu / 4 (dispatcher spreads 1-wave groups over SIMDs)
  double per_simd = ns_per_instr / ((waves_per_cu + 3) / 4);
  printf("%-10s waves/CU=%2d  %8.3f ms   %.3f ns per wave-instr per wave  (%.2f cyc @2.4GHz; per-SIMD issue interval %.2f cyc)\n",
         name, waves_per_cu, ms, ns_per_instr, ns_per_instr * 2.4, per_simd * 2.4);
  CHECK(hipFree(out));
  return 0;
}

template <int GEN>
__global__ __launch_bounds__(64) void gen_kernel(uint32_t *out) {
  // one generator per lane, 4 outputs per iteration (what one MC step consumes)
  uint32_t s0 = threadIdx.x * 2654435761u + 1, s1 = s0 ^ 0x9E3779B9u, s2 = s0 * 3u + 7u, s3 = ~s0;
  uint32_t x = s0, c = s1 & 0x7fffffffu;
  uint32_t acc = 0;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (GEN == 0) {  // xoshiro128++
        uint32_t a = s0 + s3; uint32_t r = ((a << 7) | (a >> 25)) + s0; uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = (s3 << 11) | (s3 >> 21);
        acc += r >> 9;
      } else {         // MWC64X: one 32x32+64 multiply-add and one xor per output
        uint32_t r = x ^ c;
        unsigned long long t = (unsigned long long)x * 4294883355ull + c;
        x = (uint32_t)t; c = (uint32_t)(t >> 32);
        acc += r >> 9;
      }
    }
  }
  if (acc == 0x12345u) out[0] = acc;
}
template <int GEN>
int run_gen(const char *name, int waves_per_cu) {
  uint32_t *out; CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int grid = 256 * waves_per_cu;
  hipLaunchKernelGGL(gen_kernel<GEN>, dim3(grid), dim3(64), 0, 0, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(gen_kernel<GEN>, dim3(grid), dim3(64), 0, 0, out);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("%-14s waves/CU=%2d  %8.3f ms  -> %.1f cycles @2.4GHz per 4 outputs (one MC step) per wave\n", name, waves_per_cu, ms, ms * 1e6 * 2.4 / ITERS);
  CHECK(hipFree(out));
  return 0;
}

__global__ void trig_err_kernel(int n, double *maxerr) {
  // max abs error over x in [0, 2pi): [0]=__sinf [1]=__cosf [2]=v_sin(turns) [3]=sincosf(sin) [4]=__expf rel on [-20,5]
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = 6.283185307179586 * (i + 0.5) / n;
  float xf = (float)x;
  double xs = (double)xf;
  double e0 = fabs((double)__sinf(xf) - sin(xs));
  double e1 = fabs((double)__cosf(xf) - cos(xs));
  float turns = (float)((i + 0.5) / n);
  double e2 = fabs((double)__builtin_amdgcn_sinf(turns) - sin(6.283185307179586 * (double)turns));
  float s, c; sincosf(xf, &s, &c);
  double e3 = fabs((double)s - sin(xs));
  float y = -20.0f + 25.0f * (float)((i + 0.5) / n);
  double e4 = fabs((double)__expf(y) / exp((double)y) - 1.0);
  double e[5] = {e0, e1, e2, e3, e4};
  for (int k = 0; k < 5; ++k) {
    unsigned long long *p = (unsigned long long *)&maxerr[k];
    unsigned long long v = __double_as_longlong(e[k]);
    atomicMax(p, v);  // non-negative doubles order like their bit patterns
  }
}

int main(int argc, char **argv) {
  const bool only_ops = argc > 1 && argv[1][0] == 'o';   // "ops": just the operand-source rates
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  LDS/CU=%zu\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.maxSharedMemoryPerMultiProcessor);
  if (!only_ops) for (int w : {4, 8, 16}) {
    run<FMA32>("fma_f32", w); run<MULLO>("mul_lo_u32", w); run<MULHI>("mul_hi_u32", w);
    run<SIN>("v_sin_f32", w); run<EXP2>("v_exp_f32", w); run<RCP>("v_rcp_f32", w);
    run<FMA64>("fma_f64", w); run<ADD64>("add_f64", w); run<XORSHIFT>("xor/shl/rot", w);
    run<CVT>("cvt+add", w); run<PKFMA>("pk_fma_f32 (pair assembled per op)", w);
    run<PKFMA_IND>("pk_fma_f32 independent", w); run<PKADD_IND>("pk_add_f32 independent", w);
  }
  for (int w : {4, 8, 12, 16}) {
    run<FMA_1V1S>("fma d,v,s,c", w); run<MUL_2V>("mul d,v,v", w); run<FMA_3V>("fma d,v,v,v", w);
    run<FMAC_3V>("fmac acc,v,v", w); run<PKFMA_3V>("pk_fma d,v,v,v", w);
  }
  if (only_ops) return 0;
  for (int w : {4, 8}) { run<MAD64>("mad_u64_u32", w); run<ALIGNBIT>("alignbit", w); run<XOR32>("xor_b32", w); run<ADDU32>("add_u32", w); }
  for (int w : {4, 12, 16}) { run<DPP_WAVE_ROR>("dpp wave_ror", w); run<DPP_ROW_ROR>("dpp row_ror", w); run<BPERMUTE>("ds_bpermute", w); run<DPP_WAVE_ROR_IND>("wave_ror+add", w); }
  for (int w : {4, 8}) { run_gen<0>("xoshiro128++", w); run_gen<1>("mwc64x", w); }
  for (int w : {1, 2, 3}) run<LDSB64>("lds_b64_rand", w, 100);
  run<LDSB64>("lds_b64_rand", 8, 30);
  double *d; CHECK(hipMalloc(&d, 5 * 8)); CHECK(hipMemset(d, 0, 5 * 8));
  int n = 1 << 22;
  hipLaunchKernelGGL(trig_err_kernel, dim3(n / 256), dim3(256), 0, 0, n, d);
  double h[5]; CHECK(hipMemcpy(h, d, 5 * 8, hipMemcpyDeviceToHost));
  printf("max abs err on [0,2pi): __sinf %.3e  __cosf %.3e  v_sin(turns) %.3e  sincosf %.3e ; __expf max rel err on [-20,5] %.3e\n", h[0], h[1], h[2], h[3], h[4]);
  return 0;
}
