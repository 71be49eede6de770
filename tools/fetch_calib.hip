// fetch_calib.hip -- what do FETCH_SIZE and the TCC_EA0_RDREQ_* counters report for the f64 sweep's access shape?
//
// MI355X_MICROARCH.md (HBM) calibrates FETCH_SIZE for WIDE COALESCED STREAMING reads only (it reads half of the bytes: a
// 128-byte request is tallied at 64) and calls every other access width uncalibrated.  The f64 sweep kernel reads
// SCATTERED 16-byte cells: lane l of a wave reads (theta, phi) of a random monomer, i.e. 16 bytes at [row r_l][lane l] of a
// [rows][64] double2 table -- every lane a different 1 KiB row.  This program issues a KNOWN number of each kind of read
// from a table far larger than L2 + Infinity Cache, so that (nearly) every access goes to the fabric:
//
//   stream   : every lane reads consecutive 16-byte cells, once each                         -> bytes = table size
//   scatter  : every lane reads 16 bytes of a uniformly random row (its own column), N times -> N * 64 lanes accesses
//
//   hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- /tmp/fetch_calib
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace ... -- /tmp/fetch_calib
// prints the expected counts; tools/summarize_calib.py puts them beside the counters (profiles/r03/fetch_calib.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void stream_kernel(const double2 *__restrict__ t, size_t cells, double *out) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = t[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) out[0] = acc;   // (keeps the loads alive)
}

// one wave per workgroup like the sweep kernel; row = a 64-bit LCG per lane, column = the lane
__global__ __launch_bounds__(64) void scatter_kernel(const double2 *__restrict__ t, uint32_t rows, int per_lane, double *out) {
  uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(blockIdx.x * 64 + threadIdx.x + 1);
  double acc = 0;
  for (int k = 0; k < per_lane; k += 4) {      // four independent loads in flight per lane
    double2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const uint32_t r = (uint32_t)(((s >> 32) * (uint64_t)rows) >> 32);
      v[j] = t[(size_t)r * 64 + threadIdx.x];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].y;
  }
  if (acc == 12345.678) out[0] = acc;
}

// the same scattered reads through a buffer resource with cache-policy bits (sc0 = 1, nt = 2, sc1 = 16): does any policy make
// L2 request less than a whole 128-byte line from the fabric?
typedef int v4i_ __attribute__((ext_vector_type(4)));
template <int AUX>
__global__ __launch_bounds__(64) void scatter_aux_kernel(const double2 *__restrict__ t, uint32_t rows, int per_lane, double *out) {
  uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(blockIdx.x * 64 + threadIdx.x + 1);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)t, 0, 0x7FFFFFFF, 0x00020000);
  int acc = 0;
  for (int k = 0; k < per_lane; k += 4) {
    v4i_ v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const uint32_t r = (uint32_t)(((s >> 32) * (uint64_t)rows) >> 32);
      v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (r * 64u + threadIdx.x) * 16u, 0, AUX);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += v[j].x ^ v[j].w;
  }
  if (acc == 0x12345678) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)2 << 30;                       // 2 GiB table: 8x the Infinity Cache
  const size_t cells = bytes / sizeof(double2);
  const uint32_t rows = (uint32_t)(cells / 64);
  double2 *t = nullptr;
  double *out = nullptr;
  CHECK(hipMalloc((void **)&t, bytes));
  CHECK(hipMalloc((void **)&out, 64));
  CHECK(hipMemset(t, 0, bytes));
  CHECK(hipDeviceSynchronize());
  const int waves = 4096, per_lane = 4096;
  for (int rep = 0; rep < 2; ++rep) {                         // the first dispatch of each kernel is the warm-up
    hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, t, cells, out);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(scatter_kernel, dim3(waves), dim3(64), 0, 0, t, rows, per_lane, out);
    CHECK(hipDeviceSynchronize());
  }
  {   // policy variants on the first 1 GiB of the table (buffer offsets are 32-bit), warm-up + measured dispatch each
    const uint32_t rows1 = (uint32_t)(((size_t)1 << 30) / 1024);
#define RUN_AUX(A) for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(scatter_aux_kernel<A>, dim3(waves), dim3(64), 0, 0, t, rows1, per_lane / 4, out); CHECK(hipDeviceSynchronize()); }
    RUN_AUX(0) RUN_AUX(1) RUN_AUX(2) RUN_AUX(3) RUN_AUX(16) RUN_AUX(17) RUN_AUX(18) RUN_AUX(19)
    printf("fetch_calib: scatter_aux_kernel<AUX> makes %.6g accesses of 16 B each (AUX = sc0 | nt << 1 | sc1 << 4)\n", (double)waves * 64 * (per_lane / 4));
  }
  const double accesses = (double)waves * 64 * per_lane;
  printf("fetch_calib: stream_kernel reads %.6g bytes (16 B per lane, coalesced, each cell once)\n", (double)bytes);
  printf("fetch_calib: scatter_kernel makes %.6g accesses of 16 B = %.6g algorithmic bytes; at 64 B per access %.6g, at 128 B per access %.6g\n",
         accesses, accesses * 16, accesses * 64, accesses * 128);
  (void)hipFree(t); (void)hipFree(out);
  return 0;
}
