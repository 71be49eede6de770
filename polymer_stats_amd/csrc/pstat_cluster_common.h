// pstat_cluster_common.h -- small pieces shared by the chain-per-lane kernels of the clustering main
// (pstat_cluster.hip: cells in LDS; pstat_cluster_gm.hip: f64 cells in device memory).
#pragma once

#include <hip/hip_runtime.h>

#include "pstat_math.h"

#ifndef PSTAT_CLUSTER_GM_CELL
#define PSTAT_CLUSTER_GM_CELL 40u   // bytes per monomer of pstat_cluster_gm.hip's working buffer (40, or 48 with sin(theta) cached too)
#endif

namespace pstat {

template <typename R> struct V3 { R x, y, z; };
template <typename R> __device__ __forceinline__ R dot3(const V3<R> &a, const V3<R> &b) {
  return a.x * b.x + a.y * b.y + a.z * b.z;
}

// psi_j, inc/eap_chain.jl:45-47 (acos_r: pstat_math.h)
template <typename R> __device__ __forceinline__ R bond_angle(const V3<R> &a, const V3<R> &b) {
  return acos_r(fmin((R)1, fmax((R)-1, dot3(a, b))));
}

}  // namespace pstat
