// lh_launch.hpp -- launch-shape helpers shared by the kernel translation units (lh_kernels.hip, lh_knn.hip).
#pragma once
#include "lh_device.hpp"

namespace lh {

// XCD-aware workgroup -> (job, block) map.  The dispatcher places workgroup b on XCD b % 8 (observed, used for speed
// only): all workgroups of one job are given ids congruent mod 8, so a job's tree (2 MB for 100 k points) is walked
// from ONE XCD's 4-MB L2 instead of being pulled into all eight.
// Only complete groups of 8 jobs are pinned; the remaining (njobs % 8) jobs -- e.g. a single lh_gicp_align -- use the
// plain map and spread over all XCDs.
__device__ __forceinline__ bool xcd_job_map(int njobs, int bpj, int& job, int& blk) {
  int L = blockIdx.x;
  int pinned = njobs & ~7;
  int npin = pinned * bpj;
  if (L < npin) {
    int xcd = L & 7, s = L >> 3;
    int jl = s / bpj;
    blk = s - jl * bpj;
    job = jl * 8 + xcd;
    return true;
  }
  L -= npin;
  int jl = L / bpj;
  blk = L - jl * bpj;
  job = pinned + jl;
  return job < njobs;
}
static inline int xcd_grid(int njobs, int bpj) { return njobs * bpj; }
static inline size_t stack_lds_bytes(int /*depth*/, int threads) { return (size_t)LDS_STACK * threads * sizeof(uint64_t); }

}  // namespace lh
