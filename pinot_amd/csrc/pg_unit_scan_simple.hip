// Instantiates scan_simple_kernel -- see pg_launch.h.
#include "pg_scan_simple.h"
#include "pg_launch.h"

namespace pg {

__global__ __launch_bounds__(kWideBlockThreads, PG_SIMPLE_WAVES) void scan_simple_kernel(const ScanParams p) {
  __shared__ BlockPartial red[kWideBlockThreads / 64];      // (launched with kBlockThreads or kWideBlockThreads threads)
  __shared__ uint32_t fold_flag;
  scan_simple_body(p, blockIdx.x, gridDim.x, red, &fold_flag);
}

// The same body with the one leaf a dictId set (IN / NOT IN) looked up in LDS.
__global__ __launch_bounds__(kWideBlockThreads, PG_SIMPLE_WAVES) void scan_simple_set_kernel(const ScanParams p) {
  __shared__ BlockPartial red[kWideBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  __shared__ uint32_t set_lds[kSetLdsWords];
  scan_simple_body<true>(p, blockIdx.x, gridDim.x, red, &fold_flag, set_lds);
}

void launch_scan_simple(int blocks, int threads, hipStream_t stream, const ScanParams& p, bool set_leaf) {
  if (set_leaf) scan_simple_set_kernel<<<dim3((unsigned)blocks), dim3((unsigned)threads), 0, stream>>>(p);
  else scan_simple_kernel<<<dim3((unsigned)blocks), dim3((unsigned)threads), 0, stream>>>(p);
}

int waves_scan_simple() {
  static const int cap = max_waves_per_cu_lean(scan_simple_kernel);
  return cap;
}

}  // namespace pg
