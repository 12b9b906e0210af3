// Instantiates scan_raw_kernel -- see pg_launch.h.
#include "pg_scan_raw.h"
#include "pg_launch.h"

namespace pg {

__global__ __launch_bounds__(kWideBlockThreads, PG_RAW_WAVES) void scan_raw_kernel(const ScanParams p) {
  __shared__ BlockPartial red[kWideBlockThreads / 64];      // (launched with kBlockThreads or kWideBlockThreads threads)
  __shared__ uint32_t fold_flag;
  scan_raw_body(p, blockIdx.x, gridDim.x, red, &fold_flag);
}

void launch_scan_raw(int blocks, int threads, hipStream_t stream, const ScanParams& p) {
  scan_raw_kernel<<<dim3((unsigned)blocks), dim3((unsigned)threads), 0, stream>>>(p);
}

int waves_scan_raw() {
  static const int cap = max_waves_per_cu_lean(scan_raw_kernel);
  return cap;
}

}  // namespace pg
