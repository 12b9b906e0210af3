// Instantiates scan_raw_kernel -- see pg_launch.h.
#include "pg_scan_raw.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_raw(int blocks, int threads, hipStream_t stream, const ScanParams& p) {
  scan_raw_kernel<<<dim3((unsigned)blocks), dim3((unsigned)threads), 0, stream>>>(p);
}

int waves_scan_raw() {
  static const int cap = max_waves_per_cu_lean(scan_raw_kernel);
  return cap;
}

}  // namespace pg
