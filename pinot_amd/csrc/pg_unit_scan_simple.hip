// Instantiates scan_simple_kernel -- see pg_launch.h.
#include "pg_scan_simple.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_simple(int blocks, int threads, hipStream_t stream, const ScanParams& p) {
  scan_simple_kernel<<<dim3((unsigned)blocks), dim3((unsigned)threads), 0, stream>>>(p);
}

int waves_scan_simple() {
  static const int cap = max_waves_per_cu_lean(scan_simple_kernel);
  return cap;
}

}  // namespace pg
