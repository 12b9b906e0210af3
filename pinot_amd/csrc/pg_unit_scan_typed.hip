// Instantiates scan_private_typed_kernel (raw and 8-byte aggregated columns in the lane-private layout) -- see pg_launch.h.
#include "pg_scan_typed.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_private_typed(int blocks, hipStream_t stream, const ScanParams& p) {
  scan_private_typed_kernel<<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
}

int waves_scan_private_typed() {
  static const int cap = max_waves_per_cu(scan_private_typed_kernel);
  return cap;
}

}  // namespace pg
