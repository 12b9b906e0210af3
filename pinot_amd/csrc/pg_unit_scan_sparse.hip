// Instantiates scan_sparse_kernel -- see pg_launch.h.
#include "pg_scan_sparse.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_sparse(int blocks, hipStream_t stream, const ScanParams& p) {
  scan_sparse_kernel<<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
}

int waves_scan_sparse() {
  static const int cap = max_waves_per_cu(scan_sparse_kernel);
  return cap;
}

}  // namespace pg
