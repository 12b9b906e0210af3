// Instantiates scan_sparse_kernel -- see pg_launch.h.
#include "pg_scan_sparse.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_sparse(bool one_slot, int blocks, hipStream_t stream, const ScanParams& p) {
  if (one_slot) scan_sparse_kernel<1><<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
  else scan_sparse_kernel<kMaxAggCols><<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
}

int waves_scan_sparse(bool one_slot) {
  static const int cap1 = max_waves_per_cu_lean(scan_sparse_kernel<1>);
  static const int cap4 = max_waves_per_cu_lean(scan_sparse_kernel<kMaxAggCols>);
  return one_slot ? cap1 : cap4;
}

}  // namespace pg
