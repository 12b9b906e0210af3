// Instantiates scan_private_kernel (the lane-private scan kernel) -- see pg_launch.h.
#include "pg_kernels.h"
#include "pg_launch.h"

namespace pg {

// agg_cols: how many aggregated columns the query has (the multi-column form keeps its accumulators in LDS: PrivateAccLds)
void launch_scan_private(int agg_cols, int blocks, hipStream_t stream, const ScanParams& p) {
  if (agg_cols <= 1) scan_private_kernel<1><<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
  else scan_private_kernel<kMaxAggCols><<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
}

int waves_scan_private(int agg_cols) {
  static const int cap1 = max_waves_per_cu(scan_private_kernel<1>);
  static const int cap4 = max_waves_per_cu(scan_private_kernel<kMaxAggCols>);
  return agg_cols <= 1 ? cap1 : cap4;
}

}  // namespace pg
