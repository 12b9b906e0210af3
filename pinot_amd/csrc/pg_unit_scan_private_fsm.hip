// Instantiates scan_private_fsm_kernel (the lane-private scan kernel with the transducer walk of numEntriesScannedInFilter inside) -- see pg_launch.h.
#include "pg_kernels.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_private_fsm(int agg_cols, int blocks, hipStream_t stream, const ScanParams& p) {
  if (agg_cols <= 1) scan_private_fsm_kernel<1><<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
  else scan_private_fsm_kernel<kMaxAggCols><<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
}

int waves_scan_private_fsm(int agg_cols) {
  static const int cap1 = max_waves_per_cu(scan_private_fsm_kernel<1>);
  static const int cap4 = max_waves_per_cu(scan_private_fsm_kernel<kMaxAggCols>);
  return agg_cols <= 1 ? cap1 : cap4;
}

}  // namespace pg
