// Instantiates scan_private_typed_kernel (raw and 8-byte aggregated columns in the lane-private layout) -- see pg_launch.h.
#include "pg_scan_typed.h"
#include "pg_launch.h"

namespace pg {

// agg_cols: aggregated columns of the query -- instantiated for 1, 2 and kMaxAggCols accumulator slots (a TypedAcc is ten registers:
// 0 / 4 / 31 of them spilled)
void launch_scan_private_typed(int agg_cols, int blocks, hipStream_t stream, const ScanParams& p) {
  if (agg_cols <= 1) scan_private_typed_kernel<1><<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
  else if (agg_cols == 2) scan_private_typed_kernel<2><<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
  else scan_private_typed_kernel<kMaxAggCols><<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
}

int waves_scan_private_typed(int agg_cols) {
  static const int cap1 = max_waves_per_cu(scan_private_typed_kernel<1>);
  static const int cap2 = max_waves_per_cu(scan_private_typed_kernel<2>);
  static const int cap4 = max_waves_per_cu(scan_private_typed_kernel<kMaxAggCols>);
  return agg_cols <= 1 ? cap1 : (agg_cols == 2 ? cap2 : cap4);
}

}  // namespace pg
