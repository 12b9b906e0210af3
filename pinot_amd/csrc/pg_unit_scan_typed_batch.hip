// Instantiates scan_typed_batch_kernel (pg_execute_batch's shared launch for items of scan_private_typed_kernel's shape) -- see pg_launch.h.
#include "pg_scan_typed.h"
#include "pg_launch.h"

namespace pg {

// agg_slots: 1, 2 or kMaxAggCols accumulator slots -- the most any item of the launch needs
void launch_scan_typed_batch(int agg_slots, int total_blocks, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items) {
  BatchParams bp{items, block_first, num_items, 0};
  if (agg_slots <= 1) scan_typed_batch_kernel<1><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
  else if (agg_slots == 2) scan_typed_batch_kernel<2><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
  else scan_typed_batch_kernel<kMaxAggCols><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
}

int waves_scan_typed_batch(int agg_slots) {
  static const int cap1 = max_waves_per_cu(scan_typed_batch_kernel<1>);
  static const int cap2 = max_waves_per_cu(scan_typed_batch_kernel<2>);
  static const int cap4 = max_waves_per_cu(scan_typed_batch_kernel<kMaxAggCols>);
  return agg_slots <= 1 ? cap1 : (agg_slots == 2 ? cap2 : cap4);
}

}  // namespace pg
