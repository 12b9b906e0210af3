// Instantiates scan_lean_batch_kernel -- see pg_launch.h.
#include "pg_scan_lean_batch.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_lean_batch(int kind, int total_blocks, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items) {
  BatchParams bp{items, block_first, num_items, 0};
  if (kind == 2) scan_lean_batch_kernel<2><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
  else if (kind == 13) scan_lean_batch_kernel<13><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
  else scan_lean_batch_kernel<1><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
}

int waves_scan_lean_batch(int kind) {
  static const int cap1 = max_waves_per_cu_lean(scan_lean_batch_kernel<1>), cap2 = max_waves_per_cu_lean(scan_lean_batch_kernel<2>);
  static const int cap13 = max_waves_per_cu_lean(scan_lean_batch_kernel<13>);
  return kind == 2 ? cap2 : (kind == 13 ? cap13 : cap1);
}

}  // namespace pg
