// Instantiates scan_narrow_kernel (filters over columns of at most 8 bits, four tiles per wave and iteration) -- see pg_launch.h.
#include "pg_scan_narrow.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_narrow(bool single_leaf, int blocks, hipStream_t stream, const ScanParams& p) {
  if (single_leaf) scan_narrow_single_kernel<<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
  else scan_narrow_kernel<<<dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream>>>(p);
}

void launch_scan_narrow_batch(bool single_leaf, int total_blocks, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items) {
  BatchParams bp{items, block_first, num_items, 0};
  if (single_leaf) scan_narrow_batch_kernel<true><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
  else scan_narrow_batch_kernel<false><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
}

int waves_scan_narrow_batch(bool single_leaf) {
  static const int cap = max_waves_per_cu(scan_narrow_batch_kernel<false>);
  static const int cap1 = max_waves_per_cu(scan_narrow_batch_kernel<true>);
  return single_leaf ? cap1 : cap;
}

int waves_scan_narrow(bool single_leaf) {
  static const int cap = max_waves_per_cu(scan_narrow_kernel);
  static const int cap1 = max_waves_per_cu(scan_narrow_single_kernel);
  return single_leaf ? cap1 : cap;
}

}  // namespace pg
