// Instantiates scan_private_batch_kernel (many queries of the lane-private scan kernel in one launch) -- see pg_launch.h.
#include "pg_kernels.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_private_batch(bool one_slot, int total_blocks, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items) {
  BatchParams bp{items, block_first, num_items, 0};
  if (one_slot) scan_private_batch_kernel<1><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
  else scan_private_batch_kernel<kMaxAggCols><<<dim3((unsigned)total_blocks), dim3(kBlockThreads), 0, stream>>>(bp);
}

}  // namespace pg
