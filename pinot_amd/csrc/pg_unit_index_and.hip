// Instantiates index_and_kernel (the inverted-index children of a root AND, intersected window by window) -- see pg_launch.h.
#include "pg_index_and.h"
#include "pg_launch.h"

namespace pg {

void launch_index_and_kernel(int blocks, hipStream_t stream, const IndexAndParams& ap, uint32_t num_windows) {
  index_and_kernel<<<dim3((unsigned)blocks), dim3(64 * kAndBlockWaves), 0, stream>>>(ap, num_windows);
}

// pg_execute_batch's shared launch for index-led items (items / block_first are device memory)
void launch_index_and_batch(int total_blocks, hipStream_t stream, const IndexAndParams* items, const uint32_t* block_first, int num_items) {
  IndexAndBatchParams bp;
  bp.items = items; bp.block_first = block_first; bp.num_items = num_items; bp.reserved = 0;
  index_and_batch_kernel<<<dim3((unsigned)total_blocks), dim3(64 * kAndBatchBlockWaves), 0, stream>>>(bp);
}

// Wavefronts (a window in flight each) per CU: bounded by the registers and by the ~10 KB of LDS a wavefront scatters into and keeps its
// guesses in.  index_and_kernel: that many one-wave workgroups; index_and_batch_kernel: workgroups of index_and_batch_block_waves().
int waves_index_and() {
  static const int cap = std::max(1, std::min(max_waves_per_cu_lean(index_and_kernel), (int)(160 * 1024 / (8192 + 1536))));
  return cap;
}
int index_and_batch_blocks_per_cu() {
  static const int cap = std::max(1, std::min(max_waves_per_cu_lean(index_and_batch_kernel) / kAndBatchBlockWaves, (int)(160 * 1024 / ((8192 + 1536 + 256) * kAndBatchBlockWaves))));
  return cap;
}
int index_and_batch_block_waves() { return kAndBatchBlockWaves; }

}  // namespace pg
