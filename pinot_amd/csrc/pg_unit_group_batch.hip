// Instantiates group_lds_batch_kernel (pg_execute_batch's shared launch for group-bys of the LDS-table form) -- see pg_launch.h.
#include "pg_kernels.h"
#include "pg_launch.h"

namespace pg {

void launch_group_lds_batch(int total_blocks, int threads, size_t lds, hipStream_t stream, const GroupParams* items, const uint32_t* block_first, int num_items) {
  GroupBatchParams bp{items, block_first, num_items, 0};
  set_dynamic_lds(group_lds_batch_kernel<false>, lds);
  group_lds_batch_kernel<false><<<dim3((unsigned)total_blocks), dim3((unsigned)threads), lds, stream>>>(bp);
}

int waves_group_lds_batch() {
  static const int cap = max_waves_per_cu(group_lds_batch_kernel<false>);
  return cap;
}

}  // namespace pg
