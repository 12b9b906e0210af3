// Instantiates group_private_kernel (the lane-private group-by kernel) -- see pg_launch.h.
#include "pg_kernels.h"
#include "pg_launch.h"

namespace pg {

void launch_group_private(bool lds_table, int blocks, int threads, size_t lds, hipStream_t stream, const GroupParams& gp) {
  const dim3 grid((unsigned)blocks), block((unsigned)threads);
  if (gp.hash_kind != 0) group_private_kernel<false, false, true><<<grid, block, lds, stream>>>(gp);      // Long / ArrayMap holders: hashed table (`lds`: the filter's dictId-set area, if any)
  else if (lds_table) { set_dynamic_lds(group_private_kernel<true>, lds); group_private_kernel<true><<<grid, block, lds, stream>>>(gp); }
  else if (gp.wide_keys) group_private_kernel<false, true><<<grid, block, lds, stream>>>(gp);      // key spaces above 2^24: 32-bit key multiplies
  else group_private_kernel<false><<<grid, block, lds, stream>>>(gp);
}

int waves_group_private() {
  static const int cap = max_waves_per_cu(group_private_kernel<true>);
  return cap;
}

}  // namespace pg
