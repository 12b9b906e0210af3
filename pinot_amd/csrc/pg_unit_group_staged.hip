// Instantiates scan_group_kernel (the LDS-staged group-by kernel) -- see pg_launch.h.
#include "pg_kernels.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_group(bool dma, bool lds_table, int blocks, int threads, size_t lds, hipStream_t stream, const GroupParams& gp) {
  const dim3 grid((unsigned)blocks), block((unsigned)threads);
#define PG_LAUNCH(K) do { set_dynamic_lds(K, lds); K<<<grid, block, lds, stream>>>(gp); } while (0)
  if (lds_table) { if (dma) PG_LAUNCH((scan_group_kernel<true, true>)); else PG_LAUNCH((scan_group_kernel<false, true>)); }
  else if (gp.wide_keys) { if (dma) PG_LAUNCH((scan_group_kernel<true, false, true>)); else PG_LAUNCH((scan_group_kernel<false, false, true>)); }
  else { if (dma) PG_LAUNCH((scan_group_kernel<true, false>)); else PG_LAUNCH((scan_group_kernel<false, false>)); }
#undef PG_LAUNCH
}

int waves_scan_group() {
  static const int cap = max_waves_per_cu(scan_group_kernel<true, true>);
  return cap;
}

}  // namespace pg
