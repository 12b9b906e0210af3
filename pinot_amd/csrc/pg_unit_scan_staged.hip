// Instantiates scan_agg_kernel (the LDS-staged scan kernel) -- see pg_launch.h.
#include "pg_kernels.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_agg(bool dma, bool one_slot, bool typed, int blocks, int threads, size_t lds, hipStream_t stream, const ScanParams& p) {
  const dim3 grid((unsigned)blocks), block((unsigned)threads);
#define PG_LAUNCH(K) do { set_dynamic_lds(K, lds); K<<<grid, block, lds, stream>>>(p); } while (0)
  if (typed) { if (dma) PG_LAUNCH((scan_agg_kernel<true, kMaxAggCols, true>)); else PG_LAUNCH((scan_agg_kernel<false, kMaxAggCols, true>)); }
  else if (one_slot) { if (dma) PG_LAUNCH((scan_agg_kernel<true, 1>)); else PG_LAUNCH((scan_agg_kernel<false, 1>)); }
  else { if (dma) PG_LAUNCH((scan_agg_kernel<true, kMaxAggCols>)); else PG_LAUNCH((scan_agg_kernel<false, kMaxAggCols>)); }
#undef PG_LAUNCH
}

int waves_scan_agg(bool one_slot, bool typed) {
  static const int cap1 = max_waves_per_cu(scan_agg_kernel<true, 1>);
  static const int cap4 = max_waves_per_cu(scan_agg_kernel<true, kMaxAggCols>);
  static const int cap_typed = max_waves_per_cu(scan_agg_kernel<true, kMaxAggCols, true>);
  return typed ? cap_typed : (one_slot ? cap1 : cap4);
}

}  // namespace pg
