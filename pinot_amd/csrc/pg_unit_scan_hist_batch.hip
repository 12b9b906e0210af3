// Instantiates scan_hist_batch_kernel (pg_execute_batch's shared launch for items of scan_hist_kernel's shape) -- see pg_launch.h.
#include "pg_scan_hist.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_hist_batch(int counter_bits, int total_blocks, size_t lds, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items) {
  BatchParams bp{items, block_first, num_items, 0};
  if (counter_bits == 32) {
    set_dynamic_lds(scan_hist_batch_kernel<32>, lds);
    scan_hist_batch_kernel<32><<<dim3((unsigned)total_blocks), dim3(kHistBlockThreads), lds, stream>>>(bp);
  } else if (counter_bits == 16) {
    set_dynamic_lds(scan_hist_batch_kernel<16>, lds);
    scan_hist_batch_kernel<16><<<dim3((unsigned)total_blocks), dim3(kHistBlockThreads), lds, stream>>>(bp);
  } else {
    set_dynamic_lds(scan_hist_batch_kernel<8>, lds);
    scan_hist_batch_kernel<8><<<dim3((unsigned)total_blocks), dim3(kHistBlockThreads), lds, stream>>>(bp);
  }
}

int waves_scan_hist_batch(int counter_bits) {
  static const int cap32 = max_waves_per_cu(scan_hist_batch_kernel<32>);
  static const int cap16 = max_waves_per_cu(scan_hist_batch_kernel<16>);
  static const int cap8 = max_waves_per_cu(scan_hist_batch_kernel<8>);
  return counter_bits == 32 ? cap32 : (counter_bits == 16 ? cap16 : cap8);
}

}  // namespace pg
