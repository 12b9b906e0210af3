// Instantiates scan_hist_kernel (SUM through an LDS histogram of the matching dictIds) -- see pg_launch.h.
#include "pg_scan_hist.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_hist(int counter_bits, int blocks, size_t lds, hipStream_t stream, const ScanParams& p) {
  if (counter_bits == 32) {
    set_dynamic_lds(scan_hist_kernel<32>, lds);
    scan_hist_kernel<32><<<dim3((unsigned)blocks), dim3(kHistBlockThreads), lds, stream>>>(p);
  } else if (counter_bits == 16) {
    set_dynamic_lds(scan_hist_kernel<16>, lds);
    scan_hist_kernel<16><<<dim3((unsigned)blocks), dim3(kHistBlockThreads), lds, stream>>>(p);
  } else {
    set_dynamic_lds(scan_hist_kernel<8>, lds);
    scan_hist_kernel<8><<<dim3((unsigned)blocks), dim3(kHistBlockThreads), lds, stream>>>(p);
  }
}

int waves_scan_hist(int counter_bits) {
  static const int cap32 = max_waves_per_cu(scan_hist_kernel<32>);
  static const int cap16 = max_waves_per_cu(scan_hist_kernel<16>);
  static const int cap8 = max_waves_per_cu(scan_hist_kernel<8>);
  return counter_bits == 32 ? cap32 : (counter_bits == 16 ? cap16 : cap8);
}

}  // namespace pg
