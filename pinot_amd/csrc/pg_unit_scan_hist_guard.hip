// Instantiates scan_hist_kernel's guarded tier (returning adds + guard-bit claims: exact under heavy skew) -- see pg_scan_hist.h.
#include "pg_scan_hist.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_hist_guarded(int counter_bits, int blocks, size_t lds, hipStream_t stream, const ScanParams& p) {
  if (counter_bits == 16) {
    set_dynamic_lds(scan_hist_kernel<16, true>, lds);
    scan_hist_kernel<16, true><<<dim3((unsigned)blocks), dim3(kHistBlockThreads), lds, stream>>>(p);
  } else {
    set_dynamic_lds(scan_hist_kernel<8, true>, lds);
    scan_hist_kernel<8, true><<<dim3((unsigned)blocks), dim3(kHistBlockThreads), lds, stream>>>(p);
  }
}

int waves_scan_hist_guarded(int counter_bits) {
  static const int cap16 = max_waves_per_cu(scan_hist_kernel<16, true>);
  static const int cap8 = max_waves_per_cu(scan_hist_kernel<8, true>);
  return counter_bits == 16 ? cap16 : cap8;
}

}  // namespace pg
