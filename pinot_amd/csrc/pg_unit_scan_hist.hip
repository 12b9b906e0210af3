// Instantiates scan_hist_kernel (SUM through an LDS histogram of the matching dictIds), unguarded tier -- see pg_launch.h.
#include "pg_scan_hist.h"
#include "pg_launch.h"

namespace pg {

void launch_scan_hist_guarded(int counter_bits, int blocks, size_t lds, hipStream_t stream, const ScanParams& p);
int waves_scan_hist_guarded(int counter_bits);

void launch_scan_hist(int counter_bits, bool guarded, int blocks, size_t lds, hipStream_t stream, const ScanParams& p) {
  if (guarded && counter_bits < 32) { launch_scan_hist_guarded(counter_bits, blocks, lds, stream, p); return; }
  if (counter_bits == 32) {
    set_dynamic_lds(scan_hist_kernel<32, false>, lds);
    scan_hist_kernel<32, false><<<dim3((unsigned)blocks), dim3(kHistBlockThreads), lds, stream>>>(p);
  } else if (counter_bits == 16) {
    set_dynamic_lds(scan_hist_kernel<16, false>, lds);
    scan_hist_kernel<16, false><<<dim3((unsigned)blocks), dim3(kHistBlockThreads), lds, stream>>>(p);
  } else {
    set_dynamic_lds(scan_hist_kernel<8, false>, lds);
    scan_hist_kernel<8, false><<<dim3((unsigned)blocks), dim3(kHistBlockThreads), lds, stream>>>(p);
  }
}

int waves_scan_hist(int counter_bits, bool guarded) {
  if (guarded && counter_bits < 32) return waves_scan_hist_guarded(counter_bits);
  static const int cap32 = max_waves_per_cu(scan_hist_kernel<32, false>);
  static const int cap16 = max_waves_per_cu(scan_hist_kernel<16, false>);
  static const int cap8 = max_waves_per_cu(scan_hist_kernel<8, false>);
  return counter_bits == 32 ? cap32 : (counter_bits == 16 ? cap16 : cap8);
}

}  // namespace pg
