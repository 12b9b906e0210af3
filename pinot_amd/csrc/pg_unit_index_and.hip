// Instantiates index_and_kernel (the inverted-index children of a root AND, intersected window by window) -- see pg_launch.h.
#include "pg_index_and.h"
#include "pg_launch.h"

namespace pg {

void launch_index_and_kernel(int blocks, hipStream_t stream, const IndexAndParams& ap, uint32_t num_windows) {
  index_and_kernel<<<dim3((unsigned)blocks), dim3(64), 0, stream>>>(ap, num_windows);
}

// Windows in flight per CU: one wavefront each, bounded by the registers and by the 8 KB of LDS a wavefront scatters into.
int waves_index_and() {
  static const int cap = std::min(max_waves_per_cu_lean(index_and_kernel), (int)(160 * 1024 / 8192));
  return cap;
}

}  // namespace pg
