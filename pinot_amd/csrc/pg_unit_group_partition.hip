// Instantiates the partitioned group-by kernels (pg_group_partition.h) -- see pg_launch.h.
#include "pg_group_partition.h"
#include "pg_launch.h"

namespace pg {

void launch_group_partition_histogram(int blocks, hipStream_t stream, const PartitionParams& pp) {
  group_partition_histogram_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(pp);
}

void launch_group_partition_scatter(int blocks, hipStream_t stream, const PartitionParams& pp) {
  group_partition_scatter_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(pp);
}

void launch_group_partition_aggregate(int work_items, size_t lds, hipStream_t stream, const PartitionParams& pp) {
  set_dynamic_lds(group_partition_aggregate_kernel, lds);
  group_partition_aggregate_kernel<<<dim3((unsigned)work_items), dim3(256), lds, stream>>>(pp);
}

int waves_group_partition_scatter() {
  static const int cap = max_waves_per_cu(group_partition_scatter_kernel);
  return cap;
}

}  // namespace pg
