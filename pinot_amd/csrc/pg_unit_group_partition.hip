// Instantiates the partitioned group-by kernels (pg_group_partition.h) -- see pg_launch.h.
#include "pg_group_partition.h"
#include "pg_launch.h"

namespace pg {

void launch_group_partition_histogram(int blocks, hipStream_t stream, const PartitionParams& pp) {
  group_partition_histogram_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(pp);
}

void launch_group_partition_scatter(int blocks, hipStream_t stream, const PartitionParams& pp) {
  if (pp.packed_bits > 0) {
    const size_t lds = partition_scatter_packed_lds_bytes(pp.num_partitions);
    set_dynamic_lds(group_partition_scatter_packed_kernel, lds);
    group_partition_scatter_packed_kernel<<<dim3((unsigned)blocks), dim3(256), lds, stream>>>(pp);
    return;
  }
  const size_t lds = partition_scatter_lds_bytes();
  set_dynamic_lds(group_partition_scatter_kernel, lds);
  group_partition_scatter_kernel<<<dim3((unsigned)blocks), dim3(256), lds, stream>>>(pp);
}

void launch_group_partition_aggregate(int work_items, size_t lds, hipStream_t stream, const PartitionParams& pp) {
  const dim3 grid((unsigned)work_items), block(256);
  if (pp.packed_bits > 0) {
    if (pp.gp.num_group_aggs == 0) { set_dynamic_lds(group_partition_aggregate_kernel<0, true>, lds); group_partition_aggregate_kernel<0, true><<<grid, block, lds, stream>>>(pp); }
    else { set_dynamic_lds(group_partition_aggregate_kernel<1, true>, lds); group_partition_aggregate_kernel<1, true><<<grid, block, lds, stream>>>(pp); }
    return;
  }
  switch (pp.gp.num_group_aggs) {
    case 0: set_dynamic_lds(group_partition_aggregate_kernel<0>, lds); group_partition_aggregate_kernel<0><<<grid, block, lds, stream>>>(pp); break;
    case 1: set_dynamic_lds(group_partition_aggregate_kernel<1>, lds); group_partition_aggregate_kernel<1><<<grid, block, lds, stream>>>(pp); break;
    case 2: set_dynamic_lds(group_partition_aggregate_kernel<2>, lds); group_partition_aggregate_kernel<2><<<grid, block, lds, stream>>>(pp); break;
    default: set_dynamic_lds(group_partition_aggregate_kernel<3>, lds); group_partition_aggregate_kernel<3><<<grid, block, lds, stream>>>(pp); break;
  }
}

void launch_group_repartition(int num_chunks, hipStream_t stream, const RepartitionParams& rp) {
  group_repartition_count_kernel<<<dim3((unsigned)num_chunks), dim3(256), 0, stream>>>(rp);
  group_repartition_plan_kernel<<<dim3((unsigned)rp.num_coarse), dim3(1024), 0, stream>>>(rp);
  group_repartition_scatter_kernel<<<dim3((unsigned)num_chunks), dim3(256), 0, stream>>>(rp);
}

void launch_group_typed_direct(int blocks, hipStream_t stream, const GroupParams& gp) {
  if (gp.hash_kind != 0) group_typed_direct_kernel<false, true><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(gp);      // Long / ArrayMap holders: hashed table
  else if (gp.wide_keys) group_typed_direct_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(gp);
  else group_typed_direct_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(gp);
}

int waves_group_partition_scatter() {
  // registers and the 70 KB staging area (two workgroups per CU) both bound it
  static const int cap = std::min(max_waves_per_cu(group_partition_scatter_kernel), 8);
  return cap;
}

int blocks_per_cu_group_partition_scatter_packed(int num_partitions) {
  // the staging area is 48 KB + 12 B per partition: three workgroups per CU up to ~330 partitions, two beyond
  const int by_lds = (int)((156 * 1024) / partition_scatter_packed_lds_bytes(num_partitions));
  static const int by_registers = std::max(1, max_waves_per_cu(group_partition_scatter_packed_kernel) / 4);
  return std::max(1, std::min(by_lds, by_registers));
}

}  // namespace pg
