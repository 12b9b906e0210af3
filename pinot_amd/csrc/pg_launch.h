// pg_launch.h -- host-callable launchers of the big kernel templates.  Each family is instantiated in its own translation unit
// (pg_unit_*.hip) so that the four of them compile in parallel; pg_engine.hip only sees these declarations.
#ifndef PG_LAUNCH_H
#define PG_LAUNCH_H
#include <hip/hip_runtime.h>

#include <algorithm>

#include "pg_device.h"

namespace pg {

// Wavefronts per CU the register file admits for a kernel: 512 VGPRs per SIMD lane in 8-register granules, at most 6
// because these kernels use ~100 SGPRs (MI355X_MICROARCH.md: 256-thread blocks admitted = floor(800 / (sgpr granule + 16))).
template <typename K>
int max_waves_per_cu(K kernel) {
  hipFuncAttributes attr;
  if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)) != hipSuccess || attr.numRegs <= 0) return 16;
  const int alloc = ((attr.numRegs + 7) / 8) * 8;
  return std::max(1, std::min(6, 512 / alloc)) * 4;
}
// (kernels with few SGPRs: up to eight waves per SIMD)
template <typename K>
int max_waves_per_cu_lean(K kernel) {
  hipFuncAttributes attr;
  if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)) != hipSuccess || attr.numRegs <= 0) return 16;
  const int alloc = ((attr.numRegs + 7) / 8) * 8;
  return std::max(1, std::min(8, 512 / alloc)) * 4;
}
template <typename K>
void set_dynamic_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// scan_agg_kernel<kDma, slots (1 | kMaxAggCols), typed>: the LDS-staged scan -> filter -> aggregate kernel
void launch_scan_agg(bool dma, bool one_slot, bool typed, int blocks, int threads, size_t lds, hipStream_t stream, const ScanParams& p);
int waves_scan_agg(bool one_slot, bool typed);
// scan_private_kernel<slots>: the lane-private scan -> filter -> aggregate kernel
void launch_scan_private(int agg_cols, int blocks, hipStream_t stream, const ScanParams& p);      // instantiated for 1 and kMaxAggCols slots
int waves_scan_private(int agg_cols);
// scan_private_fsm_kernel<1 | kMaxAggCols>: the same with numEntriesScannedInFilter's transducer walked inside (ScanParams.fsm_*)
void launch_scan_private_fsm(int agg_cols, int blocks, hipStream_t stream, const ScanParams& p);
int waves_scan_private_fsm(int agg_cols);
// scan_private_batch_kernel<slots>: many queries in one launch (pg_execute_batch); items / block_first are device memory
void launch_scan_private_batch(bool one_slot, int total_blocks, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items);
// scan_sparse_kernel: aggregation of the docs one sparse bitmap names, eight tiles per wave and iteration (pg_scan_sparse.h)
void launch_scan_sparse(bool one_slot, int blocks, hipStream_t stream, const ScanParams& p);      // one_slot: at most one aggregated column
int waves_scan_sparse(bool one_slot);
// scan_simple_kernel: one dictionary-range leaf (or none) + at most one aggregated packed column of <= 20 bits (pg_scan_simple.h)
void launch_scan_simple(int blocks, int threads, hipStream_t stream, const ScanParams& p, bool set_leaf = false);      // threads: kBlockThreads or kWideBlockThreads; set_leaf: scan_simple_set_kernel (the one leaf is a dictId set, looked up in LDS)
int waves_scan_simple();
// scan_raw_kernel: one raw INT range leaf (or no filter) + at most one aggregated raw INT column, five waves per SIMD, coalesced reads (pg_scan_raw.h)
void launch_scan_raw(int blocks, int threads, hipStream_t stream, const ScanParams& p);
int waves_scan_raw();
// scan_lean_batch_kernel: the shared launch of pg_execute_batch for items of those two shapes (ScanParams.lean_kind), five waves per SIMD
void launch_scan_lean_batch(int kind, int total_blocks, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items);      // kind: 1 simple, 2 raw (every item)
int waves_scan_lean_batch(int kind);
// scan_narrow_kernel: COUNT(*) / docId bitmap of a filter over columns of at most 8 bits (pg_scan_narrow.h)
void launch_scan_narrow(bool single_leaf, int blocks, hipStream_t stream, const ScanParams& p);      // single_leaf: scan_narrow_single_kernel, eight tiles per iteration
int waves_scan_narrow(bool single_leaf);
// scan_narrow_batch_kernel<single>: pg_execute_batch's shared launch for items of those two shapes
void launch_scan_narrow_batch(bool single_leaf, int total_blocks, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items);
int waves_scan_narrow_batch(bool single_leaf);
// scan_typed_batch_kernel<slots>: the same for items of scan_private_typed_kernel's shape (agg_slots: the most any item needs -- 1, 2 or kMaxAggCols)
void launch_scan_typed_batch(int agg_slots, int total_blocks, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items);
int waves_scan_typed_batch(int agg_slots);
// index_and_kernel: the inverted-index children of a root AND, intersected window by window by a persistent grid of one-wave workgroups (pg_index_and.h)
void launch_index_and_kernel(int blocks, hipStream_t stream, const IndexAndParams& ap, uint32_t num_windows);
int waves_index_and();                 // index_and_kernel: wavefronts (= one-wave workgroups, a window in flight each) per CU
int index_and_batch_blocks_per_cu();   // index_and_batch_kernel: workgroups per CU ...
int index_and_batch_block_waves();     // ... of this many independent wavefronts
// index_and_batch_kernel: pg_execute_batch's shared launch for items whose whole device work is index_and_kernel (COUNT(*) over an index-only filter, the gathered aggregation)
void launch_index_and_batch(int total_blocks, hipStream_t stream, const IndexAndParams* items, const uint32_t* block_first, int num_items);
// scan_group_kernel<kDma, kLdsTable>: LDS-staged group-by
void launch_scan_group(bool dma, bool lds_table, int blocks, int threads, size_t lds, hipStream_t stream, const GroupParams& gp);
int waves_scan_group();
// group_private_kernel<kLdsTable>: lane-private group-by (no filter)
void launch_group_private(bool lds_table, int blocks, int threads, size_t lds, hipStream_t stream, const GroupParams& gp);
int waves_group_private();
// group_lds_batch_kernel: pg_execute_batch's shared launch for group-bys of the LDS-table form (items in device memory, one table slice each, `lds` = the largest item's table)
void launch_group_lds_batch(int total_blocks, int threads, size_t lds, hipStream_t stream, const GroupParams* items, const uint32_t* block_first, int num_items);
int waves_group_lds_batch();

// scan_hist_kernel<8 | 16 | 32, guarded>: lane-private scan whose SUM column is counted per dictId in an LDS histogram (pg_scan_hist.h)
void launch_scan_hist(int counter_bits, bool guarded, int blocks, size_t lds, hipStream_t stream, const ScanParams& p);
int waves_scan_hist(int counter_bits, bool guarded);
// scan_hist_batch_kernel<8 | 16 | 32>: pg_execute_batch's shared launch for items of that shape (plain counters; `lds` = the largest item's histogram)
void launch_scan_hist_batch(int counter_bits, int total_blocks, size_t lds, hipStream_t stream, const ScanParams* items, const uint32_t* block_first, int num_items);
int waves_scan_hist_batch(int counter_bits);

// scan_private_typed_kernel: lane-private scan for raw / 8-byte aggregated columns (pg_scan_typed.h)
void launch_scan_private_typed(int agg_cols, int blocks, hipStream_t stream, const ScanParams& p);      // instantiated for 1, 2 and kMaxAggCols slots
int waves_scan_private_typed(int agg_cols);
// partitioned group-by for key spaces above the LDS table (pg_group_partition.h): histogram, scatter, aggregate
void launch_group_partition_histogram(int blocks, hipStream_t stream, const PartitionParams& pp);
void launch_group_partition_scatter(int blocks, hipStream_t stream, const PartitionParams& pp);
void launch_group_partition_aggregate(int work_items, size_t lds, hipStream_t stream, const PartitionParams& pp);
// two-level runs: count / plan / scatter of pass A's records by fine partition (pg_group_partition.h)
void launch_group_repartition(int num_chunks, hipStream_t stream, const RepartitionParams& rp);
void launch_group_typed_direct(int blocks, hipStream_t stream, const GroupParams& gp);   // raw 8-byte aggregation inputs: direct HBM table
int waves_group_partition_scatter();
int blocks_per_cu_group_partition_scatter_packed(int num_partitions);

}  // namespace pg
#endif
