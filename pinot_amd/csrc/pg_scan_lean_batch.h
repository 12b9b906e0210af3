// scan_lean_batch_kernel: pg_execute_batch's shared launch for items that have scan_simple_kernel's or scan_raw_kernel's shape -- the
// small-segment regime of a real server is mostly these (`SELECT COUNT(*) / SUM(v) ... WHERE f <op> x` over a few million rows per
// segment: BASELINE.json configs[0] x many).  Workgroups [block_first[i], block_first[i + 1]) work on items[i], every item folds and
// publishes its own record, exactly like scan_private_batch_kernel -- which runs the general body at four waves per SIMD (64 VGPRs of
// filter program, mask stack and slot arrays, 28 B of scratch): 0.64-0.69 of 8 TB/s on 64 x 10 M rows where the lean kernels reach
// 0.71-0.74 alone.  Same per-tile code as the two kernels (scan_simple_body / scan_raw_body), five waves per SIMD.
// BaseCombineOperator.java:85-142 is what a server does with such a query: one task per segment on a thread pool.
#pragma once
#include "pg_scan_raw.h"
#include "pg_scan_simple.h"

namespace pg {

#ifndef PG_LEAN_BATCH_WAVES
#define PG_LEAN_BATCH_WAVES 5
#endif

// kKind: ScanParams.lean_kind of EVERY item of the launch -- 1: scan_simple_body, 2: scan_raw_body, 13: scan_simple_body<kSet> -- (one kernel with both bodies spilled 23 registers: the engine groups a batch's
// items by kind, a launch per kind)
template <int kKind>
// (the raw body keeps a whole 8 KB tile per wave in flight: with the item's fields in registers as well it wants 4 waves per SIMD -- as
//  many bytes in flight as five waves of the packed body; at five it spilled 21 registers)
__global__ __launch_bounds__(kBlockThreads, (kKind == 2 ? 4 : PG_LEAN_BATCH_WAVES)) void scan_lean_batch_kernel(const BatchParams bp) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  int lo = 0, hi = bp.num_items - 1;                // the last item whose first workgroup is at or before this one
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (bp.block_first[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const uint32_t first = bp.block_first[lo];
  // (the item through a constant-address-space reference: scalar loads, like kernel arguments -- see scan_private_batch_kernel)
  typedef const __attribute__((address_space(4))) ScanParams ConstantScanParams;
  const ConstantScanParams& item = *(ConstantScanParams*)(bp.items + lo);
  if constexpr (kKind == 2) scan_raw_body(item, blockIdx.x - first, bp.block_first[lo + 1] - first, red, &fold_flag);
  else if constexpr (kKind == 13) {
    // (scan_simple_set_kernel's body: the item's one leaf is a dictId set of at most 16 bits, staged in LDS from the batch's blob)
    __shared__ uint32_t set_lds[kSetLdsWords];
    scan_simple_body<true>(item, blockIdx.x - first, bp.block_first[lo + 1] - first, red, &fold_flag, set_lds);
  } else scan_simple_body(item, blockIdx.x - first, bp.block_first[lo + 1] - first, red, &fold_flag);
}

}  // namespace pg
