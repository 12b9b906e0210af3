// scan_sparse_kernel: aggregation of the docs a docId bitmap names, when the bitmap is SPARSE -- an index-led filter (the AND of
// inverted-index postings, index_and_kernel's output) or a selective scan filter whose docId set has been materialised first.
//
// What it replaces: ProjectionOperator over the docIds a BitmapDocIdIterator hands out (DocIdSetOperator.java:59-86 pulls 10 000 of
// them at a time; FixedBitSVForwardIndexReaderV2.readDictIds :84-99 then reads exactly those docs, one readUnchecked each) feeding
// Sum / Min / Max / AvgAggregationFunction.aggregate.  Only the matching docs' values are needed: SURVEY.md 8(d) charges such a query
// min(B(v), matches x 64 B) for the value column -- one sector per matching doc.
//
// Why a kernel of its own: scan_private_kernel takes ONE tile per wave and iteration and keeps ~130 registers per lane (the width
// switches of its filter leaves): four waves per SIMD.  With a bitmap leaf a tile is two DEPENDENT loads -- the lane's mask dword,
// then the values its set bits point at -- and nothing to compute in between: 488 K tiles of a 1 B-row segment / 4096 resident waves
// x two ~2 us round trips = 0.48 ms, whatever the selectivity (measured: C5-dense 0.49 ms for 1.0 GB of sectors + bitmap, 2.2 TB/s).
// This kernel is latency-proof instead of general: EIGHT tiles per wave and iteration -- eight mask dwords in flight, then one match of
// each of the eight tiles in flight per round (an 8-byte load at the doc's bit position, agg_sparse_private's read), rounds repeating
// while any lane has a match left -- and ~70 registers: seven waves per SIMD.  Eight times the tiles in flight per wave, 1.75 times
// the waves.
//
// Aggregations: COUNT, and SUM / MIN / MAX / AVG over columns read as bit-packed fields (dictIds for MIN / MAX, value-plane fields or
// arithmetic-progression dictIds for SUM) -- what scan_private_kernel's plane path takes.  Bit exact with it (same integer sums).
#pragma once
#include "pg_kernels.h"

namespace pg {

#ifndef PG_SPARSE_WAVES
#define PG_SPARSE_WAVES 4      // five waves (96 VGPRs, three spilled) measured: C5-dense 0.269 -> 0.308 ms -- the eight tiles in flight need the registers
#endif
// kAggSlots: 1 for queries with at most one aggregated column (fewer accumulators live across the walk), else kMaxAggCols
template <int kAggSlots>
__global__ __launch_bounds__(kBlockThreads, (kAggSlots == 1 ? PG_SPARSE_WAVES : 4)) void scan_sparse_kernel(const ScanParams p) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const long long total_waves = (long long)gridDim.x * waves_per_block;
  // Round 5: the kernel walks index_and_kernel's per-WINDOW tile masks (p.sparse_windows: 32 tiles = 65 536 docs per window) instead of a
  // tile list: the list cost a launch of its own between the two kernels (index_and_finalize_kernel: a prefix over the windows, ~10 us on a
  // query of 86).  A wave takes windows round-robin; an empty window is one scalar load, a window's set tiles go eight at a time.
  const uint32_t* __restrict__ mask_words = p.nodes[0].set_words;      // the filter is ONE bitmap leaf: dword 64 * tile + lane = the lane's 32 docs

  unsigned long long count = 0;
  unsigned long long sum[kAggSlots];
  uint32_t umin[kAggSlots], umax[kAggSlots];
#pragma unroll
  for (int a = 0; a < kAggSlots; ++a) { sum[a] = 0; umin[a] = 0xFFFFFFFFu; umax[a] = 0u; }

  // (a wave's next four windows' masks are loaded together: on the sparse AND nearly every window is empty and a wave's work is its chain of
  //  mask loads -- four independent scalar loads instead of four dependent round trips)
  constexpr int kAhead = 4;
  for (long long window0 = (long long)blockIdx.x * waves_per_block + wave_in_block; window0 < (long long)p.sparse_num_windows; window0 += total_waves * kAhead) {
   uint32_t ahead[kAhead];
#pragma unroll
   for (int k = 0; k < kAhead; ++k) ahead[k] = window0 + k * total_waves < (long long)p.sparse_num_windows ? p.sparse_windows[window0 + k * total_waves].tiles : 0u;
#pragma unroll
   for (int k = 0; k < kAhead; ++k) {
   const long long window = window0 + k * total_waves;
   uint32_t window_tiles = ahead[k];
   while (window_tiles != 0u) {
    uint32_t tile[kSparseTiles];                              // (a segment has fewer than 2^20 tiles)
    uint32_t m[kSparseTiles];
    bool in_use[kSparseTiles];
    // Every load below is UNCONDITIONAL (a lane or a tile with nothing to read points at an address that is always there): a load inside
    // an exec-masked branch is waited for before the branch is left, which made the eight loads of a round eight round trips.
    uint32_t last_tile = 0u;
#pragma unroll
    for (int i = 0; i < kSparseTiles; ++i) {
      in_use[i] = window_tiles != 0u;
      if (in_use[i]) { last_tile = (uint32_t)window * 32u + (uint32_t)__builtin_ctz(window_tiles); window_tiles &= window_tiles - 1u; }
      tile[i] = last_tile;                                   // (an unused slot repeats a tile of the window: its mask is dropped below)
    }
#pragma unroll
    for (int i = 0; i < kSparseTiles; ++i) m[i] = mask_words[(long long)tile[i] * 64 + lane];
#pragma unroll
    for (int i = 0; i < kSparseTiles; ++i) {
      const long long rem = (long long)p.num_docs - ((long long)tile[i] * 2048 + lane * 32);          // docs past numDocs (last tile only)
      m[i] &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u));
      if (!in_use[i]) m[i] = 0u;                                                            // fewer than eight tiles were left in the window
      count += (unsigned)__builtin_popcount(m[i]);
    }
    for (int a = 0; a < p.num_agg_cols; ++a) {
      const DevAggCol& ac = p.agg_cols[a];
      const uint32_t b = (uint32_t)ac.bits;
      const uint32_t field_mask = (1u << b) - 1u;
      const bool need_sum = ac.need_sum != 0, need_minmax = ac.need_minmax != 0;
      unsigned long long wsum = 0;
      uint32_t tmin = 0xFFFFFFFFu, tmax = 0u;
      uint32_t rest[kSparseTiles];
      bool any = false;
#pragma unroll
      for (int i = 0; i < kSparseTiles; ++i) { rest[i] = m[i]; any |= rest[i] != 0u; }
      while (__builtin_amdgcn_ballot_w64(any) != 0ull) {
        Dwords2 d[kSparseTiles];
        uint32_t sh[kSparseTiles];
        bool ok[kSparseTiles];
        any = false;
#pragma unroll
        for (int i = 0; i < kSparseTiles; ++i) {
          ok[i] = rest[i] != 0u;
          const uint32_t j = ok[i] ? (uint32_t)__builtin_ctz(rest[i]) : 0u;
          rest[i] &= rest[i] - 1u;
          any |= rest[i] != 0u;
          const uint32_t bit = j * b;
          sh[i] = 64u - (bit & 31u) - b;
          // (a lane without a match in this tile reads the column's first dwords: one line the whole chip shares)
          const uint32_t* at = reinterpret_cast<const uint32_t*>(ac.fwd + (long long)tile[i] * (256ll * (long long)b)) + (uint32_t)lane * b + (bit >> 5);
          d[i] = *reinterpret_cast<const Dwords2*>(ok[i] ? at : reinterpret_cast<const uint32_t*>(ac.fwd));
        }
#pragma unroll
        for (int i = 0; i < kSparseTiles; ++i) {
          const unsigned long long x = ((unsigned long long)__builtin_bswap32(d[i].x) << 32) | (unsigned long long)__builtin_bswap32(d[i].y);
          const uint32_t v = (uint32_t)(x >> sh[i]) & field_mask;
          if (ok[i]) {
            if (need_sum) wsum += v;
            if (need_minmax) { tmax = v > tmax ? v : tmax; tmin = v < tmin ? v : tmin; }
          }
        }
      }
#pragma unroll
      for (int s = 0; s < kAggSlots; ++s) {
        if (s == a) {
          sum[s] += wsum;
          umin[s] = tmin < umin[s] ? tmin : umin[s];
          umax[s] = tmax > umax[s] ? tmax : umax[s];
        }
      }
    }
   }
   }
  }

  BlockPartial mine;
  partial_identity(mine);
  mine.count = (unsigned long long)wave_sum_i64((long long)count);
#pragma unroll
  for (int a = 0; a < kAggSlots; ++a) {
    if (a >= p.num_agg_cols) continue;
    mine.sum[a] = wave_sum_i64((long long)sum[a]);
    mine.kmin[a] = wave_min_i32(umin[a] == 0xFFFFFFFFu ? 0x7FFFFFFF : (int32_t)umin[a]);
    mine.kmax[a] = wave_max_i32(count == 0ull ? (int32_t)0x80000000 : (int32_t)umax[a]);
  }
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  publish_block_partial(p, red, waves_per_block, &fold_flag);
}

}  // namespace pg
