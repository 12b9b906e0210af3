// scan_simple_kernel: ONE dictionary-range leaf (or no filter at all; scan_simple_set_kernel: one dictId-SET leaf) and at most ONE aggregated column read as bit-packed fields --
// `SELECT COUNT(*) / SUM(v) / MIN / MAX / AVG(v) ... WHERE f <op> x`: a single-predicate filter in front of a single aggregated column,
// the shape of BASELINE.json configs[1] and the commonest shape of a segment query.
//
// What it replaces: the same operators as scan_private_kernel (DocIdSetOperator + SVScanDocIdIterator over one
// FixedBitSVForwardIndexReaderV2 column, ProjectionOperator, one AggregationFunction.aggregate: core/operator/DocIdSetOperator.java:59-86,
// dociditerators/SVScanDocIdIterator.java:76-142, query/aggregation/function/SumAggregationFunction.java:78-84).
//
// Why a kernel of its own: what separates scan_private_kernel from the box's stream rate is memory-level parallelism, not instructions --
// a wave has ONE column chunk in flight at a time (load -> decode -> load -> decode), and at ~127 registers only four waves fit a SIMD:
// ~14 MB in flight on the whole chip against the ~13 MB a 2 us round trip at 6.4 TB/s needs, with half of every wave's time spent
// decoding.  Holding more of a wave's own loads in flight costs the general kernel registers it does not have (measured and dropped
// there).  This kernel carries none of the general machinery -- no filter program, no mask stack, no leaf-kind switch, no slot arrays,
// no tile lists -- and only the widths up to kSimpleMaxBits, so that the register allocation is that of one chunk plus sixteen decoded
// values: twice the waves per SIMD, i.e. twice the requests in flight, with the same per-tile code (range_private_dispatch,
// agg_private_dispatch, agg_sparse_private of pg_kernels.h).  Bit exact with scan_private_kernel (same integer sums, same masks).
#pragma once
#include "pg_kernels.h"

namespace pg {

// (kSimpleMaxBits = 20 in pg_device.h: columns of up to 2^20 distinct values / plane fields)
#ifndef PG_SIMPLE_WAVES
#define PG_SIMPLE_WAVES 5               // wavefronts per SIMD the register allocation must allow (94 VGPRs, no spills; 6: 80 VGPRs and 26 spilled, 8: 64 and 132)
#endif

template <bool kLoZero, typename WP>
__device__ __forceinline__ uint32_t simple_range_dispatch(int b, WP lane_words, uint32_t lo, uint32_t span) {
  uint32_t m = 0;
  switch (b) {
#define PG_CASE(B) case B: range16_private<B, 0, kLoZero>(lane_words, lo, span, m); range16_private<B, 1, kLoZero>(lane_words, lo, span, m); break;
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20)
#undef PG_CASE
    default: break;
  }
  return __builtin_bitreverse32(m);      // value j -> bit j
}

// The leaf as a dictId SET (InPredicateEvaluator / NotInPredicateEvaluator; round 6b): the set's words are in LDS (stage_filter_sets, zero-padded
// to the column's dictId range), a lookup is one ds_read_b32 -- eight at a time, so that the registers stay those of the range form.
template <int B, int H, typename WP>
__device__ __forceinline__ void set16_simple(WP __restrict__ lane_words, const uint32_t* set_lds, uint32_t& m) {
  uint32_t v[16];
  decode16_private<B, H>(lane_words, v);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    uint32_t x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = set_lds[v[8 * g + j] >> 5];
#pragma unroll
    for (int j = 0; j < 8; ++j) m = (m << 1) | __builtin_amdgcn_ubfe(x[j], v[8 * g + j] & 31u, 1);
  }
}
template <typename WP>
__device__ __forceinline__ uint32_t simple_set_dispatch(int b, WP lane_words, const uint32_t* set_lds) {
  uint32_t m = 0;
  switch (b) {
#define PG_CASE(B) case B: set16_simple<B, 0>(lane_words, set_lds, m); set16_simple<B, 1>(lane_words, set_lds, m); break;
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16)
#undef PG_CASE
    default: break;
  }
  return __builtin_bitreverse32(m);      // value j -> bit j
}

template <typename WP>
__device__ __forceinline__ void simple_agg_dispatch(int b, WP lane_words, uint32_t m, bool need_sum, bool need_minmax,
                                                    uint32_t& psum, unsigned long long& wsum, uint32_t& umin, uint32_t& umax) {
  switch (b) {
#define PG_CASE(B) case B: agg16_private<B, 0>(lane_words, m, need_sum, need_minmax, psum, wsum, umin, umax); \
                           agg16_private<B, 1>(lane_words, m, need_sum, need_minmax, psum, wsum, umin, umax); break;
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20)
#undef PG_CASE
    default: break;
  }
}

// `block_index` of `num_blocks`: the workgroup's place among those that work on this query (the whole grid, or one item's share of a
// batch launch: scan_lean_batch_kernel).  P: ScanParams, or its constant-address-space form there.
// kSet: the one leaf is a dictId set over a column of at most 16 bits (scan_simple_set_kernel: a kernel of its own, so that the range form's
// code and registers -- the headline's -- stay what they were); `set_lds`: kSetLdsWords words of the workgroup's LDS.
template <bool kSet = false, typename P>
__device__ __forceinline__ void scan_simple_body(const P& p, uint32_t block_index, uint32_t num_blocks, BlockPartial* red, uint32_t* fold_flag_ptr, uint32_t* set_lds = nullptr) {
  if constexpr (kSet) stage_filter_sets(p, set_lds);
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const long long total_waves = (long long)num_blocks * waves_per_block;
  const long long num_tiles = ((long long)p.num_docs + 2047) / 2048;
  const bool has_filter = p.num_nodes == 1;
  const bool has_agg = p.num_agg_cols == 1;
  const auto& L = p.nodes[0];
  const auto& ac = p.agg_cols[0];

  unsigned long long count = 0, sum = 0;
  uint32_t umin = 0xFFFFFFFFu, umax = 0u;
  // (Two tiles per iteration -- both tiles' chunks of a column loaded before either is decoded -- was measured on this kernel and lost:
  // C2b at 10 % 0.580 -> 0.601 ms, 3 % 0.543 -> 0.569, profiles/r3/ab_scan_simple_pair_*.jsonl.  A wave's own second request buys nothing
  // the fifth wave has not already bought, and the longer decode blocks cost more than they hide.)
  for (long long tile = (long long)block_index * waves_per_block + wave_in_block; tile < num_tiles; tile += total_waves) {
    uint32_t m = 0xFFFFFFFFu;
    if (has_filter) {
      const GlobalWords words = global_words(L.fwd + tile * (256ll * L.bits)) + lane * L.bits;
      if constexpr (kSet) m = simple_set_dispatch(L.bits, words, set_lds);
      else m = L.lo == 0 ? simple_range_dispatch<true>(L.bits, words, 0u, L.span) : simple_range_dispatch<false>(L.bits, words, (uint32_t)L.lo, L.span);
      if (L.exclusive) m = ~m;
    }
    const long long rem = (long long)p.num_docs - (tile * 2048 + lane * 32);        // docs past numDocs (last tile only)
    m &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u));
    count += (unsigned)__builtin_popcount(m);
    const unsigned long long lanes_with_matches = __builtin_amdgcn_ballot_w64(m != 0u);
    if (!has_agg || lanes_with_matches == 0ull) continue;
    const GlobalWords words = global_words(ac.fwd + tile * (256ll * ac.bits)) + lane * ac.bits;
    uint32_t psum = 0, tmin = 0xFFFFFFFFu, tmax = 0u;
    unsigned long long wsum = 0;
    // (scan_private_body's rules: few lanes with a match -> walk the matches; a lane without a match loads nothing)
    if (__builtin_popcountll(lanes_with_matches) <= p.sparse_lanes) agg_sparse_private(words, words - lane * ac.bits, ac.bits, m, ac.need_sum != 0, ac.need_minmax != 0, wsum, tmin, tmax);
    else if (p.lane_skip == 0 || m != 0u) simple_agg_dispatch(ac.bits, words, m, ac.need_sum != 0, ac.need_minmax != 0, psum, wsum, tmin, tmax);
    sum += wsum + psum;
    umin = tmin < umin ? tmin : umin;
    umax = tmax > umax ? tmax : umax;
  }

  BlockPartial mine;
  partial_identity(mine);
  mine.count = (unsigned long long)wave_sum_i64((long long)count);
  mine.sum[0] = wave_sum_i64((long long)sum);
  // unsigned keys below 2^31 -> the int32 keys of BlockPartial; lanes that matched nothing keep the identities
  mine.kmin[0] = wave_min_i32(umin == 0xFFFFFFFFu ? 0x7FFFFFFF : (int32_t)umin);
  mine.kmax[0] = wave_max_i32(count == 0ull ? (int32_t)0x80000000 : (int32_t)umax);
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  publish_block_partial(p, red, waves_per_block, fold_flag_ptr, block_index, num_blocks);
}

// (scan_simple_kernel itself -- the body over the whole grid -- is defined in pg_unit_scan_simple.hip: this header is also included by the batch kernel's unit)

}  // namespace pg
