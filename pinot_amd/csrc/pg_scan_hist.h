// scan_hist_kernel: SUM over a dictionary column WITHOUT reading a single dictionary value per row.
//
// What it replaces: Dictionary.readIntValues (sspi/index/reader/Dictionary.java:207-211: one dictionary lookup per matching doc)
// feeding SumAggregationFunction.aggregate (core/query/aggregation/function/SumAggregationFunction.java:69-157).  On MI355X a
// lookup in a 400 KB dictionary is an L2 gather (2.87e11 /s measured: each one moves a 128-byte line into the L1 for 4 useful
// bytes), and a value plane of a dictionary whose values span the int range is 32 bits per doc -- twice the dictId stream.
// Both stay below half the HBM roofline on `SUM(v) WHERE f < t`.  But
//
//        SUM(v) over the matching docs  =  sum over dictIds d of  matches[d] * dictionary[d],
//
// so the kernel only COUNTS the matching docs per dictId, in a histogram in the CU's LDS (160 KB), and multiplies by the
// dictionary once per workgroup at the end: HBM traffic is exactly the dictId streams (the algorithmic bytes), nothing extra is
// resident in HBM, and the values are touched C times per workgroup instead of once per row.
//
// Counters.  One workgroup per CU keeps the whole histogram:
//     CW = 32   C <= 38 912    cannot overflow (a workgroup sees < 2^31 docs)
//     CW = 16   C <= 77 824    two counters per dword
//     CW =  8   C <= 155 648   four counters per dword (C = 100 000, BASELINE's `v`, lives here: 100 KB)
// Narrow counters can overflow; the result is nevertheless EXACT OR NOT USED, in two tiers:
//   kGuard = false (first choice): plain non-returning ds_add_u32, nothing per doc beyond the add.  A counter that wraps loses
//     2^CW and hands at most 1 to its neighbour, so the sum of all counters falls short of the number of matching docs (known
//     exactly from the mask popcounts) by a positive amount for ANY pattern of wraps: `sum of counters == matches` proves that
//     no counter wrapped.  The workgroup folds that checksum while it multiplies by the dictionary anyway; a mismatch makes
//     the engine rerun the query in the guarded tier (and remember it for the column).  Measured on C2b over an irregular
//     100 000-value dictionary: 0.595 ms per 1 B rows = 71 % of 8 TB/s on the algorithmic bytes.
//   kGuard = true: the top bit of a counter is a guard.  Every add returns the old value; a wave that sees a counter at or
//     above G = 2^(CW-1) sweeps its tile and CLAIMS the guard bit with an atomic AND -- whoever clears it owns G matches of that
//     dictId and adds G * dictionary[d] to a private spill sum, so hot dictIds (skewed data) are counted exactly.  A counter can
//     only leave its field after G further adds that all return values >= 1.5 G; any such value raises the alarm flag and the
//     engine answers through the gather / value-plane path instead.  0.795 ms on the same query: the returns cost 0.2 ms.
//
// Everything else (lane-private ownership, filter program, decode at compile-time bit positions) is scan_private_kernel's.
#pragma once
#include "pg_kernels.h"

namespace pg {

template <int CW> struct HistField;
template <> struct HistField<32> {
  static __device__ __forceinline__ uint32_t word(uint32_t d) { return d; }
  static __device__ __forceinline__ uint32_t shift(uint32_t) { return 0u; }
};
template <> struct HistField<16> {
  static __device__ __forceinline__ uint32_t word(uint32_t d) { return d >> 1; }
  static __device__ __forceinline__ uint32_t shift(uint32_t d) { return (d & 1u) << 4; }
};
template <> struct HistField<8> {
  static __device__ __forceinline__ uint32_t word(uint32_t d) { return d >> 2; }
  static __device__ __forceinline__ uint32_t shift(uint32_t d) { return (d & 3u) << 3; }
};

// Sixteen docs (half H) of the lane's chunk of the summed column: one LDS add of the match bit per doc.
// mx: running maximum of the counter values the adds returned (CW < 32).
template <int B, int H, int CW, bool kGuard>
__device__ __forceinline__ void hist16_private(const uint32_t* __restrict__ lane_words, uint32_t m, uint32_t* hist, uint32_t& mx,
                                               bool need_minmax, uint32_t& umin, uint32_t& umax) {
  uint32_t v[16];
  decode16_private<B, H>(lane_words, v);
  if constexpr (!kGuard) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      __hip_atomic_fetch_add(hist + HistField<CW>::word(v[j]), __builtin_amdgcn_ubfe(m, 16 * H + j, 1) << HistField<CW>::shift(v[j]), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    uint32_t old[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      old[j] = __hip_atomic_fetch_add(hist + HistField<CW>::word(v[j]), __builtin_amdgcn_ubfe(m, 16 * H + j, 1) << HistField<CW>::shift(v[j]), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t f = __builtin_amdgcn_ubfe(old[j], HistField<CW>::shift(v[j]), CW);
      mx = f > mx ? f : mx;
    }
  }
  if (need_minmax) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t all = (uint32_t)__builtin_amdgcn_sbfe((int)m, 16 * H + j, 1);     // ~0 when the doc matches
      const uint32_t hi = v[j] & all, lo = v[j] | ~all;
      umax = hi > umax ? hi : umax;
      umin = lo < umin ? lo : umin;
    }
  }
}

// C2a shape -- SUM(col) WHERE col's dictId in [lo, lo + span): one decode feeds the range compare, the mask and the counter add
// (the compare's VCC selects 0 / 1 as the value added).  Full tiles only: the caller sends the last, partial tile down the general path.
template <int B, int H, int CW>
__device__ __forceinline__ void hist_range16_private(const uint32_t* __restrict__ lane_words, uint32_t lo, uint32_t span, uint32_t& m, uint32_t* hist) {
  uint32_t v[16];
  decode16_private<B, H>(lane_words, v);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t x = v[j] - lo;
    uint32_t bit;
    asm("v_cmp_gt_u32 vcc, %3, %2\n\tv_cndmask_b32_e64 %1, 0, 1, vcc\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m), "=&v"(bit) : "v"(x), "s"(span) : "vcc");
    __hip_atomic_fetch_add(hist + HistField<CW>::word(v[j]), bit << HistField<CW>::shift(v[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

template <int CW>
__device__ __forceinline__ uint32_t hist_range_dispatch(int b, const uint32_t* lane_words, uint32_t lo, uint32_t span, uint32_t* hist) {
  uint32_t m = 0;
  switch (b) {
#define PG_CASE(B) case B: hist_range16_private<B, 0, CW>(lane_words, lo, span, m, hist); hist_range16_private<B, 1, CW>(lane_words, lo, span, m, hist); break;
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18)
#undef PG_CASE
    default: break;
  }
  return __builtin_bitreverse32(m);      // value j -> bit j
}

// Histograms hold at most 155 648 counters: 18-bit dictIds.
template <int CW, bool kGuard>
__device__ __forceinline__ void hist_private_dispatch(int b, const uint32_t* lane_words, uint32_t m, uint32_t* hist, uint32_t& mx, bool need_minmax,
                                                      uint32_t& umin, uint32_t& umax) {
  switch (b) {
#define PG_CASE(B) case B: hist16_private<B, 0, CW, kGuard>(lane_words, m, hist, mx, need_minmax, umin, umax); \
                           hist16_private<B, 1, CW, kGuard>(lane_words, m, hist, mx, need_minmax, umin, umax); break;
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18)
#undef PG_CASE
    default: break;
  }
}

// Doc j of the lane's chunk, any width, bit position computed at run time (the rare sweep only).
__device__ __forceinline__ uint32_t decode_private_generic(const uint32_t* lane_words, int b, int j) {
  const uint32_t bit = (uint32_t)j * (uint32_t)b;
  const uint32_t w = bit >> 5, o = bit & 31u;
  const unsigned long long x = ((unsigned long long)__builtin_bswap32(lane_words[w]) << 32) | (unsigned long long)__builtin_bswap32(lane_words[w + 1]);   // w + 1 may be the next lane's (or the padding's) first dword
  return (uint32_t)(x >> (64u - o - (uint32_t)b)) & ((1u << b) - 1u);
}

// A wave saw a guarded counter: claim the guard bit of every counter its matching docs of this tile point at.
template <int CW>
__device__ __noinline__ void hist_sweep(const uint32_t* lane_words, int b, uint32_t m, uint32_t* hist, const int32_t* __restrict__ dict, long long& spill,
                                        uint32_t& alarm) {
  constexpr uint32_t G = 1u << (CW - 1), FM = (CW == 32) ? 0xFFFFFFFFu : ((1u << CW) - 1u);
#pragma unroll 1
  for (int j = 0; j < 32; ++j) {
    if (!((m >> j) & 1u)) continue;
    const uint32_t d = decode_private_generic(lane_words, b, j);
    const uint32_t w = HistField<CW>::word(d), s = HistField<CW>::shift(d);
    const uint32_t cur = (__hip_atomic_load(hist + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> s) & FM;
    if (cur < G) continue;
    const uint32_t before = (__hip_atomic_fetch_and(hist + w, ~(G << s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> s) & FM;
    if (before & G) spill += (long long)G * (long long)dict[d];      // this lane cleared the bit: it owns G matches of dictId d
    if (before >= G + G / 2) alarm = 1u;
  }
}

// `block_index` of `num_blocks`: the workgroup's place among those that work on this parameter block -- the whole grid (scan_hist_kernel),
// or one item's share of a batch launch (scan_hist_batch_kernel).  P: ScanParams, or its constant-address-space form in device memory.
template <int CW, bool kGuard, typename P>
__device__ __forceinline__ void scan_hist_body(const P& p, uint32_t block_index, uint32_t num_blocks, uint32_t* hist) {
  static_assert(!(kGuard && CW == 32), "32-bit counters need no guard");
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const long long total_waves = (long long)num_blocks * waves_per_block;
  const long long num_tiles = ((long long)p.num_docs + 2047) / 2048;
  const int C = p.hist_bins;
  constexpr int kPerWord = 32 / CW;
  const int hist_words = (C + kPerWord - 1) / kPerWord;
  for (int w = threadIdx.x; w < hist_words; w += blockDim.x) hist[w] = 0u;
  // the filter's dictId sets behind the counters (the engine made room: set_leaves_in_lds = 1 + the area's byte offset), staged once per workgroup
  uint32_t* set_lds = nullptr;
  if (p.set_leaves_in_lds > 1) { set_lds = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(hist) + (p.set_leaves_in_lds - 1)); stage_filter_sets(p, set_lds); }
  __syncthreads();

  unsigned long long count = 0;
  uint32_t umin = 0xFFFFFFFFu, umax = 0u;
  long long spill = 0;
  uint32_t alarm = 0;
  // the one aggregated column (the engine sends other shapes to scan_private_kernel): summed through the histogram, MIN / MAX on
  // its dictIds in registers
  const auto& ac = p.agg_cols[0];

  // one inclusive-range leaf on the summed column itself (C2a), plain counters, no MIN / MAX: one decode per tile instead of two
  const bool fused = !kGuard && p.num_nodes == 1 && p.nodes[0].kind == kLeafDictRange && p.nodes[0].exclusive == 0 && p.nodes[0].fwd == ac.fwd &&
                     p.nodes[0].bits == ac.bits && ac.need_minmax == 0;
  uint32_t entries = 0u;
  const bool listed = p.tile_list != nullptr;              // index-driven filters: only the tiles index_and_kernel listed hold a match
  const long long tile_limit = listed ? (long long)*p.tile_count : num_tiles;
  for (long long tile_it = (long long)block_index * waves_per_block + wave_in_block; tile_it < tile_limit; tile_it += total_waves) {
    const long long tile = listed ? (long long)p.tile_list[tile_it] : tile_it;
    if constexpr (!kGuard) {
      if (fused && (tile + 1) * 2048 <= (long long)p.num_docs) {
        const uint32_t* fwords = reinterpret_cast<const uint32_t*>(ac.fwd + tile * (256ll * ac.bits)) + lane * ac.bits;
        count += (unsigned)__builtin_popcount(hist_range_dispatch<CW>(ac.bits, fwords, (uint32_t)p.nodes[0].lo, p.nodes[0].span, hist));
        continue;
      }
    }
    uint32_t m = eval_filter_private(p, tile, lane, entries, nullptr, set_lds);
    const long long rem = (long long)p.num_docs - (tile * 2048 + lane * 32);
    m &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u));
    count += (unsigned)__builtin_popcount(m);
    if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) continue;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(ac.fwd + tile * (256ll * ac.bits)) + lane * ac.bits;
    uint32_t mx = 0;
    if (!kGuard && __builtin_popcountll(__builtin_amdgcn_ballot_w64(m != 0u)) <= p.sparse_lanes) {
      // few lanes of the tile hold a match: walk the matches, one 8-byte load and one counter add per matching doc (agg_sparse_private)
      const uint32_t field_mask = (1u << ac.bits) - 1u;
      uint32_t rest = m;
      while (__builtin_amdgcn_ballot_w64(rest != 0u) != 0ull) {
        Dwords2 d[4];
        uint32_t sh[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ok[k] = rest != 0u;
          const uint32_t j = ok[k] ? (uint32_t)__builtin_ctz(rest) : 0u;
          rest &= rest - 1u;
          const uint32_t bit = j * (uint32_t)ac.bits;
          sh[k] = 64u - (bit & 31u) - (uint32_t)ac.bits;
          d[k] = *reinterpret_cast<const Dwords2*>(ok[k] ? words + (bit >> 5) : words - lane * ac.bits);          // unconditional: see agg_sparse_private
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned long long x = ((unsigned long long)__builtin_bswap32(d[k].x) << 32) | (unsigned long long)__builtin_bswap32(d[k].y);
          const uint32_t v = (uint32_t)(x >> sh[k]) & field_mask;
          if (ok[k]) {
            __hip_atomic_fetch_add(hist + HistField<CW>::word(v), 1u << HistField<CW>::shift(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ac.need_minmax) { umax = v > umax ? v : umax; umin = v < umin ? v : umin; }
          }
        }
      }
      continue;
    }
    // (a lane without a match loads nothing: see scan_private_kernel)
    if (p.lane_skip == 0 || m != 0u) hist_private_dispatch<CW, kGuard>(ac.bits, words, m, hist, mx, ac.need_minmax != 0, umin, umax);
    if constexpr (kGuard) {
      constexpr uint32_t G = 1u << (CW - 1);
      alarm |= mx >= G + G / 2 ? 1u : 0u;
      if (__builtin_amdgcn_ballot_w64(mx >= G) != 0ull) hist_sweep<CW>(words, ac.bits, m, hist, ac.dict, spill, alarm);
    }
  }

  // SUM = sum_d matches[d] * dictionary[d]: every thread folds its share of the counters (the guard bit is part of the count)
  __syncthreads();
  long long hsum = spill;
  unsigned long long csum = 0;        // sum of the counters as read: equals the matches iff no counter wrapped (kGuard = false)
  {
    const int32_t* __restrict__ dict = ac.dict;
    constexpr uint32_t FM = (CW == 32) ? 0xFFFFFFFFu : ((1u << CW) - 1u);
    for (int w = threadIdx.x; w < hist_words; w += blockDim.x) {
      const uint32_t h = hist[w];
      if (h == 0u) continue;
#pragma unroll
      for (int k = 0; k < kPerWord; ++k) {
        const int d = w * kPerWord + k;
        const uint32_t c = (h >> (k * CW)) & FM;
        csum += c;
        if (c != 0u && d < C) hsum += (long long)c * (long long)dict[d];
      }
    }
  }

  flush_filter_entries(p, entries);
  BlockPartial mine;
  partial_identity(mine);
  mine.count = (unsigned long long)wave_sum_i64((long long)count);
  mine.entries = (unsigned long long)wave_sum_i64((long long)entries);
  mine.flags = __builtin_amdgcn_ballot_w64(alarm != 0u) != 0ull ? kPartialHistAlarm : 0ull;
  mine.sum[0] = wave_sum_i64(hsum);
  if constexpr (!kGuard && CW < 32) mine.sum[1] = wave_sum_i64((long long)csum);      // the host compares it with `count`
  mine.kmin[0] = wave_min_i32(umin == 0xFFFFFFFFu ? 0x7FFFFFFF : (int32_t)umin);
  mine.kmax[0] = wave_max_i32(count == 0ull ? (int32_t)0x80000000 : (int32_t)umax);
  __syncthreads();       // every thread is done with the counters: the start of LDS becomes the reduction scratch
  BlockPartial* red = reinterpret_cast<BlockPartial*>(hist);
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  publish_block_partial(p, red, waves_per_block, reinterpret_cast<uint32_t*>(red + waves_per_block), block_index, num_blocks);      // (the engine sizes the LDS for it)
}

template <int CW, bool kGuard>
__global__ __launch_bounds__(kHistBlockThreads) void scan_hist_kernel(const ScanParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hist[];      // the only LDS object: counter addresses need no base add
  scan_hist_body<CW, kGuard>(p, blockIdx.x, gridDim.x, hist);
}

// pg_execute_batch's shared launch for items of scan_hist_kernel's shape (SUM over a dictionary without structure -- the normal case of
// a real Pinot dictionary -- on each of a server's many small segments: BaseCombineOperator.java:85-142).  Workgroups
// [block_first[i], block_first[i + 1]) work on items[i]; each keeps the item's whole histogram in its LDS (the launch's dynamic LDS is
// the largest item's), multiplies by the item's dictionary at the end and arrives on the item's counters: every item folds and
// publishes its own pinned record.  Plain counters only: an item whose column is in the guarded tier runs scan_hist_kernel<CW, true>
// on its own, and an item whose checksum shows a wrapped counter is answered again by pg_execute (the engine checks every record).
template <int CW>
__global__ __launch_bounds__(kHistBlockThreads) void scan_hist_batch_kernel(const BatchParams bp) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
  int lo = 0, hi = bp.num_items - 1;                // the last item whose first workgroup is at or before this one
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (bp.block_first[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const uint32_t first = bp.block_first[lo];
  typedef const __attribute__((address_space(4))) ScanParams ConstantScanParams;      // (scalar loads of the item's fields: see scan_private_batch_kernel)
  const ConstantScanParams& item = *(ConstantScanParams*)(bp.items + lo);
  scan_hist_body<CW, false>(item, blockIdx.x - first, bp.block_first[lo + 1] - first, hist);
}

}  // namespace pg
