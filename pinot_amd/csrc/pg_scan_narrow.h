// scan_narrow_kernel: filters over NARROW dictionary columns (at most 8 bits per dictId), COUNT(*) and / or the docId bitmap.
//
// A 2048-doc tile of a b-bit column is 256 b bytes: 1 KB at 4 bits.  scan_private_kernel evaluates leaf after leaf, tile after tile,
// with ~130 registers per lane: four waves per SIMD, one tile's worth of one column in flight per wave -- 4 MB on the whole chip,
// against the ~16 MB a 2 us HBM round trip needs at 8 TB/s.  Measured: 4 / 6 / 8-bit columns scan at 1.8-3.0 TB/s there, and giving
// that kernel more state (four tiles of the narrow leaves up front) only pushed it into scratch (profiles/r2/README.md).
// This kernel is the other answer: nothing but the filter, FOUR tiles per wave and iteration -- for every leaf the lane's words of all
// four tiles are loaded before the first is decoded (4 b <= 32 registers), the filter program runs on four masks at a time -- and no
// aggregation state beyond a count.  Same lane-private layout as the other kernels (lane i owns docs 32 i .. 32 i + 31 of a tile, its
// mask is dword 64 tile + i of the doc-order bitmap), so the bitmap it writes is the one every other kernel reads.
// Leaves: dictId ranges (PredicateEvaluator lowering of EQ / NOT_EQ / RANGE: SVScanDocIdIterator's matcher, SVScanDocIdIterator.java:
// 108-145, over FixedBitSVForwardIndexReaderV2's stream), dictId sets (IN / NOT IN: at most eight words, kept in LDS -- round 6b), match-all / match-none; any AND / OR / NOT tree whose evaluation needs at
// most kNarrowStack masks.
#pragma once
#include "pg_kernels.h"

namespace pg {

template <int B>
__device__ __forceinline__ void narrow_leaf_quad(const uint8_t* fwd, const long long (&tiles)[kNarrowTiles], int lane, uint32_t lo, uint32_t span,
                                                 uint32_t (&m)[kNarrowTiles]) {
  uint32_t w[kNarrowTiles][B];
#pragma unroll
  for (int t = 0; t < kNarrowTiles; ++t) {
    const uint32_t* words = reinterpret_cast<const uint32_t*>(fwd + tiles[t] * (256ll * B)) + lane * B;
#pragma unroll
    for (int i = 0; i < B; ++i) w[t][i] = words[i];
  }
#pragma unroll
  for (int t = 0; t < kNarrowTiles; ++t) {
    uint32_t mm = 0;
    range16_private<B, 0, false>(w[t], lo, span, mm);
    range16_private<B, 1, false>(w[t], lo, span, mm);
    m[t] = __builtin_bitreverse32(mm);          // value j -> bit j
  }
}

// A dictId-SET leaf (InPredicateEvaluator / NotInPredicateEvaluator) over a narrow column (round 6b): the set of a column of at most 8 bits is
// at most eight words -- leaf ordinal o keeps them in words [8 o, 8 o + 8) of the workgroup's LDS (stage_narrow_sets; zero beyond the set's own
// words).  Up to five bits the whole set is ONE word held in a register: decode, bit-field extract, shift-or per doc -- a range compare's three
// instructions; six to eight bits look the word up in LDS (eight words, one bank each).
template <int B, int H>
__device__ __forceinline__ void set16_narrow(const uint32_t (&w)[B], const uint32_t* set8, uint32_t& mm) {
  uint32_t v[16];
  decode16_private<B, H>(w, v);
  if constexpr (B <= 5) {
    const uint32_t mask = set8[0];
#pragma unroll
    for (int j = 0; j < 16; ++j) mm = (mm << 1) | __builtin_amdgcn_ubfe(mask, v[j], 1);
  } else {
    uint32_t x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = set8[v[j] >> 5];
#pragma unroll
    for (int j = 0; j < 16; ++j) mm = (mm << 1) | __builtin_amdgcn_ubfe(x[j], v[j] & 31u, 1);
  }
}
template <int B>
__device__ __forceinline__ void narrow_set_quad(const uint8_t* fwd, const long long (&tiles)[kNarrowTiles], int lane, const uint32_t* set8, uint32_t (&m)[kNarrowTiles]) {
  uint32_t w[kNarrowTiles][B];
#pragma unroll
  for (int t = 0; t < kNarrowTiles; ++t) {
    const uint32_t* words = reinterpret_cast<const uint32_t*>(fwd + tiles[t] * (256ll * B)) + lane * B;
#pragma unroll
    for (int i = 0; i < B; ++i) w[t][i] = words[i];
  }
#pragma unroll
  for (int t = 0; t < kNarrowTiles; ++t) {
    uint32_t mm = 0;
    set16_narrow<B, 0>(w[t], set8, mm);
    set16_narrow<B, 1>(w[t], set8, mm);
    m[t] = __builtin_bitreverse32(mm);          // value j -> bit j
  }
}
constexpr int kNarrowSetWords = 8 * kMaxLeaves;
template <typename P>
__device__ __forceinline__ void stage_narrow_sets(const P& p, uint32_t* sets) {
  if (p.set_leaves_in_lds == 0) return;            // (uniform: no set leaf in this filter)
  int ordinal = 0;
  for (int n = 0; n < p.num_nodes; ++n) {
    const auto& nd = p.nodes[n];
    if (nd.op != PG_FILTER_LEAF) continue;
    if (nd.kind == kLeafDictSet && ordinal < kMaxLeaves && threadIdx.x < 8) sets[8 * ordinal + threadIdx.x] = (int)threadIdx.x < (nd.set_bytes >> 2) ? nd.set_words[threadIdx.x] : 0u;
    ++ordinal;
  }
  __syncthreads();
}

struct NarrowStack {                              // kNarrowStack entries of four masks; selects instead of indexing keep it in registers
  uint32_t v[kNarrowStack][kNarrowTiles];
  int sp;
  __device__ __forceinline__ void push(const uint32_t (&x)[kNarrowTiles]) {
#pragma unroll
    for (int i = 0; i < kNarrowStack; ++i)
#pragma unroll
      for (int t = 0; t < kNarrowTiles; ++t) v[i][t] = (i == sp) ? x[t] : v[i][t];
    ++sp;
  }
  __device__ __forceinline__ void pop(uint32_t (&r)[kNarrowTiles]) {
    --sp;
#pragma unroll
    for (int t = 0; t < kNarrowTiles; ++t) r[t] = 0u;
#pragma unroll
    for (int i = 0; i < kNarrowStack; ++i)
#pragma unroll
      for (int t = 0; t < kNarrowTiles; ++t) r[t] = (i == sp) ? v[i][t] : r[t];
  }
};

template <typename P>
__device__ __forceinline__ void scan_narrow_body(const P& p, uint32_t block_index, uint32_t num_blocks, BlockPartial* red, uint32_t* fold_flag_ptr, uint32_t* sets) {
  stage_narrow_sets(p, sets);
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const long long total_waves = (long long)num_blocks * waves_per_block;
  const long long num_tiles = ((long long)p.num_docs + 2047) / 2048;
  const long long num_quads = (num_tiles + kNarrowTiles - 1) / kNarrowTiles;
  unsigned long long count = 0;
  for (long long quad = (long long)block_index * waves_per_block + wave_in_block; quad < num_quads; quad += total_waves) {
    long long tiles[kNarrowTiles];
#pragma unroll
    for (int t = 0; t < kNarrowTiles; ++t) tiles[t] = quad * kNarrowTiles + t < num_tiles ? quad * kNarrowTiles + t : quad * kNarrowTiles;   // past the end: a valid tile, masked below
    NarrowStack st;
#pragma unroll
    for (int i = 0; i < kNarrowStack; ++i)
#pragma unroll
      for (int t = 0; t < kNarrowTiles; ++t) st.v[i][t] = 0u;
    st.sp = 0;
    int leaf_ordinal = 0;
    for (int n = 0; n < p.num_nodes; ++n) {
      const auto& nd = p.nodes[n];
      uint32_t top[kNarrowTiles];
      if (nd.op == PG_FILTER_LEAF) {
#pragma unroll
        for (int t = 0; t < kNarrowTiles; ++t) top[t] = nd.kind == kLeafMatchAll ? 0xFFFFFFFFu : 0u;
        if (nd.kind == kLeafDictRange) {
          const uint32_t lo = (uint32_t)nd.lo, span = nd.span;
          switch (nd.bits) {
#define PG_CASE(B) case B: narrow_leaf_quad<B>(nd.fwd, tiles, lane, lo, span, top); break;
            PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8)
#undef PG_CASE
            default: break;
          }
        } else if (nd.kind == kLeafDictSet) {
          const uint32_t* set8 = sets + 8 * leaf_ordinal;
          switch (nd.bits) {
#define PG_CASE(B) case B: narrow_set_quad<B>(nd.fwd, tiles, lane, set8, top); break;
            PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8)
#undef PG_CASE
            default: break;
          }
        }
        if (nd.exclusive) {
#pragma unroll
          for (int t = 0; t < kNarrowTiles; ++t) top[t] = ~top[t];
        }
        if (p.leaf_out_enabled != 0) {
          // the transducer pass of numEntriesScannedInFilter wants this leaf's own match bits (ScanParams.leaf_out, pg_filter_fsm.h)
          uint32_t* const leaf_bits = p.leaf_out[leaf_ordinal];
          if (leaf_bits != nullptr) {
#pragma unroll
            for (int t = 0; t < kNarrowTiles; ++t) if (quad * kNarrowTiles + t < num_tiles) leaf_bits[(quad * kNarrowTiles + t) * 64 + lane] = top[t];
          }
        }
        ++leaf_ordinal;
      } else if (nd.op == PG_FILTER_NOT) {
        st.pop(top);
#pragma unroll
        for (int t = 0; t < kNarrowTiles; ++t) top[t] = ~top[t];
      } else {
        st.pop(top);
        for (int c = 1; c < nd.num_children; ++c) {
          uint32_t o[kNarrowTiles];
          st.pop(o);
#pragma unroll
          for (int t = 0; t < kNarrowTiles; ++t) top[t] = nd.op == PG_FILTER_AND ? (top[t] & o[t]) : (top[t] | o[t]);
        }
      }
      st.push(top);
    }
    uint32_t m[kNarrowTiles];
    if (p.num_nodes == 0) {
#pragma unroll
      for (int t = 0; t < kNarrowTiles; ++t) m[t] = 0xFFFFFFFFu;
    } else {
      st.pop(m);
    }
#pragma unroll
    for (int t = 0; t < kNarrowTiles; ++t) {
      const long long tile = quad * kNarrowTiles + t;
      const long long rem = (long long)p.num_docs - (tile * 2048 + lane * 32);          // docs past numDocs, and tiles past the last one
      const uint32_t mt = m[t] & (rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u)));
      if (p.out_bitmap && tile < num_tiles) reinterpret_cast<uint32_t*>(p.out_bitmap)[tile * 64 + lane] = mt;
      count += (unsigned)__builtin_popcount(mt);
    }
  }
  BlockPartial mine;
  partial_identity(mine);
  mine.count = (unsigned long long)wave_sum_i64((long long)count);
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  publish_block_partial(p, red, waves_per_block, fold_flag_ptr, block_index, num_blocks);
}

static __global__ __launch_bounds__(kBlockThreads) void scan_narrow_kernel(const ScanParams p) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  __shared__ uint32_t sets[kNarrowSetWords];
  scan_narrow_body(p, blockIdx.x, gridDim.x, red, &fold_flag, sets);
}

// A single dictId-range leaf (WHERE dim = x, the commonest narrow filter): no mask stack, so EIGHT tiles fit per wave and iteration
// (8 b <= 64 registers of loads in flight) in a kernel of its own register budget.
template <int B, typename P, typename LN>
__device__ __forceinline__ unsigned narrow_single_octet(const P& p, const LN& L, long long first_tile, long long num_tiles, int lane) {
  uint32_t w[kNarrowSingleTiles][B];
#pragma unroll
  for (int t = 0; t < kNarrowSingleTiles; ++t) {
    const long long tile = first_tile + t < num_tiles ? first_tile + t : first_tile;       // past the end: a valid tile, masked below
    const uint32_t* words = reinterpret_cast<const uint32_t*>(L.fwd + tile * (256ll * B)) + lane * B;
#pragma unroll
    for (int i = 0; i < B; ++i) w[t][i] = words[i];
  }
  unsigned count = 0;
  const uint32_t lo = (uint32_t)L.lo, span = L.span, flip = L.exclusive ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int t = 0; t < kNarrowSingleTiles; ++t) {
    uint32_t mm = 0;
    range16_private<B, 0, false>(w[t], lo, span, mm);
    range16_private<B, 1, false>(w[t], lo, span, mm);
    const long long tile = first_tile + t;
    const long long rem = (long long)p.num_docs - (tile * 2048 + lane * 32);
    const uint32_t mt = (__builtin_bitreverse32(mm) ^ flip) & (rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u)));
    if (p.out_bitmap && tile < num_tiles) reinterpret_cast<uint32_t*>(p.out_bitmap)[tile * 64 + lane] = mt;
    count += (unsigned)__builtin_popcount(mt);
  }
  return count;
}

template <typename P>
__device__ __forceinline__ void scan_narrow_single_body(const P& p, uint32_t block_index, uint32_t num_blocks, BlockPartial* red, uint32_t* fold_flag_ptr) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const long long total_waves = (long long)num_blocks * waves_per_block;
  const long long num_tiles = ((long long)p.num_docs + 2047) / 2048;
  const long long num_octets = (num_tiles + kNarrowSingleTiles - 1) / kNarrowSingleTiles;
  const auto& L = p.nodes[0];
  unsigned long long count = 0;
  for (long long o = (long long)block_index * waves_per_block + wave_in_block; o < num_octets; o += total_waves) {
    switch (L.bits) {
#define PG_CASE(B) case B: count += narrow_single_octet<B>(p, L, o * kNarrowSingleTiles, num_tiles, lane); break;
      PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8)
#undef PG_CASE
      default: break;
    }
  }
  BlockPartial mine;
  partial_identity(mine);
  mine.count = (unsigned long long)wave_sum_i64((long long)count);
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  publish_block_partial(p, red, waves_per_block, fold_flag_ptr, block_index, num_blocks);
}

static __global__ __launch_bounds__(kBlockThreads) void scan_narrow_single_kernel(const ScanParams p) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  scan_narrow_single_body(p, blockIdx.x, gridDim.x, red, &fold_flag);
}

// pg_execute_batch's shared launch for items of the two kernels' shape (COUNT(*) under a filter over columns of at most 8 bits on a server's
// many small segments): workgroups [block_first[i], block_first[i + 1]) work on items[i] (see scan_private_batch_kernel).
// kSingle: every item is one dictionary-range leaf (scan_narrow_single_kernel's shape), else the general narrow evaluator.
template <bool kSingle>
__global__ __launch_bounds__(kBlockThreads) void scan_narrow_batch_kernel(const BatchParams bp) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  __shared__ uint32_t sets[kNarrowSetWords];
  int lo = 0, hi = bp.num_items - 1;                // the last item whose first workgroup is at or before this one
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (bp.block_first[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const uint32_t first = bp.block_first[lo];
  typedef const __attribute__((address_space(4))) ScanParams ConstantScanParams;
  const ConstantScanParams& item = *(ConstantScanParams*)(bp.items + lo);
  if constexpr (kSingle) scan_narrow_single_body(item, blockIdx.x - first, bp.block_first[lo + 1] - first, red, &fold_flag);      // (a single SET leaf takes the general body: the octet form with the lookups' registers on top drops to three waves)
  else scan_narrow_body(item, blockIdx.x - first, bp.block_first[lo + 1] - first, red, &fold_flag, sets);
}

}  // namespace pg
