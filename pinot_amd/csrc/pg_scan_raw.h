// scan_raw_kernel: the scan pair of BASELINE.json configs[0] -- `SELECT COUNT(*) ... WHERE raw BETWEEN a AND b`, `SELECT SUM / MIN / MAX /
// AVG(raw)` -- over raw (no-dictionary, PASS_THROUGH) INT columns: at most ONE raw-range leaf and at most ONE aggregated raw INT column.
//
// What it replaces: DocIdSetOperator + SVScanDocIdIterator with an IntRawValueBasedRangePredicateEvaluator over a
// FixedByteChunkSVForwardIndexReader (core/operator/DocIdSetOperator.java:59-86, dociditerators/SVScanDocIdIterator.java:76-142,
// RangePredicateEvaluatorFactory.java:331-366, FixedByteChunkSVForwardIndexReader.java:53-61) and one AggregationFunction.aggregate.
//
// Why a kernel of its own (the sibling of scan_simple_kernel for raw columns): a lone 10 M-row segment is 4 883 tiles, and at the four
// waves per SIMD of scan_private_kernel / scan_private_typed_kernel the chip holds 4 096 -- the last 787 tiles wait for a second round
// of loads behind a launch (a whole memory round trip for a sixth of the data: profiles/r3/c1_probe_final.jsonl, 24.4 / 30.5 us for
// 40 MB).  This kernel carries no filter program, no slot arrays, no typed accumulators, and small grids run two workgroups per CU
// (pg_engine.hip lean_geometry).  COUNT / SUM / MIN / MAX do not care which lane sees which doc, so the tile is read
// fully coalesced -- instruction j of a wave covers the contiguous kilobyte of docs [256 j, 256 j + 256), 16 bytes per lane -- instead
// of lane-contiguously.  Bit exact with the kernels it replaces (same integer sums and keys).
#pragma once
#include "pg_kernels.h"

namespace pg {

#ifndef PG_RAW_WAVES
#define PG_RAW_WAVES 4                  // wavefronts per SIMD the register allocation must allow: 123 VGPRs, no scratch.  Five (96 VGPRs) spills 21 registers
                                        // -- the compiler keeps the next tile's loads of both columns in flight -- and measured slower at every size
                                        // (profiles/r4/c1_probe_raw_waves_4_vs_5.json: 10 M rows COUNT 18.4 -> 15.6 us, SUM 24.8 -> 18.9; 400 M rows SUM 346 -> 264 us)
#endif

// the 32 docs a lane sees of a tile, as four-doc pieces: piece j holds docs 256 j + 4 lane .. + 3 of the tile
typedef uint32_t raw_u32x4 __attribute__((ext_vector_type(4)));
struct RawTile { raw_u32x4 q[8]; };
__device__ __forceinline__ void load_raw_tile(const uint8_t* fwd, long long tile, int lane, RawTile& t) {
  const raw_u32x4* base = reinterpret_cast<const raw_u32x4*>(fwd + tile * 8192) + lane;
#pragma unroll
  for (int j = 0; j < 8; ++j) t.q[j] = __builtin_nontemporal_load(base + 64 * j);      // (raw buffers are padded to whole 2048-doc tiles)
}
__device__ __forceinline__ uint32_t raw_value(const RawTile& t, int j, int k) {
  const uint32_t be = k == 0 ? t.q[j].x : (k == 1 ? t.q[j].y : (k == 2 ? t.q[j].z : t.q[j].w));
  return __builtin_bswap32(be);
}

// `block_index` of `num_blocks`: the workgroup's place among those that work on this query (the whole grid, or one item's share of a
// batch launch: scan_lean_batch_kernel).  P: ScanParams, or its constant-address-space form there.
template <typename P>
__device__ __forceinline__ void scan_raw_body(const P& p, uint32_t block_index, uint32_t num_blocks, BlockPartial* red, uint32_t* fold_flag_ptr) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const long long total_waves = (long long)num_blocks * waves_per_block;
  const long long num_tiles = ((long long)p.num_docs + 2047) / 2048;
  const bool has_filter = p.num_nodes == 1;
  const bool has_agg = p.num_agg_cols == 1;
  const auto& L = p.nodes[0];
  const auto& ac = p.agg_cols[0];
  const bool same_column = has_filter && has_agg && L.fwd == ac.fwd;
  const uint32_t lo = (uint32_t)L.lo, span = L.span;
  const bool need_sum = has_agg && ac.need_sum != 0, need_minmax = has_agg && ac.need_minmax != 0;

  unsigned long long count = 0;
  long long sum = 0;
  int32_t vmin = 0x7FFFFFFF, vmax = (int32_t)0x80000000;
  for (long long tile = (long long)block_index * waves_per_block + wave_in_block; tile < num_tiles; tile += total_waves) {
    const long long rem = (long long)p.num_docs - tile * 2048;           // docs of this tile that exist (the last tile: fewer than 2048)
    RawTile t;
    uint32_t m = 0xFFFFFFFFu;                                            // bit 4 j + k: doc 256 j + 4 lane + k matches
    if (has_filter) {
      load_raw_tile(L.fwd, tile, lane, t);
      m = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) m |= (((raw_value(t, j, k) - lo) <= span) ? 1u : 0u) << (4 * j + k);
      }
      if (L.exclusive) m = ~m;
    }
    if (rem < 2048) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long long left = rem - (256 * j + 4 * lane);                // docs of piece j that exist
        const uint32_t keep = left >= 4 ? 0xFu : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
        m &= ~(0xFu << (4 * j)) | (keep << (4 * j));
      }
    }
    count += (unsigned)__builtin_popcount(m);
    if (!has_agg) continue;
    if (!same_column) {
      if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) continue;         // nothing matched in the whole tile: the aggregated column is not read
      load_raw_tile(ac.fwd, tile, lane, t);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool hit = ((m >> (4 * j + k)) & 1u) != 0u;
        const int32_t x = (int32_t)raw_value(t, j, k);
        if (need_sum) sum += hit ? (long long)x : 0ll;
        if (need_minmax) {
          vmin = (hit && x < vmin) ? x : vmin;
          vmax = (hit && x > vmax) ? x : vmax;
        }
      }
    }
  }

  BlockPartial mine;
  partial_identity(mine);
  mine.count = (unsigned long long)wave_sum_i64((long long)count);
  mine.sum[0] = wave_sum_i64(sum);
  mine.kmin[0] = wave_min_i32(vmin);
  mine.kmax[0] = wave_max_i32(vmax);
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  publish_block_partial(p, red, waves_per_block, fold_flag_ptr, block_index, num_blocks);
}

// (scan_raw_kernel itself -- the body over the whole grid -- is defined in pg_unit_scan_raw.hip: this header is also included by the batch kernel's unit)

}  // namespace pg
