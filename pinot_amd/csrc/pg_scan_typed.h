// scan_private_typed_kernel: the lane-private scan -> filter -> aggregate kernel for aggregated columns outside the 32-bit
// dictionary domain of scan_private_kernel: raw INT / LONG / FLOAT / DOUBLE forward indexes (FixedByteChunkSVForwardIndexReader
// .getInt / getLong / getFloat / getDouble: big-endian value at rawDataStart + entrySize * docId) and 8-byte dictionaries
// (LongDictionary / DoubleDictionary; FLOAT dictionaries are widened exactly at open).  Aggregation semantics are those of
// agg_raw_column_typed / agg_dict_column_wide in pg_kernels.h (Sum / Min / Max / AvgAggregationFunction.aggregate per stored type).
//
// Layout: a lane owns docs 32*lane .. 32*lane+31 of a 2048-doc tile, i.e. 128 or 256 CONTIGUOUS bytes of a raw column, read with
// 16-byte loads, four of them in flight per chunk (the LDS-staged kernel issues one 4/8-byte load per lane per 64-doc step and
// waits for it: 2.1 TB/s on raw DOUBLE; the lane-contiguous read pattern streams at 6.2-6.5 TB/s in the microbenchmark).
// Queries that also aggregate a 32-bit-domain dictionary column stay in the LDS-staged kernel: folding agg_private_dispatch in
// costs 183 VGPRs (two waves per SIMD) and the raw streams lose their latency hiding.
#pragma once
#include "pg_kernels.h"

namespace pg {

struct TypedAcc {
  long long isum;
  double fsum;
  int32_t kmin, kmax;          // dictIds / raw INT values
  long long kmin64, kmax64;    // raw LONG values or order keys of raw FLOAT / DOUBLE values
};

__device__ __forceinline__ void typed_acc_identity(TypedAcc& t) {
  t.isum = 0; t.fsum = 0.0; t.kmin = 0x7FFFFFFF; t.kmax = (int32_t)0x80000000;
  t.kmin64 = 0x7FFFFFFFFFFFFFFFll; t.kmax64 = (long long)0x8000000000000000ull;
}

// one big-endian 8-byte value whose first dword is `first`
__device__ __forceinline__ long long be64(uint32_t first, uint32_t second) {
  return (long long)(((unsigned long long)__builtin_bswap32(first) << 32) | (unsigned long long)__builtin_bswap32(second));
}

template <typename AC>
__device__ __forceinline__ void typed_fold64(const AC& ac, long long bits, bool match, TypedAcc& t) {
  long long key;
  if (ac.vkind == kValI64) {
    t.isum += match ? bits : 0ll;
    t.fsum += match ? (double)bits : 0.0;
    key = bits;
  } else {
    const double v = __longlong_as_double(bits);
    t.fsum += match ? v : 0.0;
    key = f64_order_key(v);
  }
  if (ac.need_minmax) {
    t.kmin64 = (match && key < t.kmin64) ? key : t.kmin64;
    t.kmax64 = (match && key > t.kmax64) ? key : t.kmax64;
  }
}

// raw LONG / DOUBLE: 32 docs = 256 contiguous bytes per lane
template <typename AC>
__device__ __forceinline__ void agg_raw64_private(const AC& ac, long long tile, int lane, uint32_t m, TypedAcc& t) {
  const uint4* src = reinterpret_cast<const uint4*>(ac.fwd + (tile * 2048 + (long long)lane * 32) * 8);
#pragma unroll 1      // unrolled, all sixteen loads are hoisted to the top: 256 VGPRs, two waves per SIMD
  for (int c = 0; c < 4; ++c) {
    uint4 w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = src[c * 4 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = c * 8 + i * 2;
      typed_fold64(ac, be64(w[i].x, w[i].y), ((m >> j) & 1u) != 0u, t);
      typed_fold64(ac, be64(w[i].z, w[i].w), ((m >> (j + 1)) & 1u) != 0u, t);
    }
  }
}

// The same column read COALESCED: load i of the tile takes 16 bytes per lane at (64 i + lane) * 16 -- one full KB per instruction,
// eight 128-byte lines instead of the 64 half-used lines of the lane-contiguous pattern -- so the lane holds docs 128 i + 2 lane (+1)
// of the tile instead of its own 32.  The filter mask stays in the lane-private layout (bit j of lane L = doc 32 L + j); the two mask
// bits a lane needs per load are lane (4 i + lane / 16)'s bits 2 (lane % 16) (+1): one ds_bpermute per load.
template <typename AC>
__device__ __forceinline__ void agg_raw64_coalesced(const AC& ac, long long tile, int lane, uint32_t m, TypedAcc& t) {
  const uint4* src = reinterpret_cast<const uint4*>(ac.fwd + tile * 2048 * 8) + lane;
  const int bit = (2 * lane) & 31;
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    uint4 w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = src[(c * 4 + i) * 64];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t mm = (uint32_t)__shfl((int)m, 4 * (c * 4 + i) + (lane >> 4), 64) >> bit;
      typed_fold64(ac, be64(w[i].x, w[i].y), (mm & 1u) != 0u, t);
      typed_fold64(ac, be64(w[i].z, w[i].w), (mm & 2u) != 0u, t);
    }
  }
}

// raw INT / FLOAT: 32 docs = 128 contiguous bytes per lane
template <typename AC>
__device__ __forceinline__ void agg_raw32_private(const AC& ac, long long tile, int lane, uint32_t m, TypedAcc& t) {
  const uint4* src = reinterpret_cast<const uint4*>(ac.fwd + (tile * 2048 + (long long)lane * 32) * 4);
#pragma unroll 1
  for (int c = 0; c < 2; ++c) {
    uint4 w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = src[c * 4 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t dw[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool match = ((m >> (c * 16 + i * 4 + k)) & 1u) != 0u;
        const uint32_t bits = __builtin_bswap32(dw[k]);
        if (ac.vkind == kValF32) {
          const double v = (double)__uint_as_float(bits);
          t.fsum += match ? v : 0.0;
          if (ac.need_minmax) {
            const long long key = f64_order_key(v);
            t.kmin64 = (match && key < t.kmin64) ? key : t.kmin64;
            t.kmax64 = (match && key > t.kmax64) ? key : t.kmax64;
          }
        } else {
          const int32_t v = (int32_t)bits;
          t.isum += match ? (long long)v : 0ll;
          if (ac.need_minmax) {
            t.kmin = (match && v < t.kmin) ? v : t.kmin;
            t.kmax = (match && v > t.kmax) ? v : t.kmax;
          }
        }
      }
    }
  }
}

// LONG / DOUBLE dictionary: dictIds decoded at compile-time bit positions, then sixteen 8-byte dictionary gathers in flight.
// MIN / MAX run on the dictIds (the dictionary is sorted).
template <typename AC>
__device__ __forceinline__ void agg_dict64_private(const AC& ac, long long tile, int lane, uint32_t m, TypedAcc& t) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const int b = ac.bits;
  const uint32_t* words = reinterpret_cast<const uint32_t*>(ac.fwd + tile * (256ll * b)) + lane * b;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ac.dict, 0, ac.dict_bytes, 0x00020000);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t d[16];
    if (h == 0) decode16_private_dispatch<0>(b, words, d); else decode16_private_dispatch<1>(b, words, d);
    if (ac.need_sum) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {          // eight gathers in flight at a time (sixteen cost 32 more live registers)
        u32x2 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ((m >> (16 * h + 8 * q + j)) & 1u) ? d[8 * q + j] * 8u : 0xFFFFFFFFu, 0, 0);   // out of range -> 0
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long long bits = (long long)(((unsigned long long)w[j].y << 32) | (unsigned long long)w[j].x);
          if (ac.vkind == kValI64) { t.isum += bits; t.fsum += (double)bits; }
          else t.fsum += __longlong_as_double(bits);         // +0.0 for the docs that did not match
        }
      }
    }
    if (ac.need_minmax) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const bool match = ((m >> (16 * h + j)) & 1u) != 0u;
        const int32_t key = (int32_t)d[j];
        t.kmin = (match && key < t.kmin) ? key : t.kmin;
        t.kmax = (match && key > t.kmax) ? key : t.kmax;
      }
    }
  }
}

// kAggSlots: 1 for queries with at most one aggregated column (a TypedAcc is ten registers: the four-slot instantiation keeps forty of
// them live across the tile loop and spills 32 of its 128; the one-slot form does not), else kMaxAggCols.
#ifndef PG_TYPED_WAVES
#define PG_TYPED_WAVES 4
#endif
// Three and four slots: 3 wavefronts per SIMD (168 registers, no scratch) instead of 4 (128 registers, 32 of them spilled inside the tile
// loop) -- 1.280 -> 0.913 ms on four raw columns at 1 B rows (profiles/r6/spills_three_vs_four_waves_ab.txt).
#ifndef PG_TYPED_WAVES_MANY
#define PG_TYPED_WAVES_MANY 3
#endif
// `block_index` of `num_blocks`: the workgroup's place among those working on this parameter block (the whole grid, or an item's share of
// scan_typed_batch_kernel's launch).  P: ScanParams, or its constant-address-space form in device memory.
template <int kAggSlots, typename P>
__device__ __forceinline__ void scan_private_typed_body(const P& p, uint32_t block_index, uint32_t num_blocks, BlockPartial* red, uint32_t* fold_flag_ptr, uint32_t* set_lds) {
  // (the filter's dictId sets in LDS, once per workgroup: pg_kernels.h stage_filter_sets)
  if (p.set_leaves_in_lds == 0) set_lds = nullptr;
  if (set_lds != nullptr) stage_filter_sets(p, set_lds);
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const long long total_waves = (long long)num_blocks * waves_per_block;
  const long long num_tiles = ((long long)p.num_docs + 2047) / 2048;

  unsigned long long count = 0;
  TypedAcc acc[kAggSlots];
#pragma unroll
  for (int a = 0; a < kAggSlots; ++a) typed_acc_identity(acc[a]);

  uint32_t entries = 0u;
  const bool listed = p.tile_list != nullptr;              // index-driven filters: only the tiles index_and_kernel listed hold a match
  const long long tile_limit = listed ? (long long)*p.tile_count : num_tiles;
  for (long long tile_it = (long long)block_index * waves_per_block + wave_in_block; tile_it < tile_limit; tile_it += total_waves) {
    const long long tile = listed ? (long long)p.tile_list[tile_it] : tile_it;
    uint32_t m = eval_filter_private(p, tile, lane, entries, nullptr, set_lds);
    const long long rem = (long long)p.num_docs - (tile * 2048 + lane * 32);
    m &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u));
    if (p.out_bitmap) reinterpret_cast<uint32_t*>(p.out_bitmap)[tile * 64 + lane] = m;
    count += (unsigned)__builtin_popcount(m);
    if (p.num_agg_cols == 0 || __builtin_amdgcn_ballot_w64(m != 0u) == 0ull) continue;
    for (int a = 0; a < p.num_agg_cols; ++a) {
      const auto& ac = p.agg_cols[a];
      TypedAcc t;
      typed_acc_identity(t);
      if (ac.is_raw) {
        if (ac.vkind == kValI64 || ac.vkind == kValF64) { if (p.raw64_coalesced) agg_raw64_coalesced(ac, tile, lane, m, t); else agg_raw64_private(ac, tile, lane, m, t); }
        else agg_raw32_private(ac, tile, lane, m, t);
      } else {
        agg_dict64_private(ac, tile, lane, m, t);          // the host sends 32-bit-domain dictionary columns elsewhere
      }
#pragma unroll
      for (int s = 0; s < kAggSlots; ++s) {
        if (s == a) {
          acc[s].isum += t.isum;
          acc[s].fsum += t.fsum;
          acc[s].kmin = t.kmin < acc[s].kmin ? t.kmin : acc[s].kmin;
          acc[s].kmax = t.kmax > acc[s].kmax ? t.kmax : acc[s].kmax;
          acc[s].kmin64 = t.kmin64 < acc[s].kmin64 ? t.kmin64 : acc[s].kmin64;
          acc[s].kmax64 = t.kmax64 > acc[s].kmax64 ? t.kmax64 : acc[s].kmax64;
        }
      }
    }
  }

  flush_filter_entries(p, entries);
  BlockPartial mine;
  partial_identity(mine);
  mine.count = (unsigned long long)wave_sum_i64((long long)count);
  mine.entries = (unsigned long long)wave_sum_i64((long long)entries);
#pragma unroll
  for (int a = 0; a < kAggSlots; ++a) {
    if (a >= p.num_agg_cols) continue;             // unused slots keep the identities: six wave reductions less each
    const auto& ac = p.agg_cols[a];
    const bool wide_keys = ac.is_raw && ac.vkind != kValI32;
    if (ac.need_sum) {
      mine.sum[a] = wave_sum_i64(acc[a].isum);
      if (ac.vkind != kValI32) mine.fsum[a] = wave_sum_f64(acc[a].fsum);
    }
    if (ac.need_minmax && !wide_keys) { mine.kmin[a] = wave_min_i32(acc[a].kmin); mine.kmax[a] = wave_max_i32(acc[a].kmax); }
    if (ac.need_minmax && wide_keys) { mine.kmin64[a] = wave_min_i64(acc[a].kmin64); mine.kmax64[a] = wave_max_i64(acc[a].kmax64); }
  }
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  publish_block_partial(p, red, waves_per_block, fold_flag_ptr, block_index, num_blocks);
}

template <int kAggSlots>
__global__ __launch_bounds__(kBlockThreads, (kAggSlots == 1 ? PG_TYPED_WAVES : (kAggSlots == 2 ? 4 : PG_TYPED_WAVES_MANY))) void scan_private_typed_kernel(const ScanParams p) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  __shared__ uint32_t set_lds[kSetLdsWords];
  scan_private_typed_body<kAggSlots>(p, blockIdx.x, gridDim.x, red, &fold_flag, set_lds);
}

// pg_execute_batch's shared launch for items of this kernel's shape (aggregations over raw INT / LONG / FLOAT / DOUBLE columns and 8-byte
// dictionaries on a server's many small segments): workgroups [block_first[i], block_first[i + 1]) work on items[i], every item folds and
// publishes its own record (see scan_private_batch_kernel).
template <int kAggSlots>
__global__ __launch_bounds__(kBlockThreads, (kAggSlots == 1 ? PG_TYPED_WAVES : (kAggSlots == 2 ? 4 : PG_TYPED_WAVES_MANY))) void scan_typed_batch_kernel(const BatchParams bp) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  __shared__ uint32_t set_lds[kSetLdsWords];
  int lo = 0, hi = bp.num_items - 1;                // the last item whose first workgroup is at or before this one
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (bp.block_first[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const uint32_t first = bp.block_first[lo];
  typedef const __attribute__((address_space(4))) ScanParams ConstantScanParams;
  const ConstantScanParams& item = *(ConstantScanParams*)(bp.items + lo);
  scan_private_typed_body<kAggSlots>(item, blockIdx.x - first, bp.block_first[lo + 1] - first, red, &fold_flag, set_lds);
}

}  // namespace pg
