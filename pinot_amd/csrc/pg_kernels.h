// pg_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels for the segment scan-filter-aggregate path.
//
// What each kernel replaces in the reference (paths under /root/reference, see include/pinot_gpu.h):
//   scan_agg_kernel     FixedBitSVForwardIndexReaderV2.readDictIds + FixedBitIntReader.read32 (K1),
//                       Dictionary.readIntValues (K2), PredicateEvaluator.applySV compaction (K3),
//                       SVScanDocIdIterator.next (K4), AndDocIdSet / OrDocIdSet (K5), DocIdSetOperator (K6),
//                       Sum/Min/Max/Count/Avg.aggregate (K7)                         -- SURVEY.md section 2.4
//   scan_group_kernel   + DictionaryBasedGroupKeyGenerator.ArrayBasedHolder (K8), aggregateGroupBySV (K9)
//   roaring_expand_kernel  ImmutableRoaringBitmap postings -> docId bitmap (BitmapInvertedIndexReader.getDocIds,
//                       InvertedIndexFilterOperator.getTrues OR of postings)
//   gather_*_kernel     ForwardIndexReader.readDictIds / Dictionary.read{Int,Double}Values for arbitrary docIds
//
// Data layout: columns stay in HBM byte-for-byte as Pinot writes them (big-endian, MSB-first bit stream,
// PinotDataBitSet.java:143-170).  A wavefront owns a tile of 2048 docs = 256*b bytes of a b-bit column; it
// pulls the tile with coalesced 16 B/lane LDS-DMA loads (global_load_lds_dwordx4) into its private LDS slot,
// then every lane extracts doc 64k+lane of step k with one ds_read2_b32 + v_perm_b32 (big-endian byte
// gather) + v_bfe_u32 -- the per-lane byte selector and bit offset are loop invariant because 64*b bits is a
// whole number of dwords.  Consecutive lanes read consecutive (or identical) LDS dwords, so the reads are
// bank-conflict free for every bit width.  No MFMA: this path is integer / gather bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pinot_gpu.h"
#include "pg_device.h"

namespace pg {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Stage one tile of a packed column into this wave's LDS slot as a linear byte image.
// tile_bytes is a multiple of 256; tile_src is 256-byte aligned.
template <bool kDma>
__device__ __forceinline__ void stage_tile(const uint8_t* __restrict__ tile_src, uint8_t* slot, int tile_bytes, int lane) {
  for (int base = 0; base < tile_bytes; base += 1024) {
    const int off = base + lane * 16;
    if (off < tile_bytes) {
      if constexpr (kDma) {
        // LDS destination = M0 base (wave-uniform) + lane * 16; global source is per lane.
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(tile_src + off), (lds_void_t*)(slot + base), 16, 0, 0);
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(tile_src + off);
        *reinterpret_cast<uint4*>(slot + off) = v;
      }
    }
  }
}

// Loop-invariant per-lane decode constants for a b-bit column.
struct LaneDec {
  uint32_t off;    // byte offset of the lane's first dword inside a step's 8*b bytes
  uint32_t s;      // bit offset of the value inside that dword pair (0..31)
  uint32_t sel;    // v_perm_b32 selector: 4 big-endian bytes starting at byte s>>3
  uint32_t shift;  // v_bfe_u32 offset (b <= 25)
};

__device__ __forceinline__ LaneDec make_lane_dec(int b, int lane) {
  LaneDec d;
  const uint32_t bit = (uint32_t)lane * (uint32_t)b;
  d.off = (bit >> 5) * 4u;
  d.s = bit & 31u;
  const uint32_t q = d.s >> 3;
  d.sel = 0x00010203u + q * 0x01010101u;
  d.shift = 32u - (d.s & 7u) - (uint32_t)b;
  return d;
}

// Value of doc (64*k + lane) of the staged tile.  kWide handles 26..31-bit columns whose value may span 5 bytes.
template <bool kWide>
__device__ __forceinline__ uint32_t decode_step(const uint8_t* slot, const LaneDec& L, int k, int b) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(slot + L.off + (uint32_t)k * 8u * (uint32_t)b);
  const uint32_t w0 = p[0];
  const uint32_t w1 = p[1];
  if constexpr (!kWide) {
    // bytes B0..B7 of the big-endian stream: w0 = B0..B3, w1 = B4..B7 (little-endian dword loads).
    const uint32_t win = __builtin_amdgcn_perm(w1, w0, L.sel);   // {B[q],B[q+1],B[q+2],B[q+3]} as a BE number
    return __builtin_amdgcn_ubfe(win, L.shift, (uint32_t)b);
  } else {
    const uint32_t hi = __builtin_bswap32(w0);
    const uint32_t lo = __builtin_bswap32(w1);
    const uint32_t end = L.s + (uint32_t)b;                      // 26..62
    const uint32_t v = end <= 32u ? (hi >> (32u - end)) : __builtin_amdgcn_alignbit(hi, lo, 64u - end);
    return v & ((1u << b) - 1u);
  }
}

// Register stack with a wave-uniform stack pointer (no scratch: every index is a compile-time constant).
struct MaskStack {
  uint32_t v[kStackDepth];
  int sp;
  __device__ __forceinline__ void push(uint32_t x) {
#pragma unroll
    for (int i = 0; i < kStackDepth; ++i) v[i] = (i == sp) ? x : v[i];
    ++sp;
  }
  __device__ __forceinline__ uint32_t pop() {
    --sp;
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < kStackDepth; ++i) r = (i == sp) ? v[i] : r;
    return r;
  }
};

__device__ __forceinline__ uint32_t valid_lane_mask(int num_docs, int tile, int lane) {
  const long long rem = (long long)num_docs - (long long)tile * kTileDocs;  // docs remaining from the tile start
  if (rem >= kTileDocs) return 0xFFFFFFFFu;
  long long nk = (rem - lane + 63) >> 6;                                     // steps k with 64k + lane < rem
  if (nk <= 0) return 0u;
  if (nk >= 32) return 0xFFFFFFFFu;
  return (1u << (int)nk) - 1u;
}

// Evaluate one scan leaf over the staged tile -> lane mask (bit k = doc 64k+lane matches).
template <bool kWide>
__device__ __forceinline__ uint32_t eval_dict_leaf_loop(const DevLeaf& L, const uint8_t* slot, const LaneDec& dec, int b) {
  uint32_t m = 0;
  if (L.kind == kLeafDictRange) {
    const uint32_t lo = (uint32_t)L.lo, span = L.span;
#pragma unroll 16
    for (int k = 0; k < kTileSteps; ++k) {
      const uint32_t d = decode_step<kWide>(slot, dec, k, b);
      m = (m << 1) | ((d - lo) < span ? 1u : 0u);
    }
  } else {  // kLeafDictSet
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)L.set_words, 0, L.set_bytes, 0x00020000);
#pragma unroll 8
    for (int k = 0; k < kTileSteps; ++k) {
      const uint32_t d = decode_step<kWide>(slot, dec, k, b);
      const uint32_t w = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (d >> 5) * 4u, 0, 0);  // OOB -> 0
      m = (m << 1) | ((w >> (d & 31u)) & 1u);
    }
  }
  return __builtin_bitreverse32(m);   // step k was shifted in first -> bit 31-k; restore bit k
}

template <bool kDma>
__device__ uint32_t eval_leaf(const ScanParams& p, const DevLeaf& L, int tile, uint8_t* wave_lds, int lane) {
  uint32_t m;
  switch (L.kind) {
    case kLeafMatchAll: m = 0xFFFFFFFFu; break;
    case kLeafMatchNone: m = 0u; break;
    case kLeafDictRange:
    case kLeafDictSet: {
      const DevColumn& c = p.cols[L.col];
      const int b = c.bits;
      const uint8_t* slot = wave_lds + c.slot_off;   // wave_lds = current staging buffer, staged by the caller
      const LaneDec dec = make_lane_dec(b, lane);
      m = b <= 25 ? eval_dict_leaf_loop<false>(L, slot, dec, b) : eval_dict_leaf_loop<true>(L, slot, dec, b);
      break;
    }
    case kLeafRawRange: {
      const DevColumn& c = p.cols[L.col];
      const long long base_doc = (long long)tile * kTileDocs;
      const long long last = (long long)p.num_docs - 1;
      const uint32_t lo = (uint32_t)L.lo, span = L.span;
      m = 0;
#pragma unroll 8
      for (int k = 0; k < kTileSteps; ++k) {
        long long doc = base_doc + k * 64 + lane;
        doc = doc > last ? last : doc;
        const uint32_t v = __builtin_bswap32(*reinterpret_cast<const uint32_t*>(c.fwd + doc * 4));
        m = (m << 1) | ((v - lo) <= span ? 1u : 0u);
      }
      m = __builtin_bitreverse32(m);
      break;
    }
    default: {  // kLeafBitmap: doc-order 64-bit words; word k of the tile covers docs 64k..64k+63
      const unsigned long long w = L.bitmap[(long long)tile * kTileSteps + (lane & 31)];
      const uint32_t wlo = (uint32_t)w, whi = (uint32_t)(w >> 32);
      m = 0;
#pragma unroll
      for (int k = 0; k < kTileSteps; ++k) {
        const uint32_t klo = __builtin_amdgcn_readlane(wlo, k);
        const uint32_t khi = __builtin_amdgcn_readlane(whi, k);
        const uint32_t half = lane < 32 ? klo : khi;
        m |= ((half >> (lane & 31)) & 1u) << k;
      }
      break;
    }
  }
  return L.exclusive ? ~m : m;
}

// Filter program (postfix) -> lane mask of the tile.  Scan-leaf columns must already be staged.
template <bool kDma>
__device__ __forceinline__ uint32_t eval_filter(const ScanParams& p, int tile, uint8_t* wave_lds, int lane) {
  if (p.num_nodes == 0) return 0xFFFFFFFFu;
  MaskStack st;
#pragma unroll
  for (int i = 0; i < kStackDepth; ++i) st.v[i] = 0;
  st.sp = 0;
  for (int n = 0; n < p.num_nodes; ++n) {
    const DevNode& nd = p.nodes[n];
    if (nd.op == PG_FILTER_LEAF) {
      st.push(eval_leaf<kDma>(p, p.leaves[nd.leaf], tile, wave_lds, lane));
    } else if (nd.op == PG_FILTER_NOT) {
      st.push(~st.pop());
    } else {
      uint32_t acc = st.pop();
      for (int c = 1; c < nd.num_children; ++c) {
        const uint32_t o = st.pop();
        acc = nd.op == PG_FILTER_AND ? (acc & o) : (acc | o);
      }
      st.push(acc);
    }
  }
  return st.pop();
}

// Stage every dictionary column of the tile that satisfies (in_filter / in_agg-only) selection.
template <bool kDma>
__device__ __forceinline__ void stage_columns(const ScanParams& p, int tile, uint8_t* wave_lds, int lane, bool filter_cols, bool agg_only_cols) {
  for (int c = 0; c < p.num_cols; ++c) {
    const DevColumn& col = p.cols[c];
    if (col.is_raw) continue;
    const bool is_filter = col.in_filter != 0;
    if ((is_filter && filter_cols) || (!is_filter && agg_only_cols)) {
      const int tile_bytes = 256 * col.bits;
      stage_tile<kDma>(col.fwd + (long long)tile * tile_bytes, wave_lds + col.slot_off, tile_bytes, lane);
    }
  }
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int32_t t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return v;
}

// Write the tile's docId bitmap (32 doc-order 64-bit words) from the lane masks.
__device__ __forceinline__ void store_tile_bitmap(unsigned long long* out, int tile, uint32_t m, int lane) {
  unsigned long long mine = 0;
#pragma unroll
  for (int k = 0; k < kTileSteps; ++k) {
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(((m >> k) & 1u) != 0u);
    mine = (lane == k) ? bal : mine;
  }
  if (lane < kTileSteps) out[(long long)tile * kTileSteps + lane] = mine;
}

// Wave-private queue of matching dictIds waiting for their dictionary gather.  Matches are compacted into it
// (ballot + mbcnt prefix), and it is drained 256 entries at a time with four back-to-back dense buffer loads and
// ONE wait, so the number of gather instructions is (matches / 64) instead of (rows / 64) and the L2 round trip is
// paid once per 256 matches instead of once per 8 steps.  `count` is wave-uniform.
struct GatherQueue {
  uint32_t* q;
  int cap;
  int count;
};

__device__ __forceinline__ void drain_queue(GatherQueue& gq, const __amdgpu_buffer_rsrc_t rsrc, int lane, long long& sum, int keep) {
  // gathers entries [0, n) where n = count - keep rounded down to what is there; `keep` = 0 drains everything.
  const int n = gq.count - keep;
  __builtin_amdgcn_wave_barrier();
  for (int base = 0; base < n; base += 256) {
    int32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = base + j * 64 + lane;
      const uint32_t d = gq.q[idx < n ? idx : 0];
      // Dictionary.readIntValues gather; out-of-range offset => the buffer load returns 0 and touches no memory.
      v[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, idx < n ? d * 4u : 0xFFFFFFFFu, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += (long long)v[j];
  }
  __builtin_amdgcn_wave_barrier();
  gq.count = 0;
}

// Per-column aggregation of the matching docs of one staged tile: every lane walks the set bits of its mask, four at
// a time so that four LDS reads are in flight (a separate straight 32-step path for dense tiles cost 30 VGPRs and
// one wavefront per SIMD of occupancy, which lost more than it gained).
template <bool kWide>
__device__ __forceinline__ void agg_dict_column(const DevColumn& col, const DevAggCol& ac, const uint8_t* slot, uint32_t m,
                                                int lane, long long& sum, int32_t& kmin, int32_t& kmax, GatherQueue& gq) {
  const int b = col.bits;
  const LaneDec dec = make_lane_dec(b, lane);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)col.dict, 0, col.dict_bytes, 0x00020000);
  uint32_t rem = m;
  for (;;) {
    if (__builtin_amdgcn_ballot_w64(rem != 0u) == 0ull) break;
    bool active[4];
    int k[4];
    uint32_t d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      active[j] = rem != 0u;
      k[j] = active[j] ? __builtin_ctz(rem) : 0;
      rem &= rem - 1u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = decode_step<kWide>(slot, dec, k[j], b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ac.need_minmax) {
        // sorted dictionary => min/max of the value is min/max of the dictId; the lookup happens once on the host
        const int32_t key = (int32_t)d[j];
        kmin = (active[j] && key < kmin) ? key : kmin;
        kmax = (active[j] && key > kmax) ? key : kmax;
      }
      if (col.is_plane) {
        // value plane: the decoded field IS (value - base); no dictionary, no gather
        if (ac.need_sum) sum += active[j] ? (long long)d[j] : 0ll;
      } else if (ac.need_sum) {
        const unsigned long long amask = __builtin_amdgcn_ballot_w64(active[j]);
        const uint32_t pos = (uint32_t)gq.count + __builtin_amdgcn_mbcnt_hi((uint32_t)(amask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)amask, 0u));
        if (active[j]) gq.q[pos] = d[j];
        gq.count += __builtin_popcountll(amask);
        if (gq.count > gq.cap - 64) drain_queue(gq, rsrc, lane, sum, 0);
      }
    }
  }
}

__device__ __forceinline__ void agg_raw_column(const DevColumn& col, const DevAggCol& ac, int num_docs, int tile, uint32_t m,
                                               int lane, long long& sum, int32_t& kmin, int32_t& kmax) {
  const long long base_doc = (long long)tile * kTileDocs;
  const long long last = (long long)num_docs - 1;
#pragma unroll 8
  for (int k = 0; k < kTileSteps; ++k) {
    long long doc = base_doc + k * 64 + lane;
    doc = doc > last ? last : doc;
    const int32_t v = (int32_t)__builtin_bswap32(*reinterpret_cast<const uint32_t*>(col.fwd + doc * 4));
    const bool match = ((m >> k) & 1u) != 0u;
    sum += match ? (long long)v : 0ll;
    kmin = (match && v < kmin) ? v : kmin;
    kmax = (match && v > kmax) ? v : kmax;
  }
  (void)ac;
}

// ------------------------------------------------------------------------------------------------
// Fused scan -> filter -> aggregate.  One wavefront per tile, grid-stride over tiles.
// ------------------------------------------------------------------------------------------------
template <bool kDma>
__global__ __launch_bounds__(kBlockThreads) void scan_agg_kernel(const ScanParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  uint8_t* wave_lds = smem + wave_in_block * p.wave_lds_bytes;
  const int total_waves = gridDim.x * waves_per_block;

  unsigned long long count = 0;
  long long sum[kMaxAggCols];
  int32_t kmin[kMaxAggCols], kmax[kMaxAggCols];
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a) { sum[a] = 0; kmin[a] = 0x7FFFFFFF; kmax[a] = (int32_t)0x80000000; }

  GatherQueue gq;
  gq.q = reinterpret_cast<uint32_t*>(wave_lds + p.queue_off);
  gq.cap = p.queue_cap;
  gq.count = 0;
  // the queue may stay filled across tiles only when exactly one column is summed (its entries are all that column's)
  int num_sum_cols = 0;
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a)
    num_sum_cols += (a < p.num_agg_cols && p.agg_cols[a].need_sum && !p.cols[p.agg_cols[a].col].is_raw && !p.cols[p.agg_cols[a].col].is_plane) ? 1 : 0;

  // Double-buffered tile pipeline: while tile t is decoded from staging buffer `buf`, the LDS-DMA loads of this
  // wave's next tile are already in flight into the other buffer, so the memory pipe never idles behind VALU work.
  unsigned long long cyc_wait = 0, cyc_filter = 0, cyc_agg = 0;
  const unsigned long long cyc_start = p.profile ? __builtin_amdgcn_s_memtime() : 0ull;
  bool hot = false;        // did the last processed tile match anything? (drives speculative value-column loads)
  bool cur_has_agg = false;
  int buf = 0;
  int tile = blockIdx.x * waves_per_block + wave_in_block;
  if (tile < p.num_tiles) stage_columns<kDma>(p, tile, wave_lds, lane, true, false);
  for (; tile < p.num_tiles; tile += total_waves) {
    uint8_t* cur = wave_lds + buf * p.stage_bytes;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (p.profile) t0 = __builtin_amdgcn_s_memtime();
    if constexpr (kDma) wait_vmem();                       // the current tile has landed
    if (p.profile) t1 = __builtin_amdgcn_s_memtime();
    const int next = tile + total_waves;
    const bool next_has_agg = hot && p.speculate != 0;
    if (p.double_buffer && next < p.num_tiles) stage_columns<kDma>(p, next, wave_lds + (buf ^ 1) * p.stage_bytes, lane, true, next_has_agg);
    uint32_t m = eval_filter<kDma>(p, tile, cur, lane);
    m &= valid_lane_mask(p.num_docs, tile, lane);
    if (p.out_bitmap) store_tile_bitmap(p.out_bitmap, tile, m, lane);
    count += (unsigned)__builtin_popcount(m);
    const bool any = __builtin_amdgcn_ballot_w64(m != 0u) != 0ull;
    hot = any;
    if (p.profile) t2 = __builtin_amdgcn_s_memtime();
    if (any && p.num_agg_cols > 0) {
      if (!cur_has_agg) {
        stage_columns<kDma>(p, tile, cur, lane, false, true);
        if constexpr (kDma) wait_vmem();
      }
#pragma unroll
      for (int a = 0; a < kMaxAggCols; ++a) {
        if (a < p.num_agg_cols) {
          const DevAggCol& ac = p.agg_cols[a];
          const DevColumn& col = p.cols[ac.col];
          if (col.is_raw) {
            agg_raw_column(col, ac, p.num_docs, tile, m, lane, sum[a], kmin[a], kmax[a]);
          } else {
            if (col.bits <= 25) agg_dict_column<false>(col, ac, cur + col.slot_off, m, lane, sum[a], kmin[a], kmax[a], gq);
            else agg_dict_column<true>(col, ac, cur + col.slot_off, m, lane, sum[a], kmin[a], kmax[a], gq);
            if (num_sum_cols > 1 && ac.need_sum && !col.is_plane && gq.count > 0) {
              const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)col.dict, 0, col.dict_bytes, 0x00020000);
              drain_queue(gq, rsrc, lane, sum[a], 0);
            }
          }
        }
      }
    }
    if (p.profile) {
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      cyc_wait += t1 - t0; cyc_filter += t2 - t1; cyc_agg += t3 - t2;
    }
    if (p.double_buffer) {
      cur_has_agg = next_has_agg;
      buf ^= 1;
    } else if (next < p.num_tiles) {
      // single buffer: the next tile is staged only now that this one is fully consumed
      stage_columns<kDma>(p, next, wave_lds, lane, true, next_has_agg);
      cur_has_agg = next_has_agg;
    }
  }

  // final drain of the carried queue (single summed column)
  if (num_sum_cols == 1 && gq.count > 0) {
#pragma unroll
    for (int a = 0; a < kMaxAggCols; ++a) {
      if (a < p.num_agg_cols && p.agg_cols[a].need_sum && !p.cols[p.agg_cols[a].col].is_raw && !p.cols[p.agg_cols[a].col].is_plane) {
        const DevColumn& col = p.cols[p.agg_cols[a].col];
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)col.dict, 0, col.dict_bytes, 0x00020000);
        drain_queue(gq, rsrc, lane, sum[a], 0);
      }
    }
  }

  // wave reduce -> block reduce (LDS) -> one partial record per workgroup
  __syncthreads();   // all waves are done with their staging slots; reuse the start of LDS for the reduction
  BlockPartial* red = reinterpret_cast<BlockPartial*>(smem);
  BlockPartial mine;
  mine.count = (unsigned long long)wave_sum_i64((long long)count);
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a) {
    mine.sum[a] = wave_sum_i64(sum[a]);
    mine.kmin[a] = wave_min_i32(kmin[a]);
    mine.kmax[a] = wave_max_i32(kmax[a]);
  }
  mine.cyc[0] = cyc_wait; mine.cyc[1] = cyc_filter; mine.cyc[2] = cyc_agg;
  mine.cyc[3] = p.profile ? __builtin_amdgcn_s_memtime() - cyc_start : 0ull;
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    BlockPartial acc = red[0];
    for (int w = 1; w < waves_per_block; ++w) {
      acc.count += red[w].count;
#pragma unroll
      for (int c = 0; c < 4; ++c) acc.cyc[c] += red[w].cyc[c];
#pragma unroll
      for (int a = 0; a < kMaxAggCols; ++a) {
        acc.sum[a] += red[w].sum[a];
        acc.kmin[a] = red[w].kmin[a] < acc.kmin[a] ? red[w].kmin[a] : acc.kmin[a];
        acc.kmax[a] = red[w].kmax[a] > acc.kmax[a] ? red[w].kmax[a] : acc.kmax[a];
      }
    }
    p.partials[blockIdx.x] = acc;
  }
}

// Reduce the per-workgroup partials into partials[num_blocks] (one record).
__global__ __launch_bounds__(kBlockThreads) void finalize_partials_kernel(BlockPartial* partials, int num_blocks) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  BlockPartial acc;
  acc.count = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) acc.cyc[c] = 0;
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a) { acc.sum[a] = 0; acc.kmin[a] = 0x7FFFFFFF; acc.kmax[a] = (int32_t)0x80000000; }
  for (int i = threadIdx.x; i < num_blocks; i += blockDim.x) {
    const BlockPartial b = partials[i];
    acc.count += b.count;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc.cyc[c] += b.cyc[c];
#pragma unroll
    for (int a = 0; a < kMaxAggCols; ++a) {
      acc.sum[a] += b.sum[a];
      acc.kmin[a] = b.kmin[a] < acc.kmin[a] ? b.kmin[a] : acc.kmin[a];
      acc.kmax[a] = b.kmax[a] > acc.kmax[a] ? b.kmax[a] : acc.kmax[a];
    }
  }
  acc.count = (unsigned long long)wave_sum_i64((long long)acc.count);
#pragma unroll
  for (int c = 0; c < 4; ++c) acc.cyc[c] = (unsigned long long)wave_sum_i64((long long)acc.cyc[c]);
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a) {
    acc.sum[a] = wave_sum_i64(acc.sum[a]);
    acc.kmin[a] = wave_min_i32(acc.kmin[a]);
    acc.kmax[a] = wave_max_i32(acc.kmax[a]);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    BlockPartial t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
      t.count += red[i].count;
#pragma unroll
      for (int c = 0; c < 4; ++c) t.cyc[c] += red[i].cyc[c];
#pragma unroll
      for (int a = 0; a < kMaxAggCols; ++a) {
        t.sum[a] += red[i].sum[a];
        t.kmin[a] = red[i].kmin[a] < t.kmin[a] ? red[i].kmin[a] : t.kmin[a];
        t.kmax[a] = red[i].kmax[a] > t.kmax[a] ? red[i].kmax[a] : t.kmax[a];
      }
    }
    partials[num_blocks] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// Fused scan -> filter -> group-by aggregate with a direct-indexed group table
// (DictionaryBasedGroupKeyGenerator.ArrayBasedHolder: groupId = sum dictId_j * prod_{k<j} card_k).
// Per-workgroup LDS partial table with LDS atomics, flushed with global atomics.
// ------------------------------------------------------------------------------------------------
template <int kScope>
__device__ __forceinline__ void table_update(long long* acc, int num_groups, int a, int kind, uint32_t g, long long v) {
  long long* slot = acc + (long long)a * num_groups + g;
  if (kind == kGroupSum) __hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, kScope);
  else if (kind == kGroupMin) __hip_atomic_fetch_min(slot, v, __ATOMIC_RELAXED, kScope);
  else __hip_atomic_fetch_max(slot, v, __ATOMIC_RELAXED, kScope);
}

template <bool kDma, bool kLdsTable>
__global__ __launch_bounds__(kBlockThreads) void scan_group_kernel(const GroupParams gp) {
  constexpr int kScope = kLdsTable ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const ScanParams& p = gp.scan;
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const int total_waves = gridDim.x * waves_per_block;
  const int G = gp.num_groups;
  const int NA = gp.num_group_aggs;

  // LDS: [staging slots of all waves][group table]
  uint8_t* wave_lds = smem + wave_in_block * p.wave_lds_bytes;
  unsigned long long* t_cnt;
  long long* t_acc;
  if constexpr (kLdsTable) {
    t_cnt = reinterpret_cast<unsigned long long*>(smem + waves_per_block * p.wave_lds_bytes);   // after every wave's staging buffers
    t_acc = reinterpret_cast<long long*>(t_cnt + G);
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
      t_cnt[g] = 0ull;
      for (int a = 0; a < NA; ++a) {
        const int kind = gp.group_aggs[a].kind;
        t_acc[(long long)a * G + g] = kind == kGroupSum ? 0ll : (kind == kGroupMin ? 0x7FFFFFFFFFFFFFFFll : (long long)0x8000000000000000ull);
      }
    }
    __syncthreads();
  } else {
    t_cnt = gp.table_count;
    t_acc = gp.table_acc;
  }

  int buf = 0;
  int tile = blockIdx.x * waves_per_block + wave_in_block;
  if (tile < p.num_tiles) stage_columns<kDma>(p, tile, wave_lds, lane, true, true);   // group-by touches every column of (nearly) every tile
  for (; tile < p.num_tiles; tile += total_waves) {
    uint8_t* cur = wave_lds + buf * p.stage_bytes;
    if constexpr (kDma) wait_vmem();
    const int next = tile + total_waves;
    if (p.double_buffer) {
      if (next < p.num_tiles) stage_columns<kDma>(p, next, wave_lds + (buf ^ 1) * p.stage_bytes, lane, true, true);
      buf ^= 1;
    }
    uint32_t m = eval_filter<kDma>(p, tile, cur, lane);
    m &= valid_lane_mask(p.num_docs, tile, lane);
    const bool any = __builtin_amdgcn_ballot_w64(m != 0u) != 0ull;
    if (any)

    for (int kb = 0; kb < kTileSteps; kb += 8) {
      if (__builtin_amdgcn_ballot_w64(((m >> kb) & 0xFFu) != 0u) == 0ull) continue;
      uint32_t g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = 0;
      // group id
      for (int c = 0; c < gp.num_group_cols; ++c) {
        const DevColumn& col = p.cols[gp.group_cols[c]];
        const int b = col.bits;
        const LaneDec dec = make_lane_dec(b, lane);
        const uint8_t* slot = cur + col.slot_off;
        const uint32_t mult = (uint32_t)gp.group_mult[c];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t d = b <= 25 ? decode_step<false>(slot, dec, kb + j, b) : decode_step<true>(slot, dec, kb + j, b);
          g[j] += d * mult;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((m >> (kb + j)) & 1u) __hip_atomic_fetch_add(&t_cnt[g[j]], 1ull, __ATOMIC_RELAXED, kScope);
      }
      for (int a = 0; a < NA; ++a) {
        const DevGroupAgg ga = gp.group_aggs[a];
        const DevColumn& col = p.cols[ga.col];
        long long v[8];
        if (col.is_raw) {
          const long long base_doc = (long long)tile * kTileDocs;
          const long long last = (long long)p.num_docs - 1;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            long long doc = base_doc + (kb + j) * 64 + lane;
            doc = doc > last ? last : doc;
            v[j] = (long long)(int32_t)__builtin_bswap32(*reinterpret_cast<const uint32_t*>(col.fwd + doc * 4));
          }
        } else {
          const int b = col.bits;
          const LaneDec dec = make_lane_dec(b, lane);
          const uint8_t* slot = cur + col.slot_off;
          if (ga.kind == kGroupSum && !col.is_plane) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)col.dict, 0, col.dict_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const uint32_t d = b <= 25 ? decode_step<false>(slot, dec, kb + j, b) : decode_step<true>(slot, dec, kb + j, b);
              const bool match = ((m >> (kb + j)) & 1u) != 0u;
              v[j] = (long long)(int32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, match ? d * 4u : 0xFFFFFFFFu, 0, 0);
            }
          } else {
            // value plane: the decoded field is (value - base).  MIN / MAX on a sorted dictionary: aggregate the
            // dictId (monotone in the value), look the value up on the host at the end.
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              v[j] = (long long)(b <= 25 ? decode_step<false>(slot, dec, kb + j, b) : decode_step<true>(slot, dec, kb + j, b));
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if ((m >> (kb + j)) & 1u) table_update<kScope>(t_acc, G, a, ga.kind, g[j], v[j]);
        }
      }
    }
    if (!p.double_buffer && next < p.num_tiles) stage_columns<kDma>(p, next, wave_lds, lane, true, true);
  }

  if constexpr (kLdsTable) {
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
      const unsigned long long c = t_cnt[g];
      if (c == 0ull) continue;
      __hip_atomic_fetch_add(&gp.table_count[g], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int a = 0; a < NA; ++a)
        table_update<__HIP_MEMORY_SCOPE_AGENT>(gp.table_acc, G, a, gp.group_aggs[a].kind, (uint32_t)g, t_acc[(long long)a * G + g]);
    }
  }
}

__global__ void init_group_table_kernel(GroupParams gp) {
  const int G = gp.num_groups;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) {
    gp.table_count[g] = 0ull;
    for (int a = 0; a < gp.num_group_aggs; ++a) {
      const int kind = gp.group_aggs[a].kind;
      gp.table_acc[(long long)a * G + g] = kind == kGroupSum ? 0ll : (kind == kGroupMin ? 0x7FFFFFFFFFFFFFFFll : (long long)0x8000000000000000ull);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Roaring container -> docId bitmap (ORs into a doc-order bitmap).  One workgroup per container; containers
// of one launch have distinct keys, so the read-modify-write of the 8 KiB bitmap window is race free.
// Format: public RoaringFormatSpec (array = sorted uint16, bitset = 1024 x uint64 LE, run = {start, len-1}).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t load_u16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

__global__ __launch_bounds__(kBlockThreads) void roaring_expand_kernel(const uint8_t* __restrict__ inv, const DevContainer* __restrict__ dir,
                                                                       int first, unsigned long long* bitmap, long long num_words) {
  __shared__ unsigned long long w[1024];
  const DevContainer c = dir[first + blockIdx.x];
  const uint8_t* payload = inv + c.offset;
  for (int j = threadIdx.x; j < 1024; j += blockDim.x) w[j] = 0ull;
  __syncthreads();
  if (c.type == 0) {
    for (uint32_t i = threadIdx.x; i < c.cardinality; i += blockDim.x) {
      const uint32_t v = load_u16(payload + 2 * i);
      atomicOr(&w[v >> 6], 1ull << (v & 63u));
    }
  } else if (c.type == 1) {
    for (int j = threadIdx.x; j < 1024; j += blockDim.x) {
      const uint8_t* q = payload + 8 * j;
      const unsigned long long v = (unsigned long long)load_u16(q) | ((unsigned long long)load_u16(q + 2) << 16) |
                                   ((unsigned long long)load_u16(q + 4) << 32) | ((unsigned long long)load_u16(q + 6) << 48);
      w[j] = v;
    }
  } else {
    for (uint32_t r = threadIdx.x; r < c.num_runs; r += blockDim.x) {
      const uint32_t start = load_u16(payload + 2 + 4 * r);
      const uint32_t end = start + load_u16(payload + 4 + 4 * r);   // inclusive
      for (uint32_t wi = start >> 6; wi <= (end >> 6); ++wi) {
        const uint32_t lo = wi == (start >> 6) ? (start & 63u) : 0u;
        const uint32_t hi = wi == (end >> 6) ? (end & 63u) : 63u;
        const unsigned long long mask = (hi - lo == 63u ? ~0ull : ((1ull << (hi - lo + 1u)) - 1ull)) << lo;
        atomicOr(&w[wi], mask);
      }
    }
  }
  __syncthreads();
  const long long base = (long long)c.key * 1024;
  for (int j = threadIdx.x; j < 1024; j += blockDim.x) {
    const unsigned long long v = w[j];
    if (v != 0ull && base + j < num_words) bitmap[base + j] |= v;
  }
}


// ------------------------------------------------------------------------------------------------
// Value-plane materialisation (one-time, per summed column): plane[doc] = dictionary[dictId[doc]] - base, bit-packed
// with `w` bits in the SAME big-endian MSB-first stream format as the forward index, so the scan kernels decode it
// with the same code.  It trades HBM capacity (288 GB) for the per-row dictionary gather, which on MI355X costs as
// much L2 capacity as streaming ~22 bytes (profiles/r1/microbench.jsonl).  w == 32 stores big-endian int32 values.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void materialize_plane_kernel(const DevColumn col, uint8_t* __restrict__ out, int w, int32_t base,
                                                                          int num_docs, int num_tiles, int in_slot_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const int out_words = 64 * w;                                   // 2048 values * w bits
  uint8_t* wave_lds = smem + wave_in_block * (in_slot_bytes + out_words * 4 + 16);
  uint32_t* W = reinterpret_cast<uint32_t*>(wave_lds + in_slot_bytes);
  const int total_waves = gridDim.x * waves_per_block;
  const int b = col.bits;
  const LaneDec dec = make_lane_dec(b, lane);
  for (int tile = blockIdx.x * waves_per_block + wave_in_block; tile < num_tiles; tile += total_waves) {
    stage_tile<false>(col.fwd + (long long)tile * 256 * b, wave_lds, 256 * b, lane);
    if (w < 32) for (int i = lane; i <= out_words; i += 64) W[i] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < kTileSteps; ++k) {
      const uint32_t d = b <= 25 ? decode_step<false>(wave_lds, dec, k, b) : decode_step<true>(wave_lds, dec, k, b);
      const long long doc = (long long)tile * kTileDocs + k * 64 + lane;
      const bool valid = doc < num_docs;
      const int32_t v = valid ? col.dict[d < (uint32_t)col.cardinality ? d : 0u] : base;
      if (w == 32) {
        if (valid) *reinterpret_cast<uint32_t*>(out + doc * 4) = __builtin_bswap32((uint32_t)v);
      } else {
        const uint32_t x = (uint32_t)v - (uint32_t)base;
        const uint32_t pos = (uint32_t)(k * 64 + lane) * (uint32_t)w;
        const uint32_t j = pos >> 5, s = pos & 31u;
        if (s + (uint32_t)w <= 32u) {
          atomicOr(&W[j], x << (32u - s - (uint32_t)w));
        } else {
          const uint32_t lo_bits = s + (uint32_t)w - 32u;
          atomicOr(&W[j], x >> lo_bits);
          atomicOr(&W[j + 1], x << (32u - lo_bits));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (w < 32) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(out + (long long)tile * 256 * w);
      for (int i = lane; i < out_words; i += 64) dst[i] = __builtin_bswap32(W[i]);   // host-order BE word -> stream bytes
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void fill_words_kernel(unsigned long long* words, long long n, unsigned long long value) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) words[i] = value;
}

// ------------------------------------------------------------------------------------------------
// BlockValSet-level readers for arbitrary docIds (one thread per docId).
// FixedBitIntReader.readUnchecked restated for a padded device buffer.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t read_packed(const uint8_t* fwd, long long doc, int b) {
  const long long bit = doc * b;
  const uint8_t* p = fwd + (bit >> 3);
  const int s = (int)(bit & 7);
  unsigned long long win = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) win = (win << 8) | p[i];   // buffer is padded by >= 8 bytes at open
  return (uint32_t)((win >> (64 - s - b)) & ((1ull << b) - 1ull));
}

__global__ void gather_values_kernel(DevColumn col, const int32_t* __restrict__ doc_ids, int n, int32_t* out_dict_ids,
                                     int32_t* out_ints, double* out_doubles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long doc = doc_ids[i];
  int32_t v;
  if (col.is_raw) {
    v = (int32_t)__builtin_bswap32(*reinterpret_cast<const uint32_t*>(col.fwd + doc * 4));
    if (out_dict_ids) out_dict_ids[i] = -1;
  } else {
    const uint32_t d = read_packed(col.fwd, doc, col.bits);
    if (out_dict_ids) out_dict_ids[i] = (int32_t)d;
    v = (out_ints || out_doubles) ? col.dict[d] : 0;
  }
  if (out_ints) out_ints[i] = v;
  if (out_doubles) out_doubles[i] = (double)v;
}

}  // namespace pg
