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
//   scan_private_kernel / group_private_kernel   the same roles in the lane-private layout (the kernels that normally run)
//   group_chunk_* / group_compact_kernel / group_first_doc_kernel   IntMapBasedHolder-range results and numGroupsLimit
//   (pg_scan_typed.h: raw and 8-byte aggregated columns; pg_group_partition.h: partitioned fill of the HBM group table)
//
// Data layout: columns stay in HBM byte-for-byte as Pinot writes them (big-endian, MSB-first bit stream,
// PinotDataBitSet.java:143-170).  Two ways of dealing a 2048-doc tile to the 64 lanes of a wavefront:
// lane-private (lane i owns docs 32i..32i+31 = b consecutive dwords, decoded at compile-time bit positions straight from global
// loads) and strided / LDS-staged, described next.  In the staged kernels a wavefront owns a tile of 64*steps docs (steps = 32: 256*b bytes of a b-bit column); it
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
#include "pg_fsm_kernels.h"

namespace pg {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Stage one tile of a packed column into this wave's LDS slot as a linear byte image.
// tile_bytes is a multiple of 256; tile_src is 256-byte aligned.
template <bool kDma>
__device__ __forceinline__ void stage_tile(const uint8_t* __restrict__ tile_src, uint8_t* slot, int tile_bytes, int lane) {
  const int lane_off = lane * 16;
  const int full = tile_bytes & ~1023;
  for (int base = 0; base < full; base += 1024) {        // whole 1 KiB chunks: no per-lane guard
    if constexpr (kDma) {
      // LDS destination = M0 base (wave-uniform) + lane * 16; global source is per lane.
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(tile_src + base + lane_off), (lds_void_t*)(slot + base), 16, 0, 0);
    } else {
      const uint4 v = *reinterpret_cast<const uint4*>(tile_src + base + lane_off);
      *reinterpret_cast<uint4*>(slot + base + lane_off) = v;
    }
  }
  if (full < tile_bytes && full + lane_off < tile_bytes) {   // tail chunk (tile_bytes is a multiple of 128)
    if constexpr (kDma) {
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(tile_src + full + lane_off), (lds_void_t*)(slot + full), 16, 0, 0);
    } else {
      const uint4 v = *reinterpret_cast<const uint4*>(tile_src + full + lane_off);
      *reinterpret_cast<uint4*>(slot + full + lane_off) = v;
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

// Eight consecutive steps k0..k0+7 of one column: all eight LDS reads are issued before the first result is used (written
// out as two phases on purpose -- left to itself the scheduler serialises read -> wait -> extract under register pressure and
// the loop becomes LDS-latency bound).
__device__ __forceinline__ void decode_steps8(const uint8_t* slot, const LaneDec& L, int k0, int b, uint32_t (&out)[8]) {
  const uint8_t* base = slot + L.off + (uint32_t)k0 * 8u * (uint32_t)b;
  uint32_t w0[8], w1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(base + (uint32_t)j * 8u * (uint32_t)b);
    w0[j] = p[0];
    w1[j] = p[1];
  }
  if (b <= 25) {
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = __builtin_amdgcn_ubfe(__builtin_amdgcn_perm(w1[j], w0[j], L.sel), L.shift, (uint32_t)b);
  } else {
    const uint32_t end = L.s + (uint32_t)b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t hi = __builtin_bswap32(w0[j]), lo = __builtin_bswap32(w1[j]);
      const uint32_t v = end <= 32u ? (hi >> (32u - end)) : __builtin_amdgcn_alignbit(hi, lo, 64u - end);
      out[j] = v & ((1u << b) - 1u);
    }
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


__device__ __forceinline__ uint32_t full_mask(int steps) { return steps >= 32 ? 0xFFFFFFFFu : ((1u << steps) - 1u); }

__device__ __forceinline__ uint32_t valid_lane_mask(int num_docs, int tile, int lane, int steps) {
  const long long rem = (long long)num_docs - (long long)tile * 64 * steps;  // docs remaining from the tile start
  if (rem >= 64 * steps) return full_mask(steps);
  long long nk = (rem - lane + 63) >> 6;                                      // steps k with 64k + lane < rem
  if (nk <= 0) return 0u;
  if (nk >= steps) return full_mask(steps);
  return (1u << (int)nk) - 1u;
}

__device__ __forceinline__ uint32_t decode_auto(const uint8_t* slot, const LaneDec& L, int k, int b) {
  return b <= 25 ? decode_step<false>(slot, L, k, b) : decode_step<true>(slot, L, k, b);
}

// Shift a compare result into the lane mask with two VALU ops: v_cmp writes VCC, v_addc computes m = 2*m + carry.
// (The portable form -- compare, select, shift, or -- costs 3.5 ops per value; integer VALU ops issue at 4 cycles per
// wave64 instruction on gfx950, and this loop is what bounds the kernel: profiles/r1/microbench.jsonl "valu_rate".)
__device__ __forceinline__ void shift_in_lt(uint32_t& m, uint32_t x, uint32_t limit) {
  asm("v_cmp_gt_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(x), "s"(limit) : "vcc");
}
__device__ __forceinline__ void shift_in_le(uint32_t& m, uint32_t x, uint32_t limit) {
  asm("v_cmp_ge_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(x), "s"(limit) : "vcc");
}

// DICT_RANGE leaf with the bit width known at compile time: the LDS offsets of the 32 steps become instruction
// immediates (no per-step address add) and the field width / selector math folds away.
template <int B, bool kLoZero>
__device__ __forceinline__ uint32_t range_leaf_loop(const uint8_t* slot, int lane, uint32_t lo, uint32_t span, int steps) {
  // The lane constants of all 25 widths are loop invariant, so LLVM would hoist every one of them out of the tile
  // loop and keep ~150 VGPRs live; the empty asm makes `lane` opaque here so they are recomputed (8 ops per tile).
  asm volatile("" : "+v"(lane));
  const LaneDec dec = make_lane_dec(B, lane);
  const uint8_t* base = slot + dec.off;
  const uint32_t limit = __builtin_amdgcn_readfirstlane(span);
  uint32_t m = 0;
  for (int kb = 0; kb < steps; kb += 16) {     // steps is 16 or 32: no remainder loop
    uint32_t w0[16], w1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {              // 16 LDS reads in flight, offsets are immediates
      const uint32_t* p = reinterpret_cast<const uint32_t*>(base + (kb + j) * 8 * B);
      w0[j] = p[0];
      w1[j] = p[1];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t win = __builtin_amdgcn_perm(w1[j], w0[j], dec.sel);
      const uint32_t d = __builtin_amdgcn_ubfe(win, dec.shift, (uint32_t)B);
      shift_in_lt(m, kLoZero ? d : d - lo, limit);
    }
  }
  return __builtin_bitreverse32(m) >> (32 - steps);   // step k was shifted in first; restore bit k
}

template <bool kLoZero>
__device__ __forceinline__ uint32_t range_leaf_dispatch(int b, const uint8_t* slot, int lane, uint32_t lo, uint32_t span, int steps) {
  switch (b) {
#define PG_CASE(B) case B: return range_leaf_loop<B, kLoZero>(slot, lane, lo, span, steps);
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20)
    PG_CASE(21) PG_CASE(22) PG_CASE(23) PG_CASE(24) PG_CASE(25)
#undef PG_CASE
    default: return 0u;
  }
}

// Generic leaf loops (wide 26..31-bit streams, dictId sets).
template <bool kWide>
__device__ __forceinline__ uint32_t eval_dict_leaf_loop(const DevNode& L, const uint8_t* slot, const LaneDec& dec, int b, int steps) {
  uint32_t m = 0;
  if (L.kind == kLeafDictRange) {
    const uint32_t lo = (uint32_t)L.lo, limit = __builtin_amdgcn_readfirstlane(L.span);
#pragma unroll 8
    for (int k = 0; k < steps; ++k) {
      const uint32_t d = decode_step<kWide>(slot, dec, k, b);
      shift_in_lt(m, d - lo, limit);
    }
  } else {  // kLeafDictSet
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)L.set_words, 0, L.set_bytes, 0x00020000);
#pragma unroll 8
    for (int k = 0; k < steps; ++k) {
      const uint32_t d = decode_step<kWide>(slot, dec, k, b);
      const uint32_t w = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (d >> 5) * 4u, 0, 0);  // OOB -> 0
      m = (m << 1) | ((w >> (d & 31u)) & 1u);
    }
  }
  return __builtin_bitreverse32(m) >> (32 - steps);   // step k was shifted in first; restore bit k
}

// Order-preserving integer image of a double (signed compare of the keys == Double.compare of the values: -0.0 < +0.0,
// NaN above +Infinity).  Every NaN is mapped to the canonical one so that MAX sees it; the host turns a NaN maximum into
// a NaN minimum as well (java.lang.Math.min / max propagate NaN).  The map is its own inverse.
__device__ __forceinline__ long long f64_order_key(double v) {
  long long b = __double_as_longlong(v);
  b = (v != v) ? 0x7FF8000000000000ll : b;
  return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFll);
}

// `stage` is the wave's current column staging buffer, `bstage` its current bitmap staging buffer (already filled).
// The node record carries everything the leaf needs (one scalar load).
// (must inline: a call would force the kernel-argument block into scratch memory to take its address)
__device__ __forceinline__ uint32_t eval_leaf(const ScanParams& p, const DevNode& L, int tile, const uint8_t* stage, const uint8_t* bstage, int lane) {
  const int steps = p.tile_steps;
  uint32_t m;
  switch (L.kind) {
    case kLeafMatchAll: m = 0xFFFFFFFFu; break;
    case kLeafMatchNone: m = 0u; break;
    case kLeafDictRange:
    case kLeafDictSet: {
      const int b = L.bits;
      const uint8_t* slot = stage + L.slot_off;
      if (L.kind == kLeafDictRange && b <= 25) {
        m = L.lo == 0 ? range_leaf_dispatch<true>(b, slot, lane, 0u, L.span, steps)
                      : range_leaf_dispatch<false>(b, slot, lane, (uint32_t)L.lo, L.span, steps);
      } else {
        const LaneDec dec = make_lane_dec(b, lane);
        m = b <= 25 ? eval_dict_leaf_loop<false>(L, slot, dec, b, steps) : eval_dict_leaf_loop<true>(L, slot, dec, b, steps);
      }
      break;
    }
    case kLeafRawRange: {
      const long long base_doc = (long long)tile * 64 * steps;
      const long long last = (long long)p.num_docs - 1;
      const uint32_t lo = (uint32_t)L.lo, span = L.span;
      m = 0;
#pragma unroll 8
      for (int k = 0; k < steps; ++k) {
        long long doc = base_doc + k * 64 + lane;
        doc = doc > last ? last : doc;
        const uint32_t v = __builtin_bswap32(*reinterpret_cast<const uint32_t*>(L.fwd + doc * 4));
        shift_in_le(m, v - lo, __builtin_amdgcn_readfirstlane(span));
      }
      m = __builtin_bitreverse32(m) >> (32 - steps);
      break;
    }
    case kLeafRawRange64:
    case kLeafRawRangeF64:
    case kLeafRawRangeF32: {
      // raw LONG / DOUBLE / FLOAT column (Long / Double / FloatRawValueBasedRangePredicateEvaluator): coalesced loads straight
      // from HBM; floating-point values are compared through f64_order_key (the host adjusts zero bounds so that -0.0 == 0.0)
      const long long base_doc = (long long)tile * 64 * steps;
      const long long last = (long long)p.num_docs - 1;
      const unsigned long long lo = ((unsigned long long)(uint32_t)L.lo_hi << 32) | (uint32_t)L.lo;
      const unsigned long long span = ((unsigned long long)(uint32_t)L.set_bytes << 32) | L.span;
      m = 0;
#pragma unroll 8
      for (int k = 0; k < steps; ++k) {
        long long doc = base_doc + k * 64 + lane;
        doc = doc > last ? last : doc;
        unsigned long long v;
        if (L.kind == kLeafRawRangeF32) v = (unsigned long long)f64_order_key((double)__uint_as_float(__builtin_bswap32(*reinterpret_cast<const uint32_t*>(L.fwd + doc * 4))));
        else {
          v = __builtin_bswap64(*reinterpret_cast<const unsigned long long*>(L.fwd + doc * 8));
          if (L.kind == kLeafRawRangeF64) v = (unsigned long long)f64_order_key(__longlong_as_double((long long)v));
        }
        m |= ((v - lo) <= span ? 1u : 0u) << k;
      }
      break;
    }
    case kLeafDocRange: {
      const uint32_t first = (uint32_t)tile * 64u * (uint32_t)steps + (uint32_t)lane;     // doc of step 0
      const uint32_t lo = (uint32_t)L.lo, span = L.span;
      m = 0;
#pragma unroll 8
      for (int k = 0; k < steps; ++k) m |= ((first + 64u * (uint32_t)k - lo) <= span ? 1u : 0u) << k;
      break;
    }
    default: {  // kLeafBitmap: doc-order 64-bit words staged in LDS; word k of the tile covers docs 64k..64k+63
      // Lanes 0..31 take the low dwords of words 0..31, lanes 32..63 the high dwords; a 32x32 bit-matrix transpose
      // inside each half-wave (5 ds_swizzle butterfly stages) then leaves in lane i the mask "bit k = bit i of word k".
      const uint32_t* words = reinterpret_cast<const uint32_t*>(bstage + L.lds_off);
      const int row = lane & 31;
      uint32_t x = row < steps ? words[2 * row + (lane >> 5)] : 0u;
#define PG_TRANSPOSE_STAGE(J, M)                                                              \
      {                                                                                         \
        const uint32_t pr = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, ((J) << 10) | 0x1F);  \
        const bool lo_role = (lane & (J)) == 0;                                                 \
        const uint32_t a = lo_role ? x : pr;                                                    \
        const uint32_t bsrc = lo_role ? pr : x;                                                 \
        const uint32_t t = ((a >> (J)) ^ bsrc) & (M);                                           \
        x ^= lo_role ? (t << (J)) : t;                                                          \
      }
      PG_TRANSPOSE_STAGE(16, 0x0000FFFFu)
      PG_TRANSPOSE_STAGE(8, 0x00FF00FFu)
      PG_TRANSPOSE_STAGE(4, 0x0F0F0F0Fu)
      PG_TRANSPOSE_STAGE(2, 0x33333333u)
      PG_TRANSPOSE_STAGE(1, 0x55555555u)
#undef PG_TRANSPOSE_STAGE
      m = x;
      break;
    }
  }
  return L.exclusive ? ~m : m;
}

// Stage every packed column stream of the tile that satisfies the (filter / aggregation-only) selection.
template <bool kDma>
__device__ __forceinline__ void stage_columns(const ScanParams& p, int tile, uint8_t* stage, int lane, bool filter_cols, bool agg_only_cols) {
  for (int c = 0; c < p.num_stage; ++c) {
    const DevStage& st = p.stage[c];
    const bool is_filter = st.in_filter != 0;
    if ((is_filter && filter_cols) || (!is_filter && agg_only_cols)) {
      const int tile_bytes = 8 * st.bits * p.tile_steps;
      stage_tile<kDma>(st.fwd + (long long)tile * tile_bytes, stage + st.slot_off, tile_bytes, lane);
    }
  }
}

// Stage the tile's words of every bitmap leaf (8 bytes per step, one dword per lane).
template <bool kDma>
__device__ __forceinline__ void stage_bitmaps(const ScanParams& p, int tile, uint8_t* bstage, int lane) {
  if (p.num_bitmap_leaves == 0) return;
  const int bytes = 8 * p.tile_steps;
  for (int l = 0; l < p.num_bitmap_leaves; ++l) {
    const uint8_t* src = reinterpret_cast<const uint8_t*>(p.bitmaps[l]) + (long long)tile * bytes;
    if (lane * 4 < bytes) {
      if constexpr (kDma) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + lane * 4), (lds_void_t*)(bstage + p.bitmap_lds_off[l]), 4, 0, 0);
      } else {
        *reinterpret_cast<uint32_t*>(bstage + p.bitmap_lds_off[l] + lane * 4) = *reinterpret_cast<const uint32_t*>(src + lane * 4);
      }
    }
  }
}

// Filter program (postfix) -> lane mask of the tile.  Nodes flagged kNodeExitIfZero sit on the root AND chain: when
// the running result is zero for the whole wavefront the tile is finished.  With lazy_columns the scan columns are
// staged only after the bitmap prefix (node lazy_node) left something (index-driven queries touch few tiles).
template <bool kDma>
__device__ __forceinline__ uint32_t eval_filter(const ScanParams& p, int tile, uint8_t* stage, const uint8_t* bstage, int lane, bool stage_agg_too) {
  if (p.num_nodes == 0) return 0xFFFFFFFFu;
  if (p.num_nodes == 1) return eval_leaf(p, p.nodes[0], tile, stage, bstage, lane);   // no mask stack for the common single-leaf filter
  MaskStack st;
#pragma unroll
  for (int i = 0; i < kStackDepth; ++i) st.v[i] = 0;
  st.sp = 0;
  for (int n = 0; n < p.num_nodes; ++n) {
    const DevNode& nd = p.nodes[n];     // self-contained 64-byte record: independent scalar loads off one base
    uint32_t top;
    if (nd.op == PG_FILTER_LEAF) {
      top = eval_leaf(p, nd, tile, stage, bstage, lane);
    } else if (nd.op == PG_FILTER_NOT) {
      top = ~st.pop();
    } else {
      top = st.pop();
      for (int c = 1; c < nd.num_children; ++c) {
        const uint32_t o = st.pop();
        top = nd.op == PG_FILTER_AND ? (top & o) : (top | o);
      }
    }
    if ((nd.flags & kNodeExitIfZero) && __builtin_amdgcn_ballot_w64((top & full_mask(p.tile_steps)) != 0u) == 0ull) return 0u;
    st.push(top);
    if (p.lazy_columns && n == p.lazy_node) {
      stage_columns<kDma>(p, tile, stage, lane, true, stage_agg_too);
      if constexpr (kDma) wait_vmem();
    }
  }
  return st.pop();
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

// Write the tile's docId bitmap (one doc-order 64-bit word per step) from the lane masks.
__device__ __forceinline__ void store_tile_bitmap(unsigned long long* out, int tile, uint32_t m, int lane, int steps) {
  unsigned long long mine = 0;
#pragma unroll 16
  for (int k = 0; k < steps; ++k) {
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(((m >> k) & 1u) != 0u);
    mine = (lane == k) ? bal : mine;
  }
  if (lane < steps) out[(long long)tile * steps + lane] = mine;
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

__device__ __forceinline__ void drain_queue(GatherQueue& gq, const __amdgpu_buffer_rsrc_t rsrc, int lane, long long& sum) {
  const int n = gq.count;
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

// Dense tiles (some lane has more than 12 of its docs matching): straight decode of every step with the bit width known
// at compile time (immediate LDS offsets), masked add of the plane field.  5 VALU ops per doc, against ~12 per matching
// doc for the set-bit walk.
template <int B>
__device__ __forceinline__ uint32_t plane_sum_dense(const uint8_t* slot, int lane, uint32_t m, int steps) {
  asm volatile("" : "+v"(lane));            // keep the per-width lane constants out of the tile loop's live set
  const LaneDec dec = make_lane_dec(B, lane);
  const uint8_t* base = slot + dec.off;
  uint32_t psum = 0;
  for (int kb = 0; kb < steps; kb += 8) {
    uint32_t w0[8], w1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t* p = reinterpret_cast<const uint32_t*>(base + (kb + j) * 8 * B);
      w0[j] = p[0];
      w1[j] = p[1];
    }
    const uint32_t mk = m >> kb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t d = __builtin_amdgcn_ubfe(__builtin_amdgcn_perm(w1[j], w0[j], dec.sel), dec.shift, (uint32_t)B);
      psum += d & (uint32_t)__builtin_amdgcn_sbfe((int)mk, j, 1);   // 0 or ~0
    }
  }
  return psum;
}

__device__ __forceinline__ uint32_t plane_sum_dense_dispatch(int b, const uint8_t* slot, int lane, uint32_t m, int steps) {
  switch (b) {
#define PG_CASE(B) case B: return plane_sum_dense<B>(slot, lane, m, steps);
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20)
    PG_CASE(21) PG_CASE(22) PG_CASE(23) PG_CASE(24) PG_CASE(25)
#undef PG_CASE
    default: return 0u;
  }
}

// Per-column aggregation of the matching docs of one staged tile: every lane walks the set bits of its mask, four at
// a time so that four LDS reads are in flight (a separate straight 32-step path for dense tiles cost 30 VGPRs and
// one wavefront per SIMD of occupancy, which lost more than it gained).
template <bool kWide>
__device__ __forceinline__ void agg_dict_column(const DevAggCol& ac, const uint8_t* slot, uint32_t m,
                                                int lane, long long& sum, int32_t& kmin, int32_t& kmax, GatherQueue& gq) {
  const int b = ac.bits;
  const LaneDec dec = make_lane_dec(b, lane);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ac.dict, 0, ac.dict_bytes, 0x00020000);
  const bool narrow_plane = ac.is_plane && ac.need_sum && b <= 27;
  if (narrow_plane && !ac.need_minmax && b <= 25 && __builtin_amdgcn_ballot_w64(__builtin_popcount(m) > 12) != 0ull) {
    // 32 fields of at most 25 bits: the per-tile lane sum fits 32 bits
    sum += (long long)plane_sum_dense_dispatch(b, slot, lane, m, __builtin_amdgcn_ballot_w64((m >> 16) != 0u) != 0ull ? 32 : 16);
    return;
  }
  uint32_t psum = 0;
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
      if (ac.is_plane) {
        // value plane: the decoded field IS (value - base); no dictionary, no gather
        if (narrow_plane) psum += active[j] ? d[j] : 0u;
        else if (ac.need_sum) sum += active[j] ? (long long)d[j] : 0ll;
      } else if (ac.need_sum) {
        const unsigned long long amask = __builtin_amdgcn_ballot_w64(active[j]);
        const uint32_t pos = (uint32_t)gq.count + __builtin_amdgcn_mbcnt_hi((uint32_t)(amask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)amask, 0u));
        if (active[j]) gq.q[pos] = d[j];
        gq.count += __builtin_popcountll(amask);
        if (gq.count > gq.cap - 64) drain_queue(gq, rsrc, lane, sum);
      }
    }
  }
  sum += (long long)psum;
}

__device__ __forceinline__ void agg_raw_column(const DevAggCol& col, int num_docs, int tile, int steps, uint32_t m,
                                               int lane, long long& sum, int32_t& kmin, int32_t& kmax) {
  const long long base_doc = (long long)tile * 64 * steps;
  const long long last = (long long)num_docs - 1;
#pragma unroll 8
  for (int k = 0; k < steps; ++k) {
    long long doc = base_doc + k * 64 + lane;
    doc = doc > last ? last : doc;
    const int32_t v = (int32_t)__builtin_bswap32(*reinterpret_cast<const uint32_t*>(col.fwd + doc * 4));
    const bool match = ((m >> k) & 1u) != 0u;
    sum += match ? (long long)v : 0ll;
    kmin = (match && v < kmin) ? v : kmin;
    kmax = (match && v > kmax) ? v : kmax;
  }
}

// ---- typed values (LONG / FLOAT / DOUBLE stored types) ----
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ long long wave_min_i64(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const long long t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ long long wave_max_i64(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const long long t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return v;
}
// SUM of a dictionary column whose values are 8-byte entries (LONG with a wide range, FLOAT / DOUBLE): set-bit walk, four
// decodes then four 64-bit dictionary gathers in flight.  (MIN / MAX never come here: they run on dictIds.)
template <bool kWide>
__device__ __forceinline__ void agg_dict_column_wide(const DevAggCol& ac, const uint8_t* slot, uint32_t m, int lane,
                                                     long long& isum, double& fsum, int32_t& kmin, int32_t& kmax) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const int b = ac.bits;
  const LaneDec dec = make_lane_dec(b, lane);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ac.dict, 0, ac.dict_bytes, 0x00020000);
  uint32_t rem = m;
  for (;;) {
    if (__builtin_amdgcn_ballot_w64(rem != 0u) == 0ull) break;
    bool active[4];
    uint32_t d[4];
    u32x2 w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      active[j] = rem != 0u;
      const int k = active[j] ? __builtin_ctz(rem) : 0;
      rem &= rem - 1u;
      d[j] = decode_step<kWide>(slot, dec, k, b);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, active[j] ? d[j] * 8u : 0xFFFFFFFFu, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long bits = (long long)(((unsigned long long)w[j].y << 32) | (unsigned long long)w[j].x);   // 0 when inactive
      if (ac.need_sum) {
        // LONG: exact wrapping int64 sum, plus a double image the host falls back to when the int64 sum can overflow
        if (ac.vkind == kValI64) { isum += bits; fsum += (double)bits; }
        else fsum += __longlong_as_double(bits);     // +0.0 when inactive
      }
      if (ac.need_minmax) {
        const int32_t key = (int32_t)d[j];
        kmin = (active[j] && key < kmin) ? key : kmin;
        kmax = (active[j] && key > kmax) ? key : kmax;
      }
    }
  }
}

// Raw LONG / FLOAT / DOUBLE column: one coalesced load per step straight from HBM (FixedByteChunkSVForwardIndexReader
// .getLong / getFloat / getDouble: big-endian value at rawDataStart + entrySize * docId).
__device__ __forceinline__ void agg_raw_column_typed(const DevAggCol& ac, int num_docs, int tile, int steps, uint32_t m, int lane,
                                                     long long& isum, double& fsum, long long& kmin, long long& kmax) {
  const long long base_doc = (long long)tile * 64 * steps;
  const long long last = (long long)num_docs - 1;
#pragma unroll 4
  for (int k = 0; k < steps; ++k) {
    long long doc = base_doc + k * 64 + lane;
    doc = doc > last ? last : doc;
    const bool match = ((m >> k) & 1u) != 0u;
    long long key;
    if (ac.vkind == kValI64) {
      const long long v = (long long)__builtin_bswap64(*reinterpret_cast<const unsigned long long*>(ac.fwd + doc * 8));
      isum += match ? v : 0ll;
      fsum += match ? (double)v : 0.0;
      key = v;
    } else {
      double v;
      if (ac.vkind == kValF32) v = (double)__uint_as_float(__builtin_bswap32(*reinterpret_cast<const uint32_t*>(ac.fwd + doc * 4)));
      else v = __longlong_as_double((long long)__builtin_bswap64(*reinterpret_cast<const unsigned long long*>(ac.fwd + doc * 8)));
      fsum += match ? v : 0.0;
      key = f64_order_key(v);
    }
    kmin = (match && key < kmin) ? key : kmin;
    kmax = (match && key > kmax) ? key : kmax;
  }
}

// Reduce the per-workgroup partials into partials[num_blocks] (one record).  The order of the double additions is fixed by
// the grid size, so a given launch geometry always returns the same floating-point sum.
__device__ __forceinline__ void partial_identity(BlockPartial& acc) {
  acc.count = 0;
  acc.flags = 0;
  acc.entries = 0;
  acc.stamp = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) acc.cyc[c] = 0;
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a) {
    acc.sum[a] = 0; acc.kmin[a] = 0x7FFFFFFF; acc.kmax[a] = (int32_t)0x80000000;
    acc.fsum[a] = 0.0; acc.kmin64[a] = 0x7FFFFFFFFFFFFFFFll; acc.kmax64[a] = (long long)0x8000000000000000ull;
  }
}
__device__ __forceinline__ void partial_merge(BlockPartial& acc, const BlockPartial& b) {
  acc.count += b.count;
  acc.flags |= b.flags;
  acc.entries += b.entries;
#pragma unroll
  for (int c = 0; c < 4; ++c) acc.cyc[c] += b.cyc[c];
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a) {
    acc.sum[a] += b.sum[a];
    acc.kmin[a] = b.kmin[a] < acc.kmin[a] ? b.kmin[a] : acc.kmin[a];
    acc.kmax[a] = b.kmax[a] > acc.kmax[a] ? b.kmax[a] : acc.kmax[a];
    acc.fsum[a] += b.fsum[a];
    acc.kmin64[a] = b.kmin64[a] < acc.kmin64[a] ? b.kmin64[a] : acc.kmin64[a];
    acc.kmax64[a] = b.kmax64[a] > acc.kmax64[a] ? b.kmax64[a] : acc.kmax64[a];
  }
}

// partial_merge over the fields a query uses only (FoldFields below): what thread 0 of every workgroup does with its waves' records at
// the end of a scan -- 31 fields for a COUNT / one-column SUM that uses six was ~0.5 us per record on the tail of every query.
struct FoldFields;
__device__ __forceinline__ void partial_merge_fields(BlockPartial& acc, const BlockPartial& b, int slots, bool typed, bool cycles) {
  acc.count += b.count;
  acc.flags |= b.flags;
  acc.entries += b.entries;
  if (cycles) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc.cyc[c] += b.cyc[c];
  }
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a) {
    if (a >= slots) continue;
    acc.sum[a] += b.sum[a];
    acc.kmin[a] = b.kmin[a] < acc.kmin[a] ? b.kmin[a] : acc.kmin[a];
    acc.kmax[a] = b.kmax[a] > acc.kmax[a] ? b.kmax[a] : acc.kmax[a];
    if (typed) {
      acc.fsum[a] += b.fsum[a];
      acc.kmin64[a] = b.kmin64[a] < acc.kmin64[a] ? b.kmin64[a] : acc.kmin64[a];
      acc.kmax64[a] = b.kmax64[a] > acc.kmax64[a] ? b.kmax64[a] : acc.kmax64[a];
    }
  }
}

// What a fold has to look at: `slots` aggregation slots (BlockPartial.sum / kmin / kmax [0 .. slots)), the typed fields (fsum / kmin64 /
// kmax64) and the PG_CFG_PROFILE_WAVES cycle counters only when the query uses them.  A record has thirty 64-bit-reduced fields; a
// COUNT / one-column SUM needs five of them, and the wave-level reductions are what a one-workgroup fold spends its time on.
struct FoldFields { int slots; bool typed; bool cycles; unsigned long long stamp; };      // stamp: what every record folded must carry (kCoherent folds)
template <typename P>
__device__ __forceinline__ FoldFields fold_fields_of(const P& p) { return FoldFields{p.fold_slots, p.fold_typed != 0, p.profile != 0, p.host_seq}; }

// Every lane's record folded over the wave (all lanes return the same record).
__device__ __forceinline__ void partial_wave_reduce(BlockPartial& acc, const FoldFields ff) {
  acc.count = (unsigned long long)wave_sum_i64((long long)acc.count);
  acc.entries = (unsigned long long)wave_sum_i64((long long)acc.entries);
  acc.flags = (__builtin_amdgcn_ballot_w64((acc.flags & kPartialHistAlarm) != 0ull) != 0ull ? kPartialHistAlarm : 0ull) |      // two-bit vocabulary
              (__builtin_amdgcn_ballot_w64((acc.flags & kPartialStale) != 0ull) != 0ull ? kPartialStale : 0ull);
  if (ff.cycles) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc.cyc[c] = (unsigned long long)wave_sum_i64((long long)acc.cyc[c]);
  }
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a) {
    if (a >= ff.slots) continue;
    acc.sum[a] = wave_sum_i64(acc.sum[a]);
    acc.kmin[a] = wave_min_i32(acc.kmin[a]);
    acc.kmax[a] = wave_max_i32(acc.kmax[a]);
    if (ff.typed) {
      acc.fsum[a] = wave_sum_f64(acc.fsum[a]);
      acc.kmin64[a] = wave_min_i64(acc.kmin64[a]);
      acc.kmax64[a] = wave_max_i64(acc.kmax64[a]);
    }
  }
}

// Fold of `num_records` per-workgroup records by ONE workgroup: thread t takes records t, t + blockDim, ..., then the wave, then the
// waves through `red` (>= blockDim / 64 records of LDS).  The order of the double additions depends on the launch geometry only.
// Returns the folded record in thread 0.  (Fields outside `ff` keep their identities.)
//
// kCoherent: the records were stored by other workgroups of THIS launch with agent-scope (sc1, write-through) stores, and are read with
// agent-scope loads (sc1: past the L1, served by the L2 / memory) -- MI355X_MICROARCH.md: "sc1 loads may replace the acquire only when
// the producer stored sc1".  The agent-scope acquire fence this replaces (buffer_inv sc1) cost the folding workgroup ~1.7 us on the
// critical path of every query.  Without it (finalize_partials_kernel: a kernel boundary lies between writers and reader) plain loads.
template <bool kCoherent>
__device__ __forceinline__ unsigned long long partial_word(const BlockPartial* rec, size_t byte_off) {
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const uint8_t*>(rec) + byte_off);
  if constexpr (kCoherent) return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *w;
}
// The records folded are partials[first + j * stride], j < num_records.
template <bool kCoherent = false>
__device__ __forceinline__ BlockPartial fold_partials(const BlockPartial* partials, int first, int stride, int num_records, BlockPartial* red, const FoldFields ff) {
  static_assert(kMaxAggCols % 2 == 0, "kmin / kmax are read as pairs");
  BlockPartial acc;
  partial_identity(acc);
  for (int i = threadIdx.x; i < num_records; i += blockDim.x) {
    const BlockPartial* b = partials + first + (long long)i * stride;
    acc.count += partial_word<kCoherent>(b, offsetof(BlockPartial, count));
    acc.flags |= partial_word<kCoherent>(b, offsetof(BlockPartial, flags));
    // Records written by other workgroups of this launch are ordered against the arrival counter by the hardware's write-through
    // stores, not by a release / acquire pair of the memory model: a record that is not this launch's must not become an answer.
    if constexpr (kCoherent) acc.flags |= partial_word<kCoherent>(b, offsetof(BlockPartial, stamp)) != ff.stamp ? kPartialStale : 0ull;
    acc.entries += partial_word<kCoherent>(b, offsetof(BlockPartial, entries));
    if (ff.cycles) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc.cyc[c] += partial_word<kCoherent>(b, offsetof(BlockPartial, cyc) + 8 * c);
    }
#pragma unroll
    for (int a = 0; a < kMaxAggCols; a += 2) {
      if (a >= ff.slots) continue;
      // kmin / kmax of slots a, a + 1 are one 8-byte word each (an unused odd slot holds the identities)
      const unsigned long long mins = partial_word<kCoherent>(b, offsetof(BlockPartial, kmin) + 4 * a), maxs = partial_word<kCoherent>(b, offsetof(BlockPartial, kmax) + 4 * a);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int32_t bmin = (int32_t)(uint32_t)(mins >> (32 * h)), bmax = (int32_t)(uint32_t)(maxs >> (32 * h));
        acc.kmin[a + h] = bmin < acc.kmin[a + h] ? bmin : acc.kmin[a + h];
        acc.kmax[a + h] = bmax > acc.kmax[a + h] ? bmax : acc.kmax[a + h];
      }
    }
#pragma unroll
    for (int a = 0; a < kMaxAggCols; ++a) {
      if (a >= ff.slots) continue;
      acc.sum[a] += (long long)partial_word<kCoherent>(b, offsetof(BlockPartial, sum) + 8 * a);
      if (ff.typed) {
        acc.fsum[a] += __longlong_as_double((long long)partial_word<kCoherent>(b, offsetof(BlockPartial, fsum) + 8 * a));
        const long long bmin = (long long)partial_word<kCoherent>(b, offsetof(BlockPartial, kmin64) + 8 * a), bmax = (long long)partial_word<kCoherent>(b, offsetof(BlockPartial, kmax64) + 8 * a);
        acc.kmin64[a] = bmin < acc.kmin64[a] ? bmin : acc.kmin64[a];
        acc.kmax64[a] = bmax > acc.kmax64[a] ? bmax : acc.kmax64[a];
      }
    }
  }
  partial_wave_reduce(acc, ff);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();                 // `red` may still hold the waves' own records
  if (lane == 0) red[w] = acc;
  __syncthreads();
  BlockPartial t = red[0];
  if (threadIdx.x == 0)
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) partial_merge_fields(t, red[i], ff.slots, ff.typed, ff.cycles);
  return t;
}

// A record leaves its workgroup as ONE store instruction: the record sits in LDS (`rec`) and lane i of wave 0 stores its i-th 8-byte
// word -- 240 contiguous bytes, a handful of write-through requests, where lane 0 storing the thirty words one after the other was
// thirty (a scalar sc1 store is a fabric write of its own: MI355X_MICROARCH.md).  Called by every lane of wave 0 after lane 0 wrote
// `rec` (LDS operations of one wave execute in order).  kSystem: pinned host memory (sc0 sc1), else device memory read by another
// workgroup (sc1).  Returns once the stores are acknowledged (s_waitcnt vmcnt(0) through inline asm: the guide's compiler hazard).
constexpr int kRecordWords = (int)(sizeof(BlockPartial) / 8);
static_assert(sizeof(BlockPartial) % 8 == 0 && kRecordWords <= 64, "a record is stored as 8-byte words, one per lane of a wave");
template <bool kSystem>
__device__ __forceinline__ void store_record_by_wave0(const BlockPartial* rec, BlockPartial* out) {
  __builtin_amdgcn_wave_barrier();
  const int lane = threadIdx.x;              // wave 0
  if (lane < kRecordWords) {
    const unsigned long long w = reinterpret_cast<const unsigned long long*>(rec)[lane];
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(out) + lane;
    if constexpr (kSystem) __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// The query's record into pinned host memory, then the sequence number the host polls.  Every word is a system-scope (sc0 sc1,
// write-through) store: nothing of the record is left dirty in the L2, so "the record before the sequence number" needs no release
// fence (buffer_wbl2 sc0 sc1 writes back EVERY dirty line of the XCD's L2 -- a bitmap the same kernel wrote, say), only the stores'
// acknowledgements.  Nothing follows the sequence number: the two system fences this replaces were ~3 us at the end of every query's
// kernel.  Wave 0, `rec` in LDS (store_record_by_wave0).
__device__ __forceinline__ void store_host_record_by_wave0(HostRecord* host_out, const BlockPartial* rec, unsigned long long seq) {
  static_assert(offsetof(HostRecord, partial) == 0, "the record leads");
  store_record_by_wave0<true>(rec, &host_out->partial);
  if (threadIdx.x == 0) __hip_atomic_store(&host_out->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// `host_out`: pinned, device-mapped host memory -- the folded record goes straight to the host, no copy command follows.
// (The separate launch; the scan kernels fold their records themselves when ScanParams.done_counter is set, see publish_block_partial.)
static __global__ __launch_bounds__(kBlockThreads) void finalize_partials_kernel(BlockPartial* partials, int num_blocks, HostRecord* host_out, unsigned long long seq,
                                                                                 int slots, int typed, int cycles) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  const BlockPartial t = fold_partials(partials, 0, 1, num_blocks, red, FoldFields{slots, typed != 0, cycles != 0, 0ull});
  if (threadIdx.x == 0) {
    if (host_out) red[0] = t;
    else partials[num_blocks] = t;
  }
  if (host_out && threadIdx.x < 64) store_host_record_by_wave0(host_out, &red[0], seq);
}

// The end of every scan kernel: the waves' records are in red[0 .. waves_per_block) (written by each wave's lane 0, __syncthreads()
// done).  Thread 0 merges them into the workgroup's record.  With ScanParams.done_counter set, the workgroup then ARRIVES, and the
// one whose arrival completes the count -- every other workgroup's record is in memory by then -- folds all of them and writes the
// query's result: what finalize_partials_kernel does in a launch of its own (a dependent kernel boundary, the launch, and a
// one-workgroup kernel on an otherwise idle chip -- half the device time of a 10 M-row segment).
// Inter-workgroup visibility is MI355X_MICROARCH.md's R1 form: the record is stored WRITE-THROUGH (8-byte agent-scope stores = sc1: they
// reach memory, no L2 write-back -- a release fence per workgroup was measured at +30 us on a 1024-workgroup grid, 128 L2 write-backs
// queueing per XCD), the writing lane drains vmcnt, arrives with a relaxed agent-scope atomic; the consumer takes ONE agent-scope
// acquire after seeing the count complete, __syncthreads(), then plain loads.  Arrivals are sharded over eight counters (blockIdx & 7,
// 128 bytes apart) so that a grid finishing at once does not serialise 1024 atomics on one word; the workgroup completing a shard
// arrives on the ninth.  `flag` is one dword of LDS the caller provides (scan_hist_kernel keeps its counters as the only static LDS object).
constexpr int kFoldShards = 8, kFoldStride = 32;      // counters are kFoldStride dwords apart; kFoldShards * kFoldStride is the top counter
constexpr int kFoldExtraRecords = 1 + kFoldShards;    // behind the workgroups' records: the query's record, then one record per shard
constexpr int kFoldOneCounterMax = 64;               // grids up to this many workgroups arrive on one counter (no shard hand-off)
constexpr int kFoldOneLevel = 1536;                   // grids up to this many workgroups are folded by one workgroup in one level (publish_block_partial)
// `block_index` of `num_blocks`: the workgroup's place among those that work on this ScanParams (the whole grid, or one query's share of
// a batch launch -- scan_private_batch_kernel).
template <typename P>
__device__ __forceinline__ void publish_block_partial(const P& p, BlockPartial* red, int waves_per_block, uint32_t* flag, uint32_t block_index,
                                                      uint32_t num_blocks) {
  const bool arrive = p.done_counter != nullptr && num_blocks > 1u;
  // Small grids arrive on ONE counter: with shards the last arriver of a shard arrives a second time on the top counter -- a second
  // dependent device-scope atomic at the very end of the query.  Measured (profiles/r3/c1_probe_one_arrival_counter.jsonl): 25 workgroups
  // 14.2 -> 13.5 us, 122 equal, 489 +2 us, 1024 +5 us -- same-address device-scope atomics retire at ~5 ns apiece when a grid finishes at
  // once, which is what the shards are for.
  const bool one_counter = p.fold_one_counter != 0 && num_blocks <= (uint32_t)kFoldOneCounterMax;
  const uint32_t shard = one_counter ? 0u : block_index & (kFoldShards - 1);
  const uint32_t in_shard = one_counter ? num_blocks : (num_blocks + (kFoldShards - 1) - shard) / kFoldShards;          // workgroups b with (b & 7) == shard
  const uint32_t shards = num_blocks < (uint32_t)kFoldShards ? num_blocks : (uint32_t)kFoldShards;
  if (threadIdx.x < 64) {                                  // wave 0
    if (threadIdx.x == 0) {
      BlockPartial acc = red[0];
      for (int w = 1; w < waves_per_block; ++w) partial_merge_fields(acc, red[w], p.fold_slots, p.fold_typed != 0, p.profile != 0);
      acc.stamp = p.host_seq;
      if (p.done_counter == nullptr) p.partials[block_index] = acc;
      else if (!arrive && p.host_out == nullptr) p.partials[1] = acc;
      else red[0] = acc;
    }
    if (p.done_counter != nullptr && !arrive) {
      // the only workgroup: its record IS the query's (a segment of a few thousand docs: no store / arrive / fold round trips at all)
      if (p.host_out) store_host_record_by_wave0(p.host_out, &red[0], p.host_seq);
    } else if (arrive) {
      store_record_by_wave0<false>(&red[0], &p.partials[block_index]);
    }
    if (threadIdx.x == 0) {
      // a workgroup whose arrival completes its shard goes on; on a small grid it arrives on the top counter right away and only the
      // workgroup completing THAT goes on (one flag write per phase: the other waves read it behind the barrier below)
      uint32_t go = (arrive && __hip_atomic_fetch_add(p.done_counter + shard * kFoldStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == in_shard) ? 1u : 0u;
      if (go != 0u && !one_counter && num_blocks <= (uint32_t)kFoldOneLevel)
        go = __hip_atomic_fetch_add(p.done_counter + kFoldShards * kFoldStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == shards ? 1u : 0u;
      *flag = go;
    }
  }
  __syncthreads();
  if (*flag == 0u) return;
  const FoldFields ff = fold_fields_of(p);
  BlockPartial t;
  if (num_blocks <= (uint32_t)kFoldOneLevel) {
    // every workgroup has arrived: fold every workgroup's record (at most six per thread) and publish the query's result
    t = fold_partials<true>(p.partials, 0, 1, (int)num_blocks, red, ff);
  } else {
    // Larger grids, two levels of one pass each: the workgroup that completed a shard folds the shard's records (every eighth of the
    // grid's, one per thread up to 2048 workgroups -- ONE round of loads where one workgroup folding the whole grid walked eight records
    // per thread one after the other, 12 us at the end of a 1 B-row scan), leaves the shard's record behind the grid's and arrives on
    // the top counter; the workgroup completing that folds the eight shard records.  (On small grids the second hand-off costs more
    // than the shorter walk saves: 123 workgroups 13.7 -> 16.2 us.)
    t = fold_partials<true>(p.partials, (int)shard, kFoldShards, (int)in_shard, red, ff);
    if (threadIdx.x == 0) { t.stamp = p.host_seq; red[0] = t; }
    if (threadIdx.x < 64) {
      store_record_by_wave0<false>(&red[0], &p.partials[num_blocks + 1u + shard]);
      if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(p.done_counter + kFoldShards * kFoldStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == shards ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u) return;
    t = fold_partials<true>(p.partials, (int)num_blocks + 1, 1, (int)shards, red, ff);
  }
  if (threadIdx.x == 0) {
    for (int c = 0; c <= kFoldShards; ++c) __hip_atomic_store(p.done_counter + c * kFoldStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the context's next launch
    if (p.host_out) red[0] = t;
    else p.partials[num_blocks] = t;
  }
  if (p.host_out && threadIdx.x < 64) store_host_record_by_wave0(p.host_out, &red[0], p.host_seq);
}
template <typename P>
__device__ __forceinline__ void publish_block_partial(const P& p, BlockPartial* red, int waves_per_block, uint32_t* flag) {
  publish_block_partial(p, red, waves_per_block, flag, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// Fused scan -> filter -> aggregate.  One wavefront per tile, tiles dealt round-robin over a persistent grid.
// ------------------------------------------------------------------------------------------------
// kTyped: some aggregated column has 8-byte / floating-point values (ValueKind != kValI32); adds double sums and 64-bit
// min / max keys per slot.  The all-INT instantiations carry none of that.
template <bool kDma, int kAggSlots, bool kTyped = false>
__global__ __launch_bounds__(kBlockThreads) void scan_agg_kernel(const ScanParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  uint8_t* wave_lds = smem + wave_in_block * p.wave_lds_bytes;
  const int total_waves = gridDim.x * waves_per_block;
  const int steps = p.tile_steps;

  unsigned long long count = 0;
  long long sum[kAggSlots];
  int32_t kmin[kAggSlots], kmax[kAggSlots];
#pragma unroll
  for (int a = 0; a < kAggSlots; ++a) { sum[a] = 0; kmin[a] = 0x7FFFFFFF; kmax[a] = (int32_t)0x80000000; }
  double fsum[kTyped ? kAggSlots : 1];
  long long kmin64[kTyped ? kAggSlots : 1], kmax64[kTyped ? kAggSlots : 1];
#pragma unroll
  for (int a = 0; a < (kTyped ? kAggSlots : 1); ++a) { fsum[a] = 0.0; kmin64[a] = 0x7FFFFFFFFFFFFFFFll; kmax64[a] = (long long)0x8000000000000000ull; }

  GatherQueue gq;
  gq.q = reinterpret_cast<uint32_t*>(wave_lds + p.queue_off);
  gq.cap = p.queue_cap;
  gq.count = 0;
  // the queue may stay filled across tiles only when exactly one column is summed through its dictionary
  int num_sum_cols = 0;
#pragma unroll
  for (int a = 0; a < kAggSlots; ++a)
    num_sum_cols += (a < p.num_agg_cols && p.agg_cols[a].need_sum && !p.agg_cols[a].is_raw && !p.agg_cols[a].is_plane && p.agg_cols[a].vkind == kValI32) ? 1 : 0;

  unsigned long long cyc_wait = 0, cyc_filter = 0, cyc_agg = 0;
  const unsigned long long cyc_start = p.profile ? __builtin_amdgcn_s_memtime() : 0ull;
  const bool eager = p.lazy_columns == 0;   // stage the scan columns together with the bitmap words
  uint8_t* bitmap_lds = wave_lds + p.bitmap_off;   // two tiny buffers: the next tile's posting words are always prefetched
  bool hot = false;
  bool cur_has_agg = false;
  int buf = 0, bbuf = 0;
  int tile = blockIdx.x * waves_per_block + wave_in_block;
  if (tile < p.num_tiles) {
    stage_bitmaps<kDma>(p, tile, bitmap_lds, lane);
    if (eager) stage_columns<kDma>(p, tile, wave_lds, lane, true, false);
  }
  for (; tile < p.num_tiles; tile += total_waves) {
    const ScanParams& q = p;
    uint8_t* cur = wave_lds + buf * q.stage_bytes;
    const uint8_t* bcur = bitmap_lds + bbuf * q.bitmap_bytes;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (q.profile) t0 = __builtin_amdgcn_s_memtime();
    if constexpr (kDma) wait_vmem();                       // the current tile has landed
    if (q.profile) t1 = __builtin_amdgcn_s_memtime();
    const int next = tile + total_waves;
    const bool next_has_agg = eager && hot && q.speculate != 0;
    if (next < q.num_tiles) {
      stage_bitmaps<kDma>(q, next, bitmap_lds + (bbuf ^ 1) * q.bitmap_bytes, lane);
      if (q.double_buffer && eager) stage_columns<kDma>(q, next, wave_lds + (buf ^ 1) * q.stage_bytes, lane, true, next_has_agg);
    }
    uint32_t m = eval_filter<kDma>(q, tile, cur, bcur, lane, false);
    m &= valid_lane_mask(q.num_docs, tile, lane, steps);
    if (q.out_bitmap) store_tile_bitmap(q.out_bitmap, tile, m, lane, steps);
    count += (unsigned)__builtin_popcount(m);
    const bool any = __builtin_amdgcn_ballot_w64(m != 0u) != 0ull;
    hot = any;
    if (q.profile) t2 = __builtin_amdgcn_s_memtime();
    if (any && q.num_agg_cols > 0) {
      if (!cur_has_agg) {
        stage_columns<kDma>(q, tile, cur, lane, false, true);
        if constexpr (kDma) wait_vmem();
      }
#pragma unroll
      for (int a = 0; a < kAggSlots; ++a) {
        if (a < q.num_agg_cols) {
          const DevAggCol& ac = q.agg_cols[a];
          if (kTyped && ac.vkind != kValI32) {
            if (ac.is_raw) agg_raw_column_typed(ac, q.num_docs, tile, steps, m, lane, sum[a], fsum[kTyped ? a : 0], kmin64[kTyped ? a : 0], kmax64[kTyped ? a : 0]);
            else if (ac.bits <= 25) agg_dict_column_wide<false>(ac, cur + ac.slot_off, m, lane, sum[a], fsum[kTyped ? a : 0], kmin[a], kmax[a]);
            else agg_dict_column_wide<true>(ac, cur + ac.slot_off, m, lane, sum[a], fsum[kTyped ? a : 0], kmin[a], kmax[a]);
          } else if (ac.is_raw) {
            agg_raw_column(ac, q.num_docs, tile, steps, m, lane, sum[a], kmin[a], kmax[a]);
          } else {
            if (ac.bits <= 25) agg_dict_column<false>(ac, cur + ac.slot_off, m, lane, sum[a], kmin[a], kmax[a], gq);
            else agg_dict_column<true>(ac, cur + ac.slot_off, m, lane, sum[a], kmin[a], kmax[a], gq);
            if (num_sum_cols > 1 && ac.need_sum && !ac.is_plane && gq.count > 0) {
              const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ac.dict, 0, ac.dict_bytes, 0x00020000);
              drain_queue(gq, rsrc, lane, sum[a]);
            }
          }
        }
      }
    }
    if (q.profile) {
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      cyc_wait += t1 - t0; cyc_filter += t2 - t1; cyc_agg += t3 - t2;
    }
    bbuf ^= 1;
    if (q.double_buffer) {
      cur_has_agg = next_has_agg;
      buf ^= 1;
    } else {
      // single column buffer: the next tile's columns are staged only now that this one is fully consumed
      if (eager && next < q.num_tiles) stage_columns<kDma>(q, next, wave_lds, lane, true, next_has_agg);
      cur_has_agg = next_has_agg;
    }
  }

  // final drain of the carried queue (single dictionary-summed column)
  if (num_sum_cols == 1 && gq.count > 0) {
#pragma unroll
    for (int a = 0; a < kAggSlots; ++a) {
      if (a < p.num_agg_cols && p.agg_cols[a].need_sum && !p.agg_cols[a].is_raw && !p.agg_cols[a].is_plane && p.agg_cols[a].vkind == kValI32) {
        const DevAggCol& col = p.agg_cols[a];
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)col.dict, 0, col.dict_bytes, 0x00020000);
        drain_queue(gq, rsrc, lane, sum[a]);
      }
    }
  }

  // wave reduce -> block reduce (LDS) -> one partial record per workgroup
  __syncthreads();   // all waves are done with their staging slots; reuse the start of LDS for the reduction
  BlockPartial* red = reinterpret_cast<BlockPartial*>(smem);
  BlockPartial mine;
  mine.flags = 0ull;
  mine.entries = 0ull;
  mine.count = (unsigned long long)wave_sum_i64((long long)count);
#pragma unroll
  for (int a = 0; a < kMaxAggCols; ++a) {
    if (a < kAggSlots) {
      mine.sum[a] = wave_sum_i64(sum[a < kAggSlots ? a : 0]);
      mine.kmin[a] = wave_min_i32(kmin[a < kAggSlots ? a : 0]);
      mine.kmax[a] = wave_max_i32(kmax[a < kAggSlots ? a : 0]);
    } else {
      mine.sum[a] = 0; mine.kmin[a] = 0x7FFFFFFF; mine.kmax[a] = (int32_t)0x80000000;
    }
    if (kTyped && a < kAggSlots) {
      mine.fsum[a] = wave_sum_f64(fsum[(kTyped && a < kAggSlots) ? a : 0]);
      mine.kmin64[a] = wave_min_i64(kmin64[(kTyped && a < kAggSlots) ? a : 0]);
      mine.kmax64[a] = wave_max_i64(kmax64[(kTyped && a < kAggSlots) ? a : 0]);
    } else {
      mine.fsum[a] = 0.0; mine.kmin64[a] = 0x7FFFFFFFFFFFFFFFll; mine.kmax64[a] = (long long)0x8000000000000000ull;
    }
  }
  mine.cyc[0] = cyc_wait; mine.cyc[1] = cyc_filter; mine.cyc[2] = cyc_agg;
  mine.cyc[3] = p.profile ? __builtin_amdgcn_s_memtime() - cyc_start : 0ull;
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  // (untyped instantiations leave fsum / kmin64 / kmax64 at their identities: merging them is a no-op)
  publish_block_partial(p, red, waves_per_block, reinterpret_cast<uint32_t*>(red + waves_per_block));
}

// ------------------------------------------------------------------------------------------------
// Fused scan -> filter -> group-by aggregate with a direct-indexed group table
// (DictionaryBasedGroupKeyGenerator.ArrayBasedHolder: groupId = sum dictId_j * prod_{k<j} card_k).
// Up to 16 wavefronts of a workgroup share one LDS partial table (LDS atomics), flushed with global atomics.
// ------------------------------------------------------------------------------------------------
// Group table (LDS copy and global copy share the layout): count[G] (u64) then acc[a][G] (i64).
// The LDS copy uses 32-bit atomics where the per-workgroup value provably fits: a workgroup sees fewer than 2^31 docs,
// and MIN/MAX keys are int32 (dictIds, plane offsets or raw values); only SUM needs 64 bits.  The low dword of each
// 64-bit slot is used and the flush widens it.
template <bool kLds>
__device__ __forceinline__ void group_count(unsigned long long* cnt, uint32_t g) {
  if constexpr (kLds) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(cnt + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_add(cnt + g, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool kLds>
__device__ __forceinline__ void group_sum(long long* slot, long long v) {
  __hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, kLds ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT);
}
// (Tried and dropped, profiles/r3/README.md: reading the slot first and issuing the MIN / MAX atomic only where it would change -- a
//  running extreme moves rarely -- made C3 SLOWER, 0.98 -> 1.15 ms per 1 B rows: the read's result has to come back before the exec
//  mask of the atomic is known, so every doc pays an LDS round trip where the unconditional atomic was fire-and-forget.)
template <bool kLds>
__device__ __forceinline__ void group_min(long long* slot, int32_t v) {
  if constexpr (kLds) __hip_atomic_fetch_min(reinterpret_cast<int32_t*>(slot), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_min(slot, (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool kLds>
__device__ __forceinline__ void group_max(long long* slot, int32_t v) {
  if constexpr (kLds) __hip_atomic_fetch_max(reinterpret_cast<int32_t*>(slot), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_max(slot, (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One term of the raw key, dictId * multiplier (DictionaryBasedGroupKeyGenerator.java:437-445).  Up to 2^24 slots both factors
// fit the full-rate 24-bit multiply; kWide (GroupParams::wide_keys: its own instantiations of the HBM-table kernels) takes the
// 32-bit one.  (A run-time switch between the two inside the unrolled key loops made group_private_kernel<false> fault on the
// device -- one more reason for the template than the spare multiply.)
template <bool kWide>
__device__ __forceinline__ uint32_t key_term(uint32_t d, uint32_t mult) {
  if constexpr (kWide) return d * mult;
  else return __umul24(d, mult);
}

// Aggregate four docs per lane (steps k[0..3]) into the group table.  kAllActive: every lane owns four real matching
// docs (no exec masking around the atomics).
template <bool kLds, bool kAllActive, bool kWide = false>
__device__ __forceinline__ void group_process4(const GroupParams& gp, const uint8_t* stage, int tile, int lane, const int (&k)[4], const bool (&active)[4],
                                               unsigned long long* t_cnt, long long* t_acc) {
  const ScanParams& p = gp.scan;
  const int G = gp.num_groups;
  uint32_t g[4] = {0u, 0u, 0u, 0u};
  for (int c = 0; c < gp.num_group_cols; ++c) {
    const DevGroupKey& key = gp.group_keys[c];
    const int b = key.bits;
    const LaneDec dec = make_lane_dec(b, lane);
    const uint8_t* slot = stage + key.slot_off;
    const uint32_t mult = (uint32_t)key.mult;
    if (b <= 25) {
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] += key_term<kWide>(decode_step<false>(slot, dec, k[j], b), mult);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] += key_term<kWide>(decode_step<true>(slot, dec, k[j], b), mult);
    }
  }
  const bool packed = kLds && gp.packed_agg >= 0;
  if (!packed) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kAllActive || active[j]) group_count<kLds>(t_cnt, g[j]);
    }
  }
  for (int a = 0; a < gp.num_group_aggs; ++a) {
    const DevGroupAgg& ga = gp.group_aggs[a];     // self-contained record
    const DevGroupAgg& col = ga;
    long long* acc = t_acc + (long long)a * G;
    int32_t v[4];
    if (ga.vkind != kValI32) {
      // SUM over a dictionary with 8-byte entries (LONG with a wide range: exact int64 add; FLOAT / DOUBLE: double add)
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      const int b = col.bits;
      const LaneDec dec = make_lane_dec(b, lane);
      const uint8_t* slot = stage + ga.slot_off;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)col.dict, 0, col.dict_bytes, 0x00020000);
      u32x2 w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t d = b <= 25 ? decode_step<false>(slot, dec, k[j], b) : decode_step<true>(slot, dec, k[j], b);
        w[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (kAllActive || active[j]) ? d * 8u : 0xFFFFFFFFu, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kAllActive || active[j]) {
          const long long bits = (long long)(((unsigned long long)w[j].y << 32) | (unsigned long long)w[j].x);
          if (ga.vkind == kValI64) group_sum<kLds>(acc + g[j], bits);
          else __hip_atomic_fetch_add(reinterpret_cast<double*>(acc + g[j]), __longlong_as_double(bits), __ATOMIC_RELAXED,
                                      kLds ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      continue;
    }
    if (col.is_raw) {
      const long long base_doc = (long long)tile * 64 * p.tile_steps;
      const long long last = (long long)p.num_docs - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        long long doc = base_doc + k[j] * 64 + lane;
        doc = doc > last ? last : doc;
        v[j] = (int32_t)__builtin_bswap32(*reinterpret_cast<const uint32_t*>(col.fwd + doc * 4));
      }
    } else {
      const int b = col.bits;
      const LaneDec dec = make_lane_dec(b, lane);
      const uint8_t* slot = stage + ga.slot_off;
      uint32_t d[4];
      if (b <= 25) {
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = decode_step<false>(slot, dec, k[j], b);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = decode_step<true>(slot, dec, k[j], b);
      }
      if (ga.kind == kGroupSum && !col.is_plane) {
        // dictionary gather (small dictionaries / PINOT_GPU_VALUE_PLANE=0); inactive lanes use an out-of-range offset
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)col.dict, 0, col.dict_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (kAllActive || active[j]) ? d[j] * 4u : 0xFFFFFFFFu, 0, 0);
      } else {
        // value plane: the field is (value - base).  MIN / MAX on a sorted dictionary: the dictId is monotone in the value.
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (int32_t)d[j];
      }
    }
    if (ga.kind == kGroupSum) {
      // plane offsets are unsigned fields; gathered / raw values are signed
      const bool is_unsigned = !col.is_raw && col.is_plane;
      const long long one = (packed && a == gp.packed_agg) ? (1ll << gp.packed_shift) : 0ll;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kAllActive || active[j]) group_sum<kLds>(acc + g[j], (is_unsigned ? (long long)(uint32_t)v[j] : (long long)v[j]) + one);
      }
    } else if (ga.kind == kGroupMin) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kAllActive || active[j]) group_min<kLds>(acc + g[j], v[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kAllActive || active[j]) group_max<kLds>(acc + g[j], v[j]);
      }
    }
  }
}

// Sixteen consecutive steps of a tile whose docs all match: the group ids of the lane's sixteen docs stay in registers, the
// column descriptors are read once per sixteen steps, and each column is decoded in one straight unrolled run (sixteen
// LDS reads in flight, 3 VALU per decode) followed by its sixteen atomics.
template <bool kLds, bool kWide = false>
__device__ __forceinline__ void group_dense16(const GroupParams& gp, const uint8_t* stage, int tile, int lane, int k0,
                                              unsigned long long* t_cnt, long long* t_acc) {
  const ScanParams& p = gp.scan;
  const int G = gp.num_groups;
  uint32_t g[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) g[j] = 0u;
  for (int c = 0; c < gp.num_group_cols; ++c) {
    const DevGroupKey& key = gp.group_keys[c];
    const int b = key.bits;
    const LaneDec dec = make_lane_dec(b, lane);
    const uint8_t* slot = stage + key.slot_off;
    const uint32_t mult = (uint32_t)key.mult;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t d[8];
      decode_steps8(slot, dec, k0 + 8 * h, b, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[8 * h + j] = c == 0 ? d[j] : key_term<kWide>(d[j], mult) + g[8 * h + j];
    }
  }
  const bool packed = kLds && gp.packed_agg >= 0;
  if (!packed) {
#pragma unroll
    for (int j = 0; j < 16; ++j) group_count<kLds>(t_cnt, g[j]);
  }
  for (int a = 0; a < gp.num_group_aggs; ++a) {
    const DevGroupAgg& ga = gp.group_aggs[a];
    long long* acc = t_acc + (long long)a * G;
    int32_t v[16];
    if (ga.is_raw) {
      const long long base_doc = (long long)tile * 64 * p.tile_steps;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const long long doc = base_doc + (k0 + j) * 64 + lane;     // a full tile: every doc exists
        v[j] = (int32_t)__builtin_bswap32(*reinterpret_cast<const uint32_t*>(ga.fwd + doc * 4));
      }
    } else {
      const int b = ga.bits;
      const LaneDec dec = make_lane_dec(b, lane);
      const uint8_t* slot = stage + ga.slot_off;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t d[8];
        decode_steps8(slot, dec, k0 + 8 * h, b, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[8 * h + j] = (int32_t)d[j];
      }
      if (ga.kind == kGroupSum && !ga.is_plane) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ga.dict, 0, ga.dict_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (uint32_t)v[j] * 4u, 0, 0);
      }
    }
    if (ga.kind == kGroupSum) {
      const bool is_unsigned = !ga.is_raw && ga.is_plane;
      if (is_unsigned) {
        // the packed count lives entirely in the high dword (packed_shift >= 32): the operand is the pair {field, 1 << (shift-32)}
        const uint32_t one_hi = (packed && a == gp.packed_agg) ? (1u << (gp.packed_shift - 32)) : 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) group_sum<kLds>(acc + g[j], (long long)(((unsigned long long)one_hi << 32) | (unsigned long long)(uint32_t)v[j]));
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) group_sum<kLds>(acc + g[j], (long long)v[j]);
      }
    } else if (ga.kind == kGroupMin) {
#pragma unroll
      for (int j = 0; j < 16; ++j) group_min<kLds>(acc + g[j], v[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) group_max<kLds>(acc + g[j], v[j]);
    }
  }
}

template <bool kDma, bool kLdsTable, bool kWide = false>
__global__ __launch_bounds__(kGroupBlockThreads) void scan_group_kernel(const GroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const ScanParams& p = gp.scan;
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const int total_waves = gridDim.x * waves_per_block;
  const int G = gp.num_groups;
  const int NA = gp.num_group_aggs;
  const int steps = p.tile_steps;

  // LDS: [staging buffers of all waves][group table]
  uint8_t* wave_lds = smem + wave_in_block * p.wave_lds_bytes;
  unsigned long long* t_cnt;
  long long* t_acc;
  if constexpr (kLdsTable) {
    t_cnt = reinterpret_cast<unsigned long long*>(smem + waves_per_block * p.wave_lds_bytes);
    t_acc = reinterpret_cast<long long*>(t_cnt + G);
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
      t_cnt[g] = 0ull;
      for (int a = 0; a < NA; ++a) {
        const int kind = gp.group_aggs[a].kind;
        // MIN / MAX slots hold an int32 in their low dword
        t_acc[(long long)a * G + g] = kind == kGroupSum ? 0ll : (kind == kGroupMin ? 0x7FFFFFFFll : (long long)(uint32_t)0x80000000u);
      }
    }
    __syncthreads();
  } else {
    t_cnt = gp.table_count;
    t_acc = gp.table_acc;
  }

  const bool eager = p.lazy_columns == 0;
  const uint32_t fullm = full_mask(steps);
  uint8_t* bitmap_lds = wave_lds + p.bitmap_off;
  bool hot = false, cur_has_agg = false;
  int bbuf = 0;
  int tile = blockIdx.x * waves_per_block + wave_in_block;
  if (tile < p.num_tiles) {
    stage_bitmaps<kDma>(p, tile, bitmap_lds, lane);
    // without a filter every tile needs every column: stage them all up front
    if (eager) { stage_columns<kDma>(p, tile, wave_lds, lane, true, p.num_nodes == 0); cur_has_agg = p.num_nodes == 0; }
  }
  for (; tile < p.num_tiles; tile += total_waves) {
    if constexpr (kDma) wait_vmem();
    const int next = tile + total_waves;
    if (next < p.num_tiles) stage_bitmaps<kDma>(p, next, bitmap_lds + (bbuf ^ 1) * p.bitmap_bytes, lane);
    uint32_t m = eval_filter<kDma>(p, tile, wave_lds, bitmap_lds + bbuf * p.bitmap_bytes, lane, false);
    m &= valid_lane_mask(p.num_docs, tile, lane, steps);
    const bool any = __builtin_amdgcn_ballot_w64(m != 0u) != 0ull;
    hot = any;
    if (any) {
      if (!cur_has_agg) {
        stage_columns<kDma>(p, tile, wave_lds, lane, false, true);
        if constexpr (kDma) wait_vmem();
      }
      if (gp.dense_ok && __builtin_amdgcn_ballot_w64(m != fullm) == 0ull) {
        // every doc of the tile matches: walk the steps in order, no per-lane bookkeeping, no exec masking
        for (int kb = 0; kb < steps; kb += 16) group_dense16<kLdsTable, kWide>(gp, wave_lds, tile, lane, kb, t_cnt, t_acc);
      } else {
        uint32_t rem = m;
        for (;;) {
          if (__builtin_amdgcn_ballot_w64(rem != 0u) == 0ull) break;
          bool active[4];
          int k[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            active[j] = rem != 0u;
            k[j] = active[j] ? __builtin_ctz(rem) : 0;
            rem &= rem - 1u;
          }
          group_process4<kLdsTable, false, kWide>(gp, wave_lds, tile, lane, k, active, t_cnt, t_acc);
        }
      }
    }
    bbuf ^= 1;
    const bool next_has_agg = eager && (p.num_nodes == 0 || (hot && p.speculate != 0));
    if (eager && next < p.num_tiles) stage_columns<kDma>(p, next, wave_lds, lane, true, next_has_agg);
    cur_has_agg = next_has_agg;
  }

  if constexpr (kLdsTable) {
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
      const bool packed = gp.packed_agg >= 0;
      const unsigned long long pk = packed ? (unsigned long long)t_acc[(long long)gp.packed_agg * G + g] : 0ull;
      const unsigned long long c = packed ? (pk >> gp.packed_shift) : (unsigned long long)(uint32_t)t_cnt[g];
      if (c == 0ull) continue;
      __hip_atomic_fetch_add(&gp.table_count[g], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int a = 0; a < NA; ++a) {
        const int kind = gp.group_aggs[a].kind;
        long long* slot = gp.table_acc + (long long)a * G + g;
        long long v = t_acc[(long long)a * G + g];
        if (packed && a == gp.packed_agg) v = (long long)(pk & ((1ull << gp.packed_shift) - 1ull));
        if (kind == kGroupSum && gp.group_aggs[a].vkind == kValF64) __hip_atomic_fetch_add(reinterpret_cast<double*>(slot), __longlong_as_double(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (kind == kGroupSum) __hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (kind == kGroupMin) __hip_atomic_fetch_min(slot, (long long)(int32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_max(slot, (long long)(int32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Group-by without a filter: lane-private decode straight from HBM, no LDS staging at all.
//
// Lane i of the wave that owns a 2048-doc tile owns the 32 CONSECUTIVE docs 32*i .. 32*i+31, i.e. the B consecutive dwords
// i*B .. i*B+B-1 of the tile's 256*B-byte image of a B-bit column.  It loads them with plain global loads (every byte is used;
// the L1 keeps the lines between the wave's dwordx4 instructions: `lane_contiguous_read` in tools/microbench.hip streams at the
// same 6.2-6.5 TB/s as fully coalesced loads for every width) and extracts its 32 values from registers with shifts whose
// positions are compile-time constants: one v_bfe_u32, or v_alignbit_b32 + v_and_b32 when a value straddles two dwords, plus one
// byte swap per dword.  The LDS pipe, which bounds the staged kernel (three 8-byte reads per doc + the DMA writes + the atomics),
// is left with the group-table atomics only.
// ------------------------------------------------------------------------------------------------
// A pointer read from DEVICE MEMORY (scan_private_batch_kernel's items) is a generic pointer to the compiler: every load through it is a
// flat_load -- LDS-aperture check, counted in BOTH vmcnt and lgkmcnt so that scalar and LDS waits serialise behind it (measured: the
// batch kernel ran at 65 % of the single launch's rate, 1373 flat_loads in place of global_loads).  Kernel-argument pointers are known
// to be global; for a pointer that came from memory the LOAD has to say so -- an address_space(1) pointer type, which the decode
// helpers below take as a template parameter (a cast back to a generic pointer loses it again).
typedef const __attribute__((address_space(1))) uint32_t* GlobalWords;
typedef __attribute__((address_space(1))) uint32_t* GlobalWordsOut;
__device__ __forceinline__ GlobalWords global_words(const void* ptr) { return (GlobalWords)(const uint32_t*)ptr; }
struct __attribute__((aligned(4))) Dwords2 { uint32_t x, y; };
// the two dwords at `at` (4-byte aligned), whichever address space the pointer type names
__device__ __forceinline__ Dwords2 load_dwords2(const uint32_t* at) { return *reinterpret_cast<const Dwords2*>(at); }
typedef uint32_t DwordPair __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ Dwords2 load_dwords2(GlobalWords at) {
  const DwordPair v = *(const __attribute__((address_space(1))) DwordPair*)at;
  return Dwords2{v.x, v.y};
}

// Values 16*H .. 16*H+15 of the lane's 32 (H = 0, 1).  `lane_words` = first dword of the lane's B-dword chunk.
template <int B, int H, typename WP>
__device__ __forceinline__ void decode16_private(WP __restrict__ lane_words, uint32_t (&v)[16]) {
  constexpr int first_bit = 16 * B * H;
  constexpr int w0 = first_bit >> 5;
  constexpr int w1 = (16 * B * (H + 1) - 1) >> 5;
  constexpr int NW = w1 - w0 + 1;
  uint32_t d[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) d[i] = __builtin_bswap32(lane_words[w0 + i]);     // big-endian bit stream
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int bit = first_bit + j * B - w0 * 32;
    const int i = bit >> 5, o = bit & 31;
    if (o + B <= 32) v[j] = __builtin_amdgcn_ubfe(d[i], 32 - o - B, B);
    else v[j] = __builtin_amdgcn_alignbit(d[i], d[(i + 1) < NW ? (i + 1) : i], 64 - o - B) & ((1u << B) - 1u);
  }
}

template <int H, typename WP>
__device__ __forceinline__ void decode16_private_dispatch(int b, WP lane_words, uint32_t (&v)[16]) {
  switch (b) {
#define PG_CASE(B) case B: decode16_private<B, H>(lane_words, v); break;
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20)
    PG_CASE(21) PG_CASE(22) PG_CASE(23) PG_CASE(24) PG_CASE(25) PG_CASE(26) PG_CASE(27) PG_CASE(28) PG_CASE(29) PG_CASE(30)
    PG_CASE(31)
#undef PG_CASE
    default:
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = 0u;
  }
}

// ------------------------------------------------------------------------------------------------
// scan_private_kernel: fused scan -> filter -> aggregate in the lane-private layout.
//
// Same ownership as the group-by kernel above: lane i holds the 32 consecutive docs 32*i .. 32*i+31 of the wave's 2048-doc tile,
// so its 32-bit mask IS dword (64*tile + i) of the doc-order docId bitmap -- posting bitmaps are read and filter bitmaps written
// with one coalesced dword per lane, no ballots and no bit transposes.  Columns are read with plain global loads (no LDS, no
// DMA, nothing to size per query) and decoded from registers at compile-time bit positions:
//     range leaf      bswap per dword + (v_bfe | v_alignbit+v_and) + v_cmp + v_addc per value   (~3.3-3.7 VALU per doc)
//     plane SUM       bswap per dword + extract + v_bfe (the match bit) + v_mad_u32_u24          (~3.6 VALU per doc, any selectivity)
// against ~4.5 and ~4-12 in the LDS-staged kernel, whose decode needs a per-lane byte gather (v_perm) on top of the extract.
// ------------------------------------------------------------------------------------------------
// Sixteen values (half H) of the lane's chunk of a B-bit column: mask bits of ((value - lo) < span), most recent value in bit 0.
template <int B, int H, bool kLoZero, typename WP>
__device__ __forceinline__ void range16_private(WP __restrict__ lane_words, uint32_t lo, uint32_t span, uint32_t& m) {
  uint32_t v[16];
  decode16_private<B, H>(lane_words, v);
#pragma unroll
  for (int j = 0; j < 16; ++j) shift_in_lt(m, kLoZero ? v[j] : v[j] - lo, span);
}

// C2a shape -- SUM(col) WHERE col in range, one leaf and one summed column reading the SAME stream: decode once, and use the
// compare's VCC twice (select the value for the sum, then shift the match into the mask).  One load phase per tile instead of two.
template <int B, int H, bool kLoZero, typename WP>
__device__ __forceinline__ void range_sum16_private(WP __restrict__ lane_words, uint32_t lo, uint32_t span, uint32_t& m, uint32_t& psum,
                                                    unsigned long long& wsum) {
  uint32_t v[16];
  decode16_private<B, H>(lane_words, v);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t x = kLoZero ? v[j] : v[j] - lo;
    uint32_t picked;
    asm("v_cmp_gt_u32 vcc, %3, %2\n\tv_cndmask_b32 %1, 0, %4, vcc\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
        : "+v"(m), "=&v"(picked) : "v"(x), "s"(span), "v"(v[j]) : "vcc");
    if (B <= 27) psum += picked;          // 32 fields of <= 27 bits fit the 32-bit partial sum
    else wsum += picked;
  }
}

template <bool kLoZero, typename WP>
__device__ __forceinline__ uint32_t range_sum_private_dispatch(int b, WP lane_words, uint32_t lo, uint32_t span, unsigned long long& wsum) {
  uint32_t m = 0, psum = 0;
  switch (b) {
#define PG_CASE(B) case B: range_sum16_private<B, 0, kLoZero>(lane_words, lo, span, m, psum, wsum); range_sum16_private<B, 1, kLoZero>(lane_words, lo, span, m, psum, wsum); break;
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20)
    PG_CASE(21) PG_CASE(22) PG_CASE(23) PG_CASE(24) PG_CASE(25) PG_CASE(26) PG_CASE(27) PG_CASE(28) PG_CASE(29) PG_CASE(30)
    PG_CASE(31)
#undef PG_CASE
    default: break;
  }
  wsum += psum;
  return __builtin_bitreverse32(m);
}

template <bool kLoZero, typename WP>
__device__ __forceinline__ uint32_t range_private_dispatch(int b, WP lane_words, uint32_t lo, uint32_t span) {
  uint32_t m = 0;
  switch (b) {
#define PG_CASE(B) case B: range16_private<B, 0, kLoZero>(lane_words, lo, span, m); range16_private<B, 1, kLoZero>(lane_words, lo, span, m); break;
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20)
    PG_CASE(21) PG_CASE(22) PG_CASE(23) PG_CASE(24) PG_CASE(25) PG_CASE(26) PG_CASE(27) PG_CASE(28) PG_CASE(29) PG_CASE(30)
    PG_CASE(31)
#undef PG_CASE
    default: break;
  }
  return __builtin_bitreverse32(m);      // value j -> bit j
}

// (P / N: ScanParams / DevNode, or their constant-address-space forms in scan_private_batch_kernel)
// The dictId sets of a filter in LDS (round 6b; scan_private_body stages them, kSetLdsWords per workgroup): a set leaf of a column of b bits
// takes 2^b / 32 words there, zero beyond the set's own words, so a lookup needs no bound check -- `set_lds_fits` is the rule both the
// staging and the evaluation follow, leaf after leaf in node order (`off`: words taken so far).  A lookup is then one conflict-light
// ds_read_b32 (32 words: every bank holds one address) instead of a buffer load through the texture path per doc: measured on 64 x 10 M rows,
// SUM(v) WHERE f IN (100 of 1000): profiles/r6/set_leaf_in_lds.txt.
__device__ __forceinline__ int set_lds_words(int bits) { return bits <= 5 ? 1 : 1 << (bits - 5); }
__device__ __forceinline__ bool set_lds_fits(int bits, int off) { return bits <= 16 && off + set_lds_words(bits) <= kSetLdsWords; }
template <typename P>
__device__ __forceinline__ void stage_filter_sets(const P& p, uint32_t* set_lds) {
  int off = 0;
  for (int n = 0; n < p.num_nodes; ++n) {
    const auto& nd = p.nodes[n];
    if (nd.op != PG_FILTER_LEAF || nd.kind != kLeafDictSet || !set_lds_fits(nd.bits, off)) continue;
    const int words = set_lds_words(nd.bits), have = nd.set_bytes >> 2;
    for (int i = threadIdx.x; i < words; i += blockDim.x) set_lds[off + i] = i < have ? global_words(nd.set_words)[i] : 0u;
    off += words;
  }
  __syncthreads();
}

template <typename P, typename N>
__device__ __forceinline__ uint32_t eval_leaf_private(const P& p, const N& L, long long tile, int lane, const uint32_t* set_lds = nullptr) {
  uint32_t m;
  switch (L.kind) {
    case kLeafMatchAll: m = 0xFFFFFFFFu; break;
    case kLeafMatchNone: m = 0u; break;
    case kLeafDictRange: {
      const GlobalWords words = global_words(L.fwd + tile * (256ll * L.bits)) + lane * L.bits;
      m = L.lo == 0 ? range_private_dispatch<true>(L.bits, words, 0u, L.span) : range_private_dispatch<false>(L.bits, words, (uint32_t)L.lo, L.span);
      break;
    }
    case kLeafDictSet: {
      // InPredicateEvaluator: bit dictId of the set (the words stay L1-resident)
      const GlobalWords words = global_words(L.fwd + tile * (256ll * L.bits)) + lane * L.bits;
      m = 0;
      if (set_lds != nullptr) {
        // (the set's words are in LDS, zero-padded to the column's whole dictId range: pg_kernels.h "The dictId sets of a filter in LDS")
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint32_t d[16];
          if (h == 0) decode16_private_dispatch<0>(L.bits, words, d); else decode16_private_dispatch<1>(L.bits, words, d);
          // (eight lookups in flight at a time, as scan_simple_set_kernel: sixteen kept eight more registers live across the reads)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = set_lds[d[8 * g + j] >> 5];
#pragma unroll
            for (int j = 0; j < 8; ++j) m |= __builtin_amdgcn_ubfe(w[j], d[8 * g + j] & 31u, 1) << (16 * h + 8 * g + j);
          }
        }
        break;
      }
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)L.set_words, 0, L.set_bytes, 0x00020000);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t d[16], w[16];
        if (h == 0) decode16_private_dispatch<0>(L.bits, words, d); else decode16_private_dispatch<1>(L.bits, words, d);
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (d[j] >> 5) * 4u, 0, 0);   // out of range -> 0
#pragma unroll
        for (int j = 0; j < 16; ++j) m |= ((w[j] >> (d[j] & 31u)) & 1u) << (16 * h + j);
      }
      break;
    }
    case kLeafRawRange: {
      // raw INT column: the lane's 32 docs are 128 contiguous bytes (pg_segment_open pads raw buffers to whole 2048-doc tiles, so
      // the last tile needs no clamping; its surplus docs are masked by the caller)
      const GlobalWords vals = global_words(L.fwd) + tile * 2048 + lane * 32;
      const uint32_t lo = (uint32_t)L.lo, span = L.span;
      m = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = vals[16 * h + j];
#pragma unroll
        for (int j = 0; j < 16; ++j) shift_in_le(m, __builtin_bswap32(v[j]) - lo, span);
      }
      m = __builtin_bitreverse32(m);
      break;
    }
    case kLeafDocRange: {
      // docs first .. first+31 against [lo, lo+span]: bits [a, b) with a, b clamped to 0..32
      const long long first = tile * 2048 + lane * 32;
      const long long lo = (long long)(uint32_t)L.lo, hi = lo + (long long)L.span;
      const long long a = lo - first < 0 ? 0 : (lo - first > 32 ? 32 : lo - first);
      const long long b = hi + 1 - first < 0 ? 0 : (hi + 1 - first > 32 ? 32 : hi + 1 - first);
      const uint32_t below_b = b >= 32 ? 0xFFFFFFFFu : ((1u << (int)b) - 1u);
      const uint32_t below_a = a >= 32 ? 0xFFFFFFFFu : ((1u << (int)a) - 1u);
      m = below_b & ~below_a;
      break;
    }
    default: {  // kLeafBitmap: the lane's mask is one dword of the doc-order bitmap
      m = global_words(L.set_words)[tile * 64 + lane];
      break;
    }
  }
  return L.exclusive ? ~m : m;
}

// numEntriesScannedInFilter of `a AND b`, both scan leaves -- AndDocIdIterator.next() (dociditerators/AndDocIdIterator.java:41-74)
// leap-frogging two SVScanDocIdIterators (SVScanDocIdIterator.java:91-106: advance(t) looks at doc t, t + 1, ... until one matches, one
// entry each).  At every doc exactly ONE leaf is scanning (one entry per doc: numDocs in total, added by the host); where the scanning
// leaf matches, the other one is asked about that doc (one more entry) -- if it says no IT scans on from the next doc, if it says yes
// the doc is a result and child 0 scans on.  With s = "child 1 is scanning", per doc:
//     (a, b) = (0, 0): s stays                      (1, 0): s := 1, extra entry iff s was 0
//     (0, 1): s := 0, extra entry iff s was 1       (1, 1): s := 0, extra entry either way
// so s is the carry of a binary addition with generate = a & ~b and propagate = ~(a | b): ONE 64-bit add per lane gives the state before
// each of its 32 docs, one 64-bit scalar add over the wave's ballots gives every lane's carry-in, and three popcounts give the entries.
// Tiles are dealt round-robin, so a tile does not know the state it is entered in: its extra entries are counted for entry state 0
// (into the lane's `entries`, reduced once per wave at the end like the kNodeCountEntries count) and ONE byte per tile records what the
// chain needs: delta in {-1, 0, +1} = what entry state 1 would change (only the docs before the tile's first event can tell), the
// state after the tile, and whether the tile has no event at all.  leapfrog2_chain_kernels add the sum of delta over the tiles that are
// entered in state 1.  ~30 vector + ~15 scalar instructions per 2048 docs next to the ~230 the two leaves cost.
//     byte = (delta + 1) | (state after the tile, entered in 0) << 2 | (tile has no event) << 3
template <typename P>
__device__ __forceinline__ void leapfrog2_tile(const P& p, long long tile, int lane, uint32_t a, uint32_t b, uint32_t& entries) {
  const long long rem = (long long)p.num_docs - (tile * 2048 + lane * 32);
  const uint32_t valid = rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u));      // docs past numDocs are not there
  a &= valid; b &= valid;
  const uint32_t E = a | b, X = a & ~b, Y = b & ~a, T = a & b, prop = ~E;
  const unsigned long long sum0 = (unsigned long long)(X | prop) + (unsigned long long)X;
  const unsigned long long Gm = __builtin_amdgcn_ballot_w64((uint32_t)(sum0 >> 32) != 0u), Pm = __builtin_amdgcn_ballot_w64(E == 0u);
  unsigned long long wsum;
  const bool tile_carry = __builtin_uaddll_overflow(Gm | Pm, Gm, &wsum);
  const uint32_t cin = (uint32_t)((wsum ^ Pm) >> lane) & 1u;                // the lane is entered in state 1 (the tile in state 0)
  const uint32_t S = (uint32_t)(sum0 + cin) ^ prop;                         // state before each of the lane's docs
  entries += (uint32_t)(__builtin_popcount(T) + __builtin_popcount(X & ~S) + __builtin_popcount(Y & S));
  int delta = 0;
  const unsigned long long with_events = ~Pm;
  if (with_events != 0ull) {
    const int first = __builtin_ctzll(with_events);
    const uint32_t Ef = (uint32_t)__builtin_amdgcn_readlane((int)E, first), Xf = (uint32_t)__builtin_amdgcn_readlane((int)X, first),
                   Yf = (uint32_t)__builtin_amdgcn_readlane((int)Y, first);
    const int bit = __builtin_ctz(Ef);
    delta = (int)((Yf >> bit) & 1u) - (int)((Xf >> bit) & 1u);
  }
  if (lane == 0) ((__attribute__((address_space(1))) uint8_t*)p.leap_tables)[tile] = (uint8_t)((uint32_t)(delta + 1) | ((tile_carry ? 1u : 0u) << 2) | ((with_events == 0ull ? 1u : 0u) << 3));
}

// The chain: sum over the tiles of [tile entered in state 1] * delta(tile), the entry states being the carries of (generate, propagate)
// = (state after, no event) in tile order, entry state 0 at doc 0.  The same carry trick one level up: a wavefront takes 64 consecutive
// pieces, their generate / propagate ballots and ONE 64-bit add give every piece's entry state; the wave's own summary
//     {sum (entered in 0), delta (what entering in 1 adds: only pieces before the first event can tell), state after, no event}
// is a piece of the next level.  Level 1: a workgroup of 16 waves per 1024 tiles (leapfrog2_chain_tiles_kernel); level 2: one workgroup
// over the <= 1024 workgroup summaries (a segment has < 2^31 docs = 2^20 tiles), which adds the result to the entries counter.
struct Leap2Summary { long long sum; int delta; uint32_t g, p; };
__device__ __forceinline__ Leap2Summary leap2_wave(long long sum, int delta, uint32_t g, uint32_t pp, int lane) {
  const unsigned long long Gm = __builtin_amdgcn_ballot_w64(g != 0u), Pm = __builtin_amdgcn_ballot_w64(pp != 0u);
  unsigned long long wsum;
  const bool carry = __builtin_uaddll_overflow(Gm | Pm, Gm, &wsum);
  const bool entered_in_1 = (((wsum ^ Pm) >> lane) & 1ull) != 0ull;
  long long mine = sum + (entered_in_1 ? (long long)delta : 0ll);
  mine = wave_sum_i64(mine);
  int wave_delta = 0;
  const unsigned long long with_events = ~Pm;
  if (with_events != 0ull) wave_delta = __builtin_amdgcn_readlane(delta, __builtin_ctzll(with_events));      // pieces before it have delta 0 (no event)
  return Leap2Summary{mine, wave_delta, carry ? 1u : 0u, with_events == 0ull ? 1u : 0u};
}
// thread 0 chains the waves' summaries of its workgroup (in wave order)
__device__ __forceinline__ Leap2Summary leap2_block(const Leap2Summary w, Leap2Summary* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = w;
  __syncthreads();
  Leap2Summary all{0ll, 0, 0u, 1u};
  if (threadIdx.x == 0) {
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
      const Leap2Summary s = part[i];
      all.sum += s.sum + (all.g ? (long long)s.delta : 0ll);
      all.delta += all.p ? s.delta : 0;
      all.g = s.g | (s.p & all.g);
      all.p &= s.p;
    }
  }
  return all;
}
// One launch: every workgroup summarises its 1024 tiles, stores the summary write-through and arrives (the R1 hand-off of
// publish_block_partial); the workgroup whose arrival completes the count chains the workgroups' summaries and publishes the result --
// into the pinned host record (aggregation queries: no copy command follows) or onto the device counter (group-by queries).
static __global__ __launch_bounds__(1024) void leapfrog2_chain_kernel(const uint8_t* __restrict__ tables, long long num_tiles, Leap2Summary* block_out, uint32_t* arrivals,
                                                                     HostRecord* host_out, unsigned long long host_seq, unsigned long long* entries) {
  __shared__ Leap2Summary part[16];
  __shared__ uint32_t last_flag;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t e = t < num_tiles ? (uint32_t)tables[t] : 0x9u;           // past the end: delta 0, no event
  const Leap2Summary w = leap2_wave(0ll, (int)(e & 3u) - 1, (e >> 2) & 1u, (e >> 3) & 1u, threadIdx.x & 63);
  Leap2Summary all = leap2_block(w, part);
  if (threadIdx.x == 0) {
    uint32_t last = 1u;
    if (gridDim.x > 1u) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(&block_out[blockIdx.x]);
      static_assert(sizeof(Leap2Summary) == 24, "three 8-byte words");
      __hip_atomic_store(dst, (unsigned long long)all.sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 1, (unsigned long long)(uint32_t)all.delta | ((unsigned long long)all.g << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 2, (unsigned long long)all.p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x ? 1u : 0u;
    }
    last_flag = last;
  }
  __syncthreads();
  if (last_flag == 0u) return;
  if (gridDim.x > 1u) {
    // (a segment has < 2^31 docs = 2^20 tiles: at most 1024 workgroup summaries, one per thread)
    Leap2Summary in{0ll, 0, 0u, 1u};
    if (threadIdx.x < gridDim.x) {
      // agent-scope (sc1) loads of what was stored sc1: no acquire fence (see fold_partials)
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&block_out[threadIdx.x]);
      const unsigned long long s0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), s1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               s2 = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      in = Leap2Summary{(long long)s0, (int)(uint32_t)s1, (uint32_t)(s1 >> 32), (uint32_t)s2};
    }
    __syncthreads();
    const Leap2Summary w2 = leap2_wave(in.sum, in.delta, in.g, in.p, threadIdx.x & 63);
    all = leap2_block(w2, part);
  }
  if (threadIdx.x == 0) {
    if (gridDim.x > 1u) __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the segment is entered in state 0; a negative sum wraps the unsigned counter the right way)
    if (entries != nullptr && all.sum != 0ll) atomicAdd(entries, (unsigned long long)all.sum);
    if (host_out != nullptr) {
      // write-through system stores, their acknowledgement, then the sequence number (store_host_record)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(&host_out->leap_correction), (unsigned long long)all.sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&host_out->leap_seq, host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// `entries`: per-lane share of numEntriesScannedInFilter -- a kNodeCountEntries leaf is a scan-based child of the root AND that the
// reference and-s into the docIds left by the children before it (ScanBasedDocIdIterator.applyAnd, AndDocIdSet.java:161-163,
// SVScanDocIdIterator.java:115-145: one entry per doc of that bitmap).  On the AND chain the only mask on the stack is that bitmap.
// kCollect: every leaf's own mask also goes to w[fsm_input_of_leaf[ordinal]] (the in-kernel transducer walk: scan_private_fsm_kernel) --
// four registers picked with wave-uniform selects, no store to memory.
// `set_lds`: the workgroup's staged dictId sets (stage_filter_sets), or nullptr: set leaves read their words from memory.
template <bool kCollect = false, typename P>
__device__ __forceinline__ uint32_t eval_filter_private(const P& p, long long tile, int lane, uint32_t& entries, uint32_t (*w)[4] = nullptr, const uint32_t* set_lds = nullptr) {
  if (p.num_nodes == 0) return 0xFFFFFFFFu;
  if (p.num_nodes == 1) return eval_leaf_private(p, p.nodes[0], tile, lane, (set_lds != nullptr && p.nodes[0].kind == kLeafDictSet && set_lds_fits(p.nodes[0].bits, 0)) ? set_lds : nullptr);
  int set_off = 0;                                                        // (uniform) words of the set area the leaves so far have taken
  MaskStack st;
#pragma unroll
  for (int i = 0; i < kStackDepth; ++i) st.v[i] = 0;
  st.sp = 0;
  int leaf_ordinal = 0;
  for (int n = 0; n < p.num_nodes; ++n) {
    const auto& nd = p.nodes[n];
    uint32_t top;
    if (nd.op == PG_FILTER_LEAF) {
      if (nd.flags & kNodeCountEntries) {
        const long long rem = (long long)p.num_docs - (tile * 2048 + lane * 32);
        entries += (uint32_t)__builtin_popcount(st.v[0] & (rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u))));
      }
      const uint32_t* my_set = nullptr;
      if (set_lds != nullptr && nd.kind == kLeafDictSet && set_lds_fits(nd.bits, set_off)) { my_set = set_lds + set_off; set_off += set_lds_words(nd.bits); }
      top = eval_leaf_private(p, nd, tile, lane, my_set);
      if constexpr (kCollect) {
        const int in = p.fsm_input_of_leaf[leaf_ordinal];                  // (uniform)
#pragma unroll
        for (int i = 0; i < 4; ++i) (*w)[i] = in == i ? top : (*w)[i];
      } else
      if (p.leaf_out_enabled != 0) {
        uint32_t* const leaf_bits = p.leaf_out[leaf_ordinal];          // (uniform: one scalar load)
        if (leaf_bits != nullptr) leaf_bits[tile * 64 + lane] = top;   // one coalesced 256-byte store per leaf and tile
      }
      ++leaf_ordinal;
    } else if (nd.op == PG_FILTER_NOT) {
      top = ~st.pop();
    } else {
      top = st.pop();
      if (nd.flags & kNodeLeapfrog2) leapfrog2_tile(p, tile, lane, st.v[0], top, entries);      // the root AND of two leaves: child 0 is the stack's only entry, child 1 was on top
      for (int c = 1; c < nd.num_children; ++c) {
        const uint32_t o = st.pop();
        top = nd.op == PG_FILTER_AND ? (top & o) : (top | o);
      }
    }
    // root AND chain: nothing left in the whole tile -> the remaining leaves' columns are never read
    if ((nd.flags & kNodeExitIfZero) && __builtin_amdgcn_ballot_w64(top != 0u) == 0ull) return 0u;
    st.push(top);
  }
  return st.pop();
}

template <typename P>
__device__ __forceinline__ void flush_filter_entries(const P& p, uint32_t entries) {
  if (p.filter_entries == nullptr) return;
  const unsigned long long total = (unsigned long long)wave_sum_i64((long long)entries);
  if ((threadIdx.x & 63u) == 0u && total != 0ull) atomicAdd(p.filter_entries, total);
}

// Masked aggregation of one half (16 values) of a column chunk.  Keys (dictIds / plane fields) are below 2^31.
template <int B, int H, typename WP>
__device__ __forceinline__ void agg16_private(WP __restrict__ lane_words, uint32_t m, bool need_sum, bool need_minmax,
                                              uint32_t& psum, unsigned long long& wsum, uint32_t& umin, uint32_t& umax) {
  uint32_t v[16];
  decode16_private<B, H>(lane_words, v);
  if (need_sum) {
    if (B <= 24) {
      // match bit as 0 / 1, one full-rate 24-bit multiply-add per value; 32 fields of <= 24 bits fit the 32-bit partial sum
#pragma unroll
      for (int j = 0; j < 16; ++j) psum = __umul24(v[j], __builtin_amdgcn_ubfe(m, 16 * H + j, 1)) + psum;
    } else if (B <= 27) {
#pragma unroll
      for (int j = 0; j < 16; ++j) psum += v[j] & (uint32_t)__builtin_amdgcn_sbfe((int)m, 16 * H + j, 1);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) wsum += (unsigned long long)(v[j] & (uint32_t)__builtin_amdgcn_sbfe((int)m, 16 * H + j, 1));
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

template <typename WP>
__device__ __forceinline__ void agg_private_dispatch(int b, WP lane_words, uint32_t m, bool need_sum, bool need_minmax,
                                                     uint32_t& psum, unsigned long long& wsum, uint32_t& umin, uint32_t& umax) {
  switch (b) {
#define PG_CASE(B) case B: agg16_private<B, 0>(lane_words, m, need_sum, need_minmax, psum, wsum, umin, umax); \
                           agg16_private<B, 1>(lane_words, m, need_sum, need_minmax, psum, wsum, umin, umax); break;
    PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
    PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15) PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20)
    PG_CASE(21) PG_CASE(22) PG_CASE(23) PG_CASE(24) PG_CASE(25) PG_CASE(26) PG_CASE(27) PG_CASE(28) PG_CASE(29) PG_CASE(30)
    PG_CASE(31)
#undef PG_CASE
    default: break;
  }
}

// Sparse aggregation of a tile: the lane walks the set bits of its mask and reads each matching doc's value with ONE 8-byte load at
// the doc's bit position (two dwords of the big-endian stream always hold a value of at most 31 bits; the second dword may be the next
// lane's first, or the buffer's padding).  Four matches per lane are in flight per round; the rounds repeat while any lane has matches
// left (wave-uniform).  At 1 % selectivity a lane holds 0.32 matches on average: one round of <= 64 sector-sized reads replaces the
// 17-dword chunk load and the 32-value decode of every lane that holds a match -- the value column is touched one 64-byte sector per
// matching doc (SURVEY.md 8(d)'s min(B(v), M x 64 B)), the way the reference's projection reads only the docIds its filter left
// (SVScanDocIdIterator.java:115-142 -> ProjectionOperator).
template <typename WP>
__device__ __forceinline__ void agg_sparse_private(WP __restrict__ lane_words, WP __restrict__ tile_words, int b, uint32_t m, bool need_sum,
                                                   bool need_minmax, unsigned long long& wsum, uint32_t& umin, uint32_t& umax) {
  const uint32_t field_mask = (1u << b) - 1u;
  uint32_t rest = m;
  while (__builtin_amdgcn_ballot_w64(rest != 0u) != 0ull) {
    Dwords2 d[4];
    uint32_t sh[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ok[k] = rest != 0u;
      const uint32_t j = ok[k] ? (uint32_t)__builtin_ctz(rest) : 0u;
      rest &= rest - 1u;                                   // (0 stays 0)
      const uint32_t bit = j * (uint32_t)b;
      sh[k] = 64u - (bit & 31u) - (uint32_t)b;
      // unconditional (a lane without a k-th match reads the tile's first dwords, one sector for the whole wave): a load inside an
      // exec-masked branch is waited for before the branch is left, which would make the four loads four round trips
      d[k] = load_dwords2(ok[k] ? lane_words + (bit >> 5) : tile_words);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long x = ((unsigned long long)__builtin_bswap32(d[k].x) << 32) | (unsigned long long)__builtin_bswap32(d[k].y);
      const uint32_t v = (uint32_t)(x >> sh[k]) & field_mask;
      if (ok[k]) {
        if (need_sum) wsum += v;
        if (need_minmax) { umax = v > umax ? v : umax; umin = v < umin ? v : umin; }
      }
    }
  }
}

#ifndef PG_PRIVATE_WAVES
#ifndef PG_PRIVATE_FSM_WAVES
#define PG_PRIVATE_FSM_WAVES 4  // scan_private_fsm_kernel (the transducer walked inside the scan): its own bound
#endif
#define PG_PRIVATE_WAVES 4      // wavefronts per SIMD the register allocation must allow; 5 and 6 spill in the hot path (measured)
#endif
// The kernel's body: workgroup `block_index` of the `num_blocks` that work on `p` (the whole grid in scan_private_kernel; one query's
// share of the grid in scan_private_batch_kernel, where p is read from device memory).
// Accumulators of the multi-column instantiations live in LDS, one entry per thread and slot: sixteen more registers per lane across
// the tile loop (a 64-bit sum and two keys per slot, selected by a run-time column index) was what turned the four-slot kernel's
// 127 registers into 43 spilled ones IN the loop -- two aggregated columns ran at half the rate of one (1.59 vs 0.80 ms per 1 B
// rows, profiles/r3/ab_two_aggregated_columns.jsonl).  Three LDS operations per column and tile instead; a thread only ever touches its
// own entries, so there is nothing to synchronise.
template <int kAggSlots>
struct PrivateAccLds {
  unsigned long long sum[kAggSlots][kBlockThreads];
  uint32_t umin[kAggSlots][kBlockThreads], umax[kAggSlots][kBlockThreads];
};
template <>
struct PrivateAccLds<1> { uint32_t unused; };          // the one-slot form keeps its accumulators in registers
// kFsm: numEntriesScannedInFilter of a leap-frogging root AND is walked here, tile by tile, on the leaves' masks while they are still in
// registers (fsm_perm_tile, pg_fsm_kernels.h: machines of at most four states and four inputs) -- the separate pass wrote every leaf's
// bitmap to HBM and read it back (AndDocIdIterator.java:40-73 is what is being counted).  fsm_delta / fsm_pair: the walk's LDS tables.
struct FsmWalkLds { uint8_t delta[64]; uint2 pair_fn[256]; };
template <int kAggSlots, typename P, bool kFsm = false>
__device__ __forceinline__ void scan_private_body(const P& p, const uint32_t block_index, const uint32_t num_blocks, BlockPartial* red, uint32_t* fold_flag_ptr,
                                                  PrivateAccLds<kAggSlots>* acc_lds, FsmWalkLds* fsm_lds = nullptr, uint32_t* set_lds = nullptr) {
  if constexpr (kFsm) fsm_perm_build_tables<4>(p.fsm_delta, 4, p.fsm_states, p.fsm_inputs, fsm_lds->delta, fsm_lds->pair_fn);
  if (p.set_leaves_in_lds == 0) set_lds = nullptr;                        // (uniform; PINOT_GPU_SET_LDS=0, or a filter without a set leaf)
  if (set_lds != nullptr) stage_filter_sets(p, set_lds);
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const long long total_waves = (long long)num_blocks * waves_per_block;
  const long long num_tiles = ((long long)p.num_docs + 2047) / 2048;

  unsigned long long count = 0;
  constexpr bool kAccInLds = kAggSlots > 1;
  unsigned long long sum[kAccInLds ? 1 : kAggSlots];
  uint32_t umin[kAccInLds ? 1 : kAggSlots], umax[kAccInLds ? 1 : kAggSlots];
#pragma unroll
  for (int a = 0; a < (kAccInLds ? 1 : kAggSlots); ++a) { sum[a] = 0; umin[a] = 0xFFFFFFFFu; umax[a] = 0u; }
  if constexpr (kAccInLds) {
#pragma unroll
    for (int a = 0; a < kAggSlots; ++a) { acc_lds->sum[a][threadIdx.x] = 0ull; acc_lds->umin[a][threadIdx.x] = 0xFFFFFFFFu; acc_lds->umax[a][threadIdx.x] = 0u; }
  }

  // (Early "touch" loads of the next column chunks were tried and lost 30 %: vmcnt retires in order, so a wave's own L2 hits
  // queue behind its prefetches that are still on their way from HBM.  Latency is hidden by the other resident waves instead.)
  // Per wave the chain load -> decode -> load -> decode is what is left between this kernel and the HBM ceiling (C2a reads one
  // column twice and still takes as long as C2b).  Two ways of overlapping a wave's own loads with its own compute were tried
  // and dropped: early "touch" loads through the LDS-DMA path (-30 %: vmcnt retires in order, a wave's L2 hits queue behind
  // its prefetches still coming from HBM), and holding whole column chunks in 31-register arrays loaded one phase ahead (the
  // arrays cross the width switch, are not promoted to registers and go to scratch: 15-40x slower).  A third variant, two tiles per
  // iteration with both chunks loaded inside one width-specialised function, does put two requests in flight (+5 % on C2b within
  // the same build) but the extra code costs the other shapes about as much, and two more minutes of compile time: not kept.
  // one inclusive-range leaf and one SUM over the same stream (not exclusive, no MIN / MAX): fused decode
  const bool fused = p.num_nodes == 1 && p.num_agg_cols == 1 && p.nodes[0].kind == kLeafDictRange && p.nodes[0].exclusive == 0 &&
                     p.nodes[0].fwd == p.agg_cols[0].fwd && p.nodes[0].bits == p.agg_cols[0].bits && p.agg_cols[0].need_sum != 0 &&
                     p.agg_cols[0].need_minmax == 0 && p.out_bitmap == nullptr;
  // index-driven filters: only the tiles index_and_kernel listed hold a match (any order)
  uint32_t entries = 0u;                                   // numEntriesScannedInFilter, this lane's share
  const bool listed = p.tile_list != nullptr;
  const long long tile_limit = listed ? (long long)*global_words(p.tile_count) : num_tiles;
  for (long long tile_it = (long long)block_index * waves_per_block + wave_in_block; tile_it < tile_limit; tile_it += total_waves) {
    const long long tile = listed ? (long long)global_words(p.tile_list)[tile_it] : tile_it;
    if (fused) {
      const auto& L = p.nodes[0];
      const GlobalWords words = global_words(L.fwd + tile * (256ll * L.bits)) + lane * L.bits;
      unsigned long long wsum = 0;
      uint32_t fm = L.lo == 0 ? range_sum_private_dispatch<true>(L.bits, words, 0u, L.span, wsum) : range_sum_private_dispatch<false>(L.bits, words, (uint32_t)L.lo, L.span, wsum);
      // the padding past numDocs decodes to field 0: it can only match when lo == 0, and then it adds 0 to the sum
      const long long frem = (long long)p.num_docs - (tile * 2048 + lane * 32);
      fm &= frem >= 32 ? 0xFFFFFFFFu : (frem <= 0 ? 0u : ((1u << (int)frem) - 1u));
      count += (unsigned)__builtin_popcount(fm);
      sum[0] += wsum;
      continue;
    }
    uint32_t m;
    const long long rem = (long long)p.num_docs - (tile * 2048 + lane * 32);
    if constexpr (kFsm) {
      uint32_t fw[4] = {0u, 0u, 0u, 0u};
      m = eval_filter_private<true>(p, tile, lane, entries, &fw, set_lds);
      fsm_perm_tile<4>(fw, rem >= 32 ? 32 : (rem <= 0 ? 0 : (int)rem), fsm_lds->delta, fsm_lds->pair_fn, lane, p.fsm_states, p.fsm_tables + tile * p.fsm_states);
    } else {
      m = eval_filter_private(p, tile, lane, entries, nullptr, set_lds);
    }
    // docs past numDocs (last tile only)
    m &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u));
    if (p.out_bitmap) ((GlobalWordsOut)reinterpret_cast<uint32_t*>(p.out_bitmap))[tile * 64 + lane] = m;
    count += (unsigned)__builtin_popcount(m);
    if (p.num_agg_cols == 0 || __builtin_amdgcn_ballot_w64(m != 0u) == 0ull) continue;
    // one instance of the width dispatch for all slots (a runtime loop: unrolling it four times quadruples the code and keeps
    // ~190 VGPRs live); the per-slot accumulators are selected with wave-uniform predicates
    // A lane whose 32 docs hold no match has nothing to add: its loads are not issued (exec-masked), so at low selectivity the value
    // column is read at cache-line granularity around the matches instead of in full -- the reference reads only the surviving docs
    // (SVScanDocIdIterator.java:115-142 feeds ProjectionOperator 10 000 matching docIds at a time).  At 1 % selectivity 27 % of the
    // lanes hold a match; from ~10 % on every lane does and the branch is never taken differently by two lanes.
    const bool lane_active = p.lane_skip == 0 || m != 0u;
    // few lanes of the tile hold a match: walk the matches instead of decoding whole chunks (agg_sparse_private)
    const bool sparse_tile = __builtin_popcountll(__builtin_amdgcn_ballot_w64(m != 0u)) <= p.sparse_lanes;
    for (int a = 0; a < p.num_agg_cols; ++a) {
      const auto& ac = p.agg_cols[a];
      const GlobalWords words = global_words(ac.fwd + tile * (256ll * ac.bits)) + lane * ac.bits;
      uint32_t psum = 0, tmin = 0xFFFFFFFFu, tmax = 0u;
      unsigned long long wsum = 0;
      if (sparse_tile) agg_sparse_private(words, words - lane * ac.bits, ac.bits, m, ac.need_sum != 0, ac.need_minmax != 0, wsum, tmin, tmax);
      else if (lane_active) agg_private_dispatch(ac.bits, words, m, ac.need_sum != 0, ac.need_minmax != 0, psum, wsum, tmin, tmax);
      wsum += psum;
      if constexpr (kAccInLds) {
        acc_lds->sum[a][threadIdx.x] += wsum;
        if (ac.need_minmax != 0) {
          const uint32_t lo_key = acc_lds->umin[a][threadIdx.x], hi_key = acc_lds->umax[a][threadIdx.x];
          acc_lds->umin[a][threadIdx.x] = tmin < lo_key ? tmin : lo_key;
          acc_lds->umax[a][threadIdx.x] = tmax > hi_key ? tmax : hi_key;
        }
      } else {
        sum[0] += wsum;
        umin[0] = tmin < umin[0] ? tmin : umin[0];
        umax[0] = tmax > umax[0] ? tmax : umax[0];
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
    unsigned long long my_sum;
    uint32_t my_min, my_max;
    if constexpr (kAccInLds) { my_sum = acc_lds->sum[a][threadIdx.x]; my_min = acc_lds->umin[a][threadIdx.x]; my_max = acc_lds->umax[a][threadIdx.x]; }
    else { my_sum = sum[a]; my_min = umin[a]; my_max = umax[a]; }
    mine.sum[a] = wave_sum_i64((long long)my_sum);
    // unsigned keys below 2^31 -> the int32 keys of BlockPartial; lanes that matched nothing keep the identities
    mine.kmin[a] = wave_min_i32(my_min == 0xFFFFFFFFu ? 0x7FFFFFFF : (int32_t)my_min);
    mine.kmax[a] = wave_max_i32(count == 0ull ? (int32_t)0x80000000 : (int32_t)my_max);
  }
  if (lane == 0) red[wave_in_block] = mine;
  __syncthreads();
  publish_block_partial(p, red, waves_per_block, fold_flag_ptr, block_index, num_blocks);
}

template <int kAggSlots>
__global__ __launch_bounds__(kBlockThreads, PG_PRIVATE_WAVES) void scan_private_kernel(const ScanParams p) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  __shared__ PrivateAccLds<kAggSlots> acc;
  __shared__ uint32_t set_lds[kSetLdsWords];
  scan_private_body<kAggSlots>(p, blockIdx.x, gridDim.x, red, &fold_flag, &acc, nullptr, set_lds);
}

// The same kernel with the transducer walk inside (kFsm): a leap-frogging root AND of at most four leaves-as-inputs and four states.
template <int kAggSlots>
__global__ __launch_bounds__(kBlockThreads, PG_PRIVATE_FSM_WAVES) void scan_private_fsm_kernel(const ScanParams p) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  __shared__ PrivateAccLds<kAggSlots> acc;
  __shared__ FsmWalkLds fsm_lds;
  __shared__ uint32_t set_lds[kSetLdsWords];
  scan_private_body<kAggSlots, ScanParams, true>(p, blockIdx.x, gridDim.x, red, &fold_flag, &acc, &fsm_lds, set_lds);
}

// Many queries, one launch (pg_execute_batch): workgroups [block_first[i], block_first[i + 1]) work on items[i] -- its own columns,
// filter program, record buffer, arrival counters and pinned host record, so every query folds and publishes its result on its own
// while the others are still scanning.  BaseCombineOperator runs a query's segments as tasks of a thread pool
// (core/operator/combine/BaseCombineOperator.java:85-142); a server holds hundreds of few-million-row segments, each a ~20 us kernel
// when launched alone -- launch latency, the ramp of an empty chip and its drain are then most of the device time.
struct BatchParams {
  const ScanParams* items;           // device memory
  const uint32_t* block_first;       // [num_items + 1] device memory
  int32_t num_items;
  int32_t reserved;
};
template <int kAggSlots>
__global__ __launch_bounds__(kBlockThreads, PG_PRIVATE_WAVES) void scan_private_batch_kernel(const BatchParams bp) {
  __shared__ BlockPartial red[kBlockThreads / 64];
  __shared__ uint32_t fold_flag;
  __shared__ PrivateAccLds<kAggSlots> acc;
  __shared__ uint32_t set_lds[kSetLdsWords];
  int lo = 0, hi = bp.num_items - 1;                // the last item whose first workgroup is at or before this one
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (bp.block_first[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const uint32_t first = bp.block_first[lo];
  // The item is read through a CONSTANT-address-space reference: memory the kernel never writes, so its (wave-uniform) fields are scalar
  // loads the compiler may repeat or hoist at will, like kernel arguments.  Through a plain pointer they were VECTOR loads inside the
  // tile loop (a uniform load is only selected as s_load when no store of the kernel can have clobbered it; the loop stores bitmaps and
  // leap-frog tables): global_load + s_waitcnt vmcnt(0) + v_readfirstlane per field and tile, a third dependent round trip per tile.
  // Measured on one 1 B-row item: 0.91 ms against the single launch's 0.69.
  typedef const __attribute__((address_space(4))) ScanParams ConstantScanParams;
  const ConstantScanParams& item = *(ConstantScanParams*)(bp.items + lo);
  scan_private_body<kAggSlots>(item, blockIdx.x - first, bp.block_first[lo + 1] - first, red, &fold_flag, &acc, nullptr, set_lds);
}

// The slot of `key` in an open-addressing table of (mask + 1) slots, claiming a free one if the key is new (kHashEmpty = free).  The
// engine sizes the table to at least twice the keys that can exist, so a probe sequence always ends.
__device__ __forceinline__ uint32_t hash_slot_of(unsigned long long* keys, unsigned long long mask, unsigned long long key) {
  unsigned long long h = key * 0x9E3779B97F4A7C15ull;
  h ^= h >> 32;
  unsigned long long slot = h & mask;
  for (;;) {
    unsigned long long cur = __hip_atomic_load(keys + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kHashEmpty) {
      unsigned long long expected = kHashEmpty;
      if (__hip_atomic_compare_exchange_strong(keys + slot, &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return (uint32_t)slot;
      cur = expected;
    }
    if (cur == key) return (uint32_t)slot;
    slot = (slot + 1ull) & mask;
  }
}

// ---- group_private_kernel's LDS table: replicated and bank-interleaved ----
// A wave's 64 docs of one step go to 64 random slots: an LDS atomic is serviced in lane groups (32 lanes of a 4-byte, 16 of an 8-byte
// operation) and every extra distinct address on a bank within a group costs a cycle -- with one table, ~3 of every 4 LDS cycles of
// C3 were such conflicts (profiles/r2/pmc_c3_sq_summary.json).  The table is therefore kept R = 2^logR times, slot (g, c) at index
// g * R + c, and a lane only ever touches copy c = lane % R: lanes of different copies sit on different banks BY CONSTRUCTION (4-byte
// slots: bank (g * R + c) % 32; 8-byte: bank pair (g * R + c) % 16), so only the 32 / R (16 / R) lanes of one copy can still collide.
// The address costs what it cost before -- one v_lshl_add_u32, the shift now 2 or 3 + logR and the base a per-lane register.
// Sub-tables in order: counts (u32; absent when a SUM slot carries the count), then per aggregation SUM i64 / MIN, MAX i32 -- the
// 4-byte MIN / MAX slots are dense now (they were the low dwords of 8-byte slots: even banks only).
__host__ __device__ __forceinline__ uint32_t lds_subtable_bytes(int G, int logR, int elem) { return ((((uint32_t)G << logR) * (uint32_t)elem) + 7u) & ~7u; }
__host__ __device__ __forceinline__ uint32_t lds_group_table_bytes(const GroupParams& gp, int logR) {
  uint32_t bytes = gp.packed_agg >= 0 ? 0u : lds_subtable_bytes(gp.num_groups, logR, 4);
  for (int a = 0; a < gp.num_group_aggs; ++a) bytes += lds_subtable_bytes(gp.num_groups, logR, gp.group_aggs[a].kind == kGroupSum ? 8 : 4);
  return bytes;
}

// Long / ArrayMap holders (GroupParams.hash_kind != 0): the 64-bit key of every (matching) doc, then its slot in the hashed table; sixteen docs
// at a time.  (group_private_tile<.., kHash> and group_typed_direct_kernel<.., true>)
template <bool kMasked, typename GP>
__device__ __forceinline__ void hashed_group_slots(const GP& gp, long long tile, int lane, uint32_t m, uint32_t (&g)[32]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned long long k[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) k[j] = 0ull;
      int lvl = 0;
      for (int c = 0; c < gp.num_group_cols; ++c) {
        if (lvl < gp.hash_levels && c == gp.hash_split[lvl]) {
          // the columns so far are first table lvl's key: from here on its slot number stands for them
          unsigned long long* const keys_lvl = gp.hash_keys_lvl[lvl];
          const unsigned long long mask_lvl = gp.hash_mask_lvl[lvl];
#pragma unroll
          for (int j = 0; j < 16; ++j) if (!kMasked || ((m >> (16 * h + j)) & 1u)) k[j] = (unsigned long long)hash_slot_of(keys_lvl, mask_lvl, k[j]);
          ++lvl;
        }
        const auto& key = gp.group_keys[c];
        const int b = key.bits;
        const unsigned long long mult = gp.key_mult[c];
        const uint32_t* words = reinterpret_cast<const uint32_t*>(key.fwd + tile * (256ll * b)) + lane * b;
        uint32_t d[16];
        if (h == 0) decode16_private_dispatch<0>(b, words, d); else decode16_private_dispatch<1>(b, words, d);
#pragma unroll
        for (int j = 0; j < 16; ++j) k[j] += (unsigned long long)d[j] * mult;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        g[16 * h + j] = 0u;
        if (!kMasked || ((m >> (16 * h + j)) & 1u)) g[16 * h + j] = hash_slot_of(gp.hash_keys, gp.hash_mask, k[j]);
      }
    }
}

// PG_GROUP_MINMAX_LOOK=1: an LDS MIN / MAX slot is READ first and the atomic issued only when the doc's value beats it (a slot's extreme
// moves ~ln(n) times in n docs).  Measured on C3 (1 B rows, 1000 groups, SUM + MAX): 0.987 -> 1.028 ms -- the returning read and the
// divergent branch cost more than the atomics they save (profiles/r5/c3_look_before_atomic_ab.txt).  Off.
#ifndef PG_GROUP_MINMAX_LOOK
#define PG_GROUP_MINMAX_LOOK 0
#endif
// One 2048-doc tile.  kMasked: only the docs whose bit is set in the lane's mask `m` reach the table (a filter's result, and /
// or the docs that exist in the last, partial tile); otherwise every doc of the tile does, with no exec masking around the atomics.
// GP: GroupParams, or its constant-address-space form in device memory (an item of group_lds_batch_kernel)
template <bool kLds, bool kMasked, bool kWide = false, bool kHash = false, typename GP = GroupParams>
__device__ __forceinline__ void group_private_tile(const GP& gp, long long tile, int lane, uint32_t m, unsigned long long* t_cnt, long long* t_acc, uint8_t* lds) {
  const int G = gp.num_groups;
  const int logR = kLds ? gp.lds_log_replicas : 0;                       // (see lds_group_table_bytes)
  const uint32_t cls = (uint32_t)lane & ((1u << logR) - 1u);
  uint32_t lds_off = 0u;                                                // uniform: where the next sub-table starts
  const long long first_doc = tile * 2048 + lane * 32;
  uint32_t g[32];
  // (Round 5: the tile's columns are read one after the other -- three dependent HBM round trips per tile and wave, 72 % of a wave's cycles
  //  are waits.  One "touch" load per later column ahead of the first decode, so that the later loads find their lines on the way or in
  //  the L2, made C3 21 % SLOWER (0.985 -> 1.195 ms; profiles/r5/c3_touch_columns_ab.txt): loads return in order, the real loads queue
  //  behind the touches, and the touches compete with the other fifteen waves' demand loads for the same queues.  Round 3 saw the same in
  //  scan_private_kernel.  Latency is hidden by the resident waves, not inside a wave.)
  if constexpr (kHash) {
    hashed_group_slots<kMasked>(gp, tile, lane, m, g);
    if (gp.first_doc != nullptr) {
      // numGroupsLimit pass: which docId created every group (the reference admits keys in docId order until the limit is reached)
#pragma unroll
      for (int j = 0; j < 32; ++j) if (!kMasked || ((m >> j) & 1u)) atomicMin(gp.first_doc + g[j], (uint32_t)(first_doc + j));
      return;
    }
  } else {
  // (Round 6, measured and not kept: the lane's docs in two halves of sixteen -- keys, count and every aggregation of docs 0..15, then of
  //  16..31 -- to shorten the 32-entry group-id array's live range.  The registers it frees buy nothing (100 VGPRs instead of 101: the filter
  //  program and the decoders set the allocation) and a tile becomes SIX dependent column reads instead of three: C3 0.950 -> 1.345 ms,
  //  C3-filter 1.220 -> 1.533, C3-irregular 1.294 -> 1.913 (profiles/r6/c3_variants_ab.txt).  Bounding the kernels for five waves per SIMD
  //  (two ten-wave workgroups per CU, 96 VGPRs, 2 spilled) on top of it: 1.406 ms.  The tile's three column reads are what a wave waits
  //  for; more, shorter reads or more waves with fewer registers each make that worse, not better.)
  for (int c = 0; c < gp.num_group_cols; ++c) {
    const auto& key = gp.group_keys[c];
    const int b = key.bits;
    const uint32_t mult = (uint32_t)key.mult;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(key.fwd + tile * (256ll * b)) + lane * b;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t d[16];
      if (h == 0) decode16_private_dispatch<0>(b, words, d); else decode16_private_dispatch<1>(b, words, d);
#pragma unroll
      for (int j = 0; j < 16; ++j) g[16 * h + j] = c == 0 ? d[j] : key_term<kWide>(d[j], mult) + g[16 * h + j];
    }
  }
  }
  const bool packed = kLds && gp.packed_agg >= 0;
  if (!packed) {
    if constexpr (kLds) {
      uint32_t* cnt_lane = reinterpret_cast<uint32_t*>(lds) + cls;
#pragma unroll
      for (int j = 0; j < 32; ++j)
        if (!kMasked || ((m >> j) & 1u)) __hip_atomic_fetch_add(cnt_lane + (g[j] << logR), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      lds_off = lds_subtable_bytes(G, logR, 4);
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) if (!kMasked || ((m >> j) & 1u)) group_count<kLds>(t_cnt, g[j]);
    }
  }
  for (int a = 0; a < gp.num_group_aggs; ++a) {
    const auto& ga = gp.group_aggs[a];
    long long* acc = kLds ? reinterpret_cast<long long*>(lds + lds_off) + cls : t_acc + (long long)a * G;      // kLds: this lane's copy of a SUM sub-table,
    int32_t* acc32 = reinterpret_cast<int32_t*>(lds + lds_off) + cls;                                          //       or of a MIN / MAX one
    if constexpr (kLds) lds_off += lds_subtable_bytes(G, logR, ga.kind == kGroupSum ? 8 : 4);
    const int b = ga.bits;
    const uint32_t* words = ga.is_raw ? reinterpret_cast<const uint32_t*>(ga.fwd) + first_doc
                                      : reinterpret_cast<const uint32_t*>(ga.fwd + tile * (256ll * b)) + lane * b;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ga.dict, 0, ga.dict_bytes, 0x00020000);
    const bool is_unsigned = !ga.is_raw && ga.is_plane;
    const uint32_t one_hi = (packed && a == gp.packed_agg) ? (1u << (gp.packed_shift - 32)) : 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t d[16];
      if (ga.is_raw) {
        // raw INT column: the lane's 32 docs are 128 contiguous bytes
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = __builtin_bswap32(words[16 * h + j]);      // raw buffers are padded to whole tiles
      } else {
        if (h == 0) decode16_private_dispatch<0>(b, words, d); else decode16_private_dispatch<1>(b, words, d);
        if (ga.kind == kGroupSum && !ga.is_plane) {
          // dictionary gather (small dictionaries / PINOT_GPU_VALUE_PLANE=0)
#pragma unroll
          for (int j = 0; j < 16; ++j) d[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, d[j] * 4u, 0, 0);
        }
      }
      if (ga.kind == kGroupSum && is_unsigned) {
        // plane field; the packed count lives entirely in the high dword: the operand is the register pair {field, one_hi}
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (!kMasked || ((m >> (16 * h + j)) & 1u)) group_sum<kLds>(acc + (g[16 * h + j] << logR), (long long)(((unsigned long long)one_hi << 32) | (unsigned long long)d[j]));
      } else if (ga.kind == kGroupSum) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (!kMasked || ((m >> (16 * h + j)) & 1u)) group_sum<kLds>(acc + (g[16 * h + j] << logR), (long long)(int32_t)d[j]);
      } else if (ga.kind == kGroupMin) {
        // LDS table: LOOK before the atomic.  A slot's extreme moves ~ln(n) times in n docs (a few thousand docs per group, copy and
        // workgroup: a handful of updates), so after the first tiles nearly every doc finds the slot already at or beyond its value and the
        // read-modify-write -- the expensive LDS operation, serialised per bank -- is not issued at all.  A stale read only costs an atomic
        // that changes nothing: the extreme is monotonic.  (PG_GROUP_MINMAX_LOOK=0 at compile time: the unconditional atomic of rounds 1-4.)
        int32_t cur[16];
        if constexpr (kLds && PG_GROUP_MINMAX_LOOK) {
#pragma unroll
          for (int j = 0; j < 16; ++j) cur[j] = __hip_atomic_load(acc32 + (g[16 * h + j] << logR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (!kMasked || ((m >> (16 * h + j)) & 1u)) {
            if constexpr (kLds) { if (!PG_GROUP_MINMAX_LOOK || (int32_t)d[j] < cur[j]) __hip_atomic_fetch_min(acc32 + (g[16 * h + j] << logR), (int32_t)d[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else group_min<false>(acc + g[16 * h + j], (int32_t)d[j]);
          }
        }
      } else {
        int32_t cur[16];
        if constexpr (kLds && PG_GROUP_MINMAX_LOOK) {
#pragma unroll
          for (int j = 0; j < 16; ++j) cur[j] = __hip_atomic_load(acc32 + (g[16 * h + j] << logR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (!kMasked || ((m >> (16 * h + j)) & 1u)) {
            if constexpr (kLds) { if (!PG_GROUP_MINMAX_LOOK || (int32_t)d[j] > cur[j]) __hip_atomic_fetch_max(acc32 + (g[16 * h + j] << logR), (int32_t)d[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else group_max<false>(acc + g[16 * h + j], (int32_t)d[j]);
          }
        }
      }
    }
  }
}


// `block_index` of `num_blocks`: the workgroup's place among those that work on this parameter block (the whole grid, or one item's share
// of group_lds_batch_kernel's launch).  gp.zero_identity: MIN / MAX reach the global table as keys whose identity is 0 (see the flush).
template <bool kLdsTable, bool kWide, bool kHash, typename GP>
__device__ __forceinline__ void group_private_body(const GP& gp, uint32_t block_index, uint32_t num_blocks, uint8_t* smem) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const long long total_waves = (long long)num_blocks * waves_per_block;
  const int G = gp.num_groups;
  const int NA = gp.num_group_aggs;
  unsigned long long* t_cnt;
  long long* t_acc;
  const int logR = kLdsTable ? gp.lds_log_replicas : 0;
  if constexpr (kLdsTable) {
    t_cnt = nullptr;
    t_acc = nullptr;
    const int slots = G << logR;
    uint32_t off = 0u;
    if (gp.packed_agg < 0) {
      for (int i = threadIdx.x; i < slots; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
      off = lds_subtable_bytes(G, logR, 4);
    }
    for (int a = 0; a < NA; ++a) {
      const int kind = gp.group_aggs[a].kind;
      if (kind == kGroupSum) for (int i = threadIdx.x; i < slots; i += blockDim.x) reinterpret_cast<long long*>(smem + off)[i] = 0ll;
      else for (int i = threadIdx.x; i < slots; i += blockDim.x) reinterpret_cast<int32_t*>(smem + off)[i] = kind == kGroupMin ? 0x7FFFFFFF : (int32_t)0x80000000u;
      off += lds_subtable_bytes(G, logR, kind == kGroupSum ? 8 : 4);
    }
    __syncthreads();
  } else {
    t_cnt = gp.table_count;
    t_acc = gp.table_acc;
  }
  // the filter's dictId sets (IN lists) behind the table in the dynamic LDS, when the host made room for them (GroupParams.set_lds_off)
  uint32_t* set_lds = nullptr;
  if (gp.scan.set_leaves_in_lds != 0 && gp.set_lds_off >= 0) { set_lds = reinterpret_cast<uint32_t*>(smem + gp.set_lds_off); stage_filter_sets(gp.scan, set_lds); }
  const long long num_tiles = ((long long)gp.scan.num_docs + 2047) / 2048;
  const long long wave = (long long)block_index * waves_per_block + wave_in_block;
  uint32_t entries = 0u;
  const bool listed = gp.scan.tile_list != nullptr;        // index-driven filters: only the tiles index_and_kernel listed hold a match
  const long long tile_limit = listed ? (long long)*gp.scan.tile_count : num_tiles;
  for (long long tile_it = wave; tile_it < tile_limit; tile_it += total_waves) {
    const long long tile = listed ? (long long)gp.scan.tile_list[tile_it] : tile_it;
    // the filter (if any) in the same lane-private layout: bit j of the lane's mask = doc 32*lane + j of the tile
    uint32_t m = eval_filter_private(gp.scan, tile, lane, entries, nullptr, set_lds);
    const long long rem = (long long)gp.scan.num_docs - (tile * 2048 + lane * 32);
    m &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u));
    if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) continue;
    if (__builtin_amdgcn_ballot_w64(m != 0xFFFFFFFFu) == 0ull) group_private_tile<kLdsTable, false, kWide, kHash>(gp, tile, lane, m, t_cnt, t_acc, smem);
    else group_private_tile<kLdsTable, true, kWide, kHash>(gp, tile, lane, m, t_cnt, t_acc, smem);
  }
  flush_filter_entries(gp.scan, entries);

  if constexpr (kLdsTable) {
    // flush: thread g folds the R copies of group g (the packed count and sum fold as one add: together they stay below the
    // bounds the packing was chosen for, which are per workgroup), then one global atomic per group and accumulator
    __syncthreads();
    const int R = 1 << logR;
    const bool packed = gp.packed_agg >= 0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
      uint32_t off = 0u;
      unsigned long long c = 0ull;
      if (!packed) {
        for (int r = 0; r < R; ++r) c += (unsigned long long)reinterpret_cast<const uint32_t*>(smem)[(g << logR) + r];
        off = lds_subtable_bytes(G, logR, 4);
      } else {
        uint32_t poff = 0u;
        for (int a = 0; a < gp.packed_agg; ++a) poff += lds_subtable_bytes(G, logR, gp.group_aggs[a].kind == kGroupSum ? 8 : 4);
        unsigned long long pk = 0ull;
        for (int r = 0; r < R; ++r) pk += reinterpret_cast<const unsigned long long*>(smem + poff)[(g << logR) + r];
        c = pk >> gp.packed_shift;
      }
      if (c != 0ull) __hip_atomic_fetch_add(&gp.table_count[g], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int a = 0; a < NA; ++a) {
        const int kind = gp.group_aggs[a].kind;
        long long* slot = gp.table_acc + (long long)a * G + g;
        if (kind == kGroupSum) {
          unsigned long long v = 0ull;
          for (int r = 0; r < R; ++r) v += reinterpret_cast<const unsigned long long*>(smem + off)[(g << logR) + r];
          if (packed && a == gp.packed_agg) v &= (1ull << gp.packed_shift) - 1ull;
          if (c != 0ull) __hip_atomic_fetch_add(slot, (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const int32_t* t = reinterpret_cast<const int32_t*>(smem + off) + (g << logR);
          int32_t v = t[0];
          for (int r = 1; r < R; ++r) v = kind == kGroupMin ? (t[r] < v ? t[r] : v) : (t[r] > v ? t[r] : v);
          if (c != 0ull) {
            // zero_identity (the shared batch launch: its tables are only ever memset): MIN as 2^31 - v, MAX as v + 2^31 + 1, both in
            // [1, 2^32] under fetch_max -- a slot nobody touched stays 0 whatever its kind (group_zero_identity_decode on the host)
            if (gp.zero_identity != 0) __hip_atomic_fetch_max(slot, kind == kGroupMin ? 0x80000000ll - (long long)v : (long long)v + 0x80000001ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (kind == kGroupMin) __hip_atomic_fetch_min(slot, (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_max(slot, (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        off += lds_subtable_bytes(G, logR, kind == kGroupSum ? 8 : 4);
      }
    }
    if (gp.host_table != nullptr) {
      // An item of the shared launch publishes ITSELF (round 6b): every workgroup arrives once its own atomics are acknowledged; the one
      // whose arrival completes the item's share reads the slice back (agent-scope loads: where the atomics were performed), writes it
      // through to the pinned host image (system-scope stores, acknowledged before the sequence number: store_host_record_by_wave0's
      // reasoning), zeroes what it read and resets the counter -- the launch leaves table and counter as it found them.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                             // (the LDS table is dead from here: its first dword is the flag)
      uint32_t* const flag = reinterpret_cast<uint32_t*>(smem);
      if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(gp.scan.done_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == num_blocks ? 1u : 0u;
      __syncthreads();
      if (*flag != 0u) {
        const int words = G * (1 + NA);
        unsigned long long* const slice = gp.table_count;         // count[G] | acc[NA][G]: one allocation (enqueue_group_launch)
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
          const unsigned long long v = __hip_atomic_load(&slice[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&gp.host_table[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (v != 0ull) __hip_atomic_store(&slice[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
          __hip_atomic_store(gp.scan.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&gp.scan.host_out->seq, gp.scan.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
}

// (the forms without an LDS table are launched with kBlockThreads threads: at that bound the register allocation has the whole file --
//  at the 1024-thread bound of the LDS-table form they spilled ~40 registers inside the tile loop)
// (the LDS-table forms -- this kernel's and group_lds_batch_kernel's -- are bounded together: most threads of a workgroup, and the waves per
//  SIMD the register allocation must allow; the engine sizes workgroups and their number per CU from the compiled kernels' registers)
#ifndef PG_GROUP_LDS_THREADS
#define PG_GROUP_LDS_THREADS kGroupBlockThreads
#define PG_GROUP_LDS_WAVES 4
#endif
template <bool kLdsTable, bool kWide = false, bool kHash = false>
__global__ __launch_bounds__(kLdsTable ? PG_GROUP_LDS_THREADS : kBlockThreads, kLdsTable ? PG_GROUP_LDS_WAVES : 1) void group_private_kernel(const GroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  group_private_body<kLdsTable, kWide, kHash>(gp, blockIdx.x, gridDim.x, smem);
}

// Many group-bys, one launch (pg_execute_batch): the LDS-table form over items -- workgroups [block_first[i], block_first[i + 1]) work
// on items[i], each item with its own key / aggregation columns, filter program and slice of ONE global table allocation (count[G] |
// acc[NA][G], all-zero before the launch: zero_identity).  The dynamic LDS is the largest item's table.  What GroupByCombineOperator
// (core/operator/combine/GroupByCombineOperator.java:102-165) gets from a task per segment, for the many small segments of a server.
struct GroupBatchParams {
  const GroupParams* items;          // device memory
  const uint32_t* block_first;       // [num_items + 1] device memory
  int32_t num_items;
  int32_t reserved;
};
template <bool kWide = false>      // (a template so that only pg_unit_group_batch.hip instantiates it)
__global__ __launch_bounds__(PG_GROUP_LDS_THREADS, PG_GROUP_LDS_WAVES) void group_lds_batch_kernel(const GroupBatchParams bp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int lo = 0, hi = bp.num_items - 1;                // the last item whose first workgroup is at or before this one
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (bp.block_first[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const uint32_t first = bp.block_first[lo];
  typedef const __attribute__((address_space(4))) GroupParams ConstantGroupParams;      // (scalar loads of the item's fields: see scan_private_batch_kernel)
  const ConstantGroupParams& item = *(ConstantGroupParams*)(bp.items + lo);
  group_private_body<true, kWide, false>(item, blockIdx.x - first, bp.block_first[lo + 1] - first, smem);
}

static __global__ void init_group_table_kernel(GroupParams gp) {
  const long long G = gp.num_groups;
  for (int lvl = 0; lvl < gp.hash_levels; ++lvl)
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g <= (long long)gp.hash_mask_lvl[lvl]; g += (long long)gridDim.x * blockDim.x) gp.hash_keys_lvl[lvl][g] = kHashEmpty;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += (long long)gridDim.x * blockDim.x) {
    if (gp.hash_kind != 0) gp.hash_keys[g] = kHashEmpty;
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

// ORs one Roaring container (array / bitset / run) into a 1024-word LDS window; the whole workgroup takes part.
__device__ __forceinline__ void expand_container_into(const uint8_t* __restrict__ inv, const DevContainer c, unsigned long long* w) {
  const uint8_t* payload = inv + c.offset;
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
      atomicOr(&w[j], v);
    }
  } else {
    for (uint32_t r = threadIdx.x; r < c.num_runs; r += blockDim.x) {
      const uint32_t start = load_u16(payload + 2 + 4 * r);
      uint32_t end = start + load_u16(payload + 4 + 4 * r);         // inclusive
      end = end > 65535u ? 65535u : end;                            // a malformed run must not write past the 1024-word window
      for (uint32_t wi = start >> 6; wi <= (end >> 6); ++wi) {
        const uint32_t lo = wi == (start >> 6) ? (start & 63u) : 0u;
        const uint32_t hi = wi == (end >> 6) ? (end & 63u) : 63u;
        const unsigned long long mask = (hi - lo == 63u ? ~0ull : ((1ull << (hi - lo + 1u)) - 1ull)) << lo;
        atomicOr(&w[wi], mask);
      }
    }
  }
}

// One workgroup per 64 Ki-doc window (key = blockIdx.x): it looks its container up in the posting's sorted directory
// slice, builds the 8 KiB window in LDS and either STORES it (first posting of a leaf: no separate zero-fill pass,
// windows without a container become zeros) or ORs it into the bitmap (further postings of an IN / range leaf).
static __global__ __launch_bounds__(kBlockThreads) void roaring_expand_kernel(const uint8_t* __restrict__ inv, const DevContainer* __restrict__ dir,
                                                                       int first, int count, unsigned long long* bitmap, long long num_words, int or_mode) {
  __shared__ unsigned long long w[1024];
  __shared__ int found;
  const uint32_t key = blockIdx.x;
  if (threadIdx.x == 0) {
    int lo = first, hi = first + count - 1, f = -1;
    while (lo <= hi) {                       // containers of one posting are sorted by key
      const int mid = (lo + hi) >> 1;
      const uint32_t k = dir[mid].key;
      if (k < key) lo = mid + 1; else if (k > key) hi = mid - 1; else { f = mid; break; }
    }
    found = f;
  }
  for (int j = threadIdx.x; j < 1024; j += blockDim.x) w[j] = 0ull;
  __syncthreads();
  const long long base = (long long)key * 1024;
  if (found < 0) {
    if (!or_mode) for (int j = threadIdx.x; j < 1024; j += blockDim.x) if (base + j < num_words) bitmap[base + j] = 0ull;
    return;
  }
  expand_container_into(inv, dir[found], w);
  __syncthreads();
  for (int j = threadIdx.x; j < 1024; j += blockDim.x) {
    if (base + j >= num_words) continue;
    const unsigned long long v = w[j];
    if (or_mode) { if (v != 0ull) bitmap[base + j] |= v; }
    else bitmap[base + j] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// index_and_kernel (the inverted-index children of a root AND, intersected container by container) lives in pg_index_and.h /
// pg_unit_index_and.hip; its helpers and the kernels behind it follow.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(4))) Dwords4 { uint32_t x, y, z, w; };

// 16 bytes at byte offset 16 * index of a stream that starts `lead` bytes into the 4-byte aligned `origin`.
__device__ __forceinline__ Dwords4 load16_stream(const uint32_t* __restrict__ origin, uint32_t lead, long long index) {
  const uint32_t* p = origin + 4 * index;
  Dwords4 a = *reinterpret_cast<const Dwords4*>(p);
  if (lead != 0u) {                                       // uniform over the wave
    const uint32_t e = p[4];                              // the column's buffer is padded: this never leaves it
    a.x = __builtin_amdgcn_alignbyte(a.y, a.x, lead);
    a.y = __builtin_amdgcn_alignbyte(a.z, a.y, lead);
    a.z = __builtin_amdgcn_alignbyte(a.w, a.z, lead);
    a.w = __builtin_amdgcn_alignbyte(e, a.w, lead);
  }
  return a;
}

__device__ __forceinline__ void or_doc(uint32_t* w32, uint32_t doc) { atomicOr(&w32[doc >> 5], 1u << (doc & 31u)); }

// Completes a sparsely stored result (IndexAndParams.sparse_out) for a reader that does not go by the tile list: zeros in every
// 2048-doc tile without a match.
static __global__ __launch_bounds__(256) void index_and_zero_unlisted_kernel(const WindowInfo* __restrict__ info, unsigned long long* __restrict__ out, long long num_words) {
  const long long pairs = num_words / 2;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += (long long)gridDim.x * blockDim.x) {
    const uint32_t mask = info[p >> 9].tiles;           // 512 pairs per window, 16 per tile
    if (!((mask >> ((p >> 4) & 31)) & 1u)) reinterpret_cast<uint4*>(out)[p] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// Turns index_and_kernel's per-window {tile mask, cardinality} into the ascending tile list, its length and the cardinality.
// Workgroup b owns windows [256 b, 256 b + 256): it sums the tile counts of everything before them (a few thousand 8-byte
// entries out of L2), scans its own and writes its slice of the list; the last workgroup also stores the totals.
static __global__ __launch_bounds__(256) void index_and_finalize_kernel(const WindowInfo* __restrict__ info, int num_windows, uint32_t* __restrict__ tile_list,
                                                                        uint32_t* __restrict__ tile_count, unsigned long long* __restrict__ cardinality) {
  __shared__ unsigned long long part_card[4];
  __shared__ uint32_t part_tiles[4];
  __shared__ uint32_t wave_total[4];
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  const int first = (int)blockIdx.x * 256;
  const bool last_block = blockIdx.x + 1 == gridDim.x;
  unsigned long long before_card = 0ull;
  uint32_t before_tiles = 0u;
  // (two entries per 16-byte load, four loads in flight per thread: the whole prefix is one or two round trips)
  const uint4* info2 = reinterpret_cast<const uint4*>(info);
  for (int w = (int)threadIdx.x; w < first / 2; w += 1024) {
    uint4 e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = w + 256 * k < first / 2 ? info2[w + 256 * k] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < 4; ++k) { before_tiles += (uint32_t)(__builtin_popcount(e[k].x) + __builtin_popcount(e[k].z)); before_card += (unsigned long long)e[k].y + e[k].w; }
  }
  before_tiles = (uint32_t)wave_sum_i64((long long)before_tiles);
  before_card = (unsigned long long)wave_sum_i64((long long)before_card);
  if (lane == 0) { part_tiles[wave] = before_tiles; part_card[wave] = before_card; }
  const int w = first + (int)threadIdx.x;
  const WindowInfo mine = w < num_windows ? info[w] : WindowInfo{0u, 0u};
  const uint32_t my_tiles = (uint32_t)__builtin_popcount(mine.tiles);
  // inclusive scan inside the wave
  uint32_t scan = my_tiles;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)scan, d, 64); if (lane >= d) scan += o; }
  if (lane == 63) wave_total[wave] = scan;
  const unsigned long long my_card = (unsigned long long)wave_sum_i64((long long)mine.docs);
  __syncthreads();
  uint32_t pos = part_tiles[0] + part_tiles[1] + part_tiles[2] + part_tiles[3];
  for (int k = 0; k < wave; ++k) pos += wave_total[k];
  pos += scan - my_tiles;
  if (tile_list != nullptr) {
    uint32_t m = mine.tiles;
    while (m) { const int t = __builtin_ctz(m); m &= m - 1u; tile_list[pos++] = (uint32_t)w * 32u + (uint32_t)t; }
  }
  if (last_block) {
    __syncthreads();
    if (lane == 0) part_card[wave] += my_card;          // each wave's own docs on top of its share of the prefix
    __syncthreads();
    if (threadIdx.x == 0) {
      *cardinality = part_card[0] + part_card[1] + part_card[2] + part_card[3];
      *tile_count = part_tiles[0] + part_tiles[1] + part_tiles[2] + part_tiles[3] + wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Value-plane materialisation (one-time, per summed column): plane[doc] = dictionary[dictId[doc]] - base, bit-packed
// with `w` bits in the SAME big-endian MSB-first stream format as the forward index, so the scan kernels decode it
// with the same code.  It trades HBM capacity (288 GB) for the per-row dictionary gather, which on MI355X costs as
// much L2 capacity as streaming ~22 bytes (profiles/r1/microbench.jsonl).  w == 32 stores big-endian int32 values.
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kBlockThreads) void materialize_plane_kernel(const DevColumn col, uint8_t* __restrict__ out, int w, int32_t base,
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
    for (int k = 0; k < kMaxTileSteps; ++k) {
      const uint32_t d = b <= 25 ? decode_step<false>(wave_lds, dec, k, b) : decode_step<true>(wave_lds, dec, k, b);
      const long long doc = (long long)tile * kMaxTileDocs + k * 64 + lane;
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

// Empirical HBM read ceiling of the box: a pure 16 B/lane read-reduce over a buffer, nothing else.
static __global__ __launch_bounds__(kBlockThreads) void stream_read_probe_kernel(const uint4* __restrict__ src, size_t n16, unsigned long long* out) {
  unsigned long long acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const uint4 v = src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x1234567ull) out[0] = acc;       // never true for the zero-filled probe buffer's checksum: keeps the loads alive
}

static __global__ void fill_words_kernel(unsigned long long* words, long long n, unsigned long long value) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) words[i] = value;
}

// ------------------------------------------------------------------------------------------------
// Direct-indexed HBM group table -> dense result (group-by key spaces above the array-based threshold, the reference's
// IntMapBasedHolder range: DictionaryBasedGroupKeyGenerator.java:164-184,415-490).  The table has one slot per raw key; the
// result keeps only the groups that exist, in ascending raw-key order, so the host never copies or walks the whole table.
//   group_chunk_count_kernel   per 2048-slot chunk: how many groups exist (and pass the first-doc cut), and the docs they hold
//   group_chunk_scan_kernel    exclusive scan of the chunk counts (one workgroup)
//   group_compact_kernel       order-preserving write of (raw key, count, accumulators[, first doc]) at chunk offset + rank
//   group_first_doc_kernel     numGroupsLimit reached: first docId of every group, to keep the groups the reference would have
//                              created first (IntGroupIdMap.getGroupId hands out ids in order of first appearance and refuses new
//                              keys once _size == groupIdUpperBound, :1022-1047)
// ------------------------------------------------------------------------------------------------
constexpr int kGroupChunk = 2048;

__device__ __forceinline__ bool group_kept(const unsigned long long* table_count, const uint32_t* first_doc, uint32_t max_first_doc, long long g, long long G) {
  if (g >= G || table_count[g] == 0ull) return false;
  return first_doc == nullptr || first_doc[g] <= max_first_doc;
}

static __global__ __launch_bounds__(256) void group_chunk_count_kernel(const unsigned long long* __restrict__ table_count, const uint32_t* __restrict__ first_doc,
                                                                        uint32_t max_first_doc, int G, uint32_t* __restrict__ chunk_counts,
                                                                        unsigned long long* __restrict__ total_docs) {
  __shared__ uint32_t s_groups;
  __shared__ unsigned long long s_docs;
  if (threadIdx.x == 0) { s_groups = 0u; s_docs = 0ull; }
  __syncthreads();
  const long long base = (long long)blockIdx.x * kGroupChunk;
  uint32_t n = 0;
  unsigned long long docs = 0;
  for (int j = threadIdx.x; j < kGroupChunk; j += 256) {
    const long long g = base + j;
    if (g < G) docs += table_count[g];                      // numDocsScanned counts the docs of dropped groups too
    n += group_kept(table_count, first_doc, max_first_doc, g, G) ? 1u : 0u;
  }
  atomicAdd(&s_groups, n);
  atomicAdd(&s_docs, docs);
  __syncthreads();
  if (threadIdx.x == 0) {
    chunk_counts[blockIdx.x] = s_groups;
    if (total_docs && s_docs) atomicAdd(total_docs, s_docs);
  }
}

// chunk_offsets[i] = sum of chunk_counts[0..i); chunk_offsets[num_chunks] = total.  One workgroup of 1024 threads.
static __global__ __launch_bounds__(1024) void group_chunk_scan_kernel(const uint32_t* __restrict__ chunk_counts, int num_chunks, uint32_t* __restrict__ chunk_offsets) {
  __shared__ uint32_t s[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0u;
  __syncthreads();
  for (int base = 0; base < num_chunks; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < num_chunks ? chunk_counts[i] : 0u;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                    // Hillis-Steele inclusive scan
      const uint32_t add = threadIdx.x >= (unsigned)d ? s[threadIdx.x - d] : 0u;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < num_chunks) chunk_offsets[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += s[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) chunk_offsets[num_chunks] = carry;
}

static __global__ __launch_bounds__(256) void group_compact_kernel(const unsigned long long* __restrict__ table_count, const long long* __restrict__ table_acc,
                                                                    int num_aggs, int G, const uint32_t* __restrict__ first_doc, uint32_t max_first_doc,
                                                                    const uint32_t* __restrict__ chunk_offsets, uint32_t total, int32_t* __restrict__ out_ids,
                                                                    unsigned long long* __restrict__ out_counts, long long* __restrict__ out_acc,
                                                                    uint32_t* __restrict__ out_first_doc) {
  __shared__ uint32_t wave_base[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * kGroupChunk;
  uint32_t running = chunk_offsets[blockIdx.x];
  for (int round = 0; round < kGroupChunk / 256; ++round) {         // consecutive slots per round keep the output in raw-key order
    const long long g = base + round * 256 + threadIdx.x;
    const bool keep = group_kept(table_count, first_doc, max_first_doc, g, G);
    const unsigned long long ballot = __builtin_amdgcn_ballot_w64(keep);
    const uint32_t rank_in_wave = (uint32_t)__builtin_popcountll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_base[wave] = (uint32_t)__builtin_popcountll(ballot);
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < 4; ++w) { const uint32_t c = wave_base[w]; if (w < wave) before += c; all += c; }
    if (keep) {
      const uint32_t k = running + before + rank_in_wave;
      if (out_ids) out_ids[k] = (int32_t)g;
      if (out_counts) out_counts[k] = table_count[g];
      if (out_acc) for (int a = 0; a < num_aggs; ++a) out_acc[(size_t)a * total + k] = table_acc[(long long)a * G + g];
      if (out_first_doc) out_first_doc[k] = first_doc[g];
    }
    running += all;
    __syncthreads();
  }
}

__device__ __forceinline__ uint32_t read_packed(const uint8_t* fwd, long long doc, int b);

// keys[ids[i]] -> out[i] (the 64-bit keys of the slots that hold a group)
static __global__ __launch_bounds__(256) void gather_u64_kernel(const unsigned long long* __restrict__ src, const int32_t* __restrict__ ids, int n, unsigned long long mask,
                                                                 unsigned long long* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = src[(unsigned long long)(uint32_t)ids[i] & mask];
}
static __global__ __launch_bounds__(256) void gather_u64_by_key_kernel(const unsigned long long* __restrict__ src, const unsigned long long* __restrict__ keys, int n,
                                                                        unsigned long long mask, unsigned long long* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = src[keys[i] & mask];
}

// One thread per doc: atomicMin of the docId into the slot of the doc's raw key.  `bitmap` is the filter result in doc order
// (nullptr = every doc matches).  Only runs when a query created more groups than numGroupsLimit.
static __global__ __launch_bounds__(256) void group_first_doc_kernel(const GroupParams gp, const unsigned long long* __restrict__ bitmap, uint32_t* __restrict__ first_doc,
                                                                      long long doc_lo, long long doc_hi) {
  for (long long doc = doc_lo + (long long)blockIdx.x * blockDim.x + threadIdx.x; doc < doc_hi; doc += (long long)gridDim.x * blockDim.x) {
    if (bitmap && !((bitmap[doc >> 6] >> (doc & 63)) & 1ull)) continue;
    uint32_t key = 0;
    for (int c = 0; c < gp.num_group_cols; ++c) key += read_packed(gp.group_keys[c].fwd, doc, gp.group_keys[c].bits) * (uint32_t)gp.group_keys[c].mult;
    atomicMin(&first_doc[key], (uint32_t)doc);
  }
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

// Null-key image of a nullable dictionary column (pg_engine.hip, GROUP BY under enableNullHandling): the same fixed-bit stream with
// dictId = cardinality wherever the doc is null, `bits_out` wide.  Lane-private layout like the scan kernels: a lane owns 32 docs of a
// 2048-doc tile = bits_in dwords in, bits_out dwords out, written MSB-first like PinotDataBitSet.writeInt.
static __global__ __launch_bounds__(256) void build_nullkey_fwd_kernel(const uint8_t* __restrict__ fwd, int bits_in, const unsigned long long* __restrict__ nulls,
                                                                       uint8_t* __restrict__ out, int bits_out, uint32_t null_id, int num_tiles) {
  const int lane = threadIdx.x & 63;
  for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < (long long)num_tiles; tile += (long long)gridDim.x * 4) {
    const uint32_t* words = reinterpret_cast<const uint32_t*>(fwd + tile * (256ll * bits_in)) + lane * bits_in;
    uint32_t d[32];
    decode16_private_dispatch<0>(bits_in, words, *reinterpret_cast<uint32_t(*)[16]>(&d[0]));
    decode16_private_dispatch<1>(bits_in, words, *reinterpret_cast<uint32_t(*)[16]>(&d[16]));
    const uint32_t null_mask = reinterpret_cast<const uint32_t*>(nulls)[tile * 64 + lane];      // the lane's 32 docs are one dword of the doc-order bitmap
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + tile * (256ll * bits_out)) + lane * bits_out;
    unsigned long long acc = 0ull;
    int have = 0, k = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const uint32_t id = ((null_mask >> j) & 1u) ? null_id : d[j];
      acc = (acc << bits_out) | (unsigned long long)id;
      have += bits_out;
      if (have >= 32) { dst[k++] = __builtin_bswap32((uint32_t)(acc >> (have - 32))); have -= 32; }
    }
  }
}

// ---- raw INT / LONG columns as group keys (NoDictionarySingleColumnGroupKeyGenerator / NoDictionaryMultiColumnGroupKeyGenerator) ----
// The reference keys such a column by VALUE (value -> group id in order of first appearance, NoDictionarySingleColumnGroupKeyGenerator
// .java:100-113, 240-247).  Here the column gets a KEY IMAGE the first time it is grouped by: the fixed-bit stream of (value - min), as
// if it had the dense dictionary {min .. max} -- every group-by kernel then reads it like any dictionary column, and a key comes back as
// min + digit.  HBM capacity spent on a derived stream instead of a value-keyed hash in the kernels, like the value planes.
static __global__ __launch_bounds__(256) void raw_min_max_kernel(const uint8_t* __restrict__ raw, int value_bytes, long long num_docs, long long* __restrict__ out_min_max) {
  long long lo = 0x7FFFFFFFFFFFFFFFll, hi = (long long)0x8000000000000000ull;
  for (long long doc = (long long)blockIdx.x * blockDim.x + threadIdx.x; doc < num_docs; doc += (long long)gridDim.x * blockDim.x) {
    const long long v = value_bytes == 4 ? (long long)(int32_t)__builtin_bswap32(reinterpret_cast<const uint32_t*>(raw)[doc])
                                         : (long long)__builtin_bswap64(reinterpret_cast<const unsigned long long*>(raw)[doc]);
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  lo = wave_min_i64(lo);
  hi = wave_max_i64(hi);
  if ((threadIdx.x & 63) == 0) { atomicMin(out_min_max, lo); atomicMax(out_min_max + 1, hi); }
}

// Lane-private layout like the scan kernels: a lane owns 32 docs of a 2048-doc tile and writes their bits_out-bit digits MSB-first
// (PinotDataBitSet.writeInt).  Docs past numDocs get digit 0 (the raw buffer is padded to whole tiles with zeros, which need not be >= base).
static __global__ __launch_bounds__(256) void build_raw_key_image_kernel(const uint8_t* __restrict__ raw, int value_bytes, long long base, uint8_t* __restrict__ out, int bits_out,
                                                                          int num_tiles, long long num_docs) {
  const int lane = threadIdx.x & 63;
  for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < (long long)num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048 + (long long)lane * 32;
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + tile * (256ll * bits_out)) + lane * bits_out;
    unsigned long long acc = 0ull;
    int have = 0, k = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const long long doc = first + j;
      const long long v = value_bytes == 4 ? (long long)(int32_t)__builtin_bswap32(reinterpret_cast<const uint32_t*>(raw)[doc])
                                           : (long long)__builtin_bswap64(reinterpret_cast<const unsigned long long*>(raw)[doc]);
      const uint32_t id = doc < num_docs ? (uint32_t)(unsigned long long)(v - base) : 0u;
      acc = (acc << bits_out) | (unsigned long long)id;
      have += bits_out;
      if (have >= 32) { dst[k++] = __builtin_bswap32((uint32_t)(acc >> (have - 32))); have -= 32; }
    }
  }
}

// Wide value plane (pg_engine.hip want_wide_plane): out[doc] = the 8-byte dictionary entry of the doc's dictId, big-endian like the
// value area of a raw LONG / DOUBLE forward index.  One-time build per column: a plain gather.
static __global__ __launch_bounds__(256) void materialize_wide_plane_kernel(const uint8_t* __restrict__ fwd, int bits, const unsigned long long* __restrict__ dict64,
                                                                            unsigned long long* __restrict__ out, int num_docs) {
  for (long long doc = (long long)blockIdx.x * blockDim.x + threadIdx.x; doc < (long long)num_docs; doc += (long long)gridDim.x * blockDim.x)
    out[doc] = __builtin_bswap64(dict64[read_packed(fwd, doc, bits)]);
}

static __global__ void gather_values_kernel(DevColumn col, long long value_base, const int32_t* __restrict__ doc_ids, int n, int32_t* out_dict_ids,
                                     int32_t* out_ints, long long* out_longs, double* out_doubles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long doc = doc_ids[i];
  long long lv = 0;
  double dv = 0.0;
  const bool integral = col.vkind == kValI32 || col.vkind == kValI64;
  const bool want_value = out_ints || out_longs || out_doubles;
  if (col.is_raw) {
    if (out_dict_ids) out_dict_ids[i] = -1;
    if (col.vkind == kValI32) lv = (int32_t)__builtin_bswap32(*reinterpret_cast<const uint32_t*>(col.fwd + doc * 4));
    else if (col.vkind == kValI64) lv = (long long)__builtin_bswap64(*reinterpret_cast<const unsigned long long*>(col.fwd + doc * 8));
    else if (col.vkind == kValF32) dv = (double)__uint_as_float(__builtin_bswap32(*reinterpret_cast<const uint32_t*>(col.fwd + doc * 4)));
    else dv = __longlong_as_double((long long)__builtin_bswap64(*reinterpret_cast<const unsigned long long*>(col.fwd + doc * 8)));
  } else {
    const uint32_t d = read_packed(col.fwd, doc, col.bits);
    if (out_dict_ids) out_dict_ids[i] = (int32_t)d;
    if (want_value) {
      if (col.vkind == kValI32) lv = value_base + (long long)col.dict[d];
      else if (col.vkind == kValI64) lv = reinterpret_cast<const long long*>(col.dict)[d];
      else dv = reinterpret_cast<const double*>(col.dict)[d];
    }
  }
  if (integral) dv = (double)lv;
  else {
    // Java (long) double: NaN -> 0, saturating
    lv = (dv != dv) ? 0ll : (dv >= 9.2233720368547758e18 ? 0x7FFFFFFFFFFFFFFFll : (dv <= -9.2233720368547758e18 ? (long long)0x8000000000000000ull : (long long)dv));
  }
  if (out_ints) out_ints[i] = (int32_t)lv;
  if (out_longs) out_longs[i] = lv;
  if (out_doubles) out_doubles[i] = dv;
}

}  // namespace pg
