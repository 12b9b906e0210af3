// pg_index_and.h -- index_and_kernel: AND of inverted-index leaves at RoaringBitmap-container granularity -- AndDocIdSet.iterator's
// index-based branch (core/operator/docidsets/AndDocIdSet.java:127-172: the bitmaps of all index-based children are and-ed, smallest
// first) and BitmapCollection (core/operator/filter/BitmapCollection.java:58-128) for inverted (NOT_EQ / NOT_IN) members; the postings
// are what InvertedIndexFilterOperator (core/operator/filter/InvertedIndexFilterOperator.java:60-145) would hand it.
//
// A 65 536-doc WINDOW is the unit of work and one WAVEFRONT works on it (a workgroup is a single wave: its barriers cost nothing and
// nothing is shared with other waves).  The accumulator, 1024 words, stays in REGISTERS: lane l owns the word pairs (2l + 128 i,
// 2l + 1 + 128 i), i = 0..7, so bitset containers, dense children and the result move as one 16-byte access per lane and never touch
// LDS.  Array and run containers (and children that OR several postings: a child is the OR of the postings of its matching dictIds)
// are scattered into the wave's 8 KB LDS window with ds_or, read back by the owning lanes and re-zeroed in the same pass.  Serialized
// containers start at any byte offset (odd array lengths, a run-flag bitset): they are read with aligned dword loads and re-aligned
// with v_alignbyte.  A window is finished as soon as the accumulator is empty: later children's containers are never read.
//
// Round 6: the kernel is bound by what a window WAITS for, not by bytes (profiles/r5: it read 0.72x its charge in 3.6x the roofline
// time; ~11.6 us per window, 60 % of it waiting).  Three changes, all about dependent round trips:
//   * PERSISTENT waves: the grid is what is resident (launch_index_and), a wave takes windows key, key + grid, ...: no workgroup
//     launch, kernel-argument fetch and LDS zeroing per window (one wave per window left ~30 % of the wave slots empty between a
//     retiring workgroup and its successor), and the NEXT window's directory lookups (one posting per lane, an interpolated guess
//     confirmed by one 24-byte load) are issued before this window's containers are touched -- the first of a window's dependent
//     round trips leaves the critical path;
//   * an ARRAY container's pieces are all requested before the first is scattered (it was one 1 KB piece per loop trip: a 4096-entry
//     array was eight dependent HBM round trips);
//   * fewer instructions around them: the guess is a float multiply (it was a 64-bit multiply and divide, ~150 instructions), the
//     tail masking of docs past numDocs runs in the last window only.
// Per window the kernel leaves {mask of its 32 2048-doc tiles that hold a match, cardinality}; index_and_finalize_kernel (pg_kernels.h)
// turns those into the ascending list of matching tiles where somebody reads one.
#ifndef PG_INDEX_AND_H
#define PG_INDEX_AND_H
#include "pg_kernels.h"

namespace pg {

#ifndef PG_INDEX_AND_PIECES
#define PG_INDEX_AND_PIECES 8          // 1 KB pieces of a container requested before the first is used (a whole bitset; one address register for all of them)
#endif
constexpr int kAndPiecesInFlight = PG_INDEX_AND_PIECES;
#ifndef PG_INDEX_AND_WAVES
#define PG_INDEX_AND_WAVES 4           // wavefronts (= windows in flight) per SIMD the register allocation must allow
#endif

// ap.first / ap.count of this lane's posting, its directory and count / windows (postings of frequent values have a container in
// (nearly) every window: the slot of window `key` is close to key * count / windows).
struct AndLanePosting {
  const DevContainer* dir;
  int first, count;
  float per_window;
};
static_assert(sizeof(DevContainer) == 24, "index_and_kernel moves a directory entry as six dwords");

__device__ __forceinline__ int and_guess_slot(const AndLanePosting& lp, uint32_t key) {
  int s = (int)((float)key * lp.per_window);
  s = s > lp.count - 1 ? lp.count - 1 : s;
  return lp.first + s;
}

// The directory entry guessed for window `key`, requested a window ahead: six LDS-DMA dword loads (global_load_lds_dword: the data goes
// from the memory system straight into LDS, dword f of lane l's entry at guess[64 f + l]) -- nothing waits for them and no register holds
// them while the window in front is worked on.  (Held in registers, the 24 bytes were spilled by the allocator right behind the loads:
// an s_waitcnt vmcnt(0) that put the round trip back on the critical path.)
__device__ __forceinline__ void and_issue_guess(const AndLanePosting& lp, uint32_t key, uint32_t* guess) {
  if (lp.count > 0) {
    const uint32_t* e = reinterpret_cast<const uint32_t*>(lp.dir + and_guess_slot(lp, key));
#pragma unroll
    for (int f = 0; f < 6; ++f) __builtin_amdgcn_global_load_lds((gbl_void_t*)(e + f), (lds_void_t*)(guess + 64 * f), 4, 0, 0);
  }
}

// The container of window `key`, given the guess that has landed in LDS (the caller waited for vmcnt(0)); a binary search over what the
// guess leaves otherwise.  Entry = {key, cardinality | type, num_runs | offset}.
__device__ __forceinline__ bool and_resolve(const AndLanePosting& lp, uint32_t key, const uint32_t* guess, uint32_t lane, uint2* key_card, uint2* type_runs, uint2* offset) {
  if (lp.count <= 0) return false;
  const uint32_t gkey = guess[lane];
  if (gkey == key) {
    *key_card = make_uint2(gkey, guess[64 + lane]); *type_runs = make_uint2(guess[128 + lane], guess[192 + lane]); *offset = make_uint2(guess[256 + lane], guess[320 + lane]);
    return true;
  }
  const int slot = and_guess_slot(lp, key);
  int lo = lp.first, hi = lp.first + lp.count - 1;
  // keys are distinct and ascending, so the slot is no further from the guess than the keys are apart
  if (gkey < key) { lo = slot + 1; const long long far = (long long)slot + (long long)(key - gkey); hi = far < hi ? (int)far : hi; }
  else { hi = slot - 1; const long long far = (long long)slot - (long long)(gkey - key); lo = far > lo ? (int)far : lo; }
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const uint32_t k = lp.dir[mid].key;
    if (k < key) lo = mid + 1;
    else if (k > key) hi = mid - 1;
    else {
      const uint2* e = reinterpret_cast<const uint2*>(lp.dir + mid);
      *key_card = e[0]; *type_runs = e[1]; *offset = e[2];
      return true;
    }
  }
  return false;
}

// index_and_kernel (one query): one-wavefront workgroups.  index_and_batch_kernel: a workgroup is kAndBatchBlockWaves wavefronts that share
// NOTHING but the record they publish together at the very end (a quarter of the records for the item's folding workgroup to read).  Inside
// the window loop a wave only ever reads LDS it wrote itself: LDS operations of one wave execute in order, so what separates its scatter
// from its read-back is a compiler fence and the LDS counter, never a workgroup barrier (the waves run different numbers of windows and
// children: an s_barrier there would deadlock).  Measured on one query (C5 at 1 B rows, profiles/r6): four-wave workgroups lose 8 us of 62
// to one-wave ones -- the dispatcher places a workgroup only where four wave slots and 38 KB of LDS are free at once.
constexpr int kAndBlockWaves = 1;
constexpr int kAndBatchBlockWaves = 4;
// the sum of a 32-bit figure over the wavefront (a window holds at most 65 536 docs), the same in every lane
__device__ __forceinline__ uint32_t and_wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
  return v;
}
__device__ __forceinline__ void and_wave_sync() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// acc &= bits ^ flip; returns what is left of the four words, or-ed together
__device__ __forceinline__ uint32_t and_take(uint4& acc, const uint4& bits, uint32_t flip) {
  acc.x &= bits.x ^ flip; acc.y &= bits.y ^ flip; acc.z &= bits.z ^ flip; acc.w &= bits.w ^ flip;
  return acc.x | acc.y | acc.z | acc.w;
}

// A 16-byte piece of a serialized container in two phases, so that several pieces are REQUESTED before the first is used.  The container is
// a raw buffer (base = the 4-byte aligned, uniform `origin` the payload starts `lead` bytes into, num_records = its bytes): a piece is the
// 16 bytes at `voffset` (per lane) + `soffset` (uniform) -- ONE address register per piece, or one for all of them, where global loads
// took a 64-bit pair each -- and, when lead != 0, the dword behind them; a lane whose offset lies outside the container reads zeros and
// touches no memory (which is how pieces nobody wants, and the lanes past the end of a short array, are skipped without a branch or a
// clamp).  Phase two re-aligns with v_alignbyte.  kLead is a template parameter, not a branch: a uniform `if (lead != 0)` between the
// loads of an unrolled loop put an s_waitcnt vmcnt(0) behind every piece -- eight dependent HBM round trips for one 8 KB bitset (rounds 2-5).
typedef uint32_t and_u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kAndNoPiece = 0x80000000u;            // a voffset outside every container
__device__ __forceinline__ __amdgpu_buffer_rsrc_t and_container(const uint8_t* origin, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, (int)((bytes + 3u) & ~3u), 0x00020000);
}
template <bool kLead>
__device__ __forceinline__ void and_request16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, uint32_t soffset, Dwords4* a, uint32_t* e) {
  const and_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voffset, (int)soffset, 0);
  a->x = v.x; a->y = v.y; a->z = v.z; a->w = v.w;
  if constexpr (kLead) *e = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voffset, (int)(soffset + 16u), 0); else *e = 0u;
}
template <bool kLead>
__device__ __forceinline__ Dwords4 and_aligned16(Dwords4 a, uint32_t e, uint32_t lead) {
  if constexpr (kLead) {
    a.x = __builtin_amdgcn_alignbyte(a.y, a.x, lead);
    a.y = __builtin_amdgcn_alignbyte(a.z, a.y, lead);
    a.z = __builtin_amdgcn_alignbyte(a.w, a.z, lead);
    a.w = __builtin_amdgcn_alignbyte(e, a.w, lead);
  }
  return a;
}

// Eight 16-bit docIds of an array container into the window.  and_scatter8_all: every one of them exists.  and_scatter8_some: only the
// first `left` (1 .. 8) do -- the lane that holds the container's end; a doc past it is replaced by the lane's first (setting a bit twice
// changes nothing), so nothing runs under a per-doc exec mask: round 6's SQ counters had 606 scalar instructions and 137 branches per
// window, most of them the eight `left > k` guards of every scattered piece.
__device__ __forceinline__ void and_scatter8_all(uint32_t* w32, const Dwords4& d) {
  or_doc(w32, d.x & 0xffffu); or_doc(w32, d.x >> 16);
  or_doc(w32, d.y & 0xffffu); or_doc(w32, d.y >> 16);
  or_doc(w32, d.z & 0xffffu); or_doc(w32, d.z >> 16);
  or_doc(w32, d.w & 0xffffu); or_doc(w32, d.w >> 16);
}
__device__ __forceinline__ void and_scatter8_some(uint32_t* w32, const Dwords4& d, uint32_t left) {
  const uint32_t first = d.x & 0xffffu;
  or_doc(w32, first);
  or_doc(w32, left > 1u ? d.x >> 16 : first);
  or_doc(w32, left > 2u ? d.y & 0xffffu : first);
  or_doc(w32, left > 3u ? d.y >> 16 : first);
  or_doc(w32, left > 4u ? d.z & 0xffffu : first);
  or_doc(w32, left > 5u ? d.z >> 16 : first);
  or_doc(w32, left > 6u ? d.w & 0xffffu : first);
  or_doc(w32, left > 7u ? d.w >> 16 : first);
}

// A single bitset container, straight from HBM into the owning lanes.  Behind the smaller children only the pieces where something still
// stands are wanted (`probe`; three postings of 1 / 256, 1 / 64 and 1 / 16 of the docs: ~4 docs of the window are left when the 8 KB
// bitset of the largest comes up -- AndDocIdSet.java:127-165 and-s smallest first for the same reason): a lane whose piece is not wanted
// asks for kAndNoPiece -- zeros, no memory access, and the loads stay unconditional (a load under an exec mask is waited for before the
// branch is left).  Returns what is left of the accumulator, or-ed together.
template <bool kLead>
__device__ __forceinline__ uint32_t and_single_bitset(__amdgpu_buffer_rsrc_t rsrc, uint32_t lead, uint32_t lane16, bool probe, uint32_t flip, uint4 (&acc)[8]) {
  uint32_t any = 0u;
#pragma unroll
  for (int h = 0; h < 8; h += kAndPiecesInFlight) {
    Dwords4 d[kAndPiecesInFlight];
    uint32_t e[kAndPiecesInFlight];
#pragma unroll
    for (int j = 0; j < kAndPiecesInFlight; ++j) {
      const int i = h + j;
      const bool wanted = !probe || (acc[i].x | acc[i].y | acc[i].z | acc[i].w) != 0u;
      and_request16<kLead>(rsrc, wanted ? lane16 : kAndNoPiece, 1024u * (uint32_t)i, &d[j], &e[j]);
    }
#pragma unroll
    for (int j = 0; j < kAndPiecesInFlight; ++j) {
      const Dwords4 x = and_aligned16<kLead>(d[j], e[j], lead);
      any |= and_take(acc[h + j], make_uint4(x.x, x.y, x.z, x.w), flip);
    }
  }
  return any;
}

// An array container -- sorted 16-bit docIds, eight per lane and 1 KB piece -- scattered into the wave's LDS window; kAndPiecesInFlight pieces
// are requested before the first of them is scattered (it was one piece per loop trip: a 4096-entry array was eight dependent round trips).
template <bool kLead>
__device__ __forceinline__ void and_scatter_array(__amdgpu_buffer_rsrc_t rsrc, uint32_t lead, uint32_t n, uint32_t lane, uint32_t lane16, uint32_t* w32) {
  for (uint32_t p0 = 0u; 512u * p0 < n; p0 += (uint32_t)kAndPiecesInFlight) {
    Dwords4 d[kAndPiecesInFlight];
    uint32_t e[kAndPiecesInFlight];
#pragma unroll
    for (int j = 0; j < kAndPiecesInFlight; ++j) and_request16<kLead>(rsrc, lane16, 1024u * (p0 + (uint32_t)j), &d[j], &e[j]);
#pragma unroll
    for (int j = 0; j < kAndPiecesInFlight; ++j) {
      const uint32_t piece_first = 512u * (p0 + (uint32_t)j);              // uniform
      if (piece_first + 512u <= n) {
        and_scatter8_all(w32, and_aligned16<kLead>(d[j], e[j], lead));      // a whole piece: every lane has eight docs
      } else if (piece_first < n) {
        const uint32_t e0 = piece_first + 8u * lane;                       // the container's last piece: the lanes past its end sit out
        if (e0 < n) and_scatter8_some(w32, and_aligned16<kLead>(d[j], e[j], lead), n - e0);
      }
    }
  }
}

// A bitset container OR-ed into the window (one of several postings of a child).
template <bool kLead>
__device__ __forceinline__ void and_or_bitset(__amdgpu_buffer_rsrc_t rsrc, uint32_t lead, uint32_t lane, uint32_t lane16, uint4* window) {
#pragma unroll
  for (int h = 0; h < 8; h += kAndPiecesInFlight) {
    Dwords4 d[kAndPiecesInFlight];
    uint32_t e[kAndPiecesInFlight];
#pragma unroll
    for (int j = 0; j < kAndPiecesInFlight; ++j) and_request16<kLead>(rsrc, lane16, 1024u * (uint32_t)(h + j), &d[j], &e[j]);
#pragma unroll
    for (int j = 0; j < kAndPiecesInFlight; ++j) {
      const Dwords4 x = and_aligned16<kLead>(d[j], e[j], lead);
      uint4 o = window[lane + 64u * (uint32_t)(h + j)];      // this lane's own words: no other lane writes them between the two accesses
      o.x |= x.x; o.y |= x.y; o.z |= x.z; o.w |= x.w;
      window[lane + 64u * (uint32_t)(h + j)] = o;
    }
  }
}

// The kernel's body over a parameter block P -- IndexAndParams out of the kernel arguments, or its constant-address-space form in device
// memory (an item of index_and_batch_kernel: its wave-uniform fields are scalar loads either way).  Wavefront `first_wave` of the `num_waves`
// that work on this block takes windows first_wave, first_wave + num_waves, ...; `window` / `guess` / `red` / `fold_flag`: the wave's LDS.
template <bool kRecord, int kWaves, typename P>
__device__ __forceinline__ void index_and_body(const P& ap, const uint32_t num_windows, const uint32_t block_index, const uint32_t num_blocks, uint4* window, uint32_t* guess,
                                               BlockPartial* red, uint32_t* fold_flag) {
  const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (uniform: the wave's LDS bases are scalars)
  const uint32_t first_wave = block_index * (uint32_t)kWaves + wave_in_block, num_waves = num_blocks * (uint32_t)kWaves;
  window += 512u * wave_in_block;                       // this wave's 8 KB and its guesses
  guess += 6u * 64u * wave_in_block;
  uint32_t* w32 = reinterpret_cast<uint32_t*>(window);
  const int lane0 = (int)(threadIdx.x & 63u);

  // ---- lane t looks posting t up, in every window of this wave ----
  AndLanePosting lp;
  lp.dir = nullptr; lp.first = 0; lp.count = 0; lp.per_window = 0.0f;
  if (lane0 < ap.num_postings) {
    lp.dir = ap.child[ap.posting_child[lane0]].dir;
    lp.first = ap.first[lane0];
    lp.count = ap.count[lane0];
    lp.per_window = (float)lp.count / (float)num_windows;
  }
  uint32_t key = first_wave;
  if (key < num_windows) and_issue_guess(lp, key, guess);
#pragma unroll
  for (int i = 0; i < 8; ++i) window[lane0 + 64 * i] = make_uint4(0u, 0u, 0u, 0u);
  and_wave_sync();

  // record mode (an item of a batch): what the wave found in all of its windows -- wave-UNIFORM totals (scalar registers: per-lane
  // accumulators held across the window loop cost nine vector registers, 23 more of them in scratch inside the loop)
  unsigned long long u_card = 0ull;
  long long u_sum[kMaxAndGather] = {0ll, 0ll};
  int32_t u_min[kMaxAndGather] = {0x7FFFFFFF, 0x7FFFFFFF}, u_max[kMaxAndGather] = {(int32_t)0x80000000, (int32_t)0x80000000};
  // gather mode: the survivors' values of this lane in ONE window
  unsigned long long gsum[kMaxAndGather];
  uint32_t gmin[kMaxAndGather], gmax[kMaxAndGather];

  for (; key < num_windows; key += num_waves) {
    // (the lane number is made opaque once per window: left alone, LICM hoists every lane-dependent term of the window's code -- the
    //  pieces' offsets, the tail's doc numbers, the gathers' multiplies -- out of this loop, ~70 registers of them into scratch.  For
    //  the same reason everything a lane addresses is a uniform base plus ONE 32-bit offset: 64-bit per-lane indices of the eight
    //  pieces, hoisted out of the loop over the children, were another ~50.)
    uint32_t lane = (uint32_t)lane0;
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int a = 0; a < kMaxAndGather; ++a) { gsum[a] = 0ull; gmin[a] = 0xFFFFFFFFu; gmax[a] = 0u; }
    const uint32_t lane16 = lane * 16u;                   // this lane's 16 bytes of a 1 KB piece
    const long long base = (long long)key * 1024;         // first word of the window (uniform)
    const uint32_t words_here = (uint32_t)(ap.num_words - base < 1024 ? ap.num_words - base : 1024);   // a multiple of 32

    // ---- this window's containers (the guesses were issued a window ago), then the next window's guesses ----
    int c_valid = 0;
    uint32_t c_card = 0, c_type = 0, c_runs = 0, c_off_lo = 0, c_off_hi = 0;
    wait_vmem();                                           // the guesses have landed (and the last window's stores have left)
    {
      uint2 kc, tr, of;
      if (and_resolve(lp, key, guess, lane, &kc, &tr, &of)) {
        c_valid = 1; c_card = kc.y; c_type = tr.x; c_runs = tr.y;
        c_off_lo = of.x; c_off_hi = of.y;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ... and have been read, before the next ones may land on them
    if (key + num_waves < num_windows) and_issue_guess(lp, key + num_waves, guess);

    // the accumulator starts as the AND's identity; a child is taken in as (bits ^ flip) & acc, flip = all ones for NOT_EQ / NOT_IN members
    uint4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    bool alive = true;                                    // uniform
    for (int c = 0; c < ap.num_children && alive; ++c) {
      const auto& ch = ap.child[c];
      const uint32_t flip = ch.exclusive ? ~0u : 0u;
      uint32_t any = 0u;
      if (ch.dense != nullptr) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(ch.dense + base);
#pragma unroll
        for (int h = 0; h < 8; h += kAndPiecesInFlight) {
          uint4 x[kAndPiecesInFlight];
#pragma unroll
          for (int j = 0; j < kAndPiecesInFlight; ++j) {
            const uint32_t at = lane16 + 1024u * (uint32_t)(h + j);
            x[j] = make_uint4(0u, 0u, 0u, 0u);
            if (at < 8u * words_here) x[j] = *reinterpret_cast<const uint4*>(src + at);
          }
#pragma unroll
          for (int j = 0; j < kAndPiecesInFlight; ++j) any |= and_take(acc[h + j], x[j], flip);
        }
      } else {
        int found = 0, last = -1;
        for (int q = ch.posting_begin; q < ch.posting_end; ++q) if (__builtin_amdgcn_readlane(c_valid, q)) { found++; last = q; }
        if (found == 0) {
          if (!ch.exclusive) { alive = false; break; }    // this child has nothing in the window: neither has the AND
          continue;                                       // NOT (nothing here) = everything: the accumulator stays as it is
        } else if (found == 1 && __builtin_amdgcn_readlane((int)c_type, last) == 1) {
          // a single bitset container: straight from HBM into the owning lanes
          const unsigned long long off = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)c_off_lo, last) |
                                         ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)c_off_hi, last) << 32);
          const uint32_t lead = (uint32_t)(off & 3ull);
          const __amdgpu_buffer_rsrc_t rsrc = and_container(ch.inv + (off - lead), lead + 8192u);
          const bool probe = c > 0 && !ch.exclusive;
          any = lead != 0u ? and_single_bitset<true>(rsrc, lead, lane16, probe, flip, acc) : and_single_bitset<false>(rsrc, lead, lane16, probe, flip, acc);
        } else {
          for (int q = ch.posting_begin; q < ch.posting_end; ++q) {
            if (!__builtin_amdgcn_readlane(c_valid, q)) continue;
            const uint32_t type = (uint32_t)__builtin_amdgcn_readlane((int)c_type, q);
            const unsigned long long off = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)c_off_lo, q) |
                                           ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)c_off_hi, q) << 32);
            if (type == 0u) {
              const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)c_card, q);
              const uint32_t lead = (uint32_t)(off & 3ull);
              const __amdgpu_buffer_rsrc_t rsrc = and_container(ch.inv + (off - lead), lead + 2u * n);
              if (lead != 0u) and_scatter_array<true>(rsrc, lead, n, lane, lane16, w32); else and_scatter_array<false>(rsrc, lead, n, lane, lane16, w32);
            } else if (type == 1u) {
              const uint32_t lead = (uint32_t)(off & 3ull);
              const __amdgpu_buffer_rsrc_t rsrc = and_container(ch.inv + (off - lead), lead + 8192u);
              if (lead != 0u) and_or_bitset<true>(rsrc, lead, lane, lane16, window); else and_or_bitset<false>(rsrc, lead, lane, lane16, window);
            } else {
              // run container: u16 count, then (start, length - 1) pairs
              const uint32_t runs = (uint32_t)__builtin_amdgcn_readlane((int)c_runs, q);
              const unsigned long long first = off + 2ull;
              const uint32_t lead = (uint32_t)(first & 3ull);
              const uint8_t* origin = ch.inv + (first - lead);
              for (uint32_t r = lane; r < runs; r += 64u) {
                uint32_t pair = *reinterpret_cast<const uint32_t*>(origin + 4u * r);
                if (lead != 0u) pair = __builtin_amdgcn_alignbyte(*reinterpret_cast<const uint32_t*>(origin + 4u * r + 4u), pair, lead);
                const uint32_t start = pair & 0xffffu;
                uint32_t end = start + (pair >> 16);                            // inclusive
                end = end > 65535u ? 65535u : end;                              // a malformed run must not leave the window
                for (uint32_t wi = start >> 5; wi <= (end >> 5); ++wi) {
                  const uint32_t lo = wi == (start >> 5) ? (start & 31u) : 0u;
                  const uint32_t hi = wi == (end >> 5) ? (end & 31u) : 31u;
                  atomicOr(&w32[wi], (hi - lo == 31u ? ~0u : ((1u << (hi - lo + 1u)) - 1u)) << lo);
                }
              }
            }
          }
          and_wave_sync();
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const uint4 x = window[lane + 64u * (uint32_t)i];
            window[lane + 64u * (uint32_t)i] = make_uint4(0u, 0u, 0u, 0u);
            any |= and_take(acc[i], x, flip);
          }
          and_wave_sync();
        }
      }
      alive = __builtin_amdgcn_ballot_w64(any != 0u) != 0ull;
    }

    uint8_t* const out = ap.out != nullptr ? reinterpret_cast<uint8_t*>(ap.out + base) : nullptr;      // uniform
    if (!alive) {
      // nothing stands: no tile holds a match (a dense reader of the whole bitmap still gets its zeros)
      if (lane == 0u && out != nullptr) ap.window_info[key] = WindowInfo{0u, 0u};
      if (out != nullptr && !ap.sparse_out) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t at = lane16 + 1024u * (uint32_t)i;
          if (at < 8u * words_here) *reinterpret_cast<uint4*>(out + at) = make_uint4(0u, 0u, 0u, 0u);
        }
      }
      continue;
    }

    // ---- the window's result: docs past numDocs cleared (an exclusive child sets them), words, tile mask, cardinality ----
    uint32_t tiles = 0u;
    uint32_t card = 0u;
    const long long docs_left = (long long)ap.num_docs - base * 64;      // docs of the segment from this window on
    if (docs_left < 65536) {                                             // uniform: only the segment's last window
      const uint32_t left = (uint32_t)docs_left;                         // 1 .. 65 535
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t* xs = &acc[i].x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t first = 128u * (lane + 64u * (uint32_t)i) + 32u * (uint32_t)k;
          if (first >= left) xs[k] = 0u; else if (left - first < 32u) xs[k] &= (1u << (left - first)) - 1u;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint4 x = acc[i];
      const uint32_t bit0 = 128u * (lane + 64u * (uint32_t)i);           // window-relative doc of this pair's first bit
      card += (uint32_t)(__builtin_popcount(x.x) + __builtin_popcount(x.y) + __builtin_popcount(x.z) + __builtin_popcount(x.w));
      if (ap.gather_cols != 0 && (x.x | x.y | x.z | x.w) != 0u) {
        // the survivors' values, read here (a handful per window by the planner's estimate): doc -> (tile, lane, position) of the packed
        // column, as a 32-bit offset from the window's first tile (32 tiles of at most 256 * 31 bytes)
        const uint32_t xs4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t w = xs4[k];
          while (w != 0u) {
            const uint32_t j = (uint32_t)__builtin_ctz(w);
            w &= w - 1u;
            const uint32_t doc = bit0 + 32u * (uint32_t)k + j;            // window-relative: tile doc >> 11, lane (doc >> 5) & 63, position doc & 31
#pragma unroll
            for (int a = 0; a < kMaxAndGather; ++a) {
              if (a < ap.gather_cols) {
                const auto& gc = ap.gather_col[a];
                const uint32_t b = (uint32_t)gc.bits, bit = (doc & 31u) * b;
                const uint8_t* tile0 = gc.fwd + (long long)key * (32ll * 256ll) * (long long)b;      // uniform
                const uint32_t at = (doc >> 11) * (256u * b) + (((doc >> 5) & 63u) * b + (bit >> 5)) * 4u;
                const Dwords2 d = *reinterpret_cast<const Dwords2*>(tile0 + at);
                const unsigned long long x64 = ((unsigned long long)__builtin_bswap32(d.x) << 32) | (unsigned long long)__builtin_bswap32(d.y);
                const uint32_t val = (uint32_t)(x64 >> (64u - (bit & 31u) - b)) & ((1u << b) - 1u);
                gsum[a] += val;
                gmin[a] = val < gmin[a] ? val : gmin[a];
                gmax[a] = val > gmax[a] ? val : gmax[a];
              }
            }
          }
        }
      }
    }
    if (out != nullptr) {
      // (only a kernel that reads the bitmap wants the words and the window's tile mask: COUNT(*) and the gathered aggregation take
      //  neither, and the ballots, the store decisions and the mask's scalar arithmetic were a fifth of their windows' instructions)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint4 x = acc[i];
        // pair 2 (l + 64 i) lies in tile 4 i + (l >> 4)
        const unsigned long long nz = __builtin_amdgcn_ballot_w64((x.x | x.y | x.z | x.w) != 0u);
        // sparse output: only the tiles that hold a match are stored (the list-driven kernels read no others; anybody else calls
        // index_and_zero_unlisted_kernel first)
        const bool store = !ap.sparse_out || ((nz >> (16u * (lane >> 4))) & 0xffffull) != 0ull;
        const uint32_t at = lane16 + 1024u * (uint32_t)i;
        if (store && at < 8u * words_here) *reinterpret_cast<uint4*>(out + at) = x;
#pragma unroll
        for (int g = 0; g < 4; ++g) if ((nz >> (16 * g)) & 0xffffull) tiles |= 1u << (4 * i + g);
      }
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)and_wave_sum_u32(card));
    if (lane == 0u && out != nullptr) ap.window_info[key] = WindowInfo{tiles, total};
    if constexpr (kRecord) {
      u_card += (unsigned long long)total;
    } else {
      // ---- one query: the window's figures onto counter line (window & 63), fire-and-forget ----
      // (kAndCardinalityShards counters, one 128-byte line each: ONE counter made the kernel 58 -> 194 us on C5-sparse -- ~3 700 same-address
      //  device-scope atomics at ~37 ns apiece, each holding its wave's slot until it retires)
      if (lane == 0u && ap.shards != nullptr && total != 0u)
        __hip_atomic_fetch_add(ap.shards + (size_t)(key & (uint32_t)(kAndCardinalityShards - 1)) * 16, (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ap.gather_cols != 0 && total != 0u) {
#pragma unroll
      for (int a = 0; a < kMaxAndGather; ++a) {
        if (a >= ap.gather_cols) continue;
        const long long s = wave_sum_i64((long long)gsum[a]);
        // (keys are below 2^31: dictIds / plane fields -- the signed wave reductions take them as they are)
        const int32_t kmin = wave_min_i32((int32_t)(gmin[a] == 0xFFFFFFFFu ? 0x7FFFFFFF : gmin[a]));
        const int32_t kmax = wave_max_i32((int32_t)gmax[a]);
        if constexpr (kRecord) {
          u_sum[a] += (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(s >> 32)) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)s));
          const int32_t m0 = __builtin_amdgcn_readfirstlane(kmin), m1 = __builtin_amdgcn_readfirstlane(kmax);
          u_min[a] = m0 < u_min[a] ? m0 : u_min[a];
          u_max[a] = m1 > u_max[a] ? m1 : u_max[a];
        } else if (lane == 0u) {
          unsigned long long* o = ap.shards + (size_t)(key & (uint32_t)(kAndCardinalityShards - 1)) * 16 + 1 + 3 * a;      // (word 0 of the line: the cardinality)
          __hip_atomic_fetch_add(o, (unsigned long long)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_max(o + 1, (unsigned long long)(0xFFFFFFFFu - (uint32_t)kmin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_max(o + 2, (unsigned long long)(uint32_t)kmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }

  if constexpr (kRecord) {
    if (ap.pub.partials != nullptr) {
      // ---- the wavefront's record, published like a scan kernel's: the last wavefront to arrive folds them into the pinned host record ----
      BlockPartial mine;
      partial_identity(mine);
      mine.count = u_card;
#pragma unroll
      for (int a = 0; a < kMaxAndGather; ++a) {
        if (a >= ap.gather_cols) continue;
        mine.sum[a] = u_sum[a];
        mine.kmin[a] = u_min[a];
        mine.kmax[a] = u_max[a];
      }
      if (lane0 == 0) red[wave_in_block] = mine;
      __syncthreads();                                     // (every wave of the workgroup gets here exactly once)
      publish_block_partial(ap.pub, red, kWaves, fold_flag, block_index, num_blocks);
    }
  }
}

static __global__ __launch_bounds__(64 * kAndBlockWaves, PG_INDEX_AND_WAVES) void index_and_kernel(const IndexAndParams ap, const uint32_t num_windows) {
  __shared__ uint4 window[512 * kAndBlockWaves];        // per wave: 1024 64-bit words; all zero whenever no child is being expanded
  __shared__ uint32_t guess[6 * 64 * kAndBlockWaves];   // per wave: the directory entries guessed for its NEXT window (and_issue_guess)
  index_and_body<false, kAndBlockWaves>(ap, num_windows, blockIdx.x, gridDim.x, window, guess, nullptr, nullptr);
}

// Many index-led queries, one launch (pg_execute_batch): workgroups [block_first[i], block_first[i + 1]) -- kAndBatchBlockWaves wavefronts each -- work on
// items[i], every item with its own postings, gathered columns and record (IndexAndParams.pub: every item folds and publishes on its own
// while the others still intersect).  What BaseCombineOperator (core/operator/combine/BaseCombineOperator.java:85-142) gets from a task per
// segment when the filter is answered by the inverted indexes (InvertedIndexFilterOperator.java:60-145) over a server's many small segments.
struct IndexAndBatchParams {
  const IndexAndParams* items;       // device memory; items[i].num_windows windows each
  const uint32_t* block_first;       // [num_items + 1] device memory
  int32_t num_items;
  int32_t reserved;
};
static __global__ __launch_bounds__(64 * kAndBatchBlockWaves, PG_INDEX_AND_WAVES) void index_and_batch_kernel(const IndexAndBatchParams bp) {
  __shared__ uint4 window[512 * kAndBatchBlockWaves];
  __shared__ uint32_t guess[6 * 64 * kAndBatchBlockWaves];
  __shared__ BlockPartial red[kAndBatchBlockWaves];
  __shared__ uint32_t fold_flag;
  int lo = 0, hi = bp.num_items - 1;                // the last item whose first workgroup is at or before this one
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (bp.block_first[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const uint32_t first = bp.block_first[lo];
  typedef const __attribute__((address_space(4))) IndexAndParams ConstantIndexAndParams;      // (scalar loads of the item's fields: see scan_private_batch_kernel)
  const ConstantIndexAndParams& item = *(ConstantIndexAndParams*)(bp.items + lo);
  index_and_body<true, kAndBatchBlockWaves>(item, (uint32_t)item.num_windows, blockIdx.x - first, bp.block_first[lo + 1] - first, window, guess, red, &fold_flag);
}

}  // namespace pg
#endif
