// pg_fsm_kernels.h -- numEntriesScannedInFilter of a leap-frogging root AND, counted on the device at any segment size: the finite-state
// transducer of pg_filter_fsm.h (host: compile_fsm; the same arithmetic on the host: fsm_count_tiled) run over the leaves' doc-order match
// bitmaps.  AndDocIdIterator.java:41-80 / OrDocIdIterator.java:52-140 / SVScanDocIdIterator.java:76-145 walk the docs one advance() at
// a time; here every chunk of docs is a function {entry state} -> {exit state, entries}:
//   fsm_tiles_kernel   one wavefront per 2048-doc tile: lane i walks its 32 docs from EVERY entry state (S <= 16 independent chains of
//                      LDS table lookups, interleaved), the 64 lane tables are composed in lane order into the tile's table
//   fsm_chain_kernel   1024 tile tables -> one table (64 per wavefront, then the 16 wavefronts' tables)
//   fsm_finish_kernel  the <= 1024 remaining tables, staged in LDS, walked from state 0.
// A table entry is  exit state | entries << 4  in 32 bits (a 1024-tile chunk: at most 2^21 docs x 15 entries); the last level adds in 64 bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

constexpr int kFsmStates = 16, kFsmInputs = 8, kFsmChunk = 1024;

struct FsmParams {
  const uint32_t* leaf[kFsmInputs];     // doc-order bitmaps, dword tile * 64 + lane = the lane's 32 docs; padded to whole tiles
  const uint8_t* delta;                 // [S << L] next state | entries << 4
  uint32_t* tables;                     // [num_tiles * S]
  int32_t num_inputs, num_states, num_docs, num_tiles;
  uint32_t* lane_front;                 // fsm_tile_fns_kernel: [num_tiles * 64 * SMAX / 4] the function of the lanes IN FRONT of every lane within its tile (a byte per entry state)
};

// Round 4b: the walk's table entry is 32 bits --  low half: the BYTE offset of the next state's row in the table, high half: the entries
// of the step -- so that a step of a chain is one LDS read, one or (row offset | input offset: the address of the next read) and one add
// of the entry's high half (both SDWA word selects: no shift, no mask); the 16 steps of a whole lane are unrolled with immediate bit
// offsets.  The first coding kept state and entries apart (shift + add + and + shift + or per step) in a loop with a lane-dependent trip
// count: 575 us for the 488 282 tiles of 1 B rows x 3 leaves x 4 states.
template <int SMAX, int LMAX>
__global__ __launch_bounds__(256) void fsm_tiles_kernel(const FsmParams p) {
  // Up to four input bits: TWO docs per table lookup (a 2 * LMAX-bit index, leaf i's bits for docs d, d + 1 side by side at 2i, 2i + 1);
  // the walk is a chain of dependent LDS reads, half as long this way.
  constexpr bool kPair = LMAX <= 4;
  constexpr int kIndexBits = kPair ? 2 * LMAX : LMAX;
  constexpr uint32_t kRowBytes = 4u << kIndexBits;                   // SMAX rows: at most 16 x 1 KB (pairs of four leaves) or 16 x 1 KB (eight leaves)
  static_assert(SMAX * kRowBytes <= 65536u, "row offsets are 16 bits");
  __shared__ uint8_t delta[SMAX << LMAX];                            // one doc: next | entries << 4 (the last tile's partial lanes)
  __shared__ uint32_t step[SMAX << kIndexBits];                      // one step (a pair of docs, or one): next row's byte offset | entries << 16
  __shared__ uint32_t lane_tables[4][64 * SMAX];                     // state | entries << 16
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.num_inputs, S = p.num_states;
  // the table is re-laid for LMAX input bits per state (unused states / inputs: entries that are never read)
  for (int i = threadIdx.x; i < (SMAX << LMAX); i += blockDim.x) {
    const int st = i >> LMAX, in = i & ((1 << LMAX) - 1);
    delta[i] = (st < S && in < (1 << L)) ? p.delta[(st << L) | in] : (uint8_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (SMAX << kIndexBits); i += blockDim.x) {
    const int st = i >> kIndexBits, idx = i & ((1 << kIndexBits) - 1);
    uint32_t next, inc;
    if constexpr (kPair) {
      int in0 = 0, in1 = 0;
      for (int l = 0; l < LMAX; ++l) { in0 |= ((idx >> (2 * l)) & 1) << l; in1 |= ((idx >> (2 * l + 1)) & 1) << l; }
      const uint32_t t0 = delta[(st << LMAX) | in0], t1 = delta[((t0 & 15u) << LMAX) | in1];
      next = t1 & 15u; inc = (t0 >> 4) + (t1 >> 4);
    } else {
      const uint32_t t0 = delta[(st << LMAX) | idx];
      next = t0 & 15u; inc = t0 >> 4;
    }
    step[i] = (next * kRowBytes) | (inc << 16);
  }
  __syncthreads();
  uint32_t* mine = lane_tables[wave];
  const char* const step_bytes = reinterpret_cast<const char*>(step);
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < p.num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048 + lane * 32;
    const long long rem = (long long)p.num_docs - first;
    const int docs = rem >= 32 ? 32 : (rem <= 0 ? 0 : (int)rem);      // (docs past numDocs do not exist; the lanes of the last tile stop at different docs)
    uint32_t w[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; ++i) w[i] = i < L ? p.leaf[i][tile * 64 + lane] : 0u;
    uint32_t st_out[SMAX], ent[SMAX];
    if (__builtin_amdgcn_ballot_w64(docs != 32) == 0ull) {
      // ---- every lane has its 32 docs (all tiles but the last): straight-line code, immediate bit offsets ----
      uint32_t row[SMAX];                                             // byte offset of the chain's current row
#pragma unroll
      for (int s = 0; s < SMAX; ++s) { row[s] = (uint32_t)s * kRowBytes; ent[s] = 0u; }
      constexpr int kDocsPerStep = kPair ? 2 : 1;
#pragma unroll
      for (int d = 0; d < 32; d += kDocsPerStep) {
        uint32_t in4 = 0u;                                            // the step's input as a byte offset inside a row
#pragma unroll
        for (int i = 0; i < LMAX; ++i) in4 |= __builtin_amdgcn_ubfe(w[i], d, kDocsPerStep) << (kDocsPerStep * i + 2);
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {                              // SMAX independent chains: their LDS reads are in flight together
          const uint32_t t = *reinterpret_cast<const uint32_t*>(step_bytes + (row[s] | in4));
          ent[s] += t >> 16;
          row[s] = t & 0xFFFFu;
        }
      }
#pragma unroll
      for (int s = 0; s < SMAX; ++s) st_out[s] = row[s] / kRowBytes;
    } else {
#pragma unroll
      for (int s = 0; s < SMAX; ++s) { st_out[s] = (uint32_t)s; ent[s] = 0u; }
      for (int d = 0; d < docs; ++d) {
        uint32_t in = 0u;
#pragma unroll
        for (int i = 0; i < LMAX; ++i) in |= __builtin_amdgcn_ubfe(w[i], d, 1) << i;
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
          const uint32_t t = delta[(st_out[s] << LMAX) | in];
          ent[s] += t >> 4;
          st_out[s] = t & 15u;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < SMAX; ++s) mine[lane * SMAX + s] = st_out[s] | (ent[s] << 16);      // (a lane: at most 32 docs x 8 entries)
    __builtin_amdgcn_wave_barrier();
    // The 64 lane tables composed in lane order, as a tree: at level j the lanes whose low j + 1 bits are zero append the table 2^j lanes
    // further on (which by then stands for 2^j lanes) to their own.  Six dependent rounds of SMAX LDS reads instead of a 64-step walk.
    // (a tile: at most 2048 docs x 8 entries = 2^14: the 16-bit halves hold)
    uint32_t c[SMAX], e[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) { c[s] = st_out[s]; e[s] = ent[s]; }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const bool active = (lane & ((2 << j) - 1)) == 0;
      if (active) {
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
          const uint32_t t = mine[(lane + (1 << j)) * SMAX + (int)c[s]];
          e[s] += t >> 16;
          c[s] = t & 0xFFFFu;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (active) {
#pragma unroll
        for (int s = 0; s < SMAX; ++s) mine[lane * SMAX + s] = c[s] | (e[s] << 16);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
#pragma unroll
      for (int s = 0; s < SMAX; ++s) if (s < S) p.tables[tile * S + s] = c[s] | (e[s] << 4);
    }
  }
}

// At most FOUR states and four input bits (a AND b AND c, a AND (b OR c), ... after minimisation): no table walk at all.  A function
// {entry state} -> {exit state} of four states is four BYTES of one register, and composing two of them is ONE v_perm_b32 (the first
// function's bytes are the selectors into the second's); the entries per entry state ride along as four more bytes, gathered by the
// same selectors.  A step of the lane's walk: one 8-byte LDS read (the pair-of-docs functions {next, entries}, indexed by the lane's
// input bits), two v_perm_b32, one add -- against four dependent table reads of fsm_tiles_kernel, which measured LDS-bound (64 lanes
// reading 64 random entries of a 256-byte row: ~5-way bank conflicts, ~100 LDS instructions per tile: 580 us per 1 B docs whatever the
// VALU did).  The 64 lane functions are composed in lane order by a tree over ds_bpermute (no LDS storage, no conflicts), the entries
// widened to 16 bits there (a tile: at most 2048 docs x 4 entries).
// The walk's LDS tables, built by the whole workgroup (256 threads): `delta` [4 << LMAX] one doc: next | entries << 4, and `pair_fn`
// [1 << 2 LMAX] two docs with that input index: x = next state of s in byte s, y = entries of s in byte s.  `src` is the machine's delta with
// `src_shift` input bits per state (pg_filter_fsm.h: L; ScanParams.fsm_delta: 4), S states, L inputs.  Ends with __syncthreads().
template <int LMAX, typename Src>
__device__ __forceinline__ void fsm_perm_build_tables(Src src, int src_shift, int S, int L, uint8_t* delta, uint2* pair_fn) {
  constexpr int kIndexBits = 2 * LMAX;
  for (int i = threadIdx.x; i < (4 << LMAX); i += blockDim.x) {
    const int st = i >> LMAX, in = i & ((1 << LMAX) - 1);
    delta[i] = (st < S && in < (1 << L)) ? src[(st << src_shift) | in] : (uint8_t)0;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < (1 << kIndexBits); idx += blockDim.x) {
    int in0 = 0, in1 = 0;
    for (int l = 0; l < LMAX; ++l) { in0 |= ((idx >> (2 * l)) & 1) << l; in1 |= ((idx >> (2 * l + 1)) & 1) << l; }
    uint32_t next = 0u, inc = 0u;
    for (int st = 0; st < 4; ++st) {
      const uint32_t t0 = delta[(st << LMAX) | in0], t1 = delta[((t0 & 15u) << LMAX) | in1];
      next |= ((t1 & 15u) & 3u) << (8 * st);
      inc |= ((t0 >> 4) + (t1 >> 4)) << (8 * st);
    }
    pair_fn[idx] = make_uint2(next, inc);
  }
  __syncthreads();
}

// One 2048-doc tile: the lane's 32 docs (input words w[0 .. LMAX), `docs` of them exist) walked from every entry state, the 64 lane
// functions composed in lane order; lane 0 leaves the tile's table in out[0 .. S).
template <int LMAX>
__device__ __forceinline__ void fsm_perm_tile(const uint32_t (&w)[LMAX], int docs, const uint8_t* delta, const uint2* pair_fn, int lane, int S, uint32_t* out) {
    uint32_t F = 0x03020100u, E = 0u;                                 // the identity; no entries
    if (__builtin_amdgcn_ballot_w64(docs != 32) == 0ull) {
      // the leaves' words regrouped once per tile so that a step's index is one bit-field extract per two leaves: nibble k of lo_even holds
      // (leaf 0: docs 4k, 4k + 1; leaf 1: docs 4k, 4k + 1), of lo_odd the same for docs 4k + 2, 4k + 3; hi_* for leaves 2 and 3
      constexpr uint32_t kM = 0x33333333u;
      const uint32_t lo_even = (w[0] & kM) | (LMAX > 1 ? (w[LMAX > 1 ? 1 : 0] & kM) << 2 : 0u);
      const uint32_t lo_odd = ((w[0] >> 2) & kM) | (LMAX > 1 ? (w[LMAX > 1 ? 1 : 0] & ~kM) : 0u);
      const uint32_t hi_even = LMAX > 2 ? ((w[LMAX > 2 ? 2 : 0] & kM) | (LMAX > 3 ? (w[LMAX > 3 ? 3 : 0] & kM) << 2 : 0u)) : 0u;
      const uint32_t hi_odd = LMAX > 2 ? (((w[LMAX > 2 ? 2 : 0] >> 2) & kM) | (LMAX > 3 ? (w[LMAX > 3 ? 3 : 0] & ~kM) : 0u)) : 0u;
#pragma unroll
      for (int d = 0; d < 32; d += 2) {
        const uint32_t lo = (d & 2) ? lo_odd : lo_even, hi = (d & 2) ? hi_odd : hi_even;
        uint32_t idx = __builtin_amdgcn_ubfe(lo, d & ~3, 4);
        if (LMAX > 2) idx |= __builtin_amdgcn_ubfe(hi, d & ~3, 4) << 4;
        const uint2 t = pair_fn[idx];
        E += __builtin_amdgcn_perm(t.y, t.y, F);                      // entries of the transition each chain takes (a lane: at most 32 docs x 4: a byte holds)
        F = __builtin_amdgcn_perm(t.x, t.x, F);
      }
    } else {
      uint32_t st[4] = {0u, 1u, 2u, 3u}, ent[4] = {0u, 0u, 0u, 0u};
      for (int d = 0; d < docs; ++d) {
        uint32_t in = 0u;
#pragma unroll
        for (int i = 0; i < LMAX; ++i) in |= __builtin_amdgcn_ubfe(w[i], d, 1) << i;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t t = delta[(st[c] << LMAX) | in];
          ent[c] += t >> 4;
          st[c] = t & 3u;
        }
      }
      F = st[0] | (st[1] << 8) | (st[2] << 16) | (st[3] << 24);
      E = ent[0] | (ent[1] << 8) | (ent[2] << 16) | (ent[3] << 24);
    }
    // entries as 16-bit fields: EA = chains 0, 1; EB = chains 2, 3
    uint32_t EA = __builtin_amdgcn_perm(0u, E, 0x0c010c00u), EB = __builtin_amdgcn_perm(0u, E, 0x0c030c02u);
    // The 64 lane functions composed in lane order, as a tree: at level j lane l appends the function of lane l + 2^j (which by then stands
    // for 2^j lanes) to its own.  Every lane computes; only the lanes that are multiples of 2^(j+1) hold something meaningful, and they
    // read only such lanes.
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int from = ((lane + (1 << j)) & 63) << 2;
      const uint32_t G = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)F);
      const uint32_t HA = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)EA), HB = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)EB);
      // selectors of the 16-bit entries H[F[s]]: bytes (2 F[s], 2 F[s] + 1) of {HB : HA}
      const uint32_t selA = (__builtin_amdgcn_perm(F, F, 0x01010000u) << 1) + 0x01000100u;
      const uint32_t selB = (__builtin_amdgcn_perm(F, F, 0x03030202u) << 1) + 0x01000100u;
      EA += __builtin_amdgcn_perm(HB, HA, selA);
      EB += __builtin_amdgcn_perm(HB, HA, selB);
      F = __builtin_amdgcn_perm(G, G, F);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t e = ((c < 2 ? EA : EB) >> (16 * (c & 1))) & 0xFFFFu;
        if (c < S) out[c] = ((F >> (8 * c)) & 3u) | (e << 4);
      }
    }
}

template <int LMAX>
__global__ __launch_bounds__(256) void fsm_tiles_perm_kernel(const FsmParams p) {
  static_assert(LMAX <= 4, "two docs per lookup: an index of at most eight bits");
  __shared__ uint8_t delta[4 << LMAX];                               // one doc: next | entries << 4
  __shared__ uint2 pair_fn[1 << (2 * LMAX)];                         // two docs with input idx: x = next state of s in byte s, y = entries of s in byte s
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.num_inputs, S = p.num_states;
  fsm_perm_build_tables<LMAX>(p.delta, L, S, L, delta, pair_fn);
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < p.num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048 + lane * 32;
    const long long rem = (long long)p.num_docs - first;
    const int docs = rem >= 32 ? 32 : (rem <= 0 ? 0 : (int)rem);      // (docs past numDocs do not exist; the lanes of the last tile stop at different docs)
    uint32_t w[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; ++i) w[i] = i < L ? p.leaf[i][tile * 64 + lane] : 0u;
    fsm_perm_tile<LMAX>(w, docs, delta, pair_fn, lane, S, p.tables + tile * S);
  }
}

// Five to eight states (an OR with three scan members, five-leaf ANDs): the same byte-function walk with a function in TWO registers
// (states 0..3, 4..7).  A step is one 16-byte LDS read and four v_perm_b32 (selector bytes 0..7 pick from the pair of source words); the
// tree gathers each 16-bit entry from four words -- two candidates per field, picked by bit 2 of the selector.  pg_filter_fsm.h
// fsm_count_perm8 is the same arithmetic on the host.
template <int LMAX>
__global__ __launch_bounds__(256) void fsm_tiles_perm8_kernel(const FsmParams p) {
  static_assert(LMAX <= 4, "two docs per lookup: an index of at most eight bits");
  constexpr int kIndexBits = 2 * LMAX;
  __shared__ uint8_t delta[8 << LMAX];                               // one doc: next | entries << 4
  __shared__ uint4 pair_fn[1 << kIndexBits];                         // two docs with input idx: x, y = next state of s in byte s (s < 4, s >= 4); z, w = entries
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.num_inputs, S = p.num_states;
  for (int i = threadIdx.x; i < (8 << LMAX); i += blockDim.x) {
    const int st = i >> LMAX, in = i & ((1 << LMAX) - 1);
    delta[i] = (st < S && in < (1 << L)) ? p.delta[(st << L) | in] : (uint8_t)0;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < (1 << kIndexBits); idx += blockDim.x) {
    int in0 = 0, in1 = 0;
    for (int l = 0; l < LMAX; ++l) { in0 |= ((idx >> (2 * l)) & 1) << l; in1 |= ((idx >> (2 * l + 1)) & 1) << l; }
    uint4 t = make_uint4(0u, 0u, 0u, 0u);
    for (int st = 0; st < 8; ++st) {
      const uint32_t t0 = delta[(st << LMAX) | in0], t1 = delta[((t0 & 7u) << LMAX) | in1];
      const uint32_t next = t1 & 7u, inc = (t0 >> 4) + (t1 >> 4);
      if (st < 4) { t.x |= next << (8 * st); t.z |= inc << (8 * st); } else { t.y |= next << (8 * (st - 4)); t.w |= inc << (8 * (st - 4)); }
    }
    pair_fn[idx] = t;
  }
  __syncthreads();
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < p.num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048 + lane * 32;
    const long long rem = (long long)p.num_docs - first;
    const int docs = rem >= 32 ? 32 : (rem <= 0 ? 0 : (int)rem);
    uint32_t w[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; ++i) w[i] = i < L ? p.leaf[i][tile * 64 + lane] : 0u;
    uint32_t Flo = 0x03020100u, Fhi = 0x07060504u, Elo = 0u, Ehi = 0u;
    if (__builtin_amdgcn_ballot_w64(docs != 32) == 0ull) {
      constexpr uint32_t kM = 0x33333333u;
      const uint32_t lo_even = (w[0] & kM) | (LMAX > 1 ? (w[LMAX > 1 ? 1 : 0] & kM) << 2 : 0u);
      const uint32_t lo_odd = ((w[0] >> 2) & kM) | (LMAX > 1 ? (w[LMAX > 1 ? 1 : 0] & ~kM) : 0u);
      const uint32_t hi_even = LMAX > 2 ? ((w[LMAX > 2 ? 2 : 0] & kM) | (LMAX > 3 ? (w[LMAX > 3 ? 3 : 0] & kM) << 2 : 0u)) : 0u;
      const uint32_t hi_odd = LMAX > 2 ? (((w[LMAX > 2 ? 2 : 0] >> 2) & kM) | (LMAX > 3 ? (w[LMAX > 3 ? 3 : 0] & ~kM) : 0u)) : 0u;
#pragma unroll
      for (int d = 0; d < 32; d += 2) {
        const uint32_t lo = (d & 2) ? lo_odd : lo_even, hi = (d & 2) ? hi_odd : hi_even;
        uint32_t idx = __builtin_amdgcn_ubfe(lo, d & ~3, 4);
        if (LMAX > 2) idx |= __builtin_amdgcn_ubfe(hi, d & ~3, 4) << 4;
        const uint4 t = pair_fn[idx];
        Elo += __builtin_amdgcn_perm(t.w, t.z, Flo);                  // (a lane: at most 32 docs x 7: a byte holds -- the engine sends larger machines to the table walk)
        Ehi += __builtin_amdgcn_perm(t.w, t.z, Fhi);
        const uint32_t nlo = __builtin_amdgcn_perm(t.y, t.x, Flo), nhi = __builtin_amdgcn_perm(t.y, t.x, Fhi);
        Flo = nlo; Fhi = nhi;
      }
    } else {
      uint32_t st[8], ent[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { st[c] = (uint32_t)c; ent[c] = 0u; }
      for (int d = 0; d < docs; ++d) {
        uint32_t in = 0u;
#pragma unroll
        for (int i = 0; i < LMAX; ++i) in |= __builtin_amdgcn_ubfe(w[i], d, 1) << i;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const uint32_t t = delta[(st[c] << LMAX) | in];
          ent[c] += t >> 4;
          st[c] = t & 7u;
        }
      }
      Flo = st[0] | (st[1] << 8) | (st[2] << 16) | (st[3] << 24);
      Fhi = st[4] | (st[5] << 8) | (st[6] << 16) | (st[7] << 24);
      Elo = ent[0] | (ent[1] << 8) | (ent[2] << 16) | (ent[3] << 24);
      Ehi = ent[4] | (ent[5] << 8) | (ent[6] << 16) | (ent[7] << 24);
    }
    // entries as 16-bit fields: E[r] = chains 2r, 2r + 1
    uint32_t E[4] = {__builtin_amdgcn_perm(0u, Elo, 0x0c010c00u), __builtin_amdgcn_perm(0u, Elo, 0x0c030c02u),
                     __builtin_amdgcn_perm(0u, Ehi, 0x0c010c00u), __builtin_amdgcn_perm(0u, Ehi, 0x0c030c02u)};
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int from = ((lane + (1 << j)) & 63) << 2;
      const uint32_t Glo = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)Flo), Ghi = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)Fhi);
      uint32_t H[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) H[r] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)E[r]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t src = r < 2 ? Flo : Fhi;
        const uint32_t dup = __builtin_amdgcn_perm(src, src, (r & 1) ? 0x03030202u : 0x01010000u);      // [f0 f0 f1 f1], f in 0..7
        const uint32_t sel = ((dup & 0x03030303u) << 1) + 0x01000100u;
        const uint32_t lo = __builtin_amdgcn_perm(H[1], H[0], sel), hi = __builtin_amdgcn_perm(H[3], H[2], sel);
        const uint32_t m = (dup >> 2) & 0x01010101u;
        const uint32_t mask = (m << 8) - m;                            // 0xFF in the bytes whose selector names a state >= 4
        E[r] += (hi & mask) | (lo & ~mask);
      }
      const uint32_t nlo = __builtin_amdgcn_perm(Ghi, Glo, Flo), nhi = __builtin_amdgcn_perm(Ghi, Glo, Fhi);
      Flo = nlo; Fhi = nhi;
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t e = (E[c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
        if (c < S) p.tables[tile * S + c] = (((c < 4 ? Flo : Fhi) >> (8 * (c & 3))) & 7u) | (e << 4);
      }
    }
  }
}

// `count` tables of S entries each -> ceil(count / 1024) tables.  Every wavefront stages its 64 tables in LDS with coalesced loads (walking
// them straight from memory was 64 DEPENDENT loads per lane: ~35 us of a 60 us tail), thread t < S walks them from entry state t, then the
// first wavefront walks the (up to 16) wavefront tables.
static __global__ __launch_bounds__(1024) void fsm_chain_kernel(const uint32_t* __restrict__ in, long long count, int S, uint32_t* __restrict__ out) {
  __shared__ uint32_t staged[16][64 * kFsmStates];
  __shared__ uint32_t wave_tables[16 * kFsmStates];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * kFsmChunk + wave * 64;
  const long long left = count - base;
  const int here = left >= 64 ? 64 : (left <= 0 ? 0 : (int)left);
  for (int i = lane; i < here * S; i += 64) staged[wave][i] = in[base * S + i];
  __builtin_amdgcn_wave_barrier();
  if (lane < S) {
    uint32_t c = (uint32_t)lane, e = 0u;
    for (int i = 0; i < here; ++i) {
      const uint32_t t = staged[wave][i * S + (int)c];
      e += t >> 4;
      c = t & 15u;
    }
    wave_tables[wave * kFsmStates + lane] = c | (e << 4);
  }
  __syncthreads();
  if (wave == 0 && lane < S) {
    uint32_t c = (uint32_t)lane, e = 0u;
    for (int v = 0; v < 16; ++v) {
      const uint32_t t = wave_tables[v * kFsmStates + (int)c];
      e += t >> 4;
      c = t & 15u;
    }
    out[(long long)blockIdx.x * S + lane] = c | (e << 4);
  }
}

// The last level: at most 1024 tables (a segment has < 2^20 tiles: one chain level leaves at most 1024), staged in LDS; every wavefront
// walks its 64 from every entry state, one thread then walks the 16 wavefront tables from state 0 -- the entries added in 64 bits
// (a table entry carries at most 2^21 docs x 15).
static __global__ __launch_bounds__(1024) void fsm_finish_kernel(const uint32_t* __restrict__ in, int count, int S, unsigned long long* __restrict__ out_entries) {
  extern __shared__ uint32_t staged[];
  __shared__ unsigned long long wave_entries[16 * kFsmStates];
  __shared__ uint32_t wave_state[16 * kFsmStates];
  for (int i = threadIdx.x; i < count * S; i += blockDim.x) staged[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < S) {
    uint32_t c = (uint32_t)lane;
    unsigned long long e = 0ull;
    const int first = wave * 64, last = first + 64 < count ? first + 64 : count;
    for (int i = first; i < last; ++i) {
      const uint32_t t = staged[i * S + (int)c];
      e += t >> 4;
      c = t & 15u;
    }
    wave_state[wave * kFsmStates + lane] = c;
    wave_entries[wave * kFsmStates + lane] = e;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  uint32_t c = 0u;
  unsigned long long e = 0ull;
  for (int v = 0; v < 16; ++v) {
    e += wave_entries[v * kFsmStates + (int)c];
    c = wave_state[v * kFsmStates + (int)c];
  }
  *out_entries = e;
}

// ---- NOT children: the episodes (pg_filter_fsm.h "NOT children"; the host twin: fsm_episode_entries_tiled) -------------------------------
// A NOT child over a scan leaf pulls its leaf with next(): whole 256-doc batches from wherever the last advance() left it
// (NotDocIdIterator.java:45-76, SVScanDocIdIterator.java:76-112).  The machine's delta counts what is charged doc by doc; its `marks` say
// where an EPISODE of batches opens (kMarkOpen at doc m: the batches start at m + 1) and where it closes (kMarkClose at doc x: the batch
// that holds x is the last one).  Opens and closes alternate, doc 0 is entered with the episode of origin 0 open.  Pairing them needs
// every tile's entry state (the tables of the count, walked downwards) and a second walk of the docs:
//   fsm_chunk_states_kernel     the <= 1024 chunk tables walked from state 0: the state every chunk is entered in
//   fsm_tile_states_kernel      per chunk, its 1024 tile tables walked from that state: the state every tile is entered in
//   fsm_episode_tiles_kernel    one wavefront per tile: lane functions -> lane entry states -> every lane's open / close words; closes paired
//                               with the last open in front of them inside the tile, the tile's one unpaired close and its last open kept
//   fsm_episode_finish_kernel   those paired across tiles; the end of the docs closes what is open.
constexpr uint32_t kFsmMarkOpen = 1u, kFsmMarkClose = 2u;
constexpr int kFsmScanBatch = 256;                                  // BlockDocIdIterator.OPTIMAL_ITERATOR_BATCH_SIZE

struct FsmEpisodeParams {
  const uint32_t* leaf[kFsmInputs];     // as FsmParams
  const uint8_t* delta;                 // [S << L] next state | entries << 4
  const uint8_t* marks;                 // [S << L]
  const uint8_t* tile_state;            // [num_tiles] fsm_tile_states_kernel's output
  int32_t* tile_first_close;            // [num_tiles] the close of the tile that has no open in front of it inside the tile; -1
  int32_t* tile_last_open;              // [num_tiles] -1: none
  unsigned long long* episode_entries;  // += the episodes paired inside tiles
  int32_t* final_pending;               // = 1 when the state behind the last doc has an episode open
  uint32_t pending_states;
  int32_t num_inputs, num_states, num_docs, num_tiles;
};

__device__ __forceinline__ unsigned long long fsm_episode_cost(long long origin, long long close, long long num_docs) {
  if (origin >= num_docs) return 0ull;
  if (close >= num_docs) return (unsigned long long)(num_docs - origin);
  const long long batches = ((close - origin) / kFsmScanBatch + 1) * kFsmScanBatch;
  return (unsigned long long)(batches < num_docs - origin ? batches : num_docs - origin);
}

static __global__ __launch_bounds__(1024) void fsm_chunk_states_kernel(const uint32_t* __restrict__ in, int count, int S, uint8_t* __restrict__ chunk_state) {
  extern __shared__ uint32_t staged[];
  __shared__ uint32_t wave_exit[16 * kFsmStates];
  __shared__ uint32_t wave_entry[16];
  for (int i = threadIdx.x; i < count * S; i += blockDim.x) staged[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int first = wave * 64, last = first + 64 < count ? first + 64 : count;
  if (lane < S) {
    uint32_t c = (uint32_t)lane;
    for (int i = first; i < last; ++i) c = staged[i * S + (int)c] & 15u;
    wave_exit[wave * kFsmStates + lane] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t c = 0u;
    for (int v = 0; v < 16; ++v) { wave_entry[v] = c; c = wave_exit[v * kFsmStates + (int)c]; }
  }
  __syncthreads();
  if (lane == 0) {
    uint32_t c = wave_entry[wave];
    for (int i = first; i < last; ++i) { chunk_state[i] = (uint8_t)c; c = staged[i * S + (int)c] & 15u; }
  }
}

static __global__ __launch_bounds__(1024) void fsm_tile_states_kernel(const uint32_t* __restrict__ in, long long count, int S, const uint8_t* __restrict__ chunk_state,
                                                                       uint8_t* __restrict__ tile_state) {
  __shared__ uint32_t staged[16][64 * kFsmStates];
  __shared__ uint32_t wave_exit[16 * kFsmStates];
  __shared__ uint32_t wave_entry[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * kFsmChunk + wave * 64;
  const long long left = count - base;
  const int here = left >= 64 ? 64 : (left <= 0 ? 0 : (int)left);
  for (int i = lane; i < here * S; i += 64) staged[wave][i] = in[base * S + i];
  __builtin_amdgcn_wave_barrier();
  if (lane < S) {
    uint32_t c = (uint32_t)lane;
    for (int i = 0; i < here; ++i) c = staged[wave][i * S + (int)c] & 15u;
    wave_exit[wave * kFsmStates + lane] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t c = chunk_state[blockIdx.x];
    for (int v = 0; v < 16; ++v) { wave_entry[v] = c; c = wave_exit[v * kFsmStates + (int)c]; }
  }
  __syncthreads();
  if (lane == 0) {
    uint32_t c = wave_entry[wave];
    for (int i = 0; i < here; ++i) { tile_state[base + i] = (uint8_t)c; c = staged[wave][i * S + (int)c] & 15u; }
  }
}

static __global__ __launch_bounds__(256) void fsm_episode_tiles_kernel(const FsmEpisodeParams p) {
  __shared__ uint8_t dm[kFsmStates << kFsmInputs];                   // next state | mark << 4
  __shared__ uint8_t lane_next[4][64 * kFsmStates];
  __shared__ uint8_t lane_entry[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.num_inputs, S = p.num_states;
  for (int i = threadIdx.x; i < (S << L); i += blockDim.x) dm[i] = (uint8_t)((p.delta[i] & 15u) | ((uint32_t)p.marks[i] << 4));
  __syncthreads();
  unsigned long long sum = 0ull;                                       // this lane's episodes over all of the wavefront's tiles: ONE atomic per wavefront at the end
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < p.num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048 + lane * 32;
    const long long rem = (long long)p.num_docs - first;
    const int docs = rem >= 32 ? 32 : (rem <= 0 ? 0 : (int)rem);
    uint32_t w[kFsmInputs];
#pragma unroll
    for (int i = 0; i < kFsmInputs; ++i) w[i] = i < L ? p.leaf[i][tile * 64 + lane] : 0u;
    // the lane's function: its docs from every entry state
    uint32_t st[kFsmStates];
#pragma unroll
    for (int s = 0; s < kFsmStates; ++s) st[s] = (uint32_t)s;
    for (int d = 0; d < docs; ++d) {
      uint32_t in = 0u;
#pragma unroll
      for (int i = 0; i < kFsmInputs; ++i) if (i < L) in |= ((w[i] >> d) & 1u) << i;      // (L is uniform: a machine over two leaves assembles two bits, not eight)
#pragma unroll
      for (int s = 0; s < kFsmStates; ++s) if (s < S) st[s] = dm[(st[s] << L) | in] & 15u;
    }
#pragma unroll
    for (int s = 0; s < kFsmStates; ++s) lane_next[wave][lane * kFsmStates + s] = (uint8_t)st[s];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      uint32_t c = p.tile_state[tile];
      for (int l = 0; l < 64; ++l) { lane_entry[wave][l] = (uint8_t)c; c = lane_next[wave][l * kFsmStates + (int)c]; }
    }
    __builtin_amdgcn_wave_barrier();
    // the lane's docs again, from the state it is entered in: where episodes open and close
    uint32_t cur = lane_entry[wave][lane], open_word = 0u, close_word = 0u;
    for (int d = 0; d < docs; ++d) {
      uint32_t in = 0u;
#pragma unroll
      for (int i = 0; i < kFsmInputs; ++i) if (i < L) in |= ((w[i] >> d) & 1u) << i;      // (L is uniform: a machine over two leaves assembles two bits, not eight)
      const uint32_t t = dm[(cur << L) | in];
      open_word |= ((t >> 4) == kFsmMarkOpen ? 1u : 0u) << d;
      close_word |= ((t >> 4) == kFsmMarkClose ? 1u : 0u) << d;
      cur = t & 15u;
    }
    if (docs > 0 && first + docs == (long long)p.num_docs) *p.final_pending = (int32_t)((p.pending_states >> cur) & 1u);      // the lane that holds the last doc
    // the last open in front of every lane: an inclusive prefix maximum over the wavefront, shifted by one lane
    const int32_t mine = open_word ? (int32_t)(first + 31 - __builtin_clz(open_word)) : -1;
    int32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t other = __shfl_up(incl, off);
      if (lane >= off) incl = other > incl ? other : incl;
    }
    int32_t prev = __shfl_up(incl, 1);
    if (lane == 0) prev = -1;
    int32_t unpaired = -1;
    for (uint32_t c = close_word; c != 0u; c &= c - 1u) {
      const int d = __builtin_ctz(c);
      const uint32_t below = open_word & ((1u << d) - 1u);
      const int32_t open_at = below ? (int32_t)(first + 31 - __builtin_clz(below)) : prev;
      const long long x = first + d;
      if (open_at >= 0) sum += fsm_episode_cost((long long)open_at + 1, x, p.num_docs);
      else unpaired = (int32_t)x;                                      // (its open lies in an earlier tile, or it is the episode of doc 0)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int32_t other = __shfl_xor(unpaired, off);
      unpaired = other > unpaired ? other : unpaired;
    }
    const int32_t last_open = __shfl(incl, 63);
    if (lane == 0) {
      p.tile_first_close[tile] = unpaired;
      p.tile_last_open[tile] = last_open;
    }
  }
  // (one atomic per TILE on the one counter -- 488 282 of them at 1 B docs -- was the first coding)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += (unsigned long long)__shfl_xor((long long)sum, off);
  if (lane == 0 && sum != 0ull) atomicAdd(p.episode_entries, sum);
}

// ---- Round 6: the episodes of machines of at most EIGHT states over at most four inputs, without a table walk per entry state ----------------
// fsm_episode_tiles_kernel above is the general form (sixteen chains of byte reads per doc, lane 0 walking the 64 lane functions one after
// the other, one record per TILE for a one-workgroup finish): 3.23 ms per 1 B docs behind a 1.07 ms scan (profiles/r6).  Here
//   * a lane's function {entry state} -> {exit state} is the BYTES of one register (two for five to eight states) and a step over
//     kDps = 8 / 4 / 2 docs (one / two / three-four inputs) is one LDS read of the step's function and one v_perm_b32 (two): 8 - 16 steps
//     instead of 32 x S dependent byte reads;
//   * the state every lane is entered in comes out of an inclusive SCAN of the lane functions over the wavefront (six rounds of one
//     ds_bpermute + one v_perm_b32), not out of a 64-step walk by lane 0;
//   * the second walk is ONE chain from that state: a 32-bit LDS read per step gives {next state, the step's opens, its closes};
//   * a wavefront owns a CONTIGUOUS range of tiles and carries the last open across them, so what is left to pair across wavefronts is
//     one {first unpaired close, last open} per wavefront (<= 2^14 records for fsm_episode_finish_kernel, which took 0.39 ms over one
//     record per tile on its one compute unit) and no per-tile record is written at all.
// The tiles' entry states still come from the count's tables walked downwards (fsm_chunk_states_kernel, fsm_tile_states_kernel).
// Round 6c: the first two points -- the lane functions and their scan -- moved into the tile pass of these machines (fsm_tile_fns_kernel,
// below: functions only, sixteen states too), which leaves every lane's front in memory; what remains here is the one chain from the real
// entry state, which now also counts the per-doc entries (count_entries).
struct FsmEpisodeRangeParams {
  const uint32_t* leaf[4];              // as FsmParams (at most four inputs)
  const uint8_t* delta;                 // [S << L] next state | entries << 4
  const uint8_t* marks;                 // [S << L]
  const uint8_t* tile_state;            // [num_tiles] fsm_tile_states_kernel's output
  int32_t* range_first_close;           // [num_ranges] the close of the range that has no open in front of it inside the range; -1
  int32_t* range_last_open;             // [num_ranges] -1: none
  unsigned long long* episode_entries;  // += the episodes paired inside ranges
  int32_t* final_pending;               // = 1 when the state behind the last doc has an episode open
  uint32_t pending_states;
  int32_t num_inputs, num_states, num_docs, num_tiles, num_ranges;
  const uint32_t* lane_front;           // fsm_tile_fns_kernel's lane fronts: [num_tiles * 64 * SMAX / 4] (the lane functions are not built a second time here)
  int32_t count_entries;                // != 0: the per-doc entries of the walk are added too (the tile pass -- fsm_tile_fns_kernel -- built functions only)
};

template <int SMAX>
struct FsmByteFn {                                                     // {entry state} -> {exit state}, a byte per entry state
  uint32_t lo, hi;                                                     // (hi: states 4 .. 7, SMAX >= 8 only)
  uint32_t w2, w3;                                                     // (states 8 .. 11, 12 .. 15: SMAX == 16 only -- round 6c)
};
template <int SMAX>
__device__ __forceinline__ FsmByteFn<SMAX> fsm_fn_identity() { return FsmByteFn<SMAX>{0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu}; }
// Four bytes of a sixteen-entry byte table {t0 .. t3} picked by the selector bytes of `sel` (values 0 .. 15): v_perm_b32 reaches eight bytes,
// so both halves are looked up with the selectors' low three bits and bit 3 of every selector byte picks between them (v_bfi_b32).
__device__ __forceinline__ uint32_t fsm_lookup16(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t sel) {
  const uint32_t low = sel & 0x07070707u;
  const uint32_t a = __builtin_amdgcn_perm(t1, t0, low), b = __builtin_amdgcn_perm(t3, t2, low);
  const uint32_t upper = ((sel >> 3) & 0x01010101u) * 0xFFu;          // 0xFF in the bytes whose selector is 8 .. 15
  return (b & upper) | (a & ~upper);
}
// `first`, then `then`:  out[s] = then[first[s]]
template <int SMAX>
__device__ __forceinline__ FsmByteFn<SMAX> fsm_fn_then(const FsmByteFn<SMAX>& first, const FsmByteFn<SMAX>& then) {
  FsmByteFn<SMAX> out;
  out.w2 = 0u; out.w3 = 0u;
  if constexpr (SMAX <= 4) { out.lo = __builtin_amdgcn_perm(then.lo, then.lo, first.lo); out.hi = 0u; }
  else if constexpr (SMAX <= 8) { out.lo = __builtin_amdgcn_perm(then.hi, then.lo, first.lo); out.hi = __builtin_amdgcn_perm(then.hi, then.lo, first.hi); }
  else {
    out.lo = fsm_lookup16(then.lo, then.hi, then.w2, then.w3, first.lo); out.hi = fsm_lookup16(then.lo, then.hi, then.w2, then.w3, first.hi);
    out.w2 = fsm_lookup16(then.lo, then.hi, then.w2, then.w3, first.w2); out.w3 = fsm_lookup16(then.lo, then.hi, then.w2, then.w3, first.w3);
  }
  return out;
}
template <int SMAX>
__device__ __forceinline__ uint32_t fsm_fn_at(const FsmByteFn<SMAX>& f, uint32_t state) {
  if constexpr (SMAX <= 4) return (f.lo >> (8u * state)) & 0xFFu;
  else if constexpr (SMAX <= 8) return ((state < 4u ? f.lo : f.hi) >> (8u * (state & 3u))) & 0xFFu;
  else return ((state < 8u ? (state < 4u ? f.lo : f.hi) : (state < 12u ? f.w2 : f.w3)) >> (8u * (state & 3u))) & 0xFFu;
}
// A function handed to the lane `delta` lanes up the wavefront (every word of it).
template <int SMAX>
__device__ __forceinline__ FsmByteFn<SMAX> fsm_fn_shfl_up(const FsmByteFn<SMAX>& f, unsigned delta) {
  FsmByteFn<SMAX> out;
  out.lo = (uint32_t)__shfl_up((int)f.lo, delta);
  out.hi = SMAX > 4 ? (uint32_t)__shfl_up((int)f.hi, delta) : 0u;
  out.w2 = SMAX > 8 ? (uint32_t)__shfl_up((int)f.w2, delta) : 0u;
  out.w3 = SMAX > 8 ? (uint32_t)__shfl_up((int)f.w3, delta) : 0u;
  return out;
}
// A function in memory: SMAX / 4 words a lane, one vector access.
template <int SMAX>
__device__ __forceinline__ void fsm_fn_store(uint32_t* base, long long index, const FsmByteFn<SMAX>& f) {
  if constexpr (SMAX <= 4) base[index] = f.lo;
  else if constexpr (SMAX <= 8) reinterpret_cast<uint2*>(base)[index] = make_uint2(f.lo, f.hi);
  else reinterpret_cast<uint4*>(base)[index] = make_uint4(f.lo, f.hi, f.w2, f.w3);
}
template <int SMAX>
__device__ __forceinline__ FsmByteFn<SMAX> fsm_fn_load(const uint32_t* base, long long index) {
  FsmByteFn<SMAX> f;
  f.hi = 0u; f.w2 = 0u; f.w3 = 0u;
  if constexpr (SMAX <= 4) f.lo = base[index];
  else if constexpr (SMAX <= 8) { const uint2 v = reinterpret_cast<const uint2*>(base)[index]; f.lo = v.x; f.hi = v.y; }
  else { const uint4 v = reinterpret_cast<const uint4*>(base)[index]; f.lo = v.x; f.hi = v.y; f.w2 = v.z; f.w3 = v.w; }
  return f;
}
// The function of a lane's 32 docs: from the step table when the lane has all of them (whole), else doc by doc through `dm` (next state in
// the low four bits).  step_fn: [idx * kWords + word]; idx: input i's bits of the step's kDps docs at [i * kDps, (i + 1) * kDps).
template <int SMAX, int LMAX, int kDps, int kWords>
__device__ __forceinline__ FsmByteFn<SMAX> fsm_lane_fn(const uint32_t (&w)[LMAX], int docs, bool whole, const uint32_t* step_fn, const uint8_t* dm) {
  FsmByteFn<SMAX> f = fsm_fn_identity<SMAX>();
  if (whole) {
#pragma unroll
    for (int d = 0; d < 32; d += kDps) {
      uint32_t idx = 0u;
#pragma unroll
      for (int i = 0; i < LMAX; ++i) idx |= __builtin_amdgcn_ubfe(w[i], d, kDps) << (i * kDps);
      FsmByteFn<SMAX> t;
      t.lo = step_fn[idx * kWords];
      t.hi = kWords >= 2 ? step_fn[idx * kWords + (kWords >= 2 ? 1 : 0)] : 0u;
      t.w2 = kWords >= 4 ? step_fn[idx * kWords + (kWords >= 4 ? 2 : 0)] : 0u;
      t.w3 = kWords >= 4 ? step_fn[idx * kWords + (kWords >= 4 ? 3 : 0)] : 0u;
      f = fsm_fn_then<SMAX>(f, t);
    }
  } else {
    uint32_t st[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) st[s] = (uint32_t)s;
    for (int d = 0; d < docs; ++d) {
      uint32_t in = 0u;
#pragma unroll
      for (int i = 0; i < LMAX; ++i) in |= __builtin_amdgcn_ubfe(w[i], d, 1) << i;
#pragma unroll
      for (int s = 0; s < SMAX; ++s) st[s] = dm[(st[s] << LMAX) | in] & 15u;
    }
    f.lo = st[0] | (st[1] << 8) | (st[2] << 16) | (st[3] << 24);
    if constexpr (SMAX > 4) f.hi = st[4] | (st[5] << 8) | (st[6] << 16) | (st[7] << 24);
    if constexpr (SMAX > 8) { f.w2 = st[8] | (st[9] << 8) | (st[10] << 16) | (st[11] << 24); f.w3 = st[12] | (st[13] << 8) | (st[14] << 16) | (st[15] << 24); }
  }
  return f;
}

template <int SMAX, int LMAX>
__global__ __launch_bounds__(256) void fsm_episode_ranges_kernel(const FsmEpisodeRangeParams p) {
  static_assert((SMAX == 4 || SMAX == 8 || SMAX == 16) && LMAX >= 1 && LMAX <= 4, "byte functions of four, eight or sixteen states over at most four inputs");
  constexpr int kDps = LMAX == 1 ? 8 : (LMAX == 2 ? 4 : 2);           // docs per step
  constexpr int kIndexBits = kDps * LMAX;                              // input i's bits of the step's docs at [i * kDps, (i + 1) * kDps)
  __shared__ uint8_t dm[SMAX << LMAX];                                 // one doc: next state | mark << 4 (the last tile's partial lanes)
  __shared__ uint8_t de[SMAX << LMAX];                                 // one doc: its entries (count_entries)
  __shared__ uint32_t step_mark[SMAX << kIndexBits];                   // [(state << kIndexBits) | idx]: next state | opens << 4 | closes << 12 (bit j: the step's doc j) | the step's entries << 20
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.num_inputs, S = p.num_states;
  for (int i = threadIdx.x; i < (SMAX << LMAX); i += blockDim.x) {
    const int st = i >> LMAX, in = i & ((1 << LMAX) - 1);
    const bool real = st < S && in < (1 << L);
    dm[i] = real ? (uint8_t)((p.delta[(st << L) | in] & 15u) | ((uint32_t)p.marks[(st << L) | in] << 4)) : (uint8_t)0;
    de[i] = real ? (uint8_t)(p.delta[(st << L) | in] >> 4) : (uint8_t)0;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < (1 << kIndexBits); idx += blockDim.x) {
    for (int st = 0; st < SMAX; ++st) {
      uint32_t cur = (uint32_t)st, opens = 0u, closes = 0u, ents = 0u;
      for (int j = 0; j < kDps; ++j) {
        uint32_t in = 0u;
        for (int i = 0; i < LMAX; ++i) in |= (((uint32_t)idx >> (i * kDps + j)) & 1u) << i;
        const uint32_t t = dm[(cur << LMAX) | in];
        ents += de[(cur << LMAX) | in];
        if ((t >> 4) == kFsmMarkOpen) opens |= 1u << j;
        if ((t >> 4) == kFsmMarkClose) closes |= 1u << j;
        cur = t & 15u;
      }
      step_mark[(st << kIndexBits) | idx] = cur | (opens << 4) | (closes << 12) | (ents << 20);      // (eight docs x fifteen entries: seven bits)
    }
  }
  __syncthreads();

  const long long range = (long long)blockIdx.x * 4 + wave;
  const long long per_range = ((long long)p.num_tiles + p.num_ranges - 1) / p.num_ranges;
  const long long tile_begin = range * per_range < (long long)p.num_tiles ? range * per_range : (long long)p.num_tiles;
  const long long tile_end = tile_begin + per_range < (long long)p.num_tiles ? tile_begin + per_range : (long long)p.num_tiles;
  unsigned long long sum = 0ull;                                       // this lane's episodes over the range: ONE atomic per wavefront at the end
  int32_t carry_open = -1;                                             // the last open of the range's tiles so far
  int32_t first_close = -1;                                            // the range's close that has no open in front of it inside the range (at most one: opens and closes alternate)
  for (long long tile = tile_begin; tile < tile_end; ++tile) {
    const long long first = tile * 2048 + lane * 32;
    const long long rem = (long long)p.num_docs - first;
    const int docs = rem >= 32 ? 32 : (rem <= 0 ? 0 : (int)rem);
    uint32_t w[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; ++i) w[i] = i < L ? p.leaf[i][tile * 64 + lane] : 0u;
    const bool whole = __builtin_amdgcn_ballot_w64(docs != 32) == 0ull;      // every lane has its 32 docs (all tiles but the segment's last)
    // ---- the state this lane is entered in: the tile pass (fsm_tile_fns_kernel) has built every lane's function and scanned them over the
    //      wavefront; it left the function of the lanes IN FRONT of every lane (SMAX bytes a lane), applied here to the tile's entry state.
    //      (Rounds 6a-6b built and scanned the lane functions a second time in this kernel: 0.6 of its 0.99 ms at sixteen states.) ----
    const FsmByteFn<SMAX> front = fsm_fn_load<SMAX>(p.lane_front, tile * 64 + lane);
    uint32_t cur = fsm_fn_at<SMAX>(front, (uint32_t)p.tile_state[tile]);
    // ---- the lane's docs again, one chain from that state: where episodes open and close (and what the docs cost: count_entries) ----
    uint32_t open_word = 0u, close_word = 0u, ents = 0u;
    if (whole) {
#pragma unroll
      for (int d = 0; d < 32; d += kDps) {
        uint32_t idx = 0u;
#pragma unroll
        for (int i = 0; i < LMAX; ++i) idx |= __builtin_amdgcn_ubfe(w[i], d, kDps) << (i * kDps);
        const uint32_t t = step_mark[(cur << kIndexBits) | idx];
        open_word |= __builtin_amdgcn_ubfe(t, 4, kDps) << d;
        close_word |= __builtin_amdgcn_ubfe(t, 12, kDps) << d;
        ents += t >> 20;
        cur = t & 15u;
      }
    } else {
      for (int d = 0; d < docs; ++d) {
        uint32_t in = 0u;
#pragma unroll
        for (int i = 0; i < LMAX; ++i) in |= __builtin_amdgcn_ubfe(w[i], d, 1) << i;
        const uint32_t t = dm[(cur << LMAX) | in];
        ents += de[(cur << LMAX) | in];
        open_word |= ((t >> 4) == kFsmMarkOpen ? 1u : 0u) << d;
        close_word |= ((t >> 4) == kFsmMarkClose ? 1u : 0u) << d;
        cur = t & 15u;
      }
    }
    if (p.count_entries != 0) sum += ents;
    if (docs > 0 && first + docs == (long long)p.num_docs) *p.final_pending = (int32_t)((p.pending_states >> cur) & 1u);      // the lane that holds the last doc
    // ---- the last open in front of every lane: an inclusive prefix maximum over the wavefront, shifted by one lane, and the range's carry ----
    const int32_t mine = open_word ? (int32_t)(first + 31 - __builtin_clz(open_word)) : -1;
    int32_t last = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t other = __shfl_up(last, (unsigned)off);
      if (lane >= off) last = other > last ? other : last;
    }
    int32_t prev = __shfl_up(last, 1u);
    if (lane == 0) prev = -1;
    prev = prev > carry_open ? prev : carry_open;
    int32_t unpaired = -1;
    // A close whose open lies among the lane's own 32 docs ends an episode of ONE batch: (close - origin) / 256 = 0, and away from the end of
    // the docs a whole batch is scanned -- kFsmScanBatch entries each, a population count.  (The loop below took every close: a dense leaf
    // under the NOT closes an episode every few docs, eight or more iterations a lane with a 64-bit cost each -- more than the walk itself.)
    // What is left for the loop: the closes in front of the lane's first open (opens and closes alternate: at most one), and the lanes
    // within a batch of the end of the docs.
    uint32_t loop_closes = close_word;
    if (first + 32 + kFsmScanBatch <= (long long)p.num_docs) {
      const uint32_t in_front = open_word ? ((open_word & (0u - open_word)) - 1u) : 0xFFFFFFFFu;      // the bits below the lane's first open
      sum += (unsigned long long)kFsmScanBatch * (unsigned)__builtin_popcount(close_word & ~in_front);
      loop_closes = close_word & in_front;
    }
    for (uint32_t c = loop_closes; c != 0u; c &= c - 1u) {
      const int d = __builtin_ctz(c);
      const uint32_t below = open_word & ((1u << d) - 1u);
      const int32_t open_at = below ? (int32_t)(first + 31 - __builtin_clz(below)) : prev;
      const long long x = first + d;
      if (open_at >= 0) sum += fsm_episode_cost((long long)open_at + 1, x, p.num_docs);
      else unpaired = (int32_t)x;                                      // (its open lies in front of the range, or it is the episode of doc 0)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int32_t other = __shfl_xor(unpaired, off);
      unpaired = other > unpaired ? other : unpaired;
    }
    if (first_close < 0) first_close = unpaired;
    const int32_t tile_last = __shfl(last, 63);
    carry_open = tile_last > carry_open ? tile_last : carry_open;
  }
  if (lane == 0 && range < (long long)p.num_ranges) {
    p.range_first_close[range] = first_close;
    p.range_last_open[range] = carry_open;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += (unsigned long long)__shfl_xor((long long)sum, off);
  if (lane == 0 && sum != 0ull) atomicAdd(p.episode_entries, sum);
}

// Machines of nine to sixteen states with episodes (two NOT children beside a third child, NOT over an OR of two scan leaves: round 6c).
// Their tables came from fsm_tiles_kernel<16, L> -- sixteen chains of dependent LDS byte reads per doc, 3.16 ms per 1 B docs -- and their
// episodes from fsm_episode_tiles_kernel, 4.06 ms per stream.  The range kernel above takes sixteen states (a function in four registers,
// fsm_lookup16) and, since it walks every doc from the state the doc is really entered in, counts the per-doc entries on the way
// (count_entries, the first stream's pass only).  What is left for the tile pass is the tiles' FUNCTIONS {entry state} -> {exit state}
// alone: the lane functions as in the range kernel, the same inclusive scan, lane 63 holds the tile's.  Written in the tables' format
// (next state | entries << 4, entries = 0) for fsm_chain_kernel / fsm_chunk_states_kernel / fsm_tile_states_kernel.
// Round 6c, last: the same split for every machine with episodes (SMAX 4 / 8 too): the byte-function tile kernels with entries
// (fsm_tiles_perm_kernel, fsm_tiles_perm8_kernel) keep the machines WITHOUT episodes, whose count is all there is.
template <int SMAX, int LMAX>
__global__ __launch_bounds__(256) void fsm_tile_fns_kernel(const FsmParams p) {
  static_assert((SMAX == 4 || SMAX == 8 || SMAX == 16) && LMAX >= 2 && LMAX <= 4, "as fsm_episode_ranges_kernel");
  constexpr int kDps = LMAX == 2 ? 4 : 2, kIndexBits = kDps * LMAX, kWords = SMAX / 4;
  __shared__ uint8_t dm[SMAX << LMAX];
  __shared__ uint32_t step_fn[kWords << kIndexBits];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.num_inputs, S = p.num_states;
  for (int i = threadIdx.x; i < (SMAX << LMAX); i += blockDim.x) {
    const int st = i >> LMAX, in = i & ((1 << LMAX) - 1);
    dm[i] = (st < S && in < (1 << L)) ? (uint8_t)(p.delta[(st << L) | in] & 15u) : (uint8_t)0;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < (1 << kIndexBits); idx += blockDim.x) {
    uint32_t fn[4] = {0u, 0u, 0u, 0u};
    for (int st = 0; st < SMAX; ++st) {
      uint32_t cur = (uint32_t)st;
      for (int j = 0; j < kDps; ++j) {
        uint32_t in = 0u;
        for (int i = 0; i < LMAX; ++i) in |= (((uint32_t)idx >> (i * kDps + j)) & 1u) << i;
        cur = dm[(cur << LMAX) | in] & 15u;
      }
      fn[st >> 2] |= cur << (8 * (st & 3));
    }
#pragma unroll
    for (int k = 0; k < kWords; ++k) step_fn[idx * kWords + k] = fn[k];
  }
  __syncthreads();
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < p.num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048 + lane * 32;
    const long long rem = (long long)p.num_docs - first;
    const int docs = rem >= 32 ? 32 : (rem <= 0 ? 0 : (int)rem);
    uint32_t w[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; ++i) w[i] = i < L ? p.leaf[i][tile * 64 + lane] : 0u;
    const bool whole = __builtin_amdgcn_ballot_w64(docs != 32) == 0ull;
    FsmByteFn<SMAX> incl = fsm_lane_fn<SMAX, LMAX, kDps, kWords>(w, docs, whole, step_fn, dm);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const FsmByteFn<SMAX> before = fsm_fn_shfl_up<SMAX>(incl, (unsigned)off);
      if (lane >= off) incl = fsm_fn_then<SMAX>(before, incl);
    }
    if (lane == 63) {
      uint32_t* const out = p.tables + tile * S;
      for (int c = 0; c < S; ++c) out[c] = fsm_fn_at<SMAX>(incl, (uint32_t)c);
    }
    // the function of the lanes in front of this one (lane 0: none), for the range kernel's walk from the real entry state
    FsmByteFn<SMAX> front = fsm_fn_shfl_up<SMAX>(incl, 1u);
    if (lane == 0) front = fsm_fn_identity<SMAX>();
    fsm_fn_store<SMAX>(p.lane_front, tile * 64 + lane, front);
  }
}

// One workgroup of sixteen wavefronts, each a contiguous range of tiles read 64 at a time (lane l: tile base + l -- the first coding gave every
// THREAD a contiguous range: 1024 threads each pulling a 128-byte line for four bytes, through one compute unit).  The last open in front of
// a tile = the maximum of the wavefronts in front, of the 64-tile groups in front (carry) and of the lanes in front (a prefix maximum over
// the wavefront); every tile's unpaired close against it; the end of the docs closes what is open.  *episode_entries += what is found here.
// (pg_filter_fsm.h fsm_episode_entries_tiled ends with the same structure.)
static __global__ __launch_bounds__(1024) void fsm_episode_finish_kernel(const int32_t* __restrict__ tile_first_close, const int32_t* __restrict__ tile_last_open, int num_tiles,
                                                                          int num_docs, const int32_t* __restrict__ final_pending, unsigned long long* __restrict__ episode_entries) {
  __shared__ int32_t wave_last[16];
  __shared__ int32_t wave_carry[16];
  __shared__ int32_t all_last;
  __shared__ unsigned long long total;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per_wave = (num_tiles + 15) / 16;
  const int lo = wave * per_wave < num_tiles ? wave * per_wave : num_tiles, hi = lo + per_wave < num_tiles ? lo + per_wave : num_tiles;
  int32_t m = -1;
  for (int i = lo + lane; i < hi; i += 64) { const int32_t o = tile_last_open[i]; m = o > m ? o : m; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const int32_t other = __shfl_xor(m, off); m = other > m ? other : m; }
  if (lane == 0) wave_last[wave] = m;
  if (threadIdx.x == 0) total = 0ull;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = -1;
    for (int v = 0; v < 16; ++v) { wave_carry[v] = run; run = wave_last[v] > run ? wave_last[v] : run; }
    all_last = run;
  }
  __syncthreads();
  int32_t carry = wave_carry[wave];
  unsigned long long sum = 0ull;
  for (int base = lo; base < hi; base += 64) {                        // (wave-uniform trip count: the shuffles below see all 64 lanes)
    const int i = base + lane;
    int32_t incl = i < hi ? tile_last_open[i] : -1;
    const int32_t x = i < hi ? tile_first_close[i] : -1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t other = __shfl_up(incl, off);
      if (lane >= off) incl = other > incl ? other : incl;
    }
    int32_t excl = __shfl_up(incl, 1);
    if (lane == 0) excl = -1;
    const int32_t prev = carry > excl ? carry : excl;
    if (x >= 0) sum += fsm_episode_cost((long long)prev + 1, (long long)x, (long long)num_docs);
    const int32_t group_last = __shfl(incl, 63);
    carry = group_last > carry ? group_last : carry;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += (unsigned long long)__shfl_xor((long long)sum, off);
  if (lane == 0 && sum != 0ull) atomicAdd(&total, sum);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long all = total;
    if (*final_pending != 0) all += fsm_episode_cost((long long)all_last + 1, (long long)num_docs, (long long)num_docs);
    *episode_entries += all;
  }
}

}  // namespace pg
