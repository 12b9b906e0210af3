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
};

template <int SMAX, int LMAX>
__global__ __launch_bounds__(256) void fsm_tiles_kernel(const FsmParams p) {
  // Up to four input bits: TWO docs per table lookup (a 2 * LMAX-bit index, leaf i's bits for docs d, d + 1 side by side at 2i, 2i + 1 --
  // one v_bfe_u32 + one v_lshl_or_b32 per leaf and pair); the walk is a chain of dependent LDS reads, half as long this way.
  constexpr bool kPair = LMAX <= 4;
  constexpr int kIndexBits = kPair ? 2 * LMAX : LMAX;
  __shared__ uint8_t delta[SMAX << LMAX];
  __shared__ uint16_t delta2[kPair ? (SMAX << kIndexBits) : 1];
  __shared__ uint32_t lane_tables[4][64 * SMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.num_inputs, S = p.num_states;
  // the table is re-laid for LMAX input bits per state (unused states / inputs: entries that are never read)
  for (int i = threadIdx.x; i < (SMAX << LMAX); i += blockDim.x) {
    const int st = i >> LMAX, in = i & ((1 << LMAX) - 1);
    delta[i] = (st < S && in < (1 << L)) ? p.delta[(st << L) | in] : (uint8_t)0;
  }
  __syncthreads();
  if constexpr (kPair) {
    for (int i = threadIdx.x; i < (SMAX << kIndexBits); i += blockDim.x) {
      const int st = i >> kIndexBits, idx = i & ((1 << kIndexBits) - 1);
      int in0 = 0, in1 = 0;
      for (int l = 0; l < LMAX; ++l) { in0 |= ((idx >> (2 * l)) & 1) << l; in1 |= ((idx >> (2 * l + 1)) & 1) << l; }
      const uint32_t t0 = delta[(st << LMAX) | in0], t1 = delta[((t0 & 15u) << LMAX) | in1];
      delta2[i] = (uint16_t)((t1 & 15u) | (((t0 >> 4) + (t1 >> 4)) << 4));
    }
    __syncthreads();
  }
  uint32_t* mine = lane_tables[wave];
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < p.num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048 + lane * 32;
    const long long rem = (long long)p.num_docs - first;
    const int docs = rem >= 32 ? 32 : (rem <= 0 ? 0 : (int)rem);      // (docs past numDocs do not exist; the lanes of the last tile stop at different docs)
    uint32_t w[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; ++i) w[i] = i < L ? p.leaf[i][tile * 64 + lane] : 0u;
    uint32_t cur[SMAX], ent[SMAX];                 // cur: the state, kept shifted into the table index
#pragma unroll
    for (int s = 0; s < SMAX; ++s) { cur[s] = (uint32_t)s << kIndexBits; ent[s] = 0u; }
    int d = 0;
    if constexpr (kPair) {
      for (; d + 2 <= docs; d += 2) {
        uint32_t in = 0u;
#pragma unroll
        for (int i = 0; i < LMAX; ++i) in |= __builtin_amdgcn_ubfe(w[i], d, 2) << (2 * i);
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {          // SMAX independent chains: their LDS reads are in flight together
          const uint32_t t = delta2[cur[s] | in];
          ent[s] += t >> 4;
          cur[s] = (t & 15u) << kIndexBits;
        }
      }
    }
    for (; d < docs; ++d) {
      uint32_t in = 0u;
#pragma unroll
      for (int i = 0; i < LMAX; ++i) in |= __builtin_amdgcn_ubfe(w[i], d, 1) << i;
#pragma unroll
      for (int s = 0; s < SMAX; ++s) {
        const uint32_t t = delta[((cur[s] >> kIndexBits) << LMAX) | in];
        ent[s] += t >> 4;
        cur[s] = (t & 15u) << kIndexBits;
      }
    }
#pragma unroll
    for (int s = 0; s < SMAX; ++s) mine[lane * SMAX + s] = (cur[s] >> kIndexBits) | (ent[s] << 4);
    __builtin_amdgcn_wave_barrier();
    // The 64 lane tables composed in lane order, as a tree: at level j the lanes whose low j + 1 bits are zero append the table 2^j lanes
    // further on (which by then stands for 2^j lanes) to their own.  Six dependent rounds of SMAX LDS reads instead of a 64-step walk.
    uint32_t c[SMAX], e[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) { c[s] = cur[s] >> kIndexBits; e[s] = ent[s]; }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const bool active = (lane & ((2 << j) - 1)) == 0;
      if (active) {
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
          const uint32_t t = mine[(lane + (1 << j)) * SMAX + (int)c[s]];
          e[s] += t >> 4;
          c[s] = t & 15u;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (active) {
#pragma unroll
        for (int s = 0; s < SMAX; ++s) mine[lane * SMAX + s] = c[s] | (e[s] << 4);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
#pragma unroll
      for (int s = 0; s < SMAX; ++s) if (s < S) p.tables[tile * S + s] = c[s] | (e[s] << 4);
    }
  }
}

// `count` tables of S entries each -> ceil(count / 1024) tables: thread t < S of every wavefront walks the wavefront's 64 tables from
// entry state t, then the first wavefront walks the (up to 16) wavefront tables.
static __global__ __launch_bounds__(1024) void fsm_chain_kernel(const uint32_t* __restrict__ in, long long count, int S, uint32_t* __restrict__ out) {
  __shared__ uint32_t wave_tables[16 * kFsmStates];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * kFsmChunk + wave * 64;
  if (lane < S) {
    uint32_t c = (uint32_t)lane, e = 0u;
    for (int i = 0; i < 64; ++i) {
      if (base + i >= count) break;
      const uint32_t t = in[(base + i) * S + (int)c];
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

// The last level: at most 1024 tables (a segment has < 2^20 tiles: one chain level leaves at most 1024), staged in LDS and walked from
// state 0 by one thread, the entries added in 64 bits (a table entry carries at most 2^21 docs x 15).
static __global__ __launch_bounds__(1024) void fsm_finish_kernel(const uint32_t* __restrict__ in, int count, int S, unsigned long long* __restrict__ out_entries) {
  extern __shared__ uint32_t staged[];
  for (int i = threadIdx.x; i < count * S; i += blockDim.x) staged[i] = in[i];
  __syncthreads();
  if (threadIdx.x != 0) return;
  uint32_t c = 0u;
  unsigned long long e = 0ull;
  for (int i = 0; i < count; ++i) {
    const uint32_t t = staged[i * S + (int)c];
    e += t >> 4;
    c = t & 15u;
  }
  *out_entries = e;
}

}  // namespace pg
