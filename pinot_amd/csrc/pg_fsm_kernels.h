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

template <int SMAX>
__global__ __launch_bounds__(256) void fsm_tiles_kernel(const FsmParams p) {
  __shared__ uint8_t delta[kFsmStates << kFsmInputs];
  __shared__ uint32_t lane_tables[4][64 * SMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = p.num_inputs, S = p.num_states;
  for (int i = threadIdx.x; i < (S << L); i += blockDim.x) delta[i] = p.delta[i];
  __syncthreads();
  uint32_t* mine = lane_tables[wave];
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < p.num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048 + lane * 32;
    const long long rem = (long long)p.num_docs - first;
    const int docs = rem >= 32 ? 32 : (rem <= 0 ? 0 : (int)rem);
    uint32_t w[kFsmInputs];
#pragma unroll
    for (int i = 0; i < kFsmInputs; ++i) w[i] = i < L ? p.leaf[i][tile * 64 + lane] : 0u;
    uint32_t cur[SMAX], ent[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) { cur[s] = (uint32_t)s << L; ent[s] = 0u; }
    for (int d = 0; d < 32; ++d) {
      if (d >= docs) break;                       // (docs past numDocs do not exist; the lanes of the last tile stop at different docs)
      uint32_t in = 0u;
#pragma unroll
      for (int i = 0; i < kFsmInputs; ++i) in |= ((w[i] >> d) & 1u) << i;
#pragma unroll
      for (int s = 0; s < SMAX; ++s) {            // S independent chains: their LDS reads are in flight together
        if (s < S) {
          const uint32_t t = delta[cur[s] | in];
          ent[s] += t >> 4;
          cur[s] = (t & 15u) << L;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < SMAX; ++s) if (s < S) mine[lane * SMAX + s] = (cur[s] >> L) | (ent[s] << 4);
    __builtin_amdgcn_wave_barrier();
    // lanes 0 .. S-1: entry state `lane` walked through the 64 lane tables in lane order (LDS operations of one wave execute in order)
    if (lane < S) {
      uint32_t c = (uint32_t)lane, e = 0u;
      for (int l = 0; l < 64; ++l) {
        const uint32_t t = mine[l * SMAX + (int)c];
        e += t >> 4;
        c = t & 15u;
      }
      p.tables[tile * S + lane] = c | (e << 4);
    }
    __builtin_amdgcn_wave_barrier();
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
