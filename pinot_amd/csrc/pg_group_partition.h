// Partitioned group-by for key spaces that do not fit an LDS table (the reference's IntMapBasedHolder range,
// DictionaryBasedGroupKeyGenerator.java:164-184,415-490; aggregateGroupBySV of Sum / Min / Max / Avg / Count).
//
// One global atomic per doc and per aggregation is what the direct HBM table costs (measured: 23.7 G atomics/s on MI355X, i.e.
// 42 ms per atomic stream over 1 B rows, < 1 % of the HBM roofline).  This path pays streaming traffic instead:
//   pass 0  group_partition_histogram_kernel   docs per partition (partition = raw key >> shift), an upper bound without the filter
//   pass A  group_partition_scatter_kernel     filter + decode like group_private_kernel, then (raw key, value...) records are
//                                              appended to their partition's buffer.  A workgroup ranks its 8192 docs with LDS
//                                              atomics, reserves buffer space with ONE global atomic per (workgroup, partition)
//                                              -- global atomics drop by 8192 / P -- and stages the records in LDS in partition
//                                              order so that they leave as contiguous runs.
//   pass B  group_partition_aggregate_kernel   a workgroup takes a chunk of one partition: every key of the chunk falls in the
//                                              same 2^shift-slot window, which fits an LDS table; LDS atomics (~3 T/s) do the
//                                              aggregation and only the touched slots are flushed to the HBM table.
// The HBM table, its initialisation and the compaction of the result are shared with the direct path (pg_kernels.h).
// Traffic per doc: keys twice + values once + 2 x 4 B x (1 + aggregations) of record write / read.
// Packed records (PartitionParams.packed_bits): with at most one aggregation whose input is an unsigned field of b bits (a dictId for
// MIN / MAX, a value-plane field for SUM) and shift + b <= 32, the slot inside the partition and the value share ONE dword -- the
// partition itself is implied by where the record lies -- so the record traffic halves (2 x 4 B per doc), the value column needs no
// staging pass of its own (five barriers fewer per round) and the staging area shrinks from 64 KB to 48 KB (three workgroups per CU).
#pragma once
#include "pg_kernels.h"

namespace pg {

// raw key of the lane's 32 docs of a tile: sum dictId_c * mult_c (DictionaryBasedGroupKeyGenerator.java:437-445)
// (kWide = false: the full-rate 24-bit multiply while every multiplier stays below 2^24 -- GroupParams.wide_keys says when one does not:
//  key spaces above 2^24 with three or more key columns, which the two-level runs admit, take the 32-bit multiply)
template <bool kWide = false>
__device__ __forceinline__ void decode_group_keys(const GroupParams& gp, long long tile, int lane, uint32_t (&g)[32]) {
  for (int c = 0; c < gp.num_group_cols; ++c) {
    const DevGroupKey& key = gp.group_keys[c];
    const int b = key.bits;
    const uint32_t mult = (uint32_t)key.mult;
    const bool wide = kWide || gp.wide_keys != 0;            // (uniform)
    const uint32_t* words = reinterpret_cast<const uint32_t*>(key.fwd + tile * (256ll * b)) + lane * b;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t d[16];
      if (h == 0) decode16_private_dispatch<0>(b, words, d); else decode16_private_dispatch<1>(b, words, d);
      if (wide) {
#pragma unroll
        for (int j = 0; j < 16; ++j) g[16 * h + j] = c == 0 ? d[j] : key_term<true>(d[j], mult) + g[16 * h + j];
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) g[16 * h + j] = c == 0 ? d[j] : key_term<false>(d[j], mult) + g[16 * h + j];
      }
    }
  }
}

__device__ __forceinline__ uint32_t tail_mask(const GroupParams& gp, long long tile, int lane) {
  const long long rem = (long long)gp.scan.num_docs - (tile * 2048 + lane * 32);
  return rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (int)rem) - 1u));
}

static __global__ __launch_bounds__(256) void group_partition_histogram_kernel(const PartitionParams pp) {
  __shared__ uint32_t hist[kMaxPartitions];
  const GroupParams& gp = pp.gp;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < pp.num_partitions; i += 256) hist[i] = 0u;
  __syncthreads();
  const long long num_tiles = ((long long)gp.scan.num_docs + 2047) / 2048;
  const long long total_waves = (long long)gridDim.x * 4;
  for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < num_tiles; tile += total_waves) {
    const uint32_t m = tail_mask(gp, tile, lane);
    uint32_t g[32];
    decode_group_keys(gp, tile, lane, g);
#pragma unroll
    for (int j = 0; j < 32; ++j) if ((m >> j) & 1u) __hip_atomic_fetch_add(&hist[g[j] >> pp.shift], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < pp.num_partitions; i += 256) {
    const uint32_t c = hist[i];
    if (c) __hip_atomic_fetch_add(&pp.upper[i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// LDS (dynamic): hist[P], lbase[P + 1], gbase[P], then two 8192-dword staging columns (keys, current value column).
// The records of a round are first laid out in LDS in partition order (slot = partition's LDS base + rank), then copied out
// linearly: consecutive staging slots of one partition are consecutive in its global buffer, so a wavefront's store covers a
// few contiguous runs instead of 64 unrelated dwords (the direct scatter was bound by the L2's write-request rate).
constexpr int kScatterDocsPerRound = 8192;
inline size_t partition_scatter_lds_bytes() { return (size_t)(3 * kMaxPartitions + 1) * 4 + 2 * (size_t)kScatterDocsPerRound * 4; }

// (Tried and dropped: decoding the first value column ahead of the barriers and staging it together with the keys -- five
// barriers fewer per round, yet 4.6 -> 5.1 ms: the extra loads in flight queue in front of the key loads the ranking waits for.)
static __global__ __launch_bounds__(256) void group_partition_scatter_kernel(const PartitionParams pp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  uint32_t* lbase = hist + kMaxPartitions;              // [P + 1] exclusive prefix of hist; lbase[P] = records of the round
  uint32_t* gbase = lbase + kMaxPartitions + 1;         // [P] first global record slot reserved for the round
  uint32_t* skey = gbase + kMaxPartitions;
  uint32_t* sval = skey + kScatterDocsPerRound;
  const GroupParams& gp = pp.gp;
  const int P = pp.num_partitions;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long num_tiles = ((long long)gp.scan.num_docs + 2047) / 2048;
  const long long tiles_per_round = (long long)gridDim.x * 4;
  const long long rounds = (num_tiles + tiles_per_round - 1) / tiles_per_round;
  for (long long r = 0; r < rounds; ++r) {
    // the four wavefronts of the workgroup take four tiles and rank their 8192 docs together
    const long long tile = (r * gridDim.x + blockIdx.x) * 4 + wave;
    for (int i = threadIdx.x; i < P; i += 256) hist[i] = 0u;
    __syncthreads();
    uint32_t m = 0u;
    uint32_t g[32], slot[32];
    if (tile < num_tiles) {
      uint32_t entries_unused = 0u;             // (this kernel is not asked to count: ScanParams.filter_entries stays null)
      m = eval_filter_private(gp.scan, tile, lane, entries_unused) & tail_mask(gp, tile, lane);
      if (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
        decode_group_keys(gp, tile, lane, g);
#pragma unroll
        for (int j = 0; j < 32; ++j)
          if ((m >> j) & 1u) slot[j] = __hip_atomic_fetch_add(&hist[g[j] >> pp.shift], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    // reserve the global space of every partition (one atomic each) and lay the partitions out back to back in LDS
    for (int i = threadIdx.x; i < P; i += 256) {
      const uint32_t c = hist[i];
      if (c) gbase[i] = pp.offsets[i] + __hip_atomic_fetch_add(&pp.cursor[i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
      uint32_t carry = 0u;
      for (int base_i = 0; base_i < P; base_i += 64) {
        const int i = base_i + lane;
        const uint32_t c = i < P ? hist[i] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t up = __shfl_up(incl, d, 64);
          if (lane >= d) incl += up;
        }
        if (i < P) lbase[i] = carry + incl - c;
        carry += __shfl(incl, 63, 64);
      }
      if (lane == 0) lbase[P] = carry;
    }
    __syncthreads();
    const bool any = __builtin_amdgcn_ballot_w64(m != 0u) != 0ull;      // wave-uniform
    if (any) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        if ((m >> j) & 1u) {
          slot[j] += lbase[g[j] >> pp.shift];
          skey[slot[j]] = g[j];
        }
      }
    }
    __syncthreads();
    const uint32_t round_records = lbase[P];
    for (uint32_t i = threadIdx.x; i < round_records; i += 256) {
      const uint32_t key = skey[i];
      const uint32_t p = key >> pp.shift;
      pp.part_key[gbase[p] + (i - lbase[p])] = key;
    }
    const long long first_doc = tile * 2048 + lane * 32;
    for (int a = 0; a < gp.num_group_aggs; ++a) {
      const DevGroupAgg& ga = gp.group_aggs[a];
      if (any) {
        const int b = ga.bits;
        const uint32_t* words = ga.is_raw ? reinterpret_cast<const uint32_t*>(ga.fwd) + first_doc
                                          : reinterpret_cast<const uint32_t*>(ga.fwd + tile * (256ll * b)) + lane * b;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ga.dict, 0, ga.dict_bytes, 0x00020000);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint32_t d[16];
          if (ga.is_raw) {
#pragma unroll
            for (int j = 0; j < 16; ++j) d[j] = __builtin_bswap32(words[16 * h + j]);      // raw buffers are padded to whole tiles
          } else {
            if (h == 0) decode16_private_dispatch<0>(b, words, d); else decode16_private_dispatch<1>(b, words, d);
            if (ga.kind == kGroupSum && !ga.is_plane) {
#pragma unroll
              for (int j = 0; j < 16; ++j) d[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, d[j] * 4u, 0, 0);
            }
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) if ((m >> (16 * h + j)) & 1u) sval[slot[16 * h + j]] = d[j];
        }
      }
      __syncthreads();
      uint32_t* out = pp.part_val[a];
      for (uint32_t i = threadIdx.x; i < round_records; i += 256) {
        const uint32_t p = skey[i] >> pp.shift;
        out[gbase[p] + (i - lbase[p])] = sval[i];
      }
      __syncthreads();      // the next column (or the next round's keys) overwrites the staging area
    }
    __syncthreads();
  }
}

// Packed-record variant of the scatter pass (see the file header).  LDS: hist[P], lbase[P + 1], gbase[P] (sized by the actual P), the
// 8192 staged records and the partition of each (u16), so that the copy-out knows where a record goes without its key.
inline size_t partition_scatter_packed_lds_bytes(int num_partitions) {
  return (((size_t)(3 * num_partitions + 1) * 4 + 15) & ~(size_t)15) + (size_t)kScatterDocsPerRound * 4 + (size_t)kScatterDocsPerRound * 2;
}

static __global__ __launch_bounds__(256) void group_partition_scatter_packed_kernel(const PartitionParams pp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int P = pp.num_partitions;
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  uint32_t* lbase = hist + P;                           // [P + 1]
  uint32_t* gbase = lbase + P + 1;                      // [P]
  uint32_t* srec = reinterpret_cast<uint32_t*>(smem + ((((size_t)(3 * P + 1) * 4) + 15) & ~(size_t)15));
  uint16_t* spart = reinterpret_cast<uint16_t*>(srec + kScatterDocsPerRound);
  const GroupParams& gp = pp.gp;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint32_t slot_mask = (1u << pp.shift) - 1u;
  const long long num_tiles = ((long long)gp.scan.num_docs + 2047) / 2048;
  const long long tiles_per_round = (long long)gridDim.x * 4;
  const long long rounds = (num_tiles + tiles_per_round - 1) / tiles_per_round;
  for (long long r = 0; r < rounds; ++r) {
    const long long tile = (r * gridDim.x + blockIdx.x) * 4 + wave;
    for (int i = threadIdx.x; i < P; i += 256) hist[i] = 0u;
    __syncthreads();
    uint32_t m = 0u;
    uint32_t g[32], slot[32];
    if (tile < num_tiles) {
      uint32_t entries_unused = 0u;
      m = eval_filter_private(gp.scan, tile, lane, entries_unused) & tail_mask(gp, tile, lane);
      if (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
        decode_group_keys(gp, tile, lane, g);
#pragma unroll
        for (int j = 0; j < 32; ++j)
          if ((m >> j) & 1u) slot[j] = __hip_atomic_fetch_add(&hist[g[j] >> pp.shift], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += 256) {
      const uint32_t c = hist[i];
      if (c) gbase[i] = pp.offsets[i] + __hip_atomic_fetch_add(&pp.cursor[i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
      uint32_t carry = 0u;
      for (int base_i = 0; base_i < P; base_i += 64) {
        const int i = base_i + lane;
        const uint32_t c = i < P ? hist[i] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t up = __shfl_up(incl, d, 64);
          if (lane >= d) incl += up;
        }
        if (i < P) lbase[i] = carry + incl - c;
        carry += __shfl(incl, 63, 64);
      }
      if (lane == 0) lbase[P] = carry;
    }
    __syncthreads();
    // from here on gbase[p] is the global slot of LDS slot 0 of the round (global = gbase[p] + LDS index): one lookup per record
    for (int i = threadIdx.x; i < P; i += 256) if (hist[i]) gbase[i] -= lbase[i];
    if (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
      // the value of every doc joins its slot in the record (one aggregation at most; COUNT(*) has none: the value bits stay 0)
      if (gp.num_group_aggs == 1) {
        const DevGroupAgg& ga = gp.group_aggs[0];
        const int b = ga.bits;
        const uint32_t* words = reinterpret_cast<const uint32_t*>(ga.fwd + tile * (256ll * b)) + lane * b;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint32_t d[16];
          if (h == 0) decode16_private_dispatch<0>(b, words, d); else decode16_private_dispatch<1>(b, words, d);
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int jj = 16 * h + j;
            if ((m >> jj) & 1u) {
              const uint32_t p = g[jj] >> pp.shift;
              const uint32_t at = slot[jj] + lbase[p];
              srec[at] = ((g[jj] & slot_mask) << pp.packed_bits) | d[j];
              spart[at] = (uint16_t)p;
            }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          if ((m >> j) & 1u) {
            const uint32_t p = g[j] >> pp.shift;
            const uint32_t at = slot[j] + lbase[p];
            srec[at] = (g[j] & slot_mask) << pp.packed_bits;
            spart[at] = (uint16_t)p;
          }
        }
      }
    }
    __syncthreads();
    const uint32_t round_records = lbase[P];
    for (uint32_t i = threadIdx.x; i < round_records; i += 256) {
      pp.part_key[gbase[spart[i]] + i] = srec[i];
    }
    __syncthreads();      // the next round's records overwrite the staging area
  }
}

// LDS: acc[NA][S] (i64) then cnt[S] (u32), S = 1 << shift.  NA (accumulators) is a template parameter so that the record loads of
// a batch are straight-line code: with a run-time column loop and guarded loads the compiler waited for every load separately.
template <int NA, bool kPacked = false>
__global__ __launch_bounds__(256) void group_partition_aggregate_kernel(const PartitionParams pp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const GroupParams& gp = pp.gp;
  const int S = 1 << pp.shift;
  long long* acc = reinterpret_cast<long long*>(smem);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(acc + (size_t)NA * S);
  if (pp.work_count != nullptr && blockIdx.x >= *pp.work_count) return;      // (two-level runs launch the most work items there can be)
  const PartitionWork w = pp.work[blockIdx.x];
  const uint32_t written = pp.cursor[w.partition];
  if (w.start >= written) return;                        // workgroup-uniform: the filter left this chunk empty
  const uint32_t n = min(w.len, written - w.start);
  for (int s = threadIdx.x; s < S; s += 256) {
    cnt[s] = 0u;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int kind = gp.group_aggs[a].kind;
      acc[(size_t)a * S + s] = kind == kGroupSum ? 0ll : (kind == kGroupMin ? 0x7FFFFFFFll : -0x80000000ll);
    }
  }
  __syncthreads();
  const uint32_t first = pp.offsets[w.partition] + w.start;
  const uint32_t key_base = (uint32_t)w.partition << pp.shift;
  const uint32_t* __restrict__ keys = pp.part_key + first;
  // eight record loads per thread and column are in flight before the LDS atomics of the batch start; indexes past the chunk are
  // clamped (the loads stay unconditional) and their records skipped
  constexpr int kUnroll = 8;
  for (uint32_t i0 = threadIdx.x; i0 < n; i0 += 256 * kUnroll) {
    uint32_t key[kUnroll], val[NA > 0 ? NA : 1][kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t i = min(i0 + 256u * u, n - 1u);
      key[u] = keys[i];
      if constexpr (!kPacked) {
#pragma unroll
        for (int a = 0; a < NA; ++a) val[a][u] = pp.part_val[a][first + i];
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if (i0 + 256u * u >= n) continue;
      // (a packed record carries the slot inside its COARSE partition when the run is two-level: the fine slot is its low bits)
      const uint32_t slot = kPacked ? (key[u] >> pp.packed_bits) & (uint32_t)(S - 1) : key[u] - key_base;
      __hip_atomic_fetch_add(&cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const DevGroupAgg& ga = gp.group_aggs[a];
        const uint32_t v = kPacked ? key[u] & ((1u << pp.packed_bits) - 1u) : val[a][u];
        long long* slot_acc = acc + (size_t)a * S + slot;
        if (ga.kind == kGroupSum) {
          const bool is_unsigned = !ga.is_raw && ga.is_plane;
          __hip_atomic_fetch_add(slot_acc, is_unsigned ? (long long)v : (long long)(int32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (ga.kind == kGroupMin) {
          __hip_atomic_fetch_min(slot_acc, (long long)(int32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          __hip_atomic_fetch_max(slot_acc, (long long)(int32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
  __syncthreads();
  const long long G = gp.num_groups;
  for (int s = threadIdx.x; s < S; s += 256) {
    const uint32_t c = cnt[s];
    if (!c) continue;
    const long long g = (long long)key_base + s;
    __hip_atomic_fetch_add(&gp.table_count[g], (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int kind = gp.group_aggs[a].kind;
      long long* slot = gp.table_acc + (long long)a * G + g;
      const long long v = acc[(size_t)a * S + s];
      if (kind == kGroupSum) __hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (kind == kGroupMin) __hip_atomic_fetch_min(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_max(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Two-level partitioning: key spaces of more than kMaxPartitions fine partitions (2 M .. 2^31 raw keys -- the upper IntMapBasedHolder
// range, DictionaryBasedGroupKeyGenerator.java:164-181, 415-460).  One scatter pass cannot address more than a few hundred
// destinations (its ranking and staging live in LDS), and the direct HBM table costs one global atomic per doc and accumulator
// (23.7 G/s: 42 ms per atomic stream over 1 B rows).  So pass A scatters by COARSE partition (P1 <= 512, each 2^k fine partitions
// wide), and the records of every coarse partition -- contiguous in the first buffer -- are scattered once more by fine partition
// into a second buffer:
//   group_repartition_count_kernel   a workgroup per chunk of <= 65 536 records of one coarse partition: LDS histogram over the 2^k
//                                    fine partitions, one global atomic per (chunk, fine partition it touches)
//   group_repartition_plan_kernel    a workgroup per coarse partition: exclusive scan of its fine counts -> where every fine partition
//                                    starts (inside the coarse partition's own range of the second buffer), and pass B's work list
//   group_repartition_scatter_kernel the same chunks again: count, reserve (one global atomic per chunk and fine partition), then
//                                    every record goes to  fine_offsets + reserved + its rank  (the second read of the chunk comes
//                                    out of the L2)
// Pass B (group_partition_aggregate_kernel) then runs over fine partitions exactly as in a one-level run.  Per doc: 4 B x (1 + inputs)
// written and read twice more than one level, against the 42 ms of every atomic stream.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fine_of_record(const RepartitionParams& rp, uint32_t rec) {
  // unpacked: the raw key; packed: (slot inside the coarse partition) << packed_bits | value
  const uint32_t in_coarse = rp.packed_bits > 0 ? rec >> rp.packed_bits : rec;
  return (in_coarse >> rp.fine_shift) & ((1u << rp.log2_fine_per_coarse) - 1u);
}

static __global__ __launch_bounds__(256) void group_repartition_count_kernel(const RepartitionParams rp) {
  __shared__ uint32_t hist[kMaxFinePerCoarse];
  const PartitionWork w = rp.chunks[blockIdx.x];
  const uint32_t written = rp.coarse_cursor[w.partition];
  if (w.start >= written) return;                        // the filter left this chunk empty
  const uint32_t n = min(w.len, written - w.start);
  const int F = 1 << rp.log2_fine_per_coarse;
  for (int i = threadIdx.x; i < F; i += 256) hist[i] = 0u;
  __syncthreads();
  const uint32_t* __restrict__ keys = rp.src_key + rp.coarse_offsets[w.partition] + w.start;
  for (uint32_t i = threadIdx.x; i < n; i += 256) __hip_atomic_fetch_add(&hist[fine_of_record(rp, keys[i])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  uint32_t* out = rp.fine_count + ((size_t)w.partition << rp.log2_fine_per_coarse);
  for (int i = threadIdx.x; i < F; i += 256) { const uint32_t c = hist[i]; if (c) __hip_atomic_fetch_add(&out[i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

// one workgroup of 1024 threads per coarse partition (2^k <= 1024 fine partitions: one per thread)
static __global__ __launch_bounds__(1024) void group_repartition_plan_kernel(const RepartitionParams rp) {
  __shared__ uint32_t wave_sum[16], wave_items[16];
  __shared__ uint32_t items_base;
  const int F = 1 << rp.log2_fine_per_coarse;
  const int p1 = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const size_t q = ((size_t)p1 << rp.log2_fine_per_coarse) + (size_t)t;
  const uint32_t c = t < F ? rp.fine_count[q] : 0u;
  const uint32_t items = (c + rp.aggregate_chunk - 1u) / rp.aggregate_chunk;
  uint32_t incl = c, incl_items = items;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64), up_items = __shfl_up(incl_items, d, 64);
    if (lane >= d) { incl += up; incl_items += up_items; }
  }
  if (lane == 63) { wave_sum[wave] = incl; wave_items[wave] = incl_items; }
  __syncthreads();
  uint32_t before = 0u, before_items = 0u, all_items = 0u;
  for (int v = 0; v < 16; ++v) { if (v < wave) { before += wave_sum[v]; before_items += wave_items[v]; } all_items += wave_items[v]; }
  if (t == 0) items_base = all_items ? __hip_atomic_fetch_add(rp.work_count, all_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  __syncthreads();
  if (t < F) {
    rp.fine_offsets[q] = rp.coarse_offsets[p1] + before + incl - c;
    uint32_t at = items_base + before_items + incl_items - items;
    for (uint32_t s0 = 0; s0 < c; s0 += rp.aggregate_chunk) rp.work[at++] = PartitionWork{(int32_t)q, s0, min(rp.aggregate_chunk, c - s0), 0u};
  }
}

static __global__ __launch_bounds__(256) void group_repartition_scatter_kernel(const RepartitionParams rp) {
  __shared__ uint32_t hist[kMaxFinePerCoarse];          // counts, then the next free rank of every fine partition
  __shared__ uint32_t base[kMaxFinePerCoarse];          // where the chunk's records of a fine partition start in the second buffer
  const PartitionWork w = rp.chunks[blockIdx.x];
  const uint32_t written = rp.coarse_cursor[w.partition];
  if (w.start >= written) return;
  const uint32_t n = min(w.len, written - w.start);
  const int F = 1 << rp.log2_fine_per_coarse;
  for (int i = threadIdx.x; i < F; i += 256) hist[i] = 0u;
  __syncthreads();
  const uint32_t first = rp.coarse_offsets[w.partition] + w.start;
  const uint32_t* __restrict__ keys = rp.src_key + first;
  for (uint32_t i = threadIdx.x; i < n; i += 256) __hip_atomic_fetch_add(&hist[fine_of_record(rp, keys[i])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  const size_t q0 = (size_t)w.partition << rp.log2_fine_per_coarse;
  for (int i = threadIdx.x; i < F; i += 256) {
    const uint32_t c = hist[i];
    base[i] = c ? rp.fine_offsets[q0 + i] + __hip_atomic_fetch_add(&rp.fine_cursor[q0 + i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    hist[i] = 0u;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    const uint32_t rec = keys[i];
    const uint32_t f = fine_of_record(rp, rec);
    const uint32_t at = base[f] + __hip_atomic_fetch_add(&hist[f], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    rp.dst_key[at] = rec;
    for (int a = 0; a < rp.num_vals; ++a) rp.dst_val[a][at] = rp.src_val[a][first + i];
  }
}

// ------------------------------------------------------------------------------------------------
// group_typed_direct_kernel: group-by whose aggregation inputs include a RAW LONG / FLOAT / DOUBLE column
// (SumAggregationFunction.aggregateGroupBySV over getDoubleValuesSV / getLongValuesSV of a no-dictionary column,
// core/query/aggregation/function/SumAggregationFunction.java:160-179; Min / Max / Avg likewise).  Those values have no 32-bit image
// (no dictId, no plane field), so none of the LDS-table kernels takes them: this kernel keeps the lane-private filter and key decode
// and aggregates straight into the direct-indexed HBM table, one global atomic per doc and accumulator (23.7 G atomics/s: the price
// of the general case; the reference runs it at ~25 M rows/s per core).  Every other input kind rides along (dictIds, plane fields,
// gathered dictionary values of either width), so the query runs in one pass.
// Table slots: SUM of LONG / INT: int64 add; SUM of FLOAT / DOUBLE: the slot holds a double (global_atomic_add_f64); MIN / MAX of raw LONG:
// the value; of raw FLOAT / DOUBLE: its order-preserving 64-bit key (f64_order_key); of dictionary columns: the dictId.
// ------------------------------------------------------------------------------------------------
// kHash (round 6b): the keys are beyond an int -- the slots come from the hashed table (hashed_group_slots, pg_kernels.h: the Long / ArrayMap
// holders of DictionaryBasedGroupKeyGenerator.java:628-806), everything behind the slot number is the same.
template <bool kWide, bool kHash = false>
static __global__ __launch_bounds__(256) void group_typed_direct_kernel(const GroupParams gp) {
  const int lane = threadIdx.x & 63;
  const long long num_tiles = ((long long)gp.scan.num_docs + 2047) / 2048;
  const long long total_waves = (long long)gridDim.x * 4;
  const long long G = gp.num_groups;
  uint32_t entries = 0u;
  for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < num_tiles; tile += total_waves) {
    const uint32_t m = eval_filter_private(gp.scan, tile, lane, entries) & tail_mask(gp, tile, lane);
    if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) continue;
    uint32_t g[32];
    if constexpr (kHash) hashed_group_slots<true>(gp, tile, lane, m, g);
    else decode_group_keys<kWide>(gp, tile, lane, g);
    const long long first_doc = tile * 2048 + lane * 32;
#pragma unroll
    for (int j = 0; j < 32; ++j) if ((m >> j) & 1u) __hip_atomic_fetch_add(&gp.table_count[g[j]], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int a = 0; a < gp.num_group_aggs; ++a) {
      const DevGroupAgg& ga = gp.group_aggs[a];
      long long* acc = gp.table_acc + (long long)a * G;
      const bool wide_raw = ga.is_raw && (ga.vkind == kValI64 || ga.vkind == kValF64);
      const bool float_sum = ga.kind == kGroupSum && (ga.vkind == kValF64 || ga.vkind == kValF32);
      uint32_t d[32];
      if (!ga.is_raw) {
        const int b = ga.bits;
        const uint32_t* words = reinterpret_cast<const uint32_t*>(ga.fwd + tile * (256ll * b)) + lane * b;
        decode16_private_dispatch<0>(b, words, *reinterpret_cast<uint32_t(*)[16]>(&d[0]));
        decode16_private_dispatch<1>(b, words, *reinterpret_cast<uint32_t(*)[16]>(&d[16]));
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        if (!((m >> j) & 1u)) continue;
        long long bits;                  // the accumulator's operand: an integer, the bits of a double, or a key
        if (wide_raw) {
          bits = (long long)__builtin_bswap64(*reinterpret_cast<const unsigned long long*>(ga.fwd + (first_doc + j) * 8));
        } else if (ga.is_raw) {
          const uint32_t w = __builtin_bswap32(*reinterpret_cast<const uint32_t*>(ga.fwd + (first_doc + j) * 4));
          bits = ga.vkind == kValF32 ? __double_as_longlong((double)__uint_as_float(w)) : (long long)(int32_t)w;
        } else if (ga.kind == kGroupSum && !ga.is_plane) {
          bits = ga.vkind == kValI32 ? (long long)ga.dict[d[j]] : reinterpret_cast<const long long*>(ga.dict)[d[j]];
        } else {
          bits = (long long)d[j];        // dictId (MIN / MAX) or plane field (SUM)
        }
        long long* slot = acc + g[j];
        if (ga.kind == kGroupSum) {
          if (float_sum) __hip_atomic_fetch_add(reinterpret_cast<double*>(slot), __longlong_as_double(bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else __hip_atomic_fetch_add(slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const bool float_key = ga.is_raw && (ga.vkind == kValF64 || ga.vkind == kValF32);
          const long long key = float_key ? f64_order_key(__longlong_as_double(bits)) : bits;
          if (ga.kind == kGroupMin) __hip_atomic_fetch_min(slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else __hip_atomic_fetch_max(slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  flush_filter_entries(gp.scan, entries);
}

}  // namespace pg
