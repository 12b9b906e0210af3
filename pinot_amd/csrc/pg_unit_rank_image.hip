// Builds the dictionary and rank image of a raw group-by key column (pg_rank_image.h).  One-time work per column and segment, off the query
// path after the first GROUP BY on it -- but the first query that groups by such a column pays it inline.
//
// Round 5 sorted the column: 64-bit order images of all docs -> rocPRIM radix sort -> unique -> a binary search per doc.  Measured at 1 B docs
// x 100 000 distinct values (profiles/r6/rank_image_build_sort_based.txt): sort 45 ms + unique 25 ms + pack 47-96 ms = 120-236 ms -- 90-180x
// the 1.3 ms the column takes to stream -- with 24 GB of transient HBM nobody accounted for.  A group-by key has FEW distinct values
// next to its docs, so the build is now what NoDictionarySingleColumnGroupKeyGenerator itself is -- a hash map over the values
// (core/query/aggregation/groupby/NoDictionarySingleColumnGroupKeyGenerator.java:100-135) -- in four hand-written passes:
//   1. rank_hash_insert_kernel   every doc's image into an open-addressing table in HBM (16-byte slots {image, rank}; Fibonacci hash, linear
//                                probing).  A doc whose image is already there -- nearly all of them -- costs one 16-byte read of a table
//                                that the L2 / Infinity Cache hold; only a first occurrence is a 64-bit CAS.  The table starts at 2^20
//                                slots and is rebuilt 16x larger when probes grow long or more than half of it fills.
//   2. rank_hash_collect_kernel  the occupied slots -> the distinct images (one atomic per wavefront);
//      rocprim::radix_sort_keys  over the DISTINCT images only (the one library call left: C keys, not N) -> the dictionary;
//   3. rank_hash_assign_kernel   dictionary entry r finds its slot and leaves r there;
//   4. rank_image_pack_kernel    every doc again: one probe -> its rank -> the bit-packed image (lane-private tile layout).
// Transient HBM: the table (16 MB at 2^20 slots) + two arrays of C images.
#include "pg_rank_image.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>

#include "pg_device.h"

namespace pg {

__device__ __forceinline__ unsigned long long order_image_of_double_bits(unsigned long long b) {
  if ((b & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) b = 0x7FF8000000000000ull;      // every NaN is Double.NaN
  return (b >> 63) ? ~b : (b | (1ull << 63));
}

__device__ __forceinline__ unsigned long long order_image(const uint8_t* __restrict__ raw, int vkind, long long doc) {
  if (vkind == kValI32) return (unsigned long long)(long long)(int32_t)__builtin_bswap32(reinterpret_cast<const uint32_t*>(raw)[doc]) ^ (1ull << 63);
  if (vkind == kValI64) return __builtin_bswap64(reinterpret_cast<const unsigned long long*>(raw)[doc]) ^ (1ull << 63);
  if (vkind == kValF32) {
    const float f = __uint_as_float(__builtin_bswap32(reinterpret_cast<const uint32_t*>(raw)[doc]));
    return order_image_of_double_bits((unsigned long long)__double_as_longlong((double)f));      // (float -> double is exact)
  }
  return order_image_of_double_bits(__builtin_bswap64(reinterpret_cast<const unsigned long long*>(raw)[doc]));
}

// ---- the hash table: 16-byte slots {image, rank}; kRankEmpty marks a free slot (the one image equal to it travels in a flag of its own) ----
constexpr unsigned long long kRankEmpty = ~0ull;
constexpr int kRankMaxProbe = 128;                       // a probe sequence this long means the table is too full: rebuilt larger
struct RankSlot { unsigned long long key; unsigned long long rank; };
struct RankTable {
  RankSlot* slots;
  int log2_slots;
  unsigned int* flags;                                   // [0] = 1: the table overflowed (rebuild larger); [1] = 1: an image equal to kRankEmpty exists (it is the largest image)
};
__device__ __forceinline__ unsigned long long rank_hash_slot(unsigned long long key, int log2_slots) {
  return (key * 0x9E3779B97F4A7C15ull) >> (64 - log2_slots);
}
// the slot that holds `key`, or (kInsert) the slot it now occupies; ~0 when the probe sequence ran out
template <bool kInsert>
__device__ __forceinline__ unsigned long long rank_hash_find(const RankTable& t, unsigned long long key) {
  const unsigned long long mask = (1ull << t.log2_slots) - 1ull;
  unsigned long long h = rank_hash_slot(key, t.log2_slots);
  for (int probe = 0; probe < kRankMaxProbe; ++probe, h = (h + 1ull) & mask) {
    unsigned long long cur = __hip_atomic_load(&t.slots[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return h;
    if (cur == kRankEmpty) {
      if (!kInsert) return ~0ull;
      unsigned long long expected = kRankEmpty;
      if (__hip_atomic_compare_exchange_strong(&t.slots[h].key, &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || expected == key) return h;
    }
  }
  return ~0ull;
}

static __global__ __launch_bounds__(256) void rank_hash_clear_kernel(RankSlot* slots, unsigned long long num_slots) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < num_slots; i += (unsigned long long)gridDim.x * blockDim.x) {
    slots[i].key = kRankEmpty; slots[i].rank = 0ull;
  }
}

static __global__ __launch_bounds__(256) void rank_hash_insert_kernel(const uint8_t* __restrict__ raw, int vkind, long long num_docs, RankTable t) {
  int round = 0;
  for (long long doc = (long long)blockIdx.x * blockDim.x + threadIdx.x; doc < num_docs; doc += (long long)gridDim.x * blockDim.x, ++round) {
    // somebody found the table too full: this attempt is over (looked at every 32nd doc: the flag is one L2 line for the whole grid)
    if ((round & 31) == 0 && __hip_atomic_load(&t.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const unsigned long long key = order_image(raw, vkind, doc);
    if (key == kRankEmpty) { if (t.flags[1] == 0u) __hip_atomic_store(&t.flags[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); continue; }
    if (rank_hash_find<true>(t, key) == ~0ull) __hip_atomic_store(&t.flags[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// the occupied slots' images -> out[0 .. *count): one atomic per wavefront
static __global__ __launch_bounds__(256) void rank_hash_collect_kernel(const RankSlot* __restrict__ slots, unsigned long long num_slots, unsigned long long* __restrict__ out,
                                                                unsigned long long capacity, unsigned long long* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long rounds = (num_slots + stride - 1) / stride;      // (uniform trip count: the ballot below sees every lane)
  for (unsigned long long r = 0; r < rounds; ++r) {
    const unsigned long long i = r * stride + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long key = i < num_slots ? slots[i].key : kRankEmpty;
    const unsigned long long have = __builtin_amdgcn_ballot_w64(key != kRankEmpty);
    if (have == 0ull) continue;
    unsigned long long base = 0ull;
    if (lane == 0) base = __hip_atomic_fetch_add(count, (unsigned long long)__builtin_popcountll(have), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = (unsigned long long)__shfl((long long)base, 0);
    if (key != kRankEmpty) {
      const unsigned long long at = base + (unsigned long long)__builtin_popcountll(have & ((1ull << lane) - 1ull));
      if (at < capacity) out[at] = key;
    }
  }
}

// dictionary entry r -> its slot's rank
static __global__ __launch_bounds__(256) void rank_hash_assign_kernel(const unsigned long long* __restrict__ dict, int cardinality, RankTable t) {
  for (int r = (int)(blockIdx.x * blockDim.x + threadIdx.x); r < cardinality; r += (int)(gridDim.x * blockDim.x)) {
    const unsigned long long key = dict[r];
    if (key == kRankEmpty) continue;                      // (the largest image: it has no slot, its rank is cardinality - 1)
    const unsigned long long h = rank_hash_find<false>(t, key);
    if (h != ~0ull) t.slots[h].rank = (unsigned long long)r;
  }
}

// the rank of every doc's value in the sorted dictionary, packed MSB-first at `bits_out` bits per doc in the lane-private tile layout
// (the same layout as build_raw_key_image_kernel: lane l of a tile owns docs [32 l, 32 l + 32), bits_out dwords).  A wavefront per tile:
// the column is read COALESCED (round r: lane l takes doc 64 r + l of the tile -- a lane walking its own 32 docs made every load of the
// wavefront 64 lines 256 bytes apart, and the pack was 46 ms of the 57 ms build at 1 B docs), probed right there, and the ranks change
// hands through the wave's LDS (doc d at d + (d >> 5): the owners' read-back is conflict-free); the packed dwords go back through the same
// LDS and leave as whole lines.
static __global__ __launch_bounds__(256) void rank_image_pack_kernel(const uint8_t* __restrict__ raw, int vkind, RankTable t, int cardinality,
                                                              uint8_t* __restrict__ out, int bits_out, int num_tiles, long long num_docs) {
  __shared__ uint32_t ids[4][2048 + 64];
  const int lane = threadIdx.x & 63;
  uint32_t* mine = ids[threadIdx.x >> 6];
  for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < (long long)num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048;
#pragma unroll 4
    for (int r = 0; r < 32; ++r) {
      const int d = r * 64 + lane;
      const long long doc = first + d;
      uint32_t id = 0u;
      if (doc < num_docs) {
        const unsigned long long key = order_image(raw, vkind, doc);
        if (key == kRankEmpty) id = (uint32_t)(cardinality - 1);                 // the largest image
        else {
          const unsigned long long h = rank_hash_find<false>(t, key);           // the key IS in the table
          id = h != ~0ull ? (uint32_t)t.slots[h].rank : 0u;
        }
      }
      mine[d + (d >> 5)] = id;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = mine[lane * 33 + j];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long acc = 0ull;
    int have = 0, k = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      acc = (acc << bits_out) | (unsigned long long)v[j];
      have += bits_out;
      if (have >= 32) { mine[lane * bits_out + k++] = __builtin_bswap32((uint32_t)(acc >> (have - 32))); have -= 32; }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + tile * (256ll * bits_out));
    for (int i = lane; i < 64 * bits_out; i += 64) dst[i] = mine[i];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

pg_status build_rank_image(const uint8_t* d_raw, int vkind, long long num_docs, int num_tiles, int num_cus, unsigned long long** out_d_dict,
                           std::vector<unsigned long long>* out_h_dict, uint8_t** out_image, size_t* out_image_bytes, int* out_bits, int* out_cardinality,
                           const char** out_error) {
  static thread_local char message[256];
  *out_error = message;
  message[0] = 0;
  RankSlot* d_slots = nullptr;
  unsigned long long *d_keys = nullptr, *d_sorted = nullptr, *d_dict = nullptr, *d_count = nullptr;
  unsigned int* d_flags = nullptr;
  void* d_temp = nullptr;
  uint8_t* d_image = nullptr;
  hipStream_t stream = nullptr;
  auto cleanup = [&] {
    if (d_slots) (void)hipFree(d_slots);
    if (d_keys) (void)hipFree(d_keys);
    if (d_sorted) (void)hipFree(d_sorted);
    if (d_temp) (void)hipFree(d_temp);
    if (d_count) (void)hipFree(d_count);
    if (d_flags) (void)hipFree(d_flags);
    if (stream) (void)hipStreamDestroy(stream);
    d_slots = nullptr; d_keys = d_sorted = d_count = nullptr; d_temp = nullptr; d_flags = nullptr; stream = nullptr;
  };
#define PG_RANK_TRY(expr, what)                                                                                        \
  do {                                                                                                                 \
    const hipError_t e_ = (expr);                                                                                      \
    if (e_ != hipSuccess) {                                                                                            \
      snprintf(message, sizeof(message), "rank image: %s: %s", what, hipGetErrorString(e_));                           \
      (void)hipGetLastError();                                                                                         \
      cleanup();                                                                                                       \
      if (d_dict) (void)hipFree(d_dict);                                                                               \
      if (d_image) (void)hipFree(d_image);                                                                             \
      return e_ == hipErrorOutOfMemory ? PG_ERR_OUT_OF_MEMORY : PG_ERR_DEVICE;                                         \
    }                                                                                                                  \
  } while (0)
  // PINOT_GPU_RANK_TRACE=1: the phases on the host clock (a stream synchronisation behind each) and the transient allocations, on stderr
  const bool trace = getenv("PINOT_GPU_RANK_TRACE") != nullptr && getenv("PINOT_GPU_RANK_TRACE")[0] == '1';
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
  const auto t_start = now();
  PG_RANK_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "stream");
  PG_RANK_TRY(hipMalloc((void**)&d_flags, 8), "flags");
  PG_RANK_TRY(hipMalloc((void**)&d_count, 8), "count");
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((num_docs + 255) / 256, (long long)num_cus * 16));
  // ---- 1. the distinct images: a table of 2^20 slots, 16x larger whenever it fills (PINOT_GPU_RANK_SLOTS_LOG2: where it starts -- tests) ----
  int log2_slots = 20;
  if (const char* ls = getenv("PINOT_GPU_RANK_SLOTS_LOG2")) log2_slots = std::max(6, std::min(30, atoi(ls)));
  unsigned long long selected_count = 0;
  unsigned int h_flags[2] = {0u, 0u};
  double ms_insert = 0, ms_collect = 0, ms_sort = 0, ms_assign = 0, ms_pack = 0;
  int attempts = 0;
  size_t table_bytes = 0;
  for (;;) {
    ++attempts;
    const unsigned long long num_slots = 1ull << log2_slots;
    table_bytes = (size_t)num_slots * sizeof(RankSlot);
    PG_RANK_TRY(hipMalloc((void**)&d_slots, table_bytes), "hash table");
    PG_RANK_TRY(hipMemsetAsync(d_flags, 0, 8, stream), "flags clear");
    PG_RANK_TRY(hipMemsetAsync(d_count, 0, 8, stream), "count clear");
    rank_hash_clear_kernel<<<dim3((unsigned)std::min<unsigned long long>((num_slots + 255) / 256, (unsigned long long)num_cus * 16)), dim3(256), 0, stream>>>(d_slots, num_slots);
    PG_RANK_TRY(hipGetLastError(), "table clear");
    const RankTable table{d_slots, log2_slots, d_flags};
    const auto t_insert = now();
    if (num_docs > 0) rank_hash_insert_kernel<<<dim3(grid), dim3(256), 0, stream>>>(d_raw, vkind, num_docs, table);
    PG_RANK_TRY(hipGetLastError(), "insert kernel");
    PG_RANK_TRY(hipMemcpyAsync(h_flags, d_flags, 8, hipMemcpyDeviceToHost, stream), "flags copy");
    PG_RANK_TRY(hipStreamSynchronize(stream), "insert");
    ms_insert += ms_since(t_insert);
    bool too_small = h_flags[0] != 0u;
    if (!too_small) {
      // the occupied slots: counted first (capacity 0), collected once their number is known to be at most half the table
      const auto t_collect = now();
      const unsigned cgrid = (unsigned)std::min<unsigned long long>((num_slots + 255) / 256, (unsigned long long)num_cus * 16);
      rank_hash_collect_kernel<<<dim3(cgrid), dim3(256), 0, stream>>>(d_slots, num_slots, nullptr, 0ull, d_count);
      PG_RANK_TRY(hipGetLastError(), "count kernel");
      PG_RANK_TRY(hipMemcpyAsync(&selected_count, d_count, 8, hipMemcpyDeviceToHost, stream), "count copy");
      PG_RANK_TRY(hipStreamSynchronize(stream), "count");
      too_small = selected_count * 2ull > num_slots && log2_slots < 31;
      if (!too_small) {
        const unsigned long long total = selected_count + (h_flags[1] ? 1ull : 0ull);
        if (total >= 0x7FFFFFFEull) {
          snprintf(message, sizeof(message), "rank image: %llu distinct values do not fit the int dictId domain", total);
          cleanup();
          return PG_ERR_UNSUPPORTED;
        }
        PG_RANK_TRY(hipMalloc((void**)&d_keys, (size_t)std::max<unsigned long long>(total, 1) * 8), "distinct keys");
        PG_RANK_TRY(hipMalloc((void**)&d_sorted, (size_t)std::max<unsigned long long>(total, 1) * 8), "sorted keys");
        PG_RANK_TRY(hipMemsetAsync(d_count, 0, 8, stream), "count clear");
        rank_hash_collect_kernel<<<dim3(cgrid), dim3(256), 0, stream>>>(d_slots, num_slots, d_keys, selected_count, d_count);
        PG_RANK_TRY(hipGetLastError(), "collect kernel");
        if (h_flags[1]) PG_RANK_TRY(hipMemcpyAsync(d_keys + selected_count, &kRankEmpty, 8, hipMemcpyHostToDevice, stream), "largest image");
        if (trace) { PG_RANK_TRY(hipStreamSynchronize(stream), "collect"); }
        ms_collect = ms_since(t_collect);
        selected_count = total;
        break;
      }
    }
    // rebuilt larger: 16x the slots (an attempt that overflowed stopped early; one that merely passed half its slots ran to the end)
    (void)hipFree(d_slots); d_slots = nullptr;
    if (log2_slots >= 31) { snprintf(message, sizeof(message), "rank image: more distinct values than a table of 2^31 slots holds"); cleanup(); return PG_ERR_UNSUPPORTED; }
    log2_slots = std::min(31, log2_slots + 4);
  }
  const int cardinality = (int)selected_count;
  int bits = 1;
  while (bits < 31 && (1ll << bits) < (long long)cardinality) ++bits;      // PinotDataBitSet.getNumBitsPerValue(cardinality - 1)
  // ---- 2. the dictionary: the distinct images, ascending ----
  const auto t_sort = now();
  size_t temp_sort = 0;
  if (cardinality > 0) {
    PG_RANK_TRY(rocprim::radix_sort_keys(nullptr, temp_sort, d_keys, d_sorted, (size_t)cardinality, 0u, 64u, stream), "sort sizing");
    PG_RANK_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_sort, 256)), "sort scratch");
    PG_RANK_TRY(rocprim::radix_sort_keys(d_temp, temp_sort, d_keys, d_sorted, (size_t)cardinality, 0u, 64u, stream), "sort");
  }
  PG_RANK_TRY(hipMalloc((void**)&d_dict, (size_t)std::max(cardinality, 1) * 8), "dictionary");
  if (cardinality > 0) PG_RANK_TRY(hipMemcpyAsync(d_dict, d_sorted, (size_t)cardinality * 8, hipMemcpyDeviceToDevice, stream), "dictionary copy");
  out_h_dict->assign((size_t)cardinality, 0ull);
  if (cardinality > 0) PG_RANK_TRY(hipMemcpyAsync(out_h_dict->data(), d_sorted, (size_t)cardinality * 8, hipMemcpyDeviceToHost, stream), "dictionary to host");
  if (trace) { PG_RANK_TRY(hipStreamSynchronize(stream), "sort"); }
  ms_sort = ms_since(t_sort);
  // ---- 3. every dictionary entry leaves its rank in its slot; 4. every doc's rank, bit-packed ----
  const RankTable table{d_slots, log2_slots, d_flags};
  const auto t_assign = now();
  if (cardinality > 0) rank_hash_assign_kernel<<<dim3((unsigned)std::max(1, std::min((cardinality + 255) / 256, num_cus * 16))), dim3(256), 0, stream>>>(d_dict, cardinality, table);
  PG_RANK_TRY(hipGetLastError(), "assign kernel");
  if (trace) { PG_RANK_TRY(hipStreamSynchronize(stream), "assign"); }
  ms_assign = ms_since(t_assign);
  const auto t_pack = now();
  const size_t image_bytes = (size_t)std::max(num_tiles, 1) * 256 * (size_t)bits + 64;
  PG_RANK_TRY(hipMalloc((void**)&d_image, image_bytes), "image");
  PG_RANK_TRY(hipMemsetAsync(d_image, 0, image_bytes, stream), "image clear");
  if (num_docs > 0) {
    rank_image_pack_kernel<<<dim3((unsigned)std::max(1, std::min(num_tiles / 4 + 1, num_cus * 8))), dim3(256), 0, stream>>>(d_raw, vkind, table, cardinality, d_image, bits, num_tiles, num_docs);
    PG_RANK_TRY(hipGetLastError(), "pack kernel");
  }
  PG_RANK_TRY(hipStreamSynchronize(stream), "pack");
  ms_pack = ms_since(t_pack);
  if (trace)
    fprintf(stderr, "[rank image] docs %lld vkind %d cardinality %d bits %d | transient bytes: hash table %zu (2^%d slots, attempt %d) + distinct images 2 x %zu + sort scratch %zu | kept: dictionary %zu + image %zu | ms: insert %.3f collect %.3f sort %.3f assign %.3f pack %.3f total %.3f\n",
            num_docs, vkind, cardinality, bits, table_bytes, log2_slots, attempts, (size_t)std::max(cardinality, 1) * 8, std::max<size_t>(temp_sort, 256),
            (size_t)std::max(cardinality, 1) * 8, image_bytes, ms_insert, ms_collect, ms_sort, ms_assign, ms_pack, ms_since(t_start));
#undef PG_RANK_TRY
  cleanup();
  *out_d_dict = d_dict; *out_image = d_image; *out_image_bytes = image_bytes; *out_bits = bits; *out_cardinality = cardinality;
  return PG_OK;
}

}  // namespace pg
