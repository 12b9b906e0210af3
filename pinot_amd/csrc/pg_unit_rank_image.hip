// Builds the dictionary and rank image of a raw group-by key column (pg_rank_image.h): rocPRIM's radix sort and unique for the dictionary,
// a binary search per doc for the image.  One-time work per column and segment, off the query path after the first GROUP BY on it.
#include "pg_rank_image.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

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

static __global__ __launch_bounds__(256) void rank_image_keys_kernel(const uint8_t* __restrict__ raw, int vkind, long long num_docs, unsigned long long* __restrict__ out) {
  for (long long doc = (long long)blockIdx.x * blockDim.x + threadIdx.x; doc < num_docs; doc += (long long)gridDim.x * blockDim.x) out[doc] = order_image(raw, vkind, doc);
}

// the rank of every doc's value in the sorted dictionary, packed MSB-first at `bits_out` bits per doc in the lane-private tile layout
// (the same writer as build_raw_key_image_kernel: lane l of a tile owns docs [32 l, 32 l + 32), bits_out dwords)
static __global__ __launch_bounds__(256) void rank_image_pack_kernel(const uint8_t* __restrict__ raw, int vkind, const unsigned long long* __restrict__ dict, int cardinality,
                                                              uint8_t* __restrict__ out, int bits_out, int num_tiles, long long num_docs) {
  const int lane = threadIdx.x & 63;
  for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < (long long)num_tiles; tile += (long long)gridDim.x * 4) {
    const long long first = tile * 2048 + (long long)lane * 32;
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + tile * (256ll * bits_out)) + lane * bits_out;
    unsigned long long acc = 0ull;
    int have = 0, k = 0;
    for (int j = 0; j < 32; ++j) {
      const long long doc = first + j;
      uint32_t id = 0u;
      if (doc < num_docs) {
        const unsigned long long key = order_image(raw, vkind, doc);
        int lo = 0, hi = cardinality - 1;                      // the key IS in the dictionary
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (dict[mid] < key) lo = mid + 1; else hi = mid;
        }
        id = (uint32_t)lo;
      }
      acc = (acc << bits_out) | (unsigned long long)id;
      have += bits_out;
      if (have >= 32) { dst[k++] = __builtin_bswap32((uint32_t)(acc >> (have - 32))); have -= 32; }
    }
  }
}

pg_status build_rank_image(const uint8_t* d_raw, int vkind, long long num_docs, int num_tiles, int num_cus, unsigned long long** out_d_dict,
                           std::vector<unsigned long long>* out_h_dict, uint8_t** out_image, size_t* out_image_bytes, int* out_bits, int* out_cardinality,
                           const char** out_error) {
  static thread_local char message[256];
  *out_error = message;
  message[0] = 0;
  unsigned long long *d_in = nullptr, *d_out = nullptr, *d_dict = nullptr;
  void* d_temp = nullptr;
  size_t* d_selected = nullptr;
  uint8_t* d_image = nullptr;
  hipStream_t stream = nullptr;
  pg_status status = PG_OK;
  auto cleanup = [&] {
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_temp) (void)hipFree(d_temp);
    if (d_selected) (void)hipFree(d_selected);
    if (stream) (void)hipStreamDestroy(stream);
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
  const size_t n = (size_t)std::max<long long>(num_docs, 1);
  // PINOT_GPU_RANK_TRACE=1: the phases on the host clock (a stream synchronisation behind each) and the transient allocations, on stderr
  const bool trace = getenv("PINOT_GPU_RANK_TRACE") != nullptr && getenv("PINOT_GPU_RANK_TRACE")[0] == '1';
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
  const auto t_start = now();
  PG_RANK_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "stream");
  PG_RANK_TRY(hipMalloc((void**)&d_in, n * 8), "keys");
  PG_RANK_TRY(hipMalloc((void**)&d_out, n * 8), "sorted keys");
  PG_RANK_TRY(hipMalloc((void**)&d_selected, 8), "count");
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((num_docs + 255) / 256, (long long)num_cus * 16));
  rank_image_keys_kernel<<<dim3(grid), dim3(256), 0, stream>>>(d_raw, vkind, num_docs, d_in);
  PG_RANK_TRY(hipGetLastError(), "keys kernel");
  double ms_alloc = 0, ms_keys = 0, ms_sort = 0, ms_unique = 0, ms_pack = 0;
  if (trace) { ms_alloc = ms_since(t_start); const auto t = now(); PG_RANK_TRY(hipStreamSynchronize(stream), "keys"); ms_keys = ms_since(t); }
  size_t temp_sort = 0, temp_unique = 0;
  PG_RANK_TRY(rocprim::radix_sort_keys(nullptr, temp_sort, d_in, d_out, (size_t)num_docs, 0u, 64u, stream), "sort sizing");
  PG_RANK_TRY(rocprim::unique(nullptr, temp_unique, d_out, d_in, d_selected, (size_t)num_docs, rocprim::equal_to<unsigned long long>(), stream), "unique sizing");
  PG_RANK_TRY(hipMalloc(&d_temp, std::max<size_t>(std::max(temp_sort, temp_unique), 256)), "sort scratch");
  const auto t_sort = now();
  PG_RANK_TRY(rocprim::radix_sort_keys(d_temp, temp_sort, d_in, d_out, (size_t)num_docs, 0u, 64u, stream), "sort");
  if (trace) { PG_RANK_TRY(hipStreamSynchronize(stream), "sort"); ms_sort = ms_since(t_sort); }
  const auto t_unique = now();
  PG_RANK_TRY(rocprim::unique(d_temp, temp_unique, d_out, d_in, d_selected, (size_t)num_docs, rocprim::equal_to<unsigned long long>(), stream), "unique");
  size_t selected_count = 0;
  PG_RANK_TRY(hipMemcpyAsync(&selected_count, d_selected, sizeof(size_t), hipMemcpyDeviceToHost, stream), "count copy");
  PG_RANK_TRY(hipStreamSynchronize(stream), "sort / unique");
  ms_unique = ms_since(t_unique);
  const auto t_pack = now();
  long long selected = num_docs <= 0 ? 0 : (long long)selected_count;
  if (selected >= 0x7FFFFFFEll) {
    snprintf(message, sizeof(message), "rank image: %lld distinct values do not fit the int dictId domain", selected);
    cleanup();
    return PG_ERR_UNSUPPORTED;
  }
  const int cardinality = (int)selected;
  int bits = 1;
  while (bits < 31 && (1ll << bits) < (long long)cardinality) ++bits;      // PinotDataBitSet.getNumBitsPerValue(cardinality - 1)
  PG_RANK_TRY(hipMalloc((void**)&d_dict, (size_t)std::max(cardinality, 1) * 8), "dictionary");
  if (cardinality > 0) PG_RANK_TRY(hipMemcpyAsync(d_dict, d_in, (size_t)cardinality * 8, hipMemcpyDeviceToDevice, stream), "dictionary copy");
  out_h_dict->assign((size_t)cardinality, 0ull);
  if (cardinality > 0) PG_RANK_TRY(hipMemcpyAsync(out_h_dict->data(), d_in, (size_t)cardinality * 8, hipMemcpyDeviceToHost, stream), "dictionary to host");
  const size_t image_bytes = (size_t)std::max(num_tiles, 1) * 256 * (size_t)bits + 64;
  PG_RANK_TRY(hipMalloc((void**)&d_image, image_bytes), "image");
  PG_RANK_TRY(hipMemsetAsync(d_image, 0, image_bytes, stream), "image clear");
  if (num_docs > 0) {
    rank_image_pack_kernel<<<dim3((unsigned)std::max(1, std::min(num_tiles / 4 + 1, num_cus * 8))), dim3(256), 0, stream>>>(d_raw, vkind, d_dict, cardinality, d_image, bits, num_tiles, num_docs);
    PG_RANK_TRY(hipGetLastError(), "pack kernel");
  }
  PG_RANK_TRY(hipStreamSynchronize(stream), "pack");
  ms_pack = ms_since(t_pack);
  if (trace)
    fprintf(stderr, "[rank image] docs %lld vkind %d cardinality %d bits %d | transient bytes: keys %zu + sorted %zu + sort scratch %zu = %zu | kept: dictionary %zu + image %zu | ms: alloc %.3f keys %.3f sort %.3f unique+copy %.3f pack (binary search per doc) %.3f total %.3f\n",
            num_docs, vkind, cardinality, bits, n * 8, n * 8, std::max<size_t>(std::max(temp_sort, temp_unique), 256), n * 16 + std::max<size_t>(std::max(temp_sort, temp_unique), 256),
            (size_t)std::max(cardinality, 1) * 8, image_bytes, ms_alloc, ms_keys, ms_sort, ms_unique, ms_pack, ms_since(t_start));
#undef PG_RANK_TRY
  cleanup();
  *out_d_dict = d_dict; *out_image = d_image; *out_image_bytes = image_bytes; *out_bits = bits; *out_cardinality = cardinality;
  return status;
}

}  // namespace pg
