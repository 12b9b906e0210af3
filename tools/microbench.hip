// microbench.hip -- MI355X measurements that drive the kernel design (DESIGN.md section "measured constants"):
// streaming-read ceiling, L2-resident dictionary gather rate (dense / sparse), LDS gather and LDS atomic rates.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// 1. streaming read-reduce, 16 B per lane per load
__global__ __launch_bounds__(256) void stream_read(const uint4* __restrict__ src, size_t n16, unsigned long long* out) {
  unsigned long long acc = 0;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    uint4 v = src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x1234567ull) out[0] = acc;
}

// 1b. streaming through LDS-DMA: each wave pulls 4 KiB chunks into LDS and reads them back
__global__ __launch_bounds__(256) void stream_read_dma(const uint8_t* __restrict__ src, size_t nbytes, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* slot = lds + wave * 4096;
  unsigned long long acc = 0;
  size_t nchunks = nbytes / 4096;
  size_t total_waves = (size_t)gridDim.x * 4;
  for (size_t c = (size_t)blockIdx.x * 4 + wave; c < nchunks; c += total_waves) {
    const uint8_t* p = src + c * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(p + j * 1024 + lane * 16), (lds_void_t*)(slot + j * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 v = *reinterpret_cast<const uint4*>(slot + j * 1024 + lane * 16);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x1234567ull) out[0] = acc;
}

// 2. gather from a table in global memory (L2 resident): `active_pct` percent of lanes do a real gather,
// the others present an out-of-range buffer offset (returns 0, no memory access).
template <int kAux>
__global__ __launch_bounds__(256) void gather_global(const int32_t* __restrict__ table, int table_entries, int iters, int active_pct, unsigned long long* out) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, table_entries * 4, 0x00020000);
  uint32_t x = mix32(blockIdx.x * 256u + threadIdx.x + 1u);
  long long acc = 0;
  for (int it = 0; it < iters; it += 8) {
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x = mix32(x + 0x9e3779b9u);
      const uint32_t idx = (uint32_t)(((unsigned long long)x * (unsigned)table_entries) >> 32);
      const bool active = (mix32(x ^ 0x55555555u) % 100u) < (unsigned)active_pct;
      v[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, active ? idx * 4u : 0xFFFFFFFFu, 0, kAux);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  if (acc == 0x1234567ll) out[0] = acc;
}

// 3. gather from a table staged in LDS
__global__ __launch_bounds__(256) void gather_lds(const int32_t* __restrict__ table, int table_entries, int iters, unsigned long long* out) {
  extern __shared__ int32_t ltab[];
  for (int i = threadIdx.x; i < table_entries; i += blockDim.x) ltab[i] = table[i];
  __syncthreads();
  uint32_t x = mix32(blockIdx.x * 256u + threadIdx.x + 1u);
  long long acc = 0;
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x = mix32(x + 0x9e3779b9u);
      const uint32_t idx = (uint32_t)(((unsigned long long)x * (unsigned)table_entries) >> 32);
      acc += ltab[idx];
    }
  }
  if (acc == 0x1234567ll) out[0] = acc;
}


// 5. mixed: every wave streams 4 KiB through LDS-DMA and issues `gathers_per_chunk` dense 64-lane gathers from a
// 400 KB table per chunk (C2b at 10 % selectivity = ~120 gathers per 4 KiB = ~2 instructions), no decode work.
template <int kAux>
__global__ __launch_bounds__(256) void stream_plus_gather(const uint8_t* __restrict__ src, size_t nbytes, const int32_t* __restrict__ table, int table_entries,
                                                          int gather_instrs, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 2 * 4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* slot = lds + wave * 8192;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, table_entries * 4, 0x00020000);
  unsigned long long acc = 0;
  uint32_t x = mix32(blockIdx.x * 256u + threadIdx.x + 1u);
  const size_t nchunks = nbytes / 4096, total_waves = (size_t)gridDim.x * 4;
  size_t c = (size_t)blockIdx.x * 4 + wave;
  int buf = 0;
  if (c < nchunks) {
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + c * 4096 + j * 1024 + lane * 16), (lds_void_t*)(slot + j * 1024), 16, 0, 0);
  }
  for (; c < nchunks; c += total_waves, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const size_t n = c + total_waves;
    if (n < nchunks) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + n * 4096 + j * 1024 + lane * 16), (lds_void_t*)(slot + (buf ^ 1) * 4096 + j * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 v = *reinterpret_cast<const uint4*>(slot + buf * 4096 + j * 1024 + lane * 16);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    for (int g = 0; g < gather_instrs; g += 2) {
      int v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        x = mix32(x + 0x9e3779b9u);
        const uint32_t idx = (uint32_t)(((unsigned long long)x * (unsigned)table_entries) >> 32);
        v[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (g + j < gather_instrs) ? idx * 4u : 0xFFFFFFFFu, 0, kAux);
      }
      acc += (unsigned)(v[0] + v[1]);
    }
  }
  if (acc == 0x1234567ull) out[0] = acc;
}

// 4. LDS atomics on a group table: 64-bit add + 64-bit max + 64-bit count (what group-by does per row)
template <int kOps>
__global__ __launch_bounds__(256) void lds_atomics(int groups, int iters, unsigned long long* out) {
  extern __shared__ unsigned long long tab[];
  for (int i = threadIdx.x; i < groups * 3; i += blockDim.x) tab[i] = 0;
  __syncthreads();
  uint32_t x = mix32(blockIdx.x * 256u + threadIdx.x + 1u);
  for (int it = 0; it < iters; ++it) {
    x = mix32(x + 0x9e3779b9u);
    const uint32_t g = (uint32_t)(((unsigned long long)x * (unsigned)groups) >> 32);
    __hip_atomic_fetch_add(&tab[g], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kOps >= 2) __hip_atomic_fetch_add(&tab[groups + g], (unsigned long long)(x & 0xFFFFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kOps >= 3) __hip_atomic_fetch_max((long long*)&tab[2 * groups + g], (long long)(x >> 12), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (tab[threadIdx.x % (groups * 3)] == 0x1234567ull) out[0] = 1;
}

// 4b. the group-by update mixes: which LDS atomic widths / counts does a row cost?  Cheap LCG keys so the atomics dominate.
//   0: u32 add            1: u64 add           2: u32 add + u64 add + i32 max (count, sum, max)
//   3: u64 add + i32 max (count packed into the sum word)      4: u32 add + i32 max      5: two u32 adds + i32 max
template <int kMode>
__global__ __launch_bounds__(1024) void lds_atomic_mix(int groups, int iters, unsigned long long* out) {
  extern __shared__ unsigned long long tab[];
  for (int i = threadIdx.x; i < groups * 3; i += blockDim.x) tab[i] = 0;
  __syncthreads();
  uint32_t x = blockIdx.x * 1024u + threadIdx.x + 1u;
  uint32_t* t32 = reinterpret_cast<uint32_t*>(tab);
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    const uint32_t g = __umulhi(x, (uint32_t)groups);
    const uint32_t v = x & 0xFFFFFu;
    if (kMode == 0 || kMode == 2 || kMode == 4 || kMode == 5) __hip_atomic_fetch_add(&t32[2 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kMode == 1 || kMode == 2) __hip_atomic_fetch_add(&tab[groups + g], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kMode == 3) __hip_atomic_fetch_add(&tab[groups + g], (1ull << 42) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kMode == 5) __hip_atomic_fetch_add(&t32[2 * (groups + g)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kMode >= 2) __hip_atomic_fetch_max((int32_t*)&t32[2 * (2 * groups + g)], (int32_t)(x >> 12), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (tab[threadIdx.x % (groups * 3)] == 0x1234567ull) out[0] = 1;
}

// 7. Lane-contiguous direct loads: lane i of a wave reads kDw consecutive dwords at byte offset i*4*kDw of a 256*kDw-byte tile
// (what a "lane owns 32 consecutive docs of a kDw-bit column" decode would issue), dwordx4 at a time.  Every byte is used, but
// each wave-level instruction touches 64 separate 16-byte pieces spread over kDw/32 KiB: does the TCP keep the lines until the
// next instruction of the same wave comes for their other bytes?
template <int kDw>
__global__ __launch_bounds__(256) void lane_contiguous_read(const uint8_t* __restrict__ src, size_t num_tiles, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (size_t)gridDim.x * 4;
  uint32_t acc = 0;
  for (size_t t = wave; t < num_tiles; t += waves) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + t * (256 * kDw) + (size_t)lane * 4 * kDw);
    uint32_t d[kDw];
#pragma unroll
    for (int i = 0; i < kDw; ++i) d[i] = p[i];
#pragma unroll
    for (int i = 0; i < kDw; ++i) acc ^= d[i];
  }
  if (acc == 0x12345678u) out[0] = acc;
}


// 8. FETCH_SIZE calibration for sparse access: every lane reads kBytes (8 / 16) at a stride of kStride bytes -- one piece per 128-byte
// line (kStride 128), per 64-byte half (64), or every kStride-th line -- over a buffer far larger than the 256 MiB Infinity Cache.  The
// bytes the kernel asks for are known exactly; what rocprofv3's FETCH_SIZE reports for them is the calibration factor of the walks that
// read a few bytes around each matching doc (agg_sparse_private, scan_sparse_kernel).  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./tools/microbench sparse
template <int kStride, int kBytes>
__global__ __launch_bounds__(256) void sparse_read(const uint8_t* __restrict__ src, size_t pieces, unsigned long long* out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (size_t i = gid; i < pieces; i += total) {
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(src + i * kStride);
    acc ^= p[0];
    if (kBytes == 16) acc ^= p[1];
  }
  if (acc == 0x12345678ull) out[0] = acc;
}

// 6. VALU issue rate of the integer ops the decode loop is made of (is a wave64 op 2 or 4 cycles on a SIMD?)
template <int kOp>
__global__ __launch_bounds__(256) void valu_rate(int iters, unsigned long long* out) {
  uint32_t a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 2654435761u + j * 40503u + blockIdx.x;
  const uint32_t sel = 0x00010203u + (threadIdx.x & 3) * 0x01010101u, sh = threadIdx.x & 7, y = blockIdx.x | 1u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (kOp == 0) a[j] = __builtin_amdgcn_perm(a[j], a[(j + 1) & 7], sel);
        else if (kOp == 1) a[j] = __builtin_amdgcn_ubfe(a[j], sh, 17) + 0u * y;
        else if (kOp == 2) a[j] = a[j] + y;
        else if (kOp == 3) a[j] = (a[j] - y) < 1000u ? a[(j + 1) & 7] : y;
        else if (kOp == 4) a[j] = (a[j] << 1) | ((a[(j + 1) & 7] - y) < 1000u ? 1u : 0u);
        asm volatile("" : "+v"(a[j]));
      }
    }
  }
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) x ^= a[j];
  if (x == 0x12345u) out[0] = x;
}


// 7. cross-lane 32x32 bit-matrix transpose (bitmap leaf): ds_swizzle butterfly vs ds_bpermute (__shfl_xor)
template <int kMode>
__global__ __launch_bounds__(256) void transpose_rate(int iters, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#define STAGE(J, M) { uint32_t pr; if (kMode == 0) pr = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, ((J) << 10) | 0x1F); else pr = (uint32_t)__shfl_xor((int)x, J, 64); \
      const bool lo_role = (lane & (J)) == 0; const uint32_t a = lo_role ? x : pr; const uint32_t b = lo_role ? pr : x; \
      const uint32_t t = ((a >> (J)) ^ b) & (M); x ^= lo_role ? (t << (J)) : t; }
    STAGE(16, 0x0000FFFFu) STAGE(8, 0x00FF00FFu) STAGE(4, 0x0F0F0F0Fu) STAGE(2, 0x33333333u) STAGE(1, 0x55555555u)
#undef STAGE
    acc += x;
    x += it;
  }
  if (acc == 0x12345u) out[0] = acc;
}

template <typename F>
double time_ms(F launch, int reps) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  CHECK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char** argv) {
  const bool only_atomics = argc > 1 && !strcmp(argv[1], "atomics");
  const bool only_lane = argc > 1 && !strcmp(argv[1], "lane");
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"hbm_gb\": %.1f, \"lds_per_block\": %zu}\n", prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate / 1000, prop.totalGlobalMem / 1e9, prop.sharedMemPerBlock);
  const int cus = prop.multiProcessorCount;
  unsigned long long* d_out;
  CHECK(hipMalloc((void**)&d_out, 64));

  if (argc > 1 && !strcmp(argv[1], "sparse")) {
    const size_t bytes = 4ull << 30;
    uint8_t* d_buf;
    CHECK(hipMalloc((void**)&d_buf, bytes));
    CHECK(hipMemset(d_buf, 0x5a, bytes));
    const int blocks = cus * 8;
    // (one launch each, so that a per-kernel FETCH_SIZE is one dispatch; the names carry the pattern)
    double a = time_ms([&] { sparse_read<128, 8><<<blocks, 256>>>(d_buf, bytes / 128, d_out); }, 1);
    double b = time_ms([&] { sparse_read<64, 8><<<blocks, 256>>>(d_buf, bytes / 64, d_out); }, 1);
    double c = time_ms([&] { sparse_read<256, 8><<<blocks, 256>>>(d_buf, bytes / 256, d_out); }, 1);
    double d = time_ms([&] { sparse_read<512, 16><<<blocks, 256>>>(d_buf, bytes / 512, d_out); }, 1);
    double e = time_ms([&] { stream_read<<<blocks, 256>>>((const uint4*)d_buf, bytes / 16, d_out); }, 1);
    printf("{\"bench\": \"sparse_read\", \"buffer_bytes\": %zu, \"ms\": {\"8B_per_128B_line\": %.3f, \"8B_per_64B_half\": %.3f, \"8B_per_256B\": %.3f, \"16B_per_512B\": %.3f, \"stream_16B_per_lane\": %.3f},"
           " \"lines_touched\": {\"8B_per_128B_line\": %zu, \"8B_per_64B_half\": %zu, \"8B_per_256B\": %zu, \"16B_per_512B\": %zu, \"stream_16B_per_lane\": %zu}}\n",
           bytes, a, b, c, d, e, bytes / 128, bytes / 128, bytes / 256, bytes / 512, bytes / 128);
    return 0;
  }

  if (only_lane) {
    const size_t bytes = 4ull << 30;
    uint8_t* d_buf;
    CHECK(hipMalloc((void**)&d_buf, bytes));
    CHECK(hipMemset(d_buf, 0x5a, bytes));
    for (int bpc : {2, 4, 8}) {
      double m10 = time_ms([&] { lane_contiguous_read<10><<<cus * bpc, 256>>>(d_buf, bytes / (256 * 10), d_out); }, 3);
      double m16 = time_ms([&] { lane_contiguous_read<16><<<cus * bpc, 256>>>(d_buf, bytes / (256 * 16), d_out); }, 3);
      double m17 = time_ms([&] { lane_contiguous_read<17><<<cus * bpc, 256>>>(d_buf, bytes / (256 * 17), d_out); }, 3);
      double m20 = time_ms([&] { lane_contiguous_read<20><<<cus * bpc, 256>>>(d_buf, bytes / (256 * 20), d_out); }, 3);
      double m4 = time_ms([&] { lane_contiguous_read<4><<<cus * bpc, 256>>>(d_buf, bytes / (256 * 4), d_out); }, 3);
      printf("{\"bench\": \"lane_contiguous_read\", \"blocks_per_cu\": %d, \"GBps\": {\"4dw\": %.0f, \"10dw\": %.0f, \"16dw\": %.0f, \"17dw\": %.0f, \"20dw\": %.0f}}\n", bpc,
             bytes / m4 / 1e6, bytes / m10 / 1e6, bytes / m16 / 1e6, bytes / m17 / 1e6, bytes / m20 / 1e6);
    }
    return 0;
  }

  if (only_atomics) {
    for (int groups : {1000}) {
      for (int threads : {256, 1024}) {
        const int iters = 4096, blocks = cus * (1024 / threads);
        double m[6];
        m[0] = time_ms([&] { lds_atomic_mix<0><<<blocks, threads, groups * 24>>>(groups, iters, d_out); }, 3);
        m[1] = time_ms([&] { lds_atomic_mix<1><<<blocks, threads, groups * 24>>>(groups, iters, d_out); }, 3);
        m[2] = time_ms([&] { lds_atomic_mix<2><<<blocks, threads, groups * 24>>>(groups, iters, d_out); }, 3);
        m[3] = time_ms([&] { lds_atomic_mix<3><<<blocks, threads, groups * 24>>>(groups, iters, d_out); }, 3);
        m[4] = time_ms([&] { lds_atomic_mix<4><<<blocks, threads, groups * 24>>>(groups, iters, d_out); }, 3);
        m[5] = time_ms([&] { lds_atomic_mix<5><<<blocks, threads, groups * 24>>>(groups, iters, d_out); }, 3);
        const double rows = (double)blocks * threads * iters;
        printf("{\"bench\": \"lds_atomic_mix\", \"groups\": %d, \"threads\": %d, \"rows_per_s\": {\"u32\": %.3e, \"u64\": %.3e, \"u32+u64+max32\": %.3e, "
               "\"u64packed+max32\": %.3e, \"u32+max32\": %.3e, \"u32+u32+max32\": %.3e}}\n", groups, threads,
               rows / m[0] * 1e3, rows / m[1] * 1e3, rows / m[2] * 1e3, rows / m[3] * 1e3, rows / m[4] * 1e3, rows / m[5] * 1e3);
      }
    }
    return 0;
  }

  // streaming
  const size_t bytes = 4ull << 30;
  uint8_t* d_buf;
  CHECK(hipMalloc((void**)&d_buf, bytes));
  CHECK(hipMemset(d_buf, 0x5a, bytes));
  for (int bpc : {2, 4, 8}) {
    double ms = time_ms([&] { stream_read<<<cus * bpc, 256>>>((const uint4*)d_buf, bytes / 16, d_out); }, 5);
    printf("{\"bench\": \"stream_read_dwordx4\", \"blocks_per_cu\": %d, \"GBps\": %.1f}\n", bpc, bytes / ms / 1e6);
  }
  for (int bpc : {2, 4, 8}) {
    double ms = time_ms([&] { stream_read_dma<<<cus * bpc, 256>>>(d_buf, bytes, d_out); }, 5);
    printf("{\"bench\": \"stream_read_lds_dma\", \"blocks_per_cu\": %d, \"GBps\": %.1f}\n", bpc, bytes / ms / 1e6);
  }

  // gathers
  for (int entries : {4096, 32768, 100000, 1000000}) {
    std::vector<int32_t> h(entries);
    for (int i = 0; i < entries; ++i) h[i] = i * 7 + 3;
    int32_t* d_tab;
    CHECK(hipMalloc((void**)&d_tab, entries * 4));
    CHECK(hipMemcpy(d_tab, h.data(), entries * 4, hipMemcpyHostToDevice));
    for (int pct : {100, 50, 10, 1}) {
      const int iters = 2048, blocks = cus * 8;
      double ms = time_ms([&] { gather_global<0><<<blocks, 256>>>(d_tab, entries, iters, pct, d_out); }, 3);
      const double lanes = (double)blocks * 256 * iters;
      printf("{\"bench\": \"gather_global\", \"table_entries\": %d, \"active_pct\": %d, \"lane_slots_per_s\": %.3e, \"gathers_per_s\": %.3e}\n", entries, pct,
             lanes / ms * 1e3, lanes * pct / 100.0 / ms * 1e3);
    }
    if (entries * 4 <= 128 * 1024) {
      const int iters = 4096, blocks = cus * (entries * 4 <= 64 * 1024 ? 2 : 1);
      CHECK(hipFuncSetAttribute((const void*)gather_lds, hipFuncAttributeMaxDynamicSharedMemorySize, entries * 4));
      double ms = time_ms([&] { gather_lds<<<blocks, 256, entries * 4>>>(d_tab, entries, iters, d_out); }, 3);
      printf("{\"bench\": \"gather_lds\", \"table_entries\": %d, \"blocks\": %d, \"gathers_per_s\": %.3e}\n", entries, blocks, (double)blocks * 256 * iters / ms * 1e3);
    }
    CHECK(hipFree(d_tab));
  }


  // gather cache-policy variants (100000-entry table, dense)
  {
    const int entries = 100000;
    std::vector<int32_t> h(entries);
    for (int i = 0; i < entries; ++i) h[i] = i * 7 + 3;
    int32_t* d_tab;
    CHECK(hipMalloc((void**)&d_tab, entries * 4));
    CHECK(hipMemcpy(d_tab, h.data(), entries * 4, hipMemcpyHostToDevice));
    const int iters = 2048, blocks = cus * 8;
    const double lanes = (double)blocks * 256 * iters;
#define AUXRUN(A) { double ms = time_ms([&] { gather_global<A><<<blocks, 256>>>(d_tab, entries, iters, 100, d_out); }, 3); \
    printf("{\"bench\": \"gather_global_aux\", \"aux\": %d, \"gathers_per_s\": %.3e}\n", A, lanes / ms * 1e3); }
    AUXRUN(0) AUXRUN(1) AUXRUN(2) AUXRUN(3) AUXRUN(16) AUXRUN(17) AUXRUN(18) AUXRUN(19)
    // mixed stream + gather
    for (int bpc : {2, 4}) for (int gi : {0, 1, 2, 4, 8}) {
      double ms = time_ms([&] { stream_plus_gather<0><<<cus * bpc, 256>>>(d_buf, bytes, d_tab, entries, gi, d_out); }, 3);
      printf("{\"bench\": \"stream_plus_gather\", \"blocks_per_cu\": %d, \"gather_instrs_per_4KiB\": %d, \"GBps\": %.1f, \"gathers_per_s\": %.3e}\n", bpc, gi,
             bytes / ms / 1e6, (double)(bytes / 4096) * gi * 64 / ms * 1e3);
    }
    for (int gi : {2, 4}) {
      double ms = time_ms([&] { stream_plus_gather<2><<<cus * 4, 256>>>(d_buf, bytes, d_tab, entries, gi, d_out); }, 3);
      printf("{\"bench\": \"stream_plus_gather_nt\", \"gather_instrs_per_4KiB\": %d, \"GBps\": %.1f}\n", gi, bytes / ms / 1e6);
      ms = time_ms([&] { stream_plus_gather<17><<<cus * 4, 256>>>(d_buf, bytes, d_tab, entries, gi, d_out); }, 3);
      printf("{\"bench\": \"stream_plus_gather_sc0sc1\", \"gather_instrs_per_4KiB\": %d, \"GBps\": %.1f}\n", gi, bytes / ms / 1e6);
    }
    CHECK(hipFree(d_tab));
  }


  // VALU issue rate
  {
    const int iters = 2000;
    const char* names[] = {"v_perm_b32", "v_bfe_u32", "v_add_u32", "sub+cmp+cndmask", "sub+cmp+cndmask+lshl_or"};
    for (int wpe : {1, 2, 4, 8}) {
      const int blocks = cus * wpe;   // one 256-thread block = 1 wave per SIMD
      double ms[5];
      ms[0] = time_ms([&] { valu_rate<0><<<blocks, 256>>>(iters, d_out); }, 2);
      ms[1] = time_ms([&] { valu_rate<1><<<blocks, 256>>>(iters, d_out); }, 2);
      ms[2] = time_ms([&] { valu_rate<2><<<blocks, 256>>>(iters, d_out); }, 2);
      ms[3] = time_ms([&] { valu_rate<3><<<blocks, 256>>>(iters, d_out); }, 2);
      ms[4] = time_ms([&] { valu_rate<4><<<blocks, 256>>>(iters, d_out); }, 2);
      for (int o = 0; o < 5; ++o) {
        const double stmts = (double)iters * 64 * wpe;          // source statements per SIMD
        printf("{\"bench\": \"valu_rate\", \"op\": \"%s\", \"waves_per_simd\": %d, \"ns_per_statement_per_simd\": %.3f, \"cycles_at_2.4GHz\": %.2f}\n",
               names[o], wpe, ms[o] * 1e6 / stmts, ms[o] * 1e6 / stmts * 2.4);
      }
    }
  }


  // bitmap-leaf transpose rate
  for (int wpe : {1, 4}) {
    const int iters = 4000, blocks = cus * wpe;
    double m0 = time_ms([&] { transpose_rate<0><<<blocks, 256>>>(iters, d_out); }, 2);
    double m1 = time_ms([&] { transpose_rate<1><<<blocks, 256>>>(iters, d_out); }, 2);
    printf("{\"bench\": \"bit_transpose_32x32\", \"waves_per_simd\": %d, \"ns_per_transpose_per_wave_swizzle\": %.1f, \"ns_per_transpose_per_wave_bpermute\": %.1f}\n",
           wpe, m0 * 1e6 / iters, m1 * 1e6 / iters);
  }

  // LDS atomics
  for (int groups : {16, 1000, 4000}) {
    const int iters = 4096, blocks = cus * 4;
    double m1 = time_ms([&] { lds_atomics<1><<<blocks, 256, groups * 24>>>(groups, iters, d_out); }, 3);
    double m2 = time_ms([&] { lds_atomics<2><<<blocks, 256, groups * 24>>>(groups, iters, d_out); }, 3);
    double m3 = time_ms([&] { lds_atomics<3><<<blocks, 256, groups * 24>>>(groups, iters, d_out); }, 3);
    const double rows = (double)blocks * 256 * iters;
    printf("{\"bench\": \"lds_atomics_u64\", \"groups\": %d, \"rows_per_s_1op\": %.3e, \"rows_per_s_2op\": %.3e, \"rows_per_s_3op\": %.3e}\n", groups,
           rows / m1 * 1e3, rows / m2 * 1e3, rows / m3 * 1e3);
  }
  return 0;
}
