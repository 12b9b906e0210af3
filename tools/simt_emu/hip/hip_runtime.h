// tools/simt_emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE: stands in for <hip/hip_runtime.h> when a kernel header of pinot_amd/csrc is
// compiled for the HOST by tests/test_fsm_kernels_emulated_cpu.py (g++ -I tools/simt_emu).  Nothing under pinot_amd/ includes this file; the
// product is built by hipcc against the real header.
//
// What it is: a lockstep-free SIMT emulator just large enough for the transducer kernels of pg_fsm_kernels.h.  Every lane of a workgroup is
// an OS thread; a workgroup's threads run the kernel body as an ordinary function; workgroups run one after the other.
//   __shared__            the build script rewrites it to `static` (one workgroup at a time: a function-local static IS the workgroup's LDS);
//                         `extern __shared__ T name[]` becomes `extern T name[]`, defined by the driver
//   __syncthreads()       a barrier over the workgroup's threads
//   wave_barrier          a barrier over the wavefront's 64 threads -- on the device the lanes run in lockstep and the builtin only stops the
//                         compiler from reordering; the kernels put it exactly where one lane reads what another lane wrote
//   __shfl* / ds_bpermute / ballot   every lane posts its value, barrier, reads the other lane's, barrier
//   atomicAdd             __atomic_fetch_add
//   ubfe / perm           the instructions' definitions (V_BFE_U32, V_PERM_B32)
// The point: the kernels' own source -- index arithmetic, barriers, shuffles, tails -- runs on the CPU tier against the oracle, at tile counts
// the GPU tests of a round may not have reached.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <thread>
#include <vector>

namespace simt {

struct Dim3 { unsigned x = 1, y = 1, z = 1; };

struct Wave {
  pthread_barrier_t bar;
  uint64_t slot[64];
};
struct Block {
  pthread_barrier_t bar;
  std::vector<Wave> waves;
};

inline thread_local Dim3 t_thread_idx, t_block_idx;
inline thread_local Wave* t_wave = nullptr;
inline thread_local Block* t_block = nullptr;
inline thread_local int t_lane = 0;
inline Dim3 g_block_dim, g_grid_dim;

// kernel<<<grid, block>>>(...): `body` is the kernel call.  Workgroups one after the other, a thread per lane.
inline void launch(unsigned grid, unsigned block, const std::function<void()>& body) {
  g_block_dim.x = block; g_grid_dim.x = grid;
  const unsigned num_waves = (block + 63) / 64;
  for (unsigned b = 0; b < grid; ++b) {
    Block blk;
    blk.waves.resize(num_waves);
    pthread_barrier_init(&blk.bar, nullptr, block);
    for (unsigned w = 0; w < num_waves; ++w) pthread_barrier_init(&blk.waves[w].bar, nullptr, (block - w * 64) < 64 ? (block - w * 64) : 64);
    std::vector<std::thread> threads;
    threads.reserve(block);
    for (unsigned t = 0; t < block; ++t)
      threads.emplace_back([&, t, b] {
        t_thread_idx.x = t; t_block_idx.x = b;
        t_block = &blk; t_wave = &blk.waves[t / 64]; t_lane = (int)(t % 64);
        body();
      });
    for (auto& th : threads) th.join();
    for (unsigned w = 0; w < num_waves; ++w) pthread_barrier_destroy(&blk.waves[w].bar);
    pthread_barrier_destroy(&blk.bar);
  }
}

inline void wave_sync() { pthread_barrier_wait(&t_wave->bar); }

// every lane posts `v`, then reads lane `from`'s (out of range: its own)
inline uint64_t exchange(uint64_t v, int from) {
  t_wave->slot[t_lane] = v;
  wave_sync();
  const uint64_t got = (from >= 0 && from < 64) ? t_wave->slot[from] : v;
  wave_sync();
  return got;
}

}  // namespace simt

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

#define threadIdx simt::t_thread_idx
#define blockIdx simt::t_block_idx
#define blockDim simt::g_block_dim
#define gridDim simt::g_grid_dim

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

inline void __syncthreads() { pthread_barrier_wait(&simt::t_block->bar); }
inline void __builtin_amdgcn_wave_barrier() { simt::wave_sync(); }

// V_BFE_U32
inline uint32_t __builtin_amdgcn_ubfe(uint32_t v, uint32_t offset, uint32_t width) {
  offset &= 31u; width &= 31u;
  return width == 0 ? 0u : (v >> offset) & ((1u << width) - 1u);
}
// V_PERM_B32: selector byte k of `sel` picks byte k of the result from {s0 : s1} (0..3: s1, 4..7: s0), 0x0c: zero, 0x0d..: 0xff
inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
  const uint64_t both = ((uint64_t)s0 << 32) | s1;
  uint32_t out = 0;
  for (int k = 0; k < 4; ++k) {
    const uint32_t s = (sel >> (8 * k)) & 0xFFu;
    uint32_t byte;
    if (s < 8) byte = (uint32_t)(both >> (8 * s)) & 0xFFu;
    else if (s == 0x0c) byte = 0u;
    else if (s >= 0x0d) byte = 0xFFu;
    else byte = ((both >> (16 * (s - 8) + 15)) & 1u) ? 0xFFu : 0u;      // 8..11: the sign of a 16-bit half
    out |= byte << (8 * k);
  }
  return out;
}
inline int __builtin_amdgcn_ds_bpermute(int byte_addr, int v) { return (int)(uint32_t)simt::exchange((uint32_t)v, (byte_addr >> 2) & 63); }
inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) {
  simt::t_wave->slot[simt::t_lane] = pred ? 1u : 0u;
  simt::wave_sync();
  unsigned long long m = 0;
  const unsigned lanes = (simt::g_block_dim.x - (simt::t_thread_idx.x / 64) * 64) < 64 ? (simt::g_block_dim.x - (simt::t_thread_idx.x / 64) * 64) : 64;
  for (unsigned l = 0; l < lanes; ++l) m |= (unsigned long long)(simt::t_wave->slot[l] & 1u) << l;
  simt::wave_sync();
  return m;
}

template <typename T>
inline T __shfl(T v, int lane) { uint64_t bits = 0; memcpy(&bits, &v, sizeof(T)); bits = simt::exchange(bits, lane & 63); T out; memcpy(&out, &bits, sizeof(T)); return out; }
template <typename T>
inline T __shfl_up(T v, unsigned delta) { uint64_t bits = 0; memcpy(&bits, &v, sizeof(T)); bits = simt::exchange(bits, simt::t_lane - (int)delta); T out; memcpy(&out, &bits, sizeof(T)); return out; }
template <typename T>
inline T __shfl_xor(T v, int mask) { uint64_t bits = 0; memcpy(&bits, &v, sizeof(T)); bits = simt::exchange(bits, simt::t_lane ^ mask); T out; memcpy(&out, &bits, sizeof(T)); return out; }

inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
