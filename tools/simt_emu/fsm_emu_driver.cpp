// tools/simt_emu/fsm_emu_driver.cpp -- TEST INFRASTRUCTURE: the transducer kernels of pinot_amd/csrc/pg_fsm_kernels.h compiled for the HOST and
// run by the thread-per-lane emulator of tools/simt_emu/hip/hip_runtime.h, in the order pg_engine.hip's device_fsm_filter_stats launches them.
// tests/test_fsm_kernels_emulated_cpu.py builds it (the kernel header is copied with `__shared__` rewritten to `static`) and holds the count
// against the oracle's iterator objects: the kernels' own source on the CPU tier -- tails, chunk boundaries (more than 1024 tiles), the episode
// kernels of a NOT child -- without a GPU.  Nothing under pinot_amd/ includes or links this file.
#include "hip/hip_runtime.h"

#include "pg_fsm_kernels_emu.h"                     // generated: pg_fsm_kernels.h with __shared__ -> static

#include "../../pinot_amd/csrc/pg_filter_stats.h"
#include "../../pinot_amd/csrc/pg_filter_fsm.h"

namespace pg {
uint32_t staged[16384];                             // `extern __shared__ uint32_t staged[]` of fsm_finish_kernel / fsm_chunk_states_kernel: 64 KB
}

namespace {

template <int SM>
bool launch_table_walk(unsigned blocks, const pg::FsmParams& fp, int L) {
  using namespace pg;
  // (pg_engine.hip PG_FSM_LAUNCH_L: the instantiations the library holds)
  if (L <= 2 && SM <= 4) simt::launch(blocks, 256, [&] { fsm_tiles_kernel<(SM <= 4 ? SM : 4), 2>(fp); });
  else if (L <= 3) simt::launch(blocks, 256, [&] { fsm_tiles_kernel<SM, 3>(fp); });
  else if (L <= 4) simt::launch(blocks, 256, [&] { fsm_tiles_kernel<SM, 4>(fp); });
  else if (L <= 6) simt::launch(blocks, 256, [&] { fsm_tiles_kernel<SM, 6>(fp); });
  else simt::launch(blocks, 256, [&] { fsm_tiles_kernel<SM, 8>(fp); });
  return true;
}

}  // namespace

// walk: 0 the table walk (fsm_tiles_kernel), 1 the byte-function walk of <= 4 states (fsm_tiles_perm_kernel), 2 of <= 8 (fsm_tiles_perm8_kernel),
// 3 of 9 .. 16 states with episodes (fsm_tile_fns16_kernel + fsm_episode_ranges_kernel<16, 4>).
// -1: the shape does not compile; -2: the walk does not take this machine.  `blocks`: workgroups of the tile kernels (4 wavefronts each).
extern "C" __attribute__((visibility("default"))) int64_t emu_fsm_count(const pg_query* q, int32_t num_docs, const uint64_t* const* leaf_words, int32_t walk, int32_t blocks, int32_t* out_states, int32_t* out_inputs,
                                 int32_t* out_episodes) {
  using namespace pg;
  fstats::Fsm fsm;
  if (!fstats::compile_fsm(q, &fsm)) return -1;
  const int L = fsm.num_inputs, S = fsm.num_states;
  if (out_states) *out_states = S;
  if (out_inputs) *out_inputs = L;
  if (out_episodes) *out_episodes = fsm.has_episodes() ? 1 : 0;
  int max_inc = 0;
  for (uint8_t d : fsm.delta) max_inc = std::max(max_inc, (int)(d >> 4));
  if (walk == 1 && !(S <= 4 && L <= 4 && max_inc <= 7)) return -2;
  if (walk == 2 && !(S <= 8 && L <= 4 && max_inc <= 7)) return -2;
  // walk 3: machines of 9 .. 16 states with episodes -- fsm_tile_fns16_kernel (functions only) + fsm_episode_ranges_kernel<16, 4> counting the entries
  if (walk == 3 && !(S > 8 && S <= 16 && L <= 4)) return -2;
  // (pg_engine.hip: every machine with episodes over at most four inputs takes the function-only tile pass when the byte-function walks are on;
  //  so do machines of 9 .. 16 states without episodes -- count_pass: the range kernel as the counter, one pass with no marks)
  const bool count_pass = walk == 3 && !fsm.has_episodes();
  const bool fns_pass = walk != 0 && S <= 16 && L <= 4 && (fsm.has_episodes() || count_pass);
  const long long tiles = std::max<long long>(1, ((long long)num_docs + 2047) / 2048);
  const long long chunks = (tiles + kFsmChunk - 1) / kFsmChunk;
  const size_t words64 = ((size_t)num_docs + 63) / 64;
  // the leaves' doc-order bitmaps as the scan kernel leaves them: dword tile * 64 + lane, whole tiles
  std::vector<std::vector<uint32_t>> bitmaps((size_t)L, std::vector<uint32_t>((size_t)tiles * 64, 0u));
  for (int i = 0; i < L; ++i) {
    const uint64_t* w = leaf_words[fsm.input_predicate[(size_t)i]];
    for (size_t j = 0; j < (size_t)tiles * 64; ++j) bitmaps[(size_t)i][j] = j / 2 < words64 ? (uint32_t)(w[j / 2] >> (32 * (j & 1))) : 0u;
  }
  std::vector<uint32_t> tables((size_t)tiles * (size_t)S, 0xDEADBEEFu), chunk_tables((size_t)chunks * (size_t)S, 0xDEADBEEFu);
  unsigned long long entries = 0xDEADBEEFull;
  FsmParams fp;
  memset(&fp, 0, sizeof(fp));
  for (int i = 0; i < L; ++i) fp.leaf[i] = bitmaps[(size_t)i].data();
  fp.delta = fsm.delta.data(); fp.tables = tables.data();
  fp.num_inputs = L; fp.num_states = S; fp.num_docs = num_docs; fp.num_tiles = (int32_t)tiles;
  const unsigned nb = (unsigned)std::max(1, blocks);
  std::vector<uint4> lane_fronts(fns_pass ? (size_t)tiles * 64 : 0, uint4{0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu});
  if (fns_pass) {
    fp.lane_front = reinterpret_cast<uint32_t*>(lane_fronts.data());
    if (S <= 4) { if (L <= 2) simt::launch(nb, 256, [&] { fsm_tile_fns_kernel<4, 2>(fp); }); else simt::launch(nb, 256, [&] { fsm_tile_fns_kernel<4, 4>(fp); }); }
    else if (S <= 8) { if (L <= 2) simt::launch(nb, 256, [&] { fsm_tile_fns_kernel<8, 2>(fp); }); else simt::launch(nb, 256, [&] { fsm_tile_fns_kernel<8, 4>(fp); }); }
    else if (L <= 3) simt::launch(nb, 256, [&] { fsm_tile_fns_kernel<16, 3>(fp); });
    else simt::launch(nb, 256, [&] { fsm_tile_fns_kernel<16, 4>(fp); });
  } else if (walk == 1) {
    if (L <= 2) simt::launch(nb, 256, [&] { fsm_tiles_perm_kernel<2>(fp); });
    else if (L <= 3) simt::launch(nb, 256, [&] { fsm_tiles_perm_kernel<3>(fp); });
    else simt::launch(nb, 256, [&] { fsm_tiles_perm_kernel<4>(fp); });
  } else if (walk == 2) {
    if (L <= 3) simt::launch(nb, 256, [&] { fsm_tiles_perm8_kernel<3>(fp); });
    else simt::launch(nb, 256, [&] { fsm_tiles_perm8_kernel<4>(fp); });
  } else if (S <= 2) launch_table_walk<2>(nb, fp, L);
  else if (S <= 4) launch_table_walk<4>(nb, fp, L);
  else if (S <= 8) launch_table_walk<8>(nb, fp, L);
  else launch_table_walk<16>(nb, fp, L);
  simt::launch((unsigned)chunks, 1024, [&] { fsm_chain_kernel(tables.data(), tiles, S, chunk_tables.data()); });
  simt::launch(1, 1024, [&] { fsm_finish_kernel(chunk_tables.data(), (int)chunks, S, &entries); });
  unsigned long long episodes = 0;
  if (fsm.has_episodes() || count_pass) {
    std::vector<uint8_t> chunk_state((size_t)chunks, 0xEE), tile_state((size_t)tiles, 0xEE);
    std::vector<int32_t> first_close((size_t)tiles, 12345), last_open((size_t)tiles, 12345);
    simt::launch(1, 1024, [&] { fsm_chunk_states_kernel(chunk_tables.data(), (int)chunks, S, chunk_state.data()); });
    simt::launch((unsigned)chunks, 1024, [&] { fsm_tile_states_kernel(tables.data(), tiles, S, chunk_state.data(), tile_state.data()); });
    // one pass per NOT child over a scan leaf (pg_engine.hip device_fsm_filter_stats): its marks, its pending states, its final-pending flag
    const std::vector<uint8_t> no_marks((size_t)S << L, 0);
    for (int k = 0; k < (count_pass ? 1 : fsm.num_episode_streams()); ++k) {
      int32_t final_pending = 0;
      const uint8_t* const marks = count_pass ? no_marks.data() : fsm.stream_marks(k).data();
      const uint32_t pending_states = count_pass ? 0u : fsm.stream_pending(k);
      std::fill(first_close.begin(), first_close.end(), 12345);
      std::fill(last_open.begin(), last_open.end(), 12345);
      FsmEpisodeParams ep;
      memset(&ep, 0, sizeof(ep));
      for (int i = 0; i < L; ++i) ep.leaf[i] = fp.leaf[i];
      ep.delta = fsm.delta.data(); ep.marks = marks; ep.tile_state = tile_state.data();
      ep.tile_first_close = first_close.data(); ep.tile_last_open = last_open.data();
      ep.episode_entries = &episodes; ep.final_pending = &final_pending;
      ep.pending_states = pending_states;
      ep.num_inputs = L; ep.num_states = S; ep.num_docs = num_docs; ep.num_tiles = (int32_t)tiles;
      if (fns_pass) {
        // the byte-function walks' episode kernel (pg_engine.hip: PINOT_GPU_FSM_PERM on, at most eight states over at most four inputs): a
        // wavefront per contiguous RANGE of tiles, one record per range for the finish kernel
        const long long num_ranges = std::min<long long>(tiles, (long long)nb * 4);
        std::vector<int32_t> range_close((size_t)num_ranges, 12345), range_open((size_t)num_ranges, 12345);
        FsmEpisodeRangeParams rp;
        memset(&rp, 0, sizeof(rp));
        for (int i = 0; i < L; ++i) rp.leaf[i] = fp.leaf[i];
        rp.delta = fsm.delta.data(); rp.marks = marks; rp.tile_state = tile_state.data();
        rp.range_first_close = range_close.data(); rp.range_last_open = range_open.data();
        rp.episode_entries = &episodes; rp.final_pending = &final_pending;
        rp.pending_states = pending_states;
        rp.num_inputs = L; rp.num_states = S; rp.num_docs = num_docs; rp.num_tiles = (int32_t)tiles; rp.num_ranges = (int32_t)num_ranges;
        rp.count_entries = k == 0 ? 1 : 0;
        rp.lane_front = reinterpret_cast<const uint32_t*>(lane_fronts.data());
        const unsigned rb = (unsigned)((num_ranges + 3) / 4);
        if (S > 8) simt::launch(rb, 256, [&] { fsm_episode_ranges_kernel<16, 4>(rp); });
        else if (S <= 4) { if (L <= 2) simt::launch(rb, 256, [&] { fsm_episode_ranges_kernel<4, 2>(rp); }); else simt::launch(rb, 256, [&] { fsm_episode_ranges_kernel<4, 4>(rp); }); }
        else { if (L <= 2) simt::launch(rb, 256, [&] { fsm_episode_ranges_kernel<8, 2>(rp); }); else simt::launch(rb, 256, [&] { fsm_episode_ranges_kernel<8, 4>(rp); }); }
        simt::launch(1, 1024, [&] { fsm_episode_finish_kernel(range_close.data(), range_open.data(), (int)num_ranges, num_docs, &final_pending, &episodes); });
      } else {
        simt::launch(nb, 256, [&] { fsm_episode_tiles_kernel(ep); });
        simt::launch(1, 1024, [&] { fsm_episode_finish_kernel(first_close.data(), last_open.data(), (int)tiles, num_docs, &final_pending, &episodes); });
      }
    }
  }
  return (int64_t)(entries + episodes);
}
