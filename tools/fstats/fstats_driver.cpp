// CPU test driver for pinot_amd/csrc/pg_filter_stats.h (the engine's numEntriesScannedInFilter replay): the header is plain host C++,
// so tests/test_filter_stats_cpu.py builds this file with g++ (-fsanitize=address,undefined when available) and compares the replay
// with the oracle's restatement of the reference's iterators on random filter trees -- no GPU involved.
#include "../../pinot_amd/csrc/pg_filter_stats.h"
#include "../../pinot_amd/csrc/pg_filter_fsm.h"

extern "C" int64_t fstats_replay(const pg_query* q, int32_t num_docs, const uint64_t* const* leaf_words, int32_t* out_plan, int32_t* out_scan_leaves) {
  int scan_leaves = 0;
  const pg::fstats::Plan plan = pg::fstats::choose_plan(q, &scan_leaves);
  if (out_plan) *out_plan = (int32_t)plan;
  if (out_scan_leaves) *out_scan_leaves = scan_leaves;
  if (q->num_filter_nodes == 0 || pg::fstats::malformed(q)) return 0;
  const size_t words = ((size_t)num_docs + 63) / 64;
  std::vector<pg::fstats::Words> leaves((size_t)q->num_predicates);
  for (int i = 0; i < q->num_predicates; ++i)
    if (leaf_words[i]) leaves[(size_t)i] = std::make_shared<std::vector<uint64_t>>(leaf_words[i], leaf_words[i] + (words ? words : 1));
  return pg::fstats::replay(q, num_docs, leaves);
}

// The same count with the fast paths off (mode 0: the iterator objects) or forced onto `threads` threads in chunks of `chunk_docs` docs (mode 1)
extern "C" int64_t fstats_replay_mode(const pg_query* q, int32_t num_docs, const uint64_t* const* leaf_words, int32_t mode, int32_t chunk_docs, int32_t threads) {
  if (q->num_filter_nodes == 0 || pg::fstats::malformed(q)) return 0;
  const size_t words = ((size_t)num_docs + 63) / 64;
  std::vector<pg::fstats::Words> leaves((size_t)q->num_predicates);
  for (int i = 0; i < q->num_predicates; ++i)
    if (leaf_words[i]) leaves[(size_t)i] = std::make_shared<std::vector<uint64_t>>(leaf_words[i], leaf_words[i] + (words ? words : 1));
  return pg::fstats::replay(q, num_docs, leaves, mode != 0, chunk_docs, threads);
}

// Plan::kLeap2 (an AND of exactly two scan leaves) the way the device counts it: numDocs + the carry-chain count of the docs where the
// leaf that is not scanning gets asked.
extern "C" int64_t fstats_leap2(const uint64_t* a_words, const uint64_t* b_words, int32_t num_docs) {
  return (int64_t)num_docs + pg::fstats::leap2_extra_entries(a_words, b_words, num_docs);
}

// The root AND as a finite-state transducer (pg_filter_fsm.h): -1 when the shape does not compile; else the count by the doc-by-doc walk
// (mode 0), in the device's lane / tile / chain structure (mode 1: fsm_tiles_kernel's table walk) or in fsm_tiles_perm_kernel's byte-function
// arithmetic (mode 2: -1 when the machine has more than four states or inputs).  *out_states = the number of reachable states.
extern "C" int64_t fstats_fsm(const pg_query* q, int32_t num_docs, const uint64_t* const* leaf_words, int32_t mode, int32_t* out_states, int32_t* out_inputs) {
  pg::fstats::Fsm f;
  if (!pg::fstats::compile_fsm(q, &f)) return -1;
  if (out_states) *out_states = f.num_states;
  if (out_inputs) *out_inputs = f.num_inputs;
  std::vector<const uint64_t*> words;
  for (int p : f.input_predicate) words.push_back(leaf_words[p]);
  if (mode == 0) return pg::fstats::fsm_count_sequential(f, words, num_docs);
  // (a machine with a NOT child: the per-doc entries by the chosen walk, the episodes by the device's episode structure)
  int64_t per_doc;
  if (mode == 2) per_doc = pg::fstats::fsm_count_perm(f, words, num_docs);
  else if (mode == 3) per_doc = pg::fstats::fsm_count_perm8(f, words, num_docs);      // (fsm_tiles_perm8_kernel: up to eight states)
  else per_doc = pg::fstats::fsm_count_tiled(f, words, num_docs);
  return per_doc < 0 ? per_doc : per_doc + pg::fstats::fsm_episode_entries_tiled(f, words, num_docs);
}

// Which of the device's walks a root AND takes (tools/kernel_coverage.py picks its machines with this): 0 when the shape does not compile
// or is not a replayed plan; else 1 with the machine's states, inputs and the most entries one doc can cost (the byte-function walks carry
// at most 7 per doc: pg_engine.hip device_fsm_filter_stats).
extern "C" int32_t fstats_fsm_class(const pg_query* q, int32_t* out_states, int32_t* out_inputs, int32_t* out_max_inc) {
  int scan_leaves = 0;
  if (q->num_filter_nodes < 3 || pg::fstats::choose_plan(q, &scan_leaves) != pg::fstats::Plan::kReplay) return 0;
  pg::fstats::Fsm f;
  if (!pg::fstats::compile_fsm(q, &f)) return 0;
  int max_inc = 0;
  for (uint8_t d : f.delta) max_inc = std::max(max_inc, (int)(d >> 4));
  *out_states = f.num_states; *out_inputs = f.num_inputs; *out_max_inc = max_inc;
  return 1;
}
