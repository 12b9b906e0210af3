// pg_device.h -- device-side parameter blocks shared by the kernels and the host engine.
//
// Vocabulary: a *tile* is 64 * steps consecutive docIds (steps = 32 or 16, chosen per query)
// handled by one wavefront in one pass; inside a tile, *step* k covers docs [64k, 64k+64) and lane i owns doc 64k+i
// of every step, so a per-lane 32-bit register holds the match bits of the lane's docs ("lane mask") and a wave
// ballot of bit k is exactly the 64-bit word k of the tile's docId bitmap.
#pragma once
#include <stdint.h>

namespace pg {

constexpr int kWave = 64;
constexpr int kMaxTileSteps = 32;
constexpr int kMaxTileDocs = kWave * kMaxTileSteps;   // 2048: device buffers are padded to whole 2048-doc tiles
constexpr int kMaxCols = 16;                    // distinct column streams referenced by one query (the reference's nine-column ArrayMapBasedHolder golden
                                                // stages nine key streams + its aggregation planes + its filter columns)
constexpr int kMaxLeaves = 8;
constexpr int kMaxNodes = 24;
constexpr int kMaxAggCols = 4;                  // distinct aggregated columns
constexpr int kMaxGroupCols = 10;               // group-by key columns (InnerSegmentAggregationSingleValueQueriesTest.java:39-41 groups by nine)
constexpr int kMaxHashLevels = 8;               // chained first tables of a key beyond a long (GroupParams.hash_*): each takes >= 1 column, the first key >= 2
constexpr int kMaxGroupAggs = 8;                // distinct (column, SUM|MIN|MAX) pairs of a group-by query
constexpr int kStackDepth = 8;
constexpr int kSetLdsWords = 2048;              // the dictId sets of a filter in LDS (pg_kernels.h stage_filter_sets): 8 KB per workgroup -- one set over a 16-bit dictionary, or several over narrower ones
constexpr int kBlockThreads = 256;              // scan_agg_kernel: at most 4 wavefronts per workgroup
constexpr int kWideBlockThreads = 640;          // scan_simple_kernel / scan_raw_kernel on a segment whose tiles all fit the chip at once: ten wavefronts per workgroup,
                                                // two such workgroups per CU at five waves per SIMD -- 2.5x fewer records to hand to the fold, folded by 640 threads in one round
constexpr int kHistBlockThreads = 1024;         // scan_hist_kernel: 16 wavefronts share one LDS histogram
constexpr unsigned long long kPartialHistAlarm = 1ull;   // BlockPartial.flags
constexpr unsigned long long kPartialStale = 2ull;       // BlockPartial.flags of a FOLDED record: a workgroup's record did not carry this launch's stamp
constexpr int kGroupBlockThreads = 1024;        // scan_group_kernel: up to 16 wavefronts share one LDS group table

enum LeafKind : int32_t {
  kLeafMatchAll = 0,
  kLeafMatchNone = 1,
  kLeafDictRange = 2,   // (uint32)(field - lo) < span   (field = dictId, or value-plane offset)
  kLeafDictSet = 3,     // bit dictId of set_words
  kLeafRawRange = 4,    // (uint32)(value - lo) <= span  (signed inclusive range)
  kLeafBitmap = 5,      // precomputed docId bitmap (inverted-index postings expanded on device)
  kLeafRawRange64 = 6,  // raw LONG column: (uint64)(value - lo64) <= span64
  kLeafRawRangeF64 = 7, // raw DOUBLE column: the same compare on the order-preserving integer image of the value
  kLeafRawRangeF32 = 8, // raw FLOAT column (widened to double first)
  kLeafDocRange = 9     // (uint32)(docId - lo) <= span: sorted-column predicates resolved to a docId range, no column read
};

// How an aggregated column's VALUES are represented on the device.
enum ValueKind : int32_t {
  kValI32 = 0,   // 32-bit integer domain: INT, or a LONG dictionary whose value range fits 31 bits ("offset dictionary":
                 // entries are value - min, the host adds count * min back) -- dictionary gather, value plane, raw INT
  kValI64 = 1,   // 8-byte integers: LONG dictionary with a wide range (64-bit gather), raw LONG
  kValF64 = 2,   // doubles: FLOAT / DOUBLE dictionaries (FLOAT widened at open: 64-bit gather), raw DOUBLE
  kValF32 = 3    // raw FLOAT (big-endian float32 per doc, widened on load)
};

struct DevColumn {
  const uint8_t* fwd;      // dict: first byte of the packed bit stream; raw: first value byte (after chunk header)
  const int32_t* dict;     // host-order int32 dictionary values (NULL for raw / plane)
  int32_t bits;            // 1..31 for packed streams, 32 for raw
  int32_t is_raw;
  int32_t cardinality;
  int32_t dict_bytes;      // cardinality * 4 (buffer-descriptor num_records for the gather)
  int32_t in_filter;       // referenced by a scan leaf
  int32_t in_agg;          // referenced by an aggregation or a group-by key
  int32_t slot_off;        // byte offset of this column's staging slot inside one staging buffer
  int32_t is_plane;        // fwd points at the column's VALUE PLANE: bit-packed (value - plane base), same stream format
  int32_t vkind;           // ValueKind of the values behind `dict` / the raw stream
  int32_t pad;
};

struct DevLeaf {
  int32_t kind;
  int32_t col;             // index into ScanParams.cols
  int32_t exclusive;       // NOT_EQ / NOT_IN
  int32_t lo;
  uint32_t span;
  int32_t set_bytes;
  int32_t lo_hi;           // kLeafRawRange64: high dwords of the 64-bit bound / span
  uint32_t span_hi;
  const uint32_t* set_words;
  const unsigned long long* bitmap;  // kLeafBitmap: doc-order words
  int32_t lds_off;         // kLeafBitmap: byte offset of its 256-byte slot inside one bitmap staging buffer
  int32_t pad;
};

constexpr int32_t kNodeCountEntries = 2; // scan leaf on the root AND chain behind an index-based child: ScanBasedDocIdIterator.applyAnd looks at every doc still standing
constexpr int kNarrowTiles = 4, kNarrowMaxBits = 8, kNarrowStack = 4, kNarrowSingleTiles = 8;      // scan_narrow_kernel / scan_narrow_single_kernel (pg_scan_narrow.h)
constexpr int kSimpleMaxBits = 20;       // scan_simple_kernel: widest column it is instantiated for (pg_scan_simple.h)
constexpr int kSparseTiles = 8;          // scan_sparse_kernel: tiles per wave and iteration (pg_scan_sparse.h)
constexpr int32_t kNodeLeapfrog2 = 4;    // the root AND of exactly two scan leaves: its two masks also drive the leap-frog entry count (leapfrog2_tile)
constexpr int32_t kNodeExitIfZero = 1;   // root AND chain: the tile is finished (mask 0) if the running result is wave-zero

// Host-side plan node (DevColumn / DevLeaf / PlanNode are what the engine reasons with).
struct PlanNode {
  int32_t op;              // pg_filter_op
  int32_t leaf;
  int32_t num_children;
  int32_t flags;
};

// What the kernels read.  The kernel-argument block lives in memory and is read through the scalar cache; with ~100
// SGPRs live the compiler cannot keep it in registers, so every dynamically indexed field access is a dependent
// scalar load (node -> leaf -> column cost three round trips per leaf per tile).  Each filter node is therefore ONE
// self-contained 64-byte record (a single s_load_dwordx16), and staging / aggregation descriptors are only ever
// indexed with compile-time constants.
struct DevNode {
  int32_t op;              // pg_filter_op
  int32_t flags;
  int32_t num_children;
  int32_t kind;            // LeafKind (LEAF nodes)
  int32_t exclusive;
  int32_t lo;
  uint32_t span;
  int32_t bits;            // packed width of the leaf's column
  int32_t slot_off;        // column staging slot (scan leaves)
  int32_t lds_off;         // bitmap staging slot (bitmap leaves)
  int32_t set_bytes;       // kLeafDictSet: bytes of set_words ; kLeafRawRange64: high dword of the span
  int32_t lo_hi;           // kLeafRawRange64: high dword of lo
  const uint8_t* fwd;      // raw-range leaves: first value byte
  const uint32_t* set_words;
};
static_assert(sizeof(DevNode) == 64, "one scalar load per filter node");

struct DevStage {          // one packed column stream to stage per tile
  const uint8_t* fwd;
  int32_t bits;
  int32_t slot_off;
  int32_t in_filter;
  int32_t pad;
};

struct PlanAggCol {
  int32_t col;             // index into the plan's cols
  int32_t need_sum;
  int32_t need_minmax;
  int32_t pad;
};

struct DevAggCol {         // self-contained: no second lookup into a column table
  int32_t need_sum;
  int32_t need_minmax;
  int32_t bits;
  int32_t slot_off;
  int32_t is_raw;
  int32_t is_plane;
  int32_t dict_bytes;
  int32_t vkind;           // ValueKind
  const uint8_t* fwd;
  const int32_t* dict;     // kValI32: int32 entries ; kValI64 / kValF64: 8-byte entries
};

// One record per workgroup, reduced by finalize_partials.
struct BlockPartial {
  unsigned long long count;
  long long sum[kMaxAggCols];
  int32_t kmin[kMaxAggCols];   // min dictId (dictionary columns: sorted dictionary => monotone), plane offset or raw value
  int32_t kmax[kMaxAggCols];
  unsigned long long cyc[4];   // PG_CFG_PROFILE_WAVES: shader cycles per wave summed: memory wait, filter, aggregate, whole loop
  unsigned long long flags;    // OR over the workgroups: kPartialHistAlarm (pg_scan_hist.h) = a histogram counter may have left its field
  unsigned long long entries;  // numEntriesScannedInFilter counted by the kernel: applyAnd entries of kNodeCountEntries leaves / the extra entries of kNodeLeapfrog2
  unsigned long long stamp;    // in-kernel fold: ScanParams.host_seq of the launch that wrote this record (publish_block_partial checks it: a record of
                               // an earlier launch in the same slot turns into kPartialStale -> PG_ERR_INTERNAL, never into a wrong COUNT / SUM)
  // typed columns only (scan_agg_kernel<.., kTyped = true>): double sums, and 64-bit min / max keys (raw LONG value, or the
  // order-preserving integer image of a raw FLOAT / DOUBLE value)
  double fsum[kMaxAggCols];
  long long kmin64[kMaxAggCols];
  long long kmax64[kMaxAggCols];
};

struct ScanParams {
  int32_t num_docs;
  int32_t num_tiles;           // ceil(num_docs / (64 * tile_steps))
  int32_t tile_steps;          // 32 or 16
  int32_t num_cols;
  int32_t num_leaves;
  int32_t num_nodes;
  int32_t num_agg_cols;
  int32_t num_bitmap_leaves;
  int32_t stage_bytes;         // bytes of ONE column staging buffer (all column slots)
  int32_t queue_off;           // byte offset of the wave's gather queue inside its LDS region
  int32_t queue_cap;           // gather-queue capacity in entries (multiple of 64, >= 128)
  int32_t wave_lds_bytes;      // (1 or 2) * stage_bytes + gather queue
  int32_t speculate;           // 1: stage aggregation columns together with the filter columns when the last tile matched
  int32_t double_buffer;       // 1: prefetch the wave's next tile into the second staging buffer set
  int32_t lazy_columns;        // 1: index-driven filter: stage the scan columns only if the bitmap prefix left something
  int32_t lazy_node;           // node index after which the scan columns are staged when lazy_columns
  int32_t profile;             // 1: accumulate s_memtime phase counters into BlockPartial.cyc
  int32_t bitmap_off;          // byte offset of the two (always double-buffered) bitmap staging buffers in the wave's LDS region
  int32_t bitmap_bytes;        // bytes of one bitmap staging buffer (256 per bitmap leaf)
  int32_t num_stage;           // packed column streams to stage
  int32_t hist_slot;           // scan_hist_kernel: index into agg_cols of the column summed through the LDS histogram
  int32_t hist_bins;           //                   its cardinality (counters in the histogram)
  DevStage stage[kMaxCols];
  DevNode nodes[kMaxNodes];
  DevAggCol agg_cols[kMaxAggCols];
  const unsigned long long* bitmaps[kMaxLeaves];   // bitmap leaves, in leaf order
  int32_t bitmap_lds_off[kMaxLeaves];
  const uint32_t* tile_list;       // lane-private kernels: visit only these 2048-doc tiles (index_and_kernel's survivors), or nullptr = all
  const uint32_t* tile_count;      //                       [1] how many of them
  int32_t raw64_coalesced;         // scan_private_typed_kernel: raw 8-byte columns are read 1 KB per instruction across the wave (pg_scan_typed.h)
  int32_t lane_skip;               // lane-private aggregating kernels: a lane whose 32 docs hold no match does not load its value bytes
  unsigned long long* filter_entries;  // [1] numEntriesScannedInFilter of the kNodeCountEntries leaves (lane-private kernels), or nullptr
  unsigned long long* out_bitmap;  // optional doc-order bitmap output (num_tiles * tile_steps words)
  BlockPartial* partials;          // [gridDim.x] (+ 1: the folded record when host_out is null)
  // "Last block done": the workgroup whose arrival completes `done_counter` folds the per-workgroup records inside the scan kernel
  // itself (publish_block_partial in pg_kernels.h) -- no finalize launch follows.  nullptr: the records are left for finalize_partials_kernel.
  uint32_t* done_counter;          // [9 x 32] arrival counters (eight shards + the top one, 128 bytes apart), zero between launches (the folding workgroup resets them)
  struct HostRecord* host_out;     // pinned, device-mapped host record the fold is written to, or nullptr -> partials[gridDim.x]
  unsigned long long host_seq;     // value stored into host_out->seq after the record (the host may poll it instead of synchronising the stream)
  int32_t fold_slots;              // aggregation slots a fold has to reduce (BlockPartial.sum / kmin / kmax [0 .. fold_slots))
  int32_t fold_typed;              // 1: fsum / kmin64 / kmax64 are in use (typed kernels)
  int32_t sparse_lanes;            // lane-private aggregating kernels: a tile in which at most this many lanes hold a match is aggregated by walking
                                   //   the matches (one 8-byte load per matching doc) instead of decoding every lane's 32 values; 0 = never
  int32_t fold_one_counter;        // 1: grids of at most kFoldOneCounterMax workgroups arrive on ONE counter (no shard hand-off: publish_block_partial);
                                   //    0: always eight shard counters + the top one
  uint8_t* leap_tables;            // [tiles] kNodeLeapfrog2: one byte per 2048-doc tile (leapfrog2_tile), chained by the leapfrog2_chain_*_kernels
  // numEntriesScannedInFilter by the transducer pass (pg_filter_fsm.h / pg_fsm_kernels.h): the lane-private filter also leaves every
  // leaf's own match bits behind -- dword tile * 64 + lane of a doc-order bitmap per leaf, in the order the LEAF nodes are evaluated
  // (nullptr: that leaf is not wanted) -- so that the pass does not have to scan the leaves' columns again.
  uint32_t* leaf_out[kMaxLeaves];
  int32_t leaf_out_enabled;
  // numEntriesScannedInFilter walked INSIDE the scan kernel (scan_private_fsm_kernel; pg_filter_fsm.h's transducer, machines of at most
  // four states and four inputs): every leaf's mask of the tile is an input word, the tile's table {entry state -> exit state, entries}
  // goes to fsm_tables[tile * fsm_states + s]; fsm_chain_kernel / fsm_finish_kernel join the tiles behind the scan.  Nothing else set: 0 states.
  uint32_t* fsm_tables;
  int32_t fsm_states, fsm_inputs;
  int8_t fsm_input_of_leaf[kMaxLeaves];      // the transducer's input behind the filter's leaf of that ordinal (-1: none)
  // scan_sparse_kernel walks index_and_kernel's per-window tile masks directly (no tile list, no index_and_finalize_kernel in front of it)
  const struct WindowInfo* sparse_windows;
  int32_t sparse_num_windows;
  int32_t set_leaves_in_lds;       // scan_private_*: 1 = the filter's dictId sets (IN lists) are staged in LDS once per workgroup (pg_kernels.h stage_filter_sets)
  uint8_t fsm_delta[64];                     // [state << 4 | input]: next state | entries << 4 (pg_filter_fsm.h's delta, four input bits wide)
  int32_t lean_kind;               // pg_execute_batch: 0 the general lane-private body, 1 the item has scan_simple_kernel's shape, 2 scan_raw_kernel's, 13 scan_simple_set_kernel's (3..12: pg_engine.hip "kinds of shared launch")
                                   // (scan_lean_batch_kernel runs those at five waves per SIMD)
};

// What a query's scan brings back to the host: the folded record, then a sequence number written after it.
struct HostRecord {
  BlockPartial partial;
  unsigned long long seq;
  long long leap_correction;        // kNodeLeapfrog2: what leapfrog2_chain_kernel found (sum of delta over the tiles entered in state 1) ...
  unsigned long long leap_seq;      // ... and the sequence number it stored after it
  unsigned long long pad[4];
};

// Host-side plan (what lower_filter / execute build before it is flattened into ScanParams).
struct PlanParams {
  int32_t num_cols = 0, num_leaves = 0, num_nodes = 0, num_agg_cols = 0, num_bitmap_leaves = 0;
  int32_t lazy_columns = 0, lazy_node = -1;
  DevColumn cols[kMaxCols];
  DevLeaf leaves[kMaxLeaves];
  PlanNode nodes[kMaxNodes];
  PlanAggCol agg_cols[kMaxAggCols];
};

// ---- group-by ----
enum GroupAggKind : int32_t { kGroupSum = 1, kGroupMin = 2, kGroupMax = 3 };

struct PlanGroupAgg {
  int32_t col;             // index into the plan's cols
  int32_t kind;            // GroupAggKind
};

struct DevGroupAgg {       // self-contained
  int32_t kind;            // GroupAggKind
  int32_t bits;
  int32_t slot_off;
  int32_t is_raw;
  int32_t is_plane;
  int32_t dict_bytes;
  int32_t vkind;           // ValueKind (kValI64 / kValF64: 64-bit dictionary gather, 64-bit integer / double atomic add)
  int32_t pad;
  const uint8_t* fwd;
  const int32_t* dict;
};

struct DevGroupKey {
  int32_t bits;
  int32_t slot_off;
  int32_t mult;
  int32_t pad;
  const uint8_t* fwd;      // the key column's packed dictId stream (group_private_kernel reads it straight from HBM)
};

// Global (and LDS) group table layout, struct-of-arrays per group id g in [0, num_groups):
//   count[g]            : unsigned long long
//   acc[a][g]           : long long   (SUM: exact integer sum; MIN/MAX: key as signed 64-bit)
struct GroupParams {
  ScanParams scan;
  int32_t num_group_cols;
  int32_t num_group_aggs;
  int32_t num_groups;              // product of cardinalities = slots of the direct-indexed table (an int: the IntMapBasedHolder range)
  int32_t use_lds_table;
  int32_t packed_agg;              // >= 0: that SUM slot of the LDS table also carries the group's doc count in its high bits
  int32_t packed_shift;            //       (count << packed_shift) | sum ; no separate count atomic
  int32_t dense_ok;                // 1: every aggregation is in the 32-bit value domain (the dense 16-step path applies)
  int32_t wide_keys;               // 1: num_groups > 2^24, so dictIds / multipliers may not fit the full-rate 24-bit multiply
  int32_t lds_log_replicas;        // group_private_kernel<true>: log2 of the copies of the LDS table a workgroup keeps (lane l uses copy l % R; pg_kernels.h)
  int32_t zero_identity;           // 1: the global table is all-zero before the launch and MIN / MAX reach it as keys whose identity is 0 (group_lds_batch_kernel)
  int32_t set_lds_off;             // group_private_body: byte offset of the filter's dictId-set area (kSetLdsWords words) in the dynamic LDS, behind the table; -1 = none
  DevGroupKey group_keys[kMaxGroupCols];
  DevGroupAgg group_aggs[kMaxGroupAggs];
  unsigned long long* table_count; // [num_groups]
  long long* table_acc;            // [num_group_aggs * num_groups]
  // Raw keys beyond an int (the reference's LongMapBasedHolder / ArrayMapBasedHolder, DictionaryBasedGroupKeyGenerator.java:162-176,
  // 628-806, 808+): the table above is no longer indexed by the raw key but HASHED -- num_groups slots (a power of two, at least twice
  // the keys that can exist), open addressing, linear probing, the 64-bit key of a slot in hash_keys[slot] (kHashEmpty = free).
  //   hash_kind 1: the raw key sum dictId_j * key_mult[j] fits a long and is the key.
  //   hash_kind 2: it does not.  The columns are taken in order while the key still fits a long; at column hash_split[l] the key so far
  //                is handed to first table l (hash_keys_lvl[l]), which turns it into its slot number, and the sum goes on from there:
  //                key = that slot + sum over the next columns dictId_j * key_mult[j] (key_mult carries the table's size).  hash_levels
  //                such tables are chained (one is enough up to ~96 bits of key; nine 31-bit columns need seven).
  int32_t hash_kind;
  int32_t hash_levels;
  unsigned long long hash_mask;     // num_groups - 1
  unsigned long long* hash_keys;    // [num_groups]
  int32_t hash_split[kMaxHashLevels];
  unsigned long long hash_mask_lvl[kMaxHashLevels];     // slots of first table l - 1
  unsigned long long* hash_keys_lvl[kMaxHashLevels];    // [hash_mask_lvl[l] + 1]
  unsigned long long key_mult[kMaxGroupCols];
  uint32_t* first_doc;              // non-null: the numGroupsLimit pass -- no aggregation, atomicMin of the docId into first_doc[slot]
  // group_lds_batch_kernel's items (zero_identity): non-null = the workgroup whose arrival on scan.done_counter completes the item's share of
  // the launch copies the item's table slice (count[G] | acc[NA][G], contiguous from table_count) to this pinned, device-mapped host image,
  // leaves the slice all-zero again and stores scan.host_seq into scan.host_out->seq -- the host converts an item while the launch still
  // works on the others; no copy command and no memset behind the launch.
  unsigned long long* host_table;
};
constexpr unsigned long long kHashEmpty = ~0ull;

// ---- partitioned group-by (pg_group_partition.h) ----
constexpr int kMaxPartitions = 512;
constexpr int kPartitionChunk = 1 << 21;       // most records one pass-B workgroup takes
constexpr int kMaxPartitionAggs = 3;

struct PartitionWork {
  int32_t partition;
  uint32_t start;          // first record of the chunk inside the partition
  uint32_t len;            // records of the chunk (upper bound; clamped by what pass A really wrote)
  uint32_t pad;
};

struct PartitionParams {
  GroupParams gp;
  int32_t shift;                    // log2(slots per partition)
  int32_t num_partitions;
  int32_t packed_bits;              // > 0: ONE 32-bit record per doc = (slot within the partition << packed_bits) | value, value < 2^packed_bits
  int32_t reserved;                 //      (COUNT(*): packed_bits = 1, value 0); part_val is not used
  uint32_t* upper;                  // [P] pass 0 result: docs per partition ignoring the filter
  uint32_t* cursor;                 // [P] records appended by pass A
  const uint32_t* offsets;          // [P + 1] first record of each partition buffer (prefix sum of `upper`)
  uint32_t* part_key;               // [num_docs] raw keys
  uint32_t* part_val[kMaxPartitionAggs];   // [num_docs] 32-bit aggregation inputs (dictIds, plane fields or raw values)
  const PartitionWork* work;        // pass B work list
  const uint32_t* work_count;       // two-level runs: the list was built on the device (group_repartition_plan_kernel) -- workgroups past *work_count leave; else nullptr
};

// ---- two-level partitioning (key spaces of 2 M .. 2^31 raw keys: pg_group_partition.h) ----
// Pass A scatters by COARSE partition (at most kMaxPartitions of them, 2^log2_fine_per_coarse fine partitions each); the records of every
// coarse partition are then scattered once more, by fine partition, into a second buffer -- pass B aggregates fine partitions as before.
constexpr int kMaxFinePerCoarse = 1024;
constexpr uint32_t kRepartitionChunk = 1u << 16;      // records one workgroup of the re-scatter takes (counted, then placed: two reads, the second out of L2)
struct RepartitionParams {
  const uint32_t* src_key;                   // pass A's records, by coarse partition
  const uint32_t* src_val[kMaxPartitionAggs];
  uint32_t* dst_key;                         // the same records, by fine partition
  uint32_t* dst_val[kMaxPartitionAggs];
  const uint32_t* coarse_offsets;            // [P1 + 1] first record of every coarse partition's buffer
  const uint32_t* coarse_cursor;             // [P1] records pass A wrote
  uint32_t* fine_count;                      // [P1 << k] records per fine partition (counted by group_repartition_count_kernel)
  uint32_t* fine_offsets;                    // [P1 << k] first record of every fine partition inside its coarse partition's range of the second buffer
  uint32_t* fine_cursor;                     // [P1 << k] records placed so far
  PartitionWork* work;                       // pass B's work list, built by group_repartition_plan_kernel
  uint32_t* work_count;
  const PartitionWork* chunks;               // the re-scatter's own work list: (coarse partition, start, length) chunks of at most kRepartitionChunk records
  int32_t num_coarse, log2_fine_per_coarse, fine_shift, packed_bits, num_vals;
  uint32_t aggregate_chunk;                  // most records one pass-B workgroup takes
};

// ---- index-only AND (index_and_kernel): the inverted-index children of a root AND, intersected window by window ----
constexpr int kMaxAndChildren = 8;
constexpr int kMaxAndPostings = 64;      // postings of all children together: one directory lookup per lane of the window's wavefront
constexpr int kMaxChildPostings = 16;    // postings OR-ed into one child before it is expanded densely instead (EQ: 1; IN lists, short ranges)

constexpr int kMaxAndGather = 2;              // IndexAndParams.gather_col
constexpr int kAndCardinalityShards = 64;      // IndexAndParams.shards: this many counter lines, 16 words (128 bytes) apart
struct WindowInfo { uint32_t tiles; uint32_t docs; };   // mask of the window's 32 2048-doc tiles that hold a match; matching docs

struct AndChild {
  const uint8_t* inv;                    // the column's inverted-index buffer (serialized RoaringBitmaps), or nullptr for a dense child
  const struct DevContainer* dir;        // the column's parsed container directory
  const unsigned long long* dense;       // dense child: a doc-order bitmap that already exists (long IN lists expanded by roaring_expand_kernel)
  int32_t posting_begin, posting_end;    // its slice of IndexAndParams.first / count
  int32_t exclusive;                     // NOT_EQ / NOT_IN: the complement over [0, numDocs)
  int32_t reserved;
};

struct IndexAndParams {
  int32_t num_children;
  int32_t num_postings;
  int32_t num_docs;
  int32_t sparse_out;                    // store only the 2048-doc tiles that hold a match (see index_and_zero_unlisted_kernel)
  long long num_words;                   // 64-bit words of the output bitmap (2048-doc tiles * 32)
  unsigned long long* out;               // doc-order result; nullptr = only the cardinality is wanted
  struct WindowInfo* window_info;        // [windows]
  // The query's figures out of this kernel: COUNT(*) over an index-only filter (FastFilteredCountOperator.java:66-72) and the aggregation over
  // a handful of survivors per window (AndDocIdSet.java:127-172 + ProjectionOperator) are this kernel and nothing else.  Every wavefront sums
  // its windows' matching docs and, gather_cols > 0, reads the survivors' values itself: bit-packed fields of up to kMaxAndGather columns
  // (dictIds for MIN / MAX, plane fields / arithmetic-progression dictIds for SUM: what scan_sparse_kernel reads).  Two ways out:
  //   shards != nullptr (index_and_kernel, one query): the wavefront adds its totals to counter line (wave & 63) of kAndCardinalityShards
  //     128-byte lines with fire-and-forget device-scope atomics -- word 0 the cardinality, words 1 + 3 a .. per gathered column
  //     {sum, 2^32 - 1 - min key, max key}, all-zero identities; the host copies the lines back and zeroes them behind the answer;
  //   pub.partials != nullptr (index_and_batch_kernel, an item of a batch): ONE BlockPartial per wavefront, published like a scan kernel's
  //     (publish_block_partial: the wavefront whose arrival completes the item's count folds them into the item's pinned host record) --
  //     no copy and no memset per item.  For a single query the fold of ~4 000 records at the kernel's tail cost more than the copy
  //     saves (C5-sparse 58.5 -> 62 us, COUNT 39.5 -> 50 us: profiles/r6/c5_index_and_records_vs_shards.jsonl).
  unsigned long long* shards;
  struct AndPublish {
    uint32_t* done_counter;              // ExecCtx.d_done (zero between launches)
    BlockPartial* partials;              // [grid + kFoldExtraRecords]; nullptr = no record (the kernel leaves a bitmap and window masks)
    struct HostRecord* host_out;
    unsigned long long host_seq;
    int32_t fold_slots, fold_typed, profile, fold_one_counter;
  } pub;
  int32_t gather_cols;
  int32_t num_windows;                   // windows of the segment (index_and_batch_kernel reads it here; index_and_kernel takes it as an argument)
  DevAggCol gather_col[2];
  AndChild child[kMaxAndChildren];
  int32_t first[kMaxAndPostings];        // directory slice [first, first + count) of every posting
  int32_t count[kMaxAndPostings];
  uint8_t posting_child[kMaxAndPostings];
};

// ---- roaring expansion ----
struct DevContainer {
  uint32_t key;            // high 16 bits of the docIds in this container
  uint32_t cardinality;
  uint32_t type;           // 0 array, 1 bitset, 2 run
  uint32_t num_runs;
  uint64_t offset;         // byte offset of the container payload inside the column's inverted-index buffer
};

}  // namespace pg
