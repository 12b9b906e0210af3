// pg_device.h -- device-side parameter blocks shared by the kernels and the host engine.
//
// Vocabulary: a *tile* is 2048 consecutive docIds handled by one wavefront in one
// pass; inside a tile, *step* k (0..31) covers docs [64k, 64k+64) and lane i owns doc 64k+i of every
// step, so a per-lane 32-bit register holds the match bits of the lane's 32 docs ("lane mask") and a
// wave ballot of bit k is exactly the 64-bit word k of the tile's docId bitmap.
#pragma once
#include <stdint.h>

namespace pg {

constexpr int kWave = 64;
constexpr int kTileSteps = 32;
constexpr int kTileDocs = kWave * kTileSteps;   // 2048 docs per tile
constexpr int kMaxCols = 8;                     // distinct columns referenced by one query
constexpr int kMaxLeaves = 8;
constexpr int kMaxNodes = 24;
constexpr int kMaxAggCols = 4;                  // distinct aggregated columns
constexpr int kMaxGroupCols = 3;                // ArrayBasedHolder fast paths cover 1..3 keys
constexpr int kMaxGroupAggs = 8;                // distinct (column, SUM|MIN|MAX) pairs of a group-by query
constexpr int kStackDepth = 8;
constexpr int kBlockThreads = 256;           // maximum; launches may use 64 / 128 when LDS is tight

enum LeafKind : int32_t {
  kLeafMatchAll = 0,
  kLeafMatchNone = 1,
  kLeafDictRange = 2,   // (uint32)(dictId - lo) < span
  kLeafDictSet = 3,     // bit dictId of set_words
  kLeafRawRange = 4,    // (uint32)(value - lo) <= span  (signed inclusive range)
  kLeafBitmap = 5       // precomputed docId bitmap (inverted-index postings expanded on device)
};

struct DevColumn {
  const uint8_t* fwd;      // dict: first byte of the packed bit stream; raw: first value byte (after chunk header)
  const int32_t* dict;     // host-order int32 dictionary values (NULL for raw)
  int32_t bits;            // 1..31 for dictionary columns, 32 for raw
  int32_t is_raw;
  int32_t cardinality;
  int32_t dict_bytes;      // cardinality * 4 (buffer-descriptor num_records for the gather)
  int32_t in_filter;       // referenced by a scan leaf
  int32_t in_agg;          // referenced by an aggregation or a group-by key
  int32_t slot_off;        // byte offset of this column's staging slot inside the wave's LDS region
  int32_t is_plane;        // fwd points at the column's VALUE PLANE: bit-packed (value - plane base), same stream format
};

struct DevLeaf {
  int32_t kind;
  int32_t col;             // index into ScanParams.cols
  int32_t exclusive;       // NOT_EQ / NOT_IN
  int32_t lo;
  uint32_t span;
  int32_t set_bytes;
  const uint32_t* set_words;
  const unsigned long long* bitmap;  // kLeafBitmap: doc-order words
};

struct DevNode {
  int32_t op;              // pg_filter_op
  int32_t leaf;
  int32_t num_children;
  int32_t pad;
};

struct DevAggCol {
  int32_t col;             // index into ScanParams.cols
  int32_t need_sum;
  int32_t need_minmax;
  int32_t pad;
};

// One record per workgroup, reduced by finalize_partials.
struct BlockPartial {
  unsigned long long count;
  long long sum[kMaxAggCols];
  int32_t kmin[kMaxAggCols];   // min dictId (dictionary columns: sorted dictionary => monotone) or min raw value
  int32_t kmax[kMaxAggCols];
  unsigned long long cyc[4];   // PG_CFG_PROFILE_WAVES: shader cycles per wave summed: memory wait, filter, aggregate, whole loop
};

struct ScanParams {
  int32_t num_docs;
  int32_t num_tiles;
  int32_t num_cols;
  int32_t num_leaves;
  int32_t num_nodes;
  int32_t num_agg_cols;
  int32_t queue_off;           // byte offset of the wave's gather queue inside its LDS region
  int32_t wave_lds_bytes;      // staging slots (256 * bits + 16 each) + gather queue
  int32_t speculate;           // 1: issue aggregation-column loads together with the filter loads when the last tile matched
  int32_t queue_cap;           // gather-queue capacity in entries (multiple of 64, >= 128)
  int32_t stage_bytes;         // bytes of ONE staging buffer set (all column slots); the wave owns two (double buffering)
  int32_t double_buffer;       // 1: prefetch the wave's next tile into the second staging buffer set
  int32_t profile;             // 1: accumulate s_memtime phase counters into BlockPartial.cyc
  int32_t pad3;
  DevColumn cols[kMaxCols];
  DevLeaf leaves[kMaxLeaves];
  DevNode nodes[kMaxNodes];
  DevAggCol agg_cols[kMaxAggCols];
  unsigned long long* out_bitmap;  // optional doc-order bitmap output (num_tiles * 32 words)
  BlockPartial* partials;          // [gridDim.x]
};

// ---- group-by ----
enum GroupAggKind : int32_t { kGroupSum = 1, kGroupMin = 2, kGroupMax = 3 };

struct DevGroupAgg {
  int32_t col;             // index into cols
  int32_t kind;            // GroupAggKind
};

// Global (and LDS) group table layout, struct-of-arrays per group id g in [0, num_groups):
//   count[g]            : unsigned long long
//   acc[a][g]           : long long   (SUM: exact integer sum; MIN/MAX: key as signed 64-bit)
struct GroupParams {
  ScanParams scan;
  int32_t num_group_cols;
  int32_t num_group_aggs;
  int32_t num_groups;              // product of cardinalities (<= arrayBasedThreshold)
  int32_t use_lds_table;
  int32_t group_cols[kMaxGroupCols];
  int32_t group_mult[kMaxGroupCols];
  DevGroupAgg group_aggs[kMaxGroupAggs];
  unsigned long long* table_count; // [num_groups]
  long long* table_acc;            // [num_group_aggs * num_groups]
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
