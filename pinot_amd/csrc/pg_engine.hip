// pg_engine.hip -- host side of the C ABI declared in include/pinot_gpu.h: HBM-resident segments,
// query lowering to kernel parameter blocks, launches on per-query HIP streams, partial-result readback.
// Everything that computes runs in the kernels of pg_kernels.h; there is no CPU fallback in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <array>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pinot_gpu.h"
#include "pg_device.h"
#include "pg_filter_stats.h"
#include "pg_filter_fsm.h"
#include "pg_fsm_kernels.h"
#include "pg_kernels.h"
#include "pg_launch.h"
#include "pg_rank_image.h"

namespace {

using namespace pg;

thread_local std::string g_error;

pg_status fail(pg_status st, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return st;
}

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      return fail(_e == hipErrorOutOfMemory ? PG_ERR_OUT_OF_MEMORY : PG_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, \
                  hipGetErrorString(_e), __FILE__, __LINE__);                                          \
  } while (0)

struct Engine {
  bool initialized = false;
  std::atomic<uint64_t> epoch{0};      // moves with every pg_init: what was lowered under other settings is not reused (pg_segment.plan_cache)
  int device = 0;
  int blocks_per_cu = 0;
  bool raw64_coalesced = true;   // PINOT_GPU_RAW64_COALESCED=0: raw LONG / DOUBLE columns are read lane-contiguously (256 bytes per lane)
  int wide_plane = -1;       // PINOT_GPU_WIDE_PLANE: -1 unfiltered sums of 8-byte dictionaries stream a materialised value column, 0 never, 1 always
  bool partition_stats_cache = true;   // PINOT_GPU_PARTITION_STATS_CACHE=0: run the partition histogram pass in every query
  bool partition_packed = true;   // PINOT_GPU_PARTITION_PACKED=0: always separate key / value record columns
  bool plane_async = true;   // PINOT_GPU_PLANE_ASYNC=0: a query that wants a plane waits for its build
  bool staged_h2d = true;    // PINOT_GPU_STAGED_H2D=0: pg_segment_open copies with plain (pageable) hipMemcpy
  long long exact_stats_docs = 64ll << 20;   // PINOT_GPU_EXACT_FILTER_STATS_DOCS: largest segment whose leap-frogging filters are replayed for numEntriesScannedInFilter
  int flags = 0;
  bool use_dma = true;
  int value_plane = -1;      // -1 auto, 0 never, 1 always (PINOT_GPU_VALUE_PLANE)
  int tile_steps = 0;        // 0 auto, 16 or 32 forced (PINOT_GPU_TILE_STEPS)
  bool double_buffer = false;
  bool direct_result = true; // PINOT_GPU_DIRECT_RESULT=0: the folded partial is copied device -> host with a copy command
  int fold_finalize = -1;    // PINOT_GPU_FOLD_FINALIZE: 1 the scan kernel's last workgroup folds the workgroups' records itself, 0 finalize_partials_kernel
                             // does in a launch of its own, -1 (default) = 1: at every size
  int fold_one_counter = 1;  // PINOT_GPU_FOLD_ONE_COUNTER=0: grids of at most 64 workgroups also arrive on eight shard counters + the top one
  bool poll_result = true;   // PINOT_GPU_POLL_RESULT=0: pg_execute always waits with hipStreamSynchronize instead of spinning on the pinned record's sequence number
  bool set_lds = true;       // PINOT_GPU_SET_LDS=0: dictId-set leaves (IN lists) of the lane-private scan kernels read their words from memory per doc (rounds 2-6a)
  bool lane_skip = true;     // PINOT_GPU_LANE_SKIP=0: the aggregating kernels load a tile's value bytes for every lane, matches or not
  bool batch_blocks_per_cu_forced = false;
  int batch_blocks_per_cu = 4;    // PINOT_GPU_BATCH_BLOCKS_PER_CU: workgroups per CU a batch launch is cut into (all items together)
  bool scan_raw = true;      // PINOT_GPU_SCAN_RAW=0: raw INT scans stay in scan_private_kernel / scan_private_typed_kernel (four waves per SIMD)
  bool scan_simple = true;   // PINOT_GPU_SCAN_SIMPLE=0: one-leaf / one-column queries stay in scan_private_kernel (half the waves per SIMD)
  bool scan_sparse = true;   // PINOT_GPU_SCAN_SPARSE=0: index-led aggregations scan their listed tiles in scan_private_kernel (one tile per wave and iteration)
  // (read at every pg_init like the rest: they were function-local statics, fixed at their first use in the process, until the kernel
  //  coverage run of round 5 needed both settings in one process)
  bool lean_batch = true;    // PINOT_GPU_LEAN_BATCH=0: items of scan_simple_kernel's shape share the general batch launch
  bool partition_two_level = true;   // PINOT_GPU_PARTITION_TWO_LEVEL=0: key spaces above one scatter pass keep the direct HBM atomics
  bool fsm_perm = true;      // PINOT_GPU_FSM_PERM=0: the transducer pass always walks tables (fsm_tiles_kernel), never byte functions
  int index_and_waves = 0;    // PINOT_GPU_INDEX_AND_WAVES: -k = index_and_kernel's waves take k windows each (grid = windows / k); n > 0 = a persistent grid of n wavefronts per CU; 0 = of as many as are resident
  bool batch_index = true;   // PINOT_GPU_BATCH_INDEX=0: pg_execute_batch runs index-led items (COUNT(*) over an index-only filter, the gathered aggregation) as launches of their own instead of sharing index_and_batch_kernel's
  bool index_gather = true;  // PINOT_GPU_INDEX_GATHER=0: an index-led aggregation always runs scan_sparse_kernel behind index_and_kernel (never inside it)
  bool fsm_fused = true;     // PINOT_GPU_FSM_FUSED=0: the transducer always runs as a pass of its own behind the scan (leaf bitmaps through HBM)
  bool fsm_stats = true;     // PINOT_GPU_FSM_STATS=0: no transducer pass (host replay / upper bound)
  bool fsm_episodes = true;  // PINOT_GPU_FSM_EPISODES=0: a NOT child over a scan leaf stays with the host replay / upper bound (round 4's behaviour)
  bool plan_cache = true;    // PINOT_GPU_PLAN_CACHE=0: pg_execute_batch lowers every item of every call
  bool group_publish = true; // PINOT_GPU_GROUP_PUBLISH=0: the group-by items of a shared launch come back through one copy of all slices and a memset behind the launch
  bool batch_more = true;    // PINOT_GPU_BATCH_MORE=0: items of scan_narrow_kernel's / scan_private_typed_kernel's shape run their own launches
  bool group_one_launch = true;   // PINOT_GPU_GROUP_ONE_LAUNCH=0: pg_execute runs a small group-by as init + kernel + count + scan + compact launches (rounds 1-4)
  bool batch_group = true;   // PINOT_GPU_BATCH_GROUP=0: group-by items run their own launches on a worker thread
  bool batch_hist = true;    // PINOT_GPU_BATCH_HIST=0: items of scan_hist_kernel's shape run their own launch on a worker thread
  bool batch_launch = true;  // PINOT_GPU_BATCH_LAUNCH=0: pg_execute_batch runs every item as a pg_execute of its own on the worker threads (no shared launch)
  bool leap2 = true;         // PINOT_GPU_LEAP2=0: a leap-frogging `a AND b` is not counted on the device (host replay / upper bound instead)
  int sparse_lanes = 32;     // PINOT_GPU_SPARSE_LANES: tiles in which at most this many of the 64 lanes hold a match are aggregated match by match (0 = never)
  bool plane_gcd = true;     // PINOT_GPU_PLANE_GCD=0: planes hold value - min unscaled, never alias the dictId stream
  bool scan_private = true;  // PINOT_GPU_SCAN_PRIVATE=0: always the LDS-staged scan kernel
  bool group_private = true; // PINOT_GPU_GROUP_PRIVATE=0: unfiltered group-by through the LDS-staged kernel
  int group_log_replicas = 3; // PINOT_GPU_GROUP_REPLICAS=0..4: log2 of the most copies of its LDS table group_private_kernel keeps (as many as fit)
  bool group_pack = true;    // PINOT_GPU_GROUP_PACK=0: separate count atomic in the group-by LDS table
  bool scan_typed_private = true;     // PINOT_GPU_SCAN_TYPED_PRIVATE=0: raw / 8-byte aggregated columns stay in the LDS-staged kernel
  bool scan_narrow_single = true;     // PINOT_GPU_SCAN_NARROW_SINGLE=0: a single narrow leaf takes the general narrow kernel (four tiles per iteration)
  bool scan_narrow = true;            // PINOT_GPU_SCAN_NARROW=0: filters over columns of at most 8 bits stay in scan_private_kernel
  bool group_partition = true;        // PINOT_GPU_GROUP_PARTITION=0: key spaces above the LDS table always use direct HBM atomics
  long long partition_min_docs = 1ll << 22;   // ... =force: partition even tiny segments (tests)
  unsigned long long group_table_bytes = 64ull << 30;      // PINOT_GPU_GROUP_TABLE_BYTES: largest direct-indexed group table a query may ask for
  int hist = -1;             // PINOT_GPU_HIST: -1 auto, 0 never, 1 also for arithmetic-progression dictionaries (tests)
  int hist_blocks = 0;       // PINOT_GPU_HIST_BLOCKS: cap on the histogram kernel's workgroups (tests: many docs per counter from a small segment)
  bool hist_guard = false;   // PINOT_GPU_HIST_GUARD=1: start every column in the guarded tier (tests)
  int hist_bits = 0;         // PINOT_GPU_HIST_BITS: 8 / 16 force narrower counters than the cardinality needs (tests of the guard protocol)
  int group_waves = 0;       // PINOT_GPU_GROUP_WAVES: cap on wavefronts per group-by workgroup (default 16)
  int physical_devices = 1;  // hipGetDeviceCount at pg_init
  int logical_devices = 1;   // PINOT_GPU_ALIAS_DEVICES=N (> physical): device ids 0..N-1 are accepted, id d runs on HIP device d mod physical.  Every
                             // per-device structure (batch contexts, deferred launches, placement) is keyed by the LOGICAL id, so a one-GPU box
                             // executes the multi-device paths of pg_execute_batch with N contexts on one chip.
  std::mutex mu;
};
Engine g_engine;
// the HIP device behind a device id of the C ABI (pg_segment_desc.device_id, pg_config.device_id)
// (ids alias only under PINOT_GPU_ALIAS_DEVICES, i.e. when more ids are accepted than devices exist; callers range-check with device_id_ok)
inline int phys_device(int logical) { return (g_engine.logical_devices > g_engine.physical_devices && g_engine.physical_devices > 0) ? logical % g_engine.physical_devices : logical; }
// a device id the C ABI accepts: 0 .. logical_devices - 1 once pg_init has counted them; before pg_init only what HIP itself has
inline bool device_id_ok(int id) {
  if (id < 0) return false;
  if (g_engine.initialized) return id < g_engine.logical_devices;
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && id < n;
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
inline uint32_t le16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct ColumnDev {
  std::string name;
  int stored_type = 0, encoding = 0, bits = 0, cardinality = 0;
  uint8_t* d_fwd_alloc = nullptr;
  uint8_t* d_fwd = nullptr;      // first value byte
  size_t fwd_alloc_bytes = 0;
  int32_t* d_dict = nullptr;
  std::vector<int32_t> h_dict;   // host-order 32-bit values in the kernels' domain: INT values, or (LONG value - value_base)
  // Stored types other than INT.  vkind says how the kernels see the VALUES of the column:
  //   kValI32  INT; or a LONG dictionary whose range fits 31 bits: h_dict / d_dict hold value - value_base ("offset
  //            dictionary"), every 32-bit path (gather, value plane, plane-evaluated ranges) works unchanged and the host adds
  //            count * value_base to the sums;
  //   kValI64  LONG dictionary with a wider range (d_dict64: int64 entries), raw LONG;
  //   kValF64  FLOAT / DOUBLE dictionary (d_dict64: doubles, FLOAT widened exactly), raw DOUBLE;   kValF32  raw FLOAT.
  int vkind = kValI32;
  int64_t value_base = 0;
  std::vector<double> h_dict_f64;       // every dictionary: (double) of the true value (MIN / MAX / group keys / SPI readers)
  std::vector<int64_t> h_dict_i64;      // INT / LONG dictionaries: the true value
  unsigned long long* d_dict64 = nullptr;
  uint8_t* d_inv = nullptr;
  uint64_t inv_size = 0;
  DevContainer* d_dir = nullptr;
  std::vector<DevContainer> h_dir;
  std::vector<int64_t> posting_first;   // [cardinality + 1] index into the container directory
  std::vector<int64_t> posting_docs;    // [cardinality] docs of every posting list (the planner's estimates: summing 15 259 containers per posting on every query was 8 us of host time each)
  unsigned long long* d_null_bitmap = nullptr;   // null value vector expanded to a doc-order bitmap (num_tiles * 32 words), or nullptr
  int nullkey_column = -1;              // nullable dictionary column: index of its hidden null-key image (dictId = cardinality where the doc is null)
  // Raw INT / LONG column as a group key: index of its hidden KEY IMAGE -- the fixed-bit stream of (value - raw_min), cardinality
  // raw_max - raw_min + 1, built the first time the column is grouped by (ensure_key_image).  -1: not a raw INT / LONG column, or its
  // value range does not fit the int dictId domain.  On the image itself: key_image_of = the raw column, key_base = its raw_min.
  int keyimage_column = -1;
  int64_t raw_min = 0, raw_max = -1;    // raw INT / LONG columns: smallest / largest value (one pass at open)
  int key_image_of = -1;
  int64_t key_base = 0;
  int key_image_state = 0;              // 0 placeholder (no stream yet), 2 built; pg_segment::key_image_mu
  // A raw column whose values do not fit the int dictId domain (FLOAT / DOUBLE, INT / LONG over more than 31 bits) gets a RANK image instead
  // (pg_rank_image.h): a dictionary of its distinct values built on the device the first time it is grouped by, and every doc's rank in it.
  // On the image: rank_image = true; cardinality / bits are 0 until it is built; h_rank_keys / d_rank_dict hold the order images of the values.
  bool rank_image = false;
  std::vector<unsigned long long> h_rank_keys;
  unsigned long long* d_rank_dict = nullptr;
  bool borrows_dictionary = false;      // hidden image: d_dict / d_dict64 belong to the column it was made from
  int64_t num_nulls = 0;
  // value plane (built lazily on the device the first time the column is summed)
  uint8_t* d_plane = nullptr;
  int plane_bits = 0;                   // 1..31 packed, 32 = big-endian int32 values
  int64_t plane_base = 0;               // value = plane_base + plane_scale * decoded field (0 / 1 for plane_bits == 32)
  int64_t plane_scale = 1;              // gcd of (value - min): frame of reference + GCD scaling
  bool plane_is_fwd = false;            // arithmetic-progression dictionary: the dictId stream itself is the plane (d_plane == d_fwd)
  int plane_fwd_published = 0;          // 1 once plane_is_fwd and the plane_* fields above are set (release / acquire): such a plane is never built, budgeted or dropped,
                                        // so queries take it without g_planes.mu (64 items of a batch lowered side by side met on that mutex twice each)
  // what the plane of this column would look like (computed once at open: the gcd walks the whole dictionary)
  int64_t shape_base = 0, shape_scale = 1;
  int shape_bits = 0;
  bool shape_is_fwd = false;
  bool plane_ready = false;
  // residency of a materialised plane (g_planes.mu guards all of these): state, queries reading it right now, LRU stamp, bytes
  int plane_state = 0;                  // 0 none, 1 being built on the segment's plane stream, 2 ready
  int plane_users = 0;
  uint64_t plane_last_use = 0;
  size_t plane_bytes = 0;
  hipEvent_t plane_event = nullptr;     // recorded behind the build
  int32_t* d_plane_fields = nullptr;    // scaled dictionary the build kernel reads (freed when the build is seen complete)
  int hist_tier = 0;                    // histogram SUM of this column: 0 plain counters + checksum, 1 guarded counters (a counter once wrapped), 2 not
                                        // used any more (even a guarded counter ran away: the value plane / gather paths serve the column)
};

// Per-query execution context: a stream plus reusable device scratch.  Pooled per segment so that
// concurrent pg_execute calls on one handle never share mutable state.
struct ExecCtx {
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  BlockPartial* d_partials = nullptr;
  int partial_capacity = 0;
  HostRecord* h_record = nullptr;               // pinned and device-mapped: the folding workgroup (or finalize_partials_kernel) writes the result into it
  HostRecord* h_record_dev = nullptr;           // its device-side address
  BlockPartial* h_partial = nullptr;            // = &h_record->partial
  uint32_t* d_done = nullptr;                   // "last block done" arrival counter of the scan kernels (zero between launches)
  unsigned long long seq = 0;                   // sequence number of the context's last launch (HostRecord.seq)
  bool pre_started = false;                     // timed runs: ev[0] has been recorded (some kernel runs before the scan)
  bool pre_enqueued = false;                    // something has been put on the stream ahead of the scan kernel (any run, timed or not)
  // The transducer pass of numEntriesScannedInFilter (pg_filter_fsm.h): the leaves' bitmaps, the tiles' tables, the episode records.  Per
  // CONTEXT since round 6 -- rounds 4-5 kept one scratch per segment under a mutex held across the whole query, so two leap-frogging
  // queries on one segment ran back to back.  h_fsm_stage: pinned, the machine's tables on their way in and the two counts on their way out.
  uint8_t* d_fsm_scratch = nullptr;
  size_t fsm_scratch_bytes = 0;
  uint8_t* h_fsm_stage = nullptr;
  hipEvent_t ev_pass[2] = {nullptr, nullptr};
  int ev_last = 3;                              // timed runs: the event that closes the query's device work (2 when nothing follows the scan kernel)
  std::vector<unsigned long long*> d_bitmaps;   // each num_tiles*32 words
  std::vector<uint32_t*> d_sets;
  std::vector<size_t> set_capacity;
  unsigned long long* d_table = nullptr;
  size_t table_capacity = 0;                    // in 8-byte words
  unsigned long long* h_table = nullptr;        // pinned
  size_t h_table_capacity = 0;
  int32_t* d_gather_in = nullptr;
  uint8_t* d_gather_out = nullptr;
  size_t gather_capacity = 0;
  uint8_t* d_partition = nullptr;               // partitioned group-by: counters, work list and the record buffers
  size_t partition_capacity = 0;
  uint32_t* d_tile_list = nullptr;              // index_and_kernel: surviving 2048-doc tiles (one entry per tile of the segment)
  unsigned long long* d_and_counters = nullptr; // [0] cardinality (u64), [1] low dword: number of listed tiles; [2 ...] index_and_kernel's counter lines (IndexAndParams.shards)
  bool and_counter_dirty = false;               // the counter lines were added to and not yet zeroed again
  unsigned long long* h_and_shards = nullptr;   // pinned: where the lines land
  uint8_t* d_arena = nullptr;                   // DeviceScratch: per-query device scratch (compaction of group-by results)
  size_t arena_capacity = 0, arena_wanted = 0;
  uint8_t* h_groups = nullptr;                  // pinned staging of many-group results
  size_t h_groups_capacity = 0;
  unsigned long long* d_filter_entries = nullptr;   // kNodeCountEntries leaves: numEntriesScannedInFilter counted by the lane-private kernels
  unsigned long long* h_filter_entries = nullptr;   // pinned copy
  uint8_t* d_leap_tables = nullptr;             // kNodeLeapfrog2: one byte per 2048-doc tile (leapfrog2_tile), then one Leap2Summary per 1024 tiles
  Leap2Summary* d_leap_blocks = nullptr;        //   (leapfrog2_chain_tiles_kernel -> leapfrog2_chain_blocks_kernel)
  size_t leap_capacity = 0;
  WindowInfo* d_window_info = nullptr;          // index_and_kernel: {tile mask, matching docs} of every 65 536-doc window
  size_t tile_list_capacity = 0, window_info_capacity = 0;
};

}  // namespace

struct pg_segment {
  int device = 0;
  int num_docs = 0;
  int num_tiles = 0;
  int num_cus = 256;
  std::atomic<uint64_t> device_bytes{0};      // (grows after open under three different mutexes -- key images, value planes, the transducer's scratch: atomic, so that no update is lost)
  std::string name;
  std::vector<ColumnDev> cols;          // the caller's columns, then hidden images: key images of raw columns (ColumnDev.keyimage_column), null-key images (nullkey_column)
  int num_user_cols = 0;
  std::mutex key_image_mu;              // building a key image (once per raw column)
  std::mutex ctx_mu;
  // Partitioned group-by: docs per partition of a key-column set, ignoring the filter -- a property of the segment, not of the query.
  // Pass 0 (group_partition_histogram_kernel) computes it the first time a (key columns, shift) combination is grouped by; later
  // queries size their record buffers from the cached counts and skip the pass and its round trip to the host.
  std::mutex partition_stats_mu;
  std::vector<std::pair<std::vector<int>, std::vector<uint32_t>>> partition_stats;      // key = {shift, key columns...}
  hipStream_t plane_stream = nullptr;   // value planes are built here, beside the queries
  uint64_t plane_bytes = 0;             // HBM held by materialised value planes (part of device_bytes)
  std::vector<ExecCtx*> free_ctx;
  std::vector<ExecCtx*> all_ctx;
  // the transducer pass of numEntriesScannedInFilter (device_fsm_filter_stats): leaf bitmaps + tables, one query at a time per segment
  // pg_execute_batch: the last few queries lowered for the shared launch (LoweredItem), so that a server that sends the same query to
  // its segments again and again -- one batch per broker request -- does not lower it again.  plane_epoch moves whenever a value plane of
  // this segment is built or dropped: an item lowered before that reads addresses that may be gone.
  std::atomic<uint64_t> plane_epoch{0};
  std::mutex plan_cache_mu;
  std::vector<std::shared_ptr<const struct LoweredItem>> plan_cache;
};

// What a query's scan kernel leaves behind for the transducer pass: the bitmap of every input leaf it evaluated itself.

struct FsmSide {
  const pg::fstats::Fsm* fsm = nullptr;
  uint32_t* bitmap[pg::kFsmInputs] = {};        // where input i's doc-order bitmap goes
  bool mapped[pg::kFsmInputs] = {};             // the lowered filter has a LEAF node of its own for input i
  bool kernel_wrote = false;                    // the kernel that ran carries the store (eval_filter_private, no tile list)
  // the rest of the pass's scratch (prepare_fsm_side), for the walk INSIDE the scan kernel (scan_private_fsm_kernel)
  uint32_t* tables = nullptr;                   // [tiles * S] the tiles' tables
  uint32_t* chunks = nullptr;                   // [chunks * S] fsm_chain_kernel's output
  unsigned long long* entries = nullptr;        // fsm_finish_kernel's output
  long long num_tiles = 0, num_chunks = 0;
  bool fused = false;                           // out: the scan kernel walked the transducer itself; its count is in the context's pinned counter
  bool prepared = false;                        // prepare_fsm_side gave it the context's scratch
};
constexpr size_t kAndShardBytes = (size_t)pg::kAndCardinalityShards * 128;      // ExecCtx.d_and_counters + 2 / h_and_shards
constexpr size_t kFsmStageBytes = 64 + (1 + (size_t)fstats::kFsmMaxEpisodeStreams) * ((size_t)pg::kFsmStates << pg::kFsmInputs);      // ExecCtx.h_fsm_stage: the two counts | delta | marks of every episode stream

namespace {

void destroy_ctx(ExecCtx* c) {
  if (!c) return;
  // (round 5: a COUNT(*) over an index-only filter leaves a memset of its counters on the stream BEHIND the answer -- nothing of the context
  //  may be freed under it)
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->h_fsm_stage) (void)hipHostFree(c->h_fsm_stage);
  if (c->d_fsm_scratch) (void)hipFree(c->d_fsm_scratch);
  for (hipEvent_t e : c->ev_pass) if (e) (void)hipEventDestroy(e);
  if (c->d_partials) (void)hipFree(c->d_partials);
  if (c->h_record) (void)hipHostFree(c->h_record);
  if (c->d_done) (void)hipFree(c->d_done);
  for (auto* b : c->d_bitmaps) (void)hipFree(b);
  for (auto* s : c->d_sets) (void)hipFree(s);
  if (c->d_table) (void)hipFree(c->d_table);
  if (c->h_table) (void)hipHostFree(c->h_table);
  if (c->d_gather_in) (void)hipFree(c->d_gather_in);
  if (c->d_gather_out) (void)hipFree(c->d_gather_out);
  if (c->d_partition) (void)hipFree(c->d_partition);
  if (c->d_tile_list) (void)hipFree(c->d_tile_list);
  if (c->d_and_counters) (void)hipFree(c->d_and_counters);
  if (c->h_and_shards) (void)hipHostFree(c->h_and_shards);
  if (c->d_window_info) (void)hipFree(c->d_window_info);
  if (c->d_arena) (void)hipFree(c->d_arena);
  if (c->h_groups) (void)hipHostFree(c->h_groups);
  if (c->d_filter_entries) (void)hipFree(c->d_filter_entries);
  if (c->d_leap_tables) (void)hipFree(c->d_leap_tables);
  if (c->h_filter_entries) (void)hipHostFree(c->h_filter_entries);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

pg_status acquire_ctx(pg_segment* seg, ExecCtx** out) {
  {
    std::lock_guard<std::mutex> lk(seg->ctx_mu);
    if (!seg->free_ctx.empty()) {
      *out = seg->free_ctx.back();
      seg->free_ctx.pop_back();
      return PG_OK;
    }
  }
  ExecCtx* c = new ExecCtx();
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) for (auto& ev : c->ev) { e = hipEventCreate(&ev); if (e != hipSuccess) break; }
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_record, sizeof(HostRecord), hipHostMallocMapped);
  if (e == hipSuccess) { memset(c->h_record, 0, sizeof(HostRecord)); c->h_partial = &c->h_record->partial; }
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&c->h_record_dev, c->h_record, 0);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_done, 10 * 128);      // fold: eight shards + the top counter; leapfrog2_chain_kernel: one more
  if (e == hipSuccess) e = hipMemsetAsync(c->d_done, 0, 10 * 128, c->stream);      // on the context's OWN stream: ordered before its first kernel (a null-stream memset is not)
  if (e != hipSuccess) {
    destroy_ctx(c);
    return fail(PG_ERR_DEVICE, "creating execution context failed: %s", hipGetErrorString(e));
  }
  {
    std::lock_guard<std::mutex> lk(seg->ctx_mu);
    seg->all_ctx.push_back(c);
  }
  *out = c;
  return PG_OK;
}

void release_ctx(pg_segment* seg, ExecCtx* c) {
  std::lock_guard<std::mutex> lk(seg->ctx_mu);
  seg->free_ctx.push_back(c);
}

struct CtxGuard {
  pg_segment* seg;
  ExecCtx* ctx;
  ~CtxGuard() { if (ctx) release_ctx(seg, ctx); }
};

pg_status ensure_partials(ExecCtx* c, int blocks) {
  if (c->partial_capacity >= blocks + kFoldExtraRecords) return PG_OK;
  if (c->d_partials) (void)hipFree(c->d_partials);
  c->d_partials = nullptr;
  HIP_TRY(hipMalloc((void**)&c->d_partials, sizeof(BlockPartial) * (size_t)(blocks + kFoldExtraRecords)));
  c->partial_capacity = blocks + kFoldExtraRecords;
  return PG_OK;
}

pg_status ensure_bitmap(pg_segment* seg, ExecCtx* c, size_t index) {
  while (c->d_bitmaps.size() <= index) {
    unsigned long long* p = nullptr;
    size_t words = (size_t)std::max(seg->num_tiles, 1) * kMaxTileSteps;
    HIP_TRY(hipMalloc((void**)&p, words * 8));
    c->d_bitmaps.push_back(p);
  }
  return PG_OK;
}

pg_status ensure_set(ExecCtx* c, size_t index, size_t bytes) {
  while (c->d_sets.size() <= index) { c->d_sets.push_back(nullptr); c->set_capacity.push_back(0); }
  if (c->set_capacity[index] < bytes) {
    if (c->d_sets[index]) (void)hipFree(c->d_sets[index]);
    c->d_sets[index] = nullptr;
    size_t cap = std::max<size_t>(bytes, 4096);
    HIP_TRY(hipMalloc((void**)&c->d_sets[index], cap));
    c->set_capacity[index] = cap;
  }
  return PG_OK;
}

constexpr long long kBatchMaxTiles = 32768;       // items of up to 64 Mi docs may share a batch launch (pg_execute_batch); larger ones run their own kernel
constexpr int kMaxGroupSlots = 0x7FFFFFFF;        // raw keys are ints: the reference's ArrayBasedHolder + IntMapBasedHolder range (DictionaryBasedGroupKeyGenerator.java:164-184)
constexpr size_t kGroupTableKeepBytes = 1ull << 31; // a direct-indexed table above this is freed after the query instead of staying with the context
constexpr size_t kArenaKeepBytes = 1ull << 30;      // same for the scratch arena

// hipMalloc'ed scratch of one query, released when it goes out of scope (rare paths only: the hot paths reuse ExecCtx buffers)
// Device scratch of one query, carved out of the context's arena; what does not fit is allocated for the query alone and the arena
// grows before the context's next query, so a steady stream of queries allocates nothing (hipMalloc / hipFree cost ~100 us each and
// synchronise the device: they were most of the host time of a many-group result).
struct DeviceScratch {
  ExecCtx* ctx;
  size_t used = 0, wanted = 0;
  std::vector<void*> overflow;
  explicit DeviceScratch(ExecCtx* c) : ctx(c) {
    if (c->arena_wanted > c->arena_capacity) {
      if (c->d_arena) (void)hipFree(c->d_arena);
      c->d_arena = nullptr; c->arena_capacity = 0;
      const size_t bytes = c->arena_wanted + (c->arena_wanted >> 2);
      if (hipMalloc((void**)&c->d_arena, bytes) == hipSuccess) c->arena_capacity = bytes; else (void)hipGetLastError();
    }
  }
  void* alloc(size_t bytes) {
    const size_t need = (std::max<size_t>(bytes, 8) + 255) & ~(size_t)255;
    wanted += need;
    if (used + need <= ctx->arena_capacity) { void* p = ctx->d_arena + used; used += need; return p; }
    void* p = nullptr;
    if (hipMalloc(&p, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    overflow.push_back(p);
    return p;
  }
  ~DeviceScratch() {
    for (void* p : overflow) (void)hipFree(p);
    ctx->arena_wanted = std::max(ctx->arena_wanted, std::min(wanted, kArenaKeepBytes));      // larger requests stay one-off allocations
  }
};

// Pinned host staging of a many-group result (ids, counts, accumulators): the copies off the device run at link speed and the
// conversion into pg_agg_value reads them in place.
pg_status ensure_host_groups(ExecCtx* c, size_t bytes) {
  if (c->h_groups_capacity >= bytes) return PG_OK;
  if (c->h_groups) (void)hipHostFree(c->h_groups);
  c->h_groups = nullptr; c->h_groups_capacity = 0;
  const size_t want = bytes + (bytes >> 2);
  HIP_TRY(hipHostMalloc((void**)&c->h_groups, want, hipHostMallocDefault));
  c->h_groups_capacity = want;
  return PG_OK;
}

pg_status ensure_table(ExecCtx* c, size_t words, size_t host_words) {
  if (c->table_capacity < words) {
    if (c->d_table) (void)hipFree(c->d_table);
    c->d_table = nullptr; c->table_capacity = 0;
    if (hipMalloc((void**)&c->d_table, words * 8) != hipSuccess) {
      (void)hipGetLastError();
      c->d_table = nullptr;
      return fail(PG_ERR_OUT_OF_MEMORY, "group-by table of %zu bytes does not fit the device", words * 8);
    }
    c->table_capacity = words;
  }
  if (c->h_table_capacity < host_words) {
    if (c->h_table) (void)hipHostFree(c->h_table);
    c->h_table = nullptr;
    HIP_TRY(hipHostMalloc((void**)&c->h_table, host_words * 8, hipHostMallocDefault));
    c->h_table_capacity = host_words;
  }
  return PG_OK;
}

// Parse one serialized RoaringBitmap (public RoaringFormatSpec) into container descriptors whose
// offsets are relative to the start of the column's inverted-index buffer.
pg_status parse_roaring(const uint8_t* base, uint64_t start, uint64_t len, std::vector<DevContainer>* out) {
  if (len == 0) return PG_OK;
  if (len < 8) return fail(PG_ERR_INVALID_ARGUMENT, "inverted index: truncated bitmap");
  const uint8_t* data = base + start;
  uint32_t cookie = le32(data);
  uint32_t n;
  const uint8_t* run_flags = nullptr;
  uint64_t pos;
  bool has_run = false;
  if ((cookie & 0xFFFF) == 12347u) {
    has_run = true;
    n = (cookie >> 16) + 1;
    run_flags = data + 4;
    pos = 4 + (n + 7) / 8;
  } else if (cookie == 12346u) {
    n = le32(data + 4);
    pos = 8;
  } else {
    return fail(PG_ERR_INVALID_ARGUMENT, "inverted index: bad roaring cookie %u", cookie);
  }
  const uint8_t* desc = data + pos;
  pos += (uint64_t)n * 4;
  if (!has_run || n >= 4) pos += (uint64_t)n * 4;
  if (pos > len) return fail(PG_ERR_INVALID_ARGUMENT, "inverted index: truncated roaring header");
  for (uint32_t c = 0; c < n; ++c) {
    DevContainer dc;
    dc.key = le16(desc + 4 * c);
    dc.cardinality = le16(desc + 4 * c + 2) + 1;
    dc.num_runs = 0;
    dc.offset = start + pos;
    bool is_run = has_run && ((run_flags[c >> 3] >> (c & 7)) & 1);
    if (is_run) {
      if (pos + 2 > len) return fail(PG_ERR_INVALID_ARGUMENT, "inverted index: truncated run container");
      dc.type = 2;
      dc.num_runs = le16(data + pos);
      pos += 2 + 4ull * dc.num_runs;
    } else if (dc.cardinality > 4096) {
      dc.type = 1;
      pos += 8192;
    } else {
      dc.type = 0;
      pos += 2ull * dc.cardinality;
    }
    if (pos > len) return fail(PG_ERR_INVALID_ARGUMENT, "inverted index: truncated container");
    out->push_back(dc);
  }
  return PG_OK;
}

// ---- host -> HBM at segment open ----
// The caller's index buffers are pageable (mmap-ed segment files): a plain hipMemcpy stages them through one internal bounce buffer on
// one thread (~16-24 GB/s measured on the bench box).  Large buffers instead go through a small pool of pinned bounce buffers, one
// host thread and one stream per slot: each thread copies a 16 MiB chunk into its pinned buffer while its previous chunk is still on
// the wire, so the CPU-side copies of several threads and the DMA overlap and the link is the limit.  One upload at a time.
constexpr size_t kStageChunk = 16u << 20;
constexpr int kStageSlots = 8;
struct StageSlot { hipStream_t stream = nullptr; uint8_t* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; };
struct StagePool { std::mutex mu; int device = -1; StageSlot slots[kStageSlots]; };
StagePool g_stage;

void destroy_stage_pool_locked() {
  for (auto& sl : g_stage.slots) {
    for (int b = 0; b < 2; ++b) { if (sl.buf[b]) (void)hipHostFree(sl.buf[b]); if (sl.ev[b]) (void)hipEventDestroy(sl.ev[b]); sl.buf[b] = nullptr; sl.ev[b] = nullptr; }
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
    sl.stream = nullptr;
  }
  g_stage.device = -1;
}

hipError_t h2d_copy(void* dst, const void* src, size_t bytes, int device) {
  if (!g_engine.staged_h2d || bytes < 4 * kStageChunk) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
  std::lock_guard<std::mutex> lk(g_stage.mu);
  if (g_stage.device != device) {
    destroy_stage_pool_locked();
    for (auto& sl : g_stage.slots) {
      hipError_t e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
      for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        e = hipHostMalloc((void**)&sl.buf[b], kStageChunk, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev[b], hipEventDisableTiming);
      }
      if (e != hipSuccess) { destroy_stage_pool_locked(); (void)hipGetLastError(); return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice); }
    }
    g_stage.device = device;
  }
  const size_t chunks = (bytes + kStageChunk - 1) / kStageChunk;
  std::atomic<size_t> next{0};
  std::atomic<int> error{(int)hipSuccess};
  std::vector<std::thread> workers;
  const int threads = (int)std::min<size_t>(kStageSlots, chunks);
  for (int t = 0; t < threads; ++t) {
    workers.emplace_back([&, t] {
      StageSlot& sl = g_stage.slots[t];
      hipError_t e = hipSetDevice(phys_device(device));
      int b = 0;
      while (e == hipSuccess) {
        const size_t c = next.fetch_add(1);
        if (c >= chunks) break;
        const size_t off = c * kStageChunk, n = std::min(kStageChunk, bytes - off);
        e = hipEventSynchronize(sl.ev[b]);                     // the chunk this buffer carried last has left it (no-op before its first use)
        if (e != hipSuccess) break;
        memcpy(sl.buf[b], (const uint8_t*)src + off, n);
        e = hipMemcpyAsync((uint8_t*)dst + off, sl.buf[b], n, hipMemcpyHostToDevice, sl.stream);
        if (e == hipSuccess) e = hipEventRecord(sl.ev[b], sl.stream);
        b ^= 1;
      }
      const hipError_t done = hipStreamSynchronize(sl.stream);
      if (e == hipSuccess) e = done;
      if (e != hipSuccess) error.store((int)e);
    });
  }
  for (auto& w : workers) w.join();
  return (hipError_t)error.load();
}

void drop_planes_of(pg_segment* seg);

void free_segment(pg_segment* seg) {
  if (!seg) return;
  drop_planes_of(seg);
  if (seg->plane_stream) (void)hipStreamDestroy(seg->plane_stream);
  for (auto* c : seg->all_ctx) destroy_ctx(c);
  for (auto& col : seg->cols) {
    if (col.d_fwd_alloc) (void)hipFree(col.d_fwd_alloc);
    if (col.d_dict && !col.borrows_dictionary) (void)hipFree(col.d_dict);
    if (col.d_dict64 && !col.borrows_dictionary) (void)hipFree(col.d_dict64);
    if (col.d_inv) (void)hipFree(col.d_inv);
    if (col.d_dir) (void)hipFree(col.d_dir);
    if (col.d_null_bitmap) (void)hipFree(col.d_null_bitmap);
    if (col.d_rank_dict) (void)hipFree(col.d_rank_dict);
    if (col.plane_event) (void)hipEventDestroy(col.plane_event);
  }
  delete seg;
}

// The key image of raw INT / LONG column `c` (ColumnDev.keyimage_column), built on first use: one pass over the resident column on a
// stream of its own (the queries' streams are non-blocking), under the segment's key-image mutex.  *out_column = the image's index.
pg_status ensure_key_image(pg_segment* seg, int c, int* out_column) {
  ColumnDev& col = seg->cols[(size_t)c];
  if (col.keyimage_column < 0)
    return fail(PG_ERR_UNSUPPORTED, "group-by on raw column %s: only INT / LONG columns whose value range fits an int have a key image", col.name.c_str());
  ColumnDev& image = seg->cols[(size_t)col.keyimage_column];
  if (image.rank_image && __atomic_load_n(&image.key_image_state, __ATOMIC_ACQUIRE) != 2) {
    // the column's own dictionary and the docs' ranks in it: one sort + unique + binary-search pass, once per column and segment
    std::lock_guard<std::mutex> lk(seg->key_image_mu);
    if (image.key_image_state == 3) return fail(PG_ERR_UNSUPPORTED, "group-by on raw column %s: its distinct values do not fit the int dictId domain", col.name.c_str());
    if (image.key_image_state != 2) {
      HIP_TRY(hipSetDevice(phys_device(seg->device)));
      uint8_t* d = nullptr;
      size_t bytes = 0;
      int bits = 0, card = 0;
      const char* why = "";
      const pg_status rst = build_rank_image(col.d_fwd, col.vkind, seg->num_docs, seg->num_tiles, seg->num_cus, &image.d_rank_dict, &image.h_rank_keys, &d, &bytes, &bits, &card, &why);
      if (rst == PG_ERR_UNSUPPORTED) { image.key_image_state = 3; return fail(rst, "group-by on raw column %s: %s", col.name.c_str(), why); }
      if (rst != PG_OK) return fail(rst, "group-by on raw column %s: %s", col.name.c_str(), why);
      image.d_fwd_alloc = d; image.d_fwd = d; image.fwd_alloc_bytes = bytes;
      image.bits = bits; image.cardinality = card;
      seg->device_bytes += bytes + (size_t)card * 8;
      __atomic_store_n(&image.key_image_state, 2, __ATOMIC_RELEASE);
    }
  }
  if (__atomic_load_n(&image.key_image_state, __ATOMIC_ACQUIRE) != 2) {
    std::lock_guard<std::mutex> lk(seg->key_image_mu);
    if (image.key_image_state != 2) {
      const size_t bytes = (size_t)std::max(seg->num_tiles, 1) * 256 * (size_t)image.bits + 64;
      uint8_t* d = nullptr;
      hipStream_t stream = nullptr;
      hipError_t e = hipMalloc((void**)&d, bytes);
      if (e != hipSuccess) return fail(PG_ERR_OUT_OF_MEMORY, "key image of column %s: hipMalloc(%zu): %s", col.name.c_str(), bytes, hipGetErrorString(e));
      e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipMemsetAsync(d, 0, bytes, stream);
      if (e == hipSuccess) {
        build_raw_key_image_kernel<<<dim3((unsigned)std::max(1, std::min(seg->num_tiles / 4 + 1, seg->num_cus * 8))), dim3(256), 0, stream>>>(col.d_fwd, col.vkind == kValI32 ? 4 : 8, image.key_base, d,
                                                                                                                                               image.bits, seg->num_tiles, seg->num_docs);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (stream) (void)hipStreamDestroy(stream);
      if (e != hipSuccess) { (void)hipFree(d); return fail(PG_ERR_DEVICE, "key image of column %s: %s", col.name.c_str(), hipGetErrorString(e)); }
      image.d_fwd_alloc = d; image.d_fwd = d; image.fwd_alloc_bytes = bytes;
      seg->device_bytes += bytes;
      __atomic_store_n(&image.key_image_state, 2, __ATOMIC_RELEASE);
    }
  }
  if (out_column) *out_column = col.keyimage_column;
  return PG_OK;
}

// ---- value planes ----
// Policy: summing a dictionary column through its dictionary costs one L2 gather per matching row; a gather costs the
// L2 as much as streaming ~22 bytes, so unless the dictionary is tiny (L1-resident) or the plane would be much wider
// than the dictId stream, the plane wins as soon as a few percent of the rows match.
// Frame of reference + GCD: field = (value - min) / g with g = gcd of all (value - min); w = bits of the largest field.
// When the dictionary is an arithmetic progression (field[d] == d: dense ids, fixed-step values) the dictId stream already IS
// the plane and nothing is materialised.  PINOT_GPU_PLANE_GCD=0 keeps g = 1.
struct PlaneShape { int64_t base; int64_t scale; int bits; bool is_fwd; };
PlaneShape plane_shape(const ColumnDev& col) { return PlaneShape{col.shape_base, col.shape_scale, col.shape_bits, col.shape_is_fwd}; }
PlaneShape compute_plane_shape(const ColumnDev& col) {
  PlaneShape ps;
  const int64_t lo = col.h_dict.front(), hi = col.h_dict.back();
  int64_t g = 0;
  if (g_engine.plane_gcd) {
    for (size_t d = 1; d < col.h_dict.size() && g != 1; ++d) {
      int64_t a = (int64_t)col.h_dict[d] - lo, b = g;
      while (b) { const int64_t t = a % b; a = b; b = t; }
      g = a;
    }
  }
  if (g <= 0) g = 1;
  const int64_t top = (hi - lo) / g;
  int w = 1;
  while (w < 32 && (top >> w) != 0) ++w;
  ps.base = w == 32 ? 0 : lo;
  ps.scale = w == 32 ? 1 : g;
  ps.bits = w;
  ps.is_fwd = g_engine.plane_gcd && w < 32 && top == (int64_t)col.h_dict.size() - 1;     // sorted distinct multiples: field[d] == d
  if (ps.is_fwd) ps.bits = col.bits;
  return ps;
}

bool want_value_plane(const ColumnDev& col) {
  if (col.encoding != PG_FWD_FIXED_BIT_DICT || col.cardinality < 1 || col.vkind != kValI32) return false;
  if (g_engine.value_plane == 0) return false;
  if (g_engine.value_plane == 1) return true;
  const PlaneShape ps = plane_shape(col);
  return ps.is_fwd || col.cardinality > 8192 || ps.bits - col.bits <= 8;
}

// Wide plane: SUM over a LONG / FLOAT / DOUBLE dictionary is one 8-byte L2 gather per doc (2.5e11 /s: 6.6 % of the HBM roofline on an
// unfiltered 250 M-row column).  Materialising the values once -- plane[doc] = dictionary[dictId[doc]], big-endian like a raw column --
// turns the column into a raw 8-byte stream that scan_private_typed_kernel reads with 16-byte loads.  Worth its 8 bytes per doc of
// traffic when most docs match: taken when the query has no filter (PINOT_GPU_WIDE_PLANE=1: always, =0: never) and sums the column
// without taking its MIN / MAX (those run on dictIds); aggregation-only queries.
bool want_wide_plane(const pg_segment* seg, const pg_query* q, int column) {
  if (g_engine.wide_plane == 0 || q->num_group_by > 0 || column < 0 || column >= (int)seg->cols.size()) return false;
  const ColumnDev& col = seg->cols[(size_t)column];
  if (col.encoding != PG_FWD_FIXED_BIT_DICT || col.cardinality < 1 || (col.vkind != kValI64 && col.vkind != kValF64)) return false;
  if (g_engine.wide_plane != 1 && q->num_filter_nodes > 0) return false;
  bool summed = false;
  for (int a = 0; a < q->num_aggregations; ++a) {
    const pg_aggregation& ag = q->aggregations[a];
    if (ag.column != column) continue;
    if (ag.function == PG_AGG_MIN || ag.function == PG_AGG_MAX) return false;
    summed |= ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG;
  }
  return summed;
}

// ---- histogram SUM (pg_scan_hist.h) ----
// SUM(col) = sum_d matches[d] * dictionary[d]: the kernel counts the matching docs per dictId in LDS and never reads a value per
// row.  It reads exactly the dictId stream, so it is preferred over a value plane whenever the histogram fits the CU's LDS --
// except for arithmetic-progression dictionaries, whose dictId stream already is the plane (same bytes, fewer VALU ops per doc).
constexpr size_t kHistLdsBytes = 152 * 1024;      // of 160 KiB: the rest holds the workgroup's reduction records
int hist_counter_bits(const ColumnDev& col) {
  const size_t C = (size_t)std::max(col.cardinality, 1);
  int cw = C * 4 <= kHistLdsBytes ? 32 : (C * 2 <= kHistLdsBytes ? 16 : (C <= kHistLdsBytes ? 8 : 0));
  if (cw > 0 && g_engine.hist_bits > 0) cw = std::min(cw, g_engine.hist_bits);
  return cw;
}
bool want_hist(const ColumnDev& col) {
  if (col.encoding != PG_FWD_FIXED_BIT_DICT || col.cardinality < 1 || col.vkind != kValI32 || col.bits > 18) return false;
  if (g_engine.hist == 0 || __atomic_load_n(&col.hist_tier, __ATOMIC_RELAXED) >= 2) return false;
  if (hist_counter_bits(col) == 0) return false;
  return g_engine.hist == 1 || !plane_shape(col).is_fwd;
}

// ---- value-plane residency ----
// A materialised plane costs about as much HBM as the column itself, so planes live under a budget (pg_config.plane_budget_bytes,
// PINOT_GPU_PLANE_BUDGET_BYTES, pg_set_plane_budget) shared by all segments of the process: a plane is built the first time its
// column is summed, on the segment's own plane stream -- the query that asked does not wait for it, it runs the path that reads the
// dictionary instead (same result), and so does every query until the build is seen complete -- and when the budget is exceeded the
// least recently used planes nobody is reading are released first.  PINOT_GPU_PLANE_ASYNC=0 builds in line (the query waits).
struct PlaneRegistry {
  std::mutex mu;
  std::vector<std::pair<pg_segment*, int>> resident;      // planes in state 1 or 2
  uint64_t total_bytes = 0;
  uint64_t budget_bytes = ~0ull;
  uint64_t tick = 0;
};
PlaneRegistry g_planes;

// g_planes.mu held.  Frees the plane of (seg, column); it must be ready and unused.
void drop_plane_locked(pg_segment* seg, int column) {
  ColumnDev& col = seg->cols[(size_t)column];
  if (col.d_plane && !col.plane_is_fwd) (void)hipFree(col.d_plane);
  col.d_plane = nullptr;
  col.plane_ready = false;
  col.plane_state = 0;
  __atomic_store_n(&col.plane_fwd_published, 0, __ATOMIC_RELEASE);      // (only a closing segment drops a plane that aliases its forward index)
  g_planes.total_bytes -= col.plane_bytes;
  seg->plane_bytes -= col.plane_bytes;
  seg->device_bytes -= col.plane_bytes;
  col.plane_bytes = 0;
  seg->plane_epoch.fetch_add(1, std::memory_order_release);
  auto& r = g_planes.resident;
  r.erase(std::remove(r.begin(), r.end(), std::make_pair(seg, column)), r.end());
}

// Every plane of a segment that is going away (no query is running on it any more).
void drop_planes_of(pg_segment* seg) {
  std::lock_guard<std::mutex> lk(g_planes.mu);
  for (int c = 0; c < (int)seg->cols.size(); ++c) {
    ColumnDev& col = seg->cols[(size_t)c];
    if (col.plane_state == 0) continue;
    if (col.plane_event) (void)hipEventSynchronize(col.plane_event);
    if (col.d_plane_fields) { (void)hipFree(col.d_plane_fields); col.d_plane_fields = nullptr; }
    drop_plane_locked(seg, c);
  }
}

// Asks for the value plane of a column.  *ready: the plane can be read by this query (release_plane when the query is over);
// otherwise the query runs without it.  Never blocks on a build unless PINOT_GPU_PLANE_ASYNC=0.
pg_status acquire_plane(pg_segment* seg, int column, bool* ready) {
  *ready = false;
  ColumnDev& col = seg->cols[(size_t)column];
  if (__atomic_load_n(&col.plane_fwd_published, __ATOMIC_ACQUIRE) != 0) { *ready = true; return PG_OK; }
  std::lock_guard<std::mutex> lk(g_planes.mu);
  if (col.plane_state == 0 && col.vkind != kValI32) {
    // wide plane (want_wide_plane): 8 bytes per doc, padded to whole 2048-doc tiles like a raw column
    const size_t bytes = (size_t)std::max(seg->num_tiles, 1) * 2048 * 8 + 64;
    while (g_planes.total_bytes + bytes > g_planes.budget_bytes) {
      int best = -1;
      for (int i = 0; i < (int)g_planes.resident.size(); ++i) {
        const ColumnDev& c = g_planes.resident[(size_t)i].first->cols[(size_t)g_planes.resident[(size_t)i].second];
        if (c.plane_state != 2 || c.plane_users != 0 || c.plane_is_fwd) continue;
        if (best < 0 || c.plane_last_use < g_planes.resident[(size_t)best].first->cols[(size_t)g_planes.resident[(size_t)best].second].plane_last_use) best = i;
      }
      if (best < 0) return PG_OK;
      drop_plane_locked(g_planes.resident[(size_t)best].first, g_planes.resident[(size_t)best].second);
    }
    HIP_TRY(hipSetDevice(phys_device(seg->device)));
    if (!seg->plane_stream) HIP_TRY(hipStreamCreateWithFlags(&seg->plane_stream, hipStreamNonBlocking));
    if (!col.plane_event) HIP_TRY(hipEventCreateWithFlags(&col.plane_event, hipEventDisableTiming));
    uint8_t* plane = nullptr;
    if (hipMalloc((void**)&plane, bytes) != hipSuccess) { (void)hipGetLastError(); return PG_OK; }
    HIP_TRY(hipMemsetAsync(plane, 0, bytes, seg->plane_stream));
    materialize_wide_plane_kernel<<<dim3((unsigned)std::max(1, seg->num_cus * 8)), dim3(256), 0, seg->plane_stream>>>(col.d_fwd, col.bits, col.d_dict64,
                                                                                                                       reinterpret_cast<unsigned long long*>(plane), seg->num_docs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(col.plane_event, seg->plane_stream));
    col.d_plane = plane;
    col.plane_bytes = bytes;
    col.plane_state = 1;
    g_planes.total_bytes += bytes;
    seg->plane_bytes += bytes;
    seg->device_bytes += bytes;
    g_planes.resident.emplace_back(seg, column);
    seg->plane_epoch.fetch_add(1, std::memory_order_release);
  }
  if (col.plane_state == 0) {
    const PlaneShape ps = plane_shape(col);
    col.plane_bits = ps.bits;
    col.plane_base = ps.base;
    col.plane_scale = ps.scale;
    if (ps.is_fwd) {                       // arithmetic-progression dictionary: the dictId stream is the plane, nothing to build or to budget
      col.plane_is_fwd = true;
      col.d_plane = col.d_fwd;
      col.plane_ready = true;
      col.plane_state = 2;
      __atomic_store_n(&col.plane_fwd_published, 1, __ATOMIC_RELEASE);
      seg->plane_epoch.fetch_add(1, std::memory_order_release);
      *ready = true;
      return PG_OK;
    }
    const int w = ps.bits;
    const size_t bytes = (size_t)std::max(seg->num_tiles, 1) * 256 * (size_t)w + 64;
    // make room: least recently used first, only planes nobody reads
    while (g_planes.total_bytes + bytes > g_planes.budget_bytes) {
      int best = -1;
      for (int i = 0; i < (int)g_planes.resident.size(); ++i) {
        const ColumnDev& c = g_planes.resident[(size_t)i].first->cols[(size_t)g_planes.resident[(size_t)i].second];
        if (c.plane_state != 2 || c.plane_users != 0 || c.plane_is_fwd) continue;
        if (best < 0 || c.plane_last_use < g_planes.resident[(size_t)best].first->cols[(size_t)g_planes.resident[(size_t)best].second].plane_last_use) best = i;
      }
      if (best < 0) return PG_OK;          // nothing can go: this column is served without a plane
      drop_plane_locked(g_planes.resident[(size_t)best].first, g_planes.resident[(size_t)best].second);
    }
    HIP_TRY(hipSetDevice(phys_device(seg->device)));
    if (!seg->plane_stream) HIP_TRY(hipStreamCreateWithFlags(&seg->plane_stream, hipStreamNonBlocking));
    if (!col.plane_event) HIP_TRY(hipEventCreateWithFlags(&col.plane_event, hipEventDisableTiming));
    uint8_t* plane = nullptr;
    if (hipMalloc((void**)&plane, bytes) != hipSuccess) { (void)hipGetLastError(); return PG_OK; }      // no room on the device either: no plane
    HIP_TRY(hipMemsetAsync(plane, 0, bytes, seg->plane_stream));
    // the kernel writes dict'[dictId] where dict' = the scaled fields (for scale 1: value - base through the base argument)
    if (ps.scale != 1) {
      std::vector<int32_t> fields(col.h_dict.size());
      for (size_t d = 0; d < fields.size(); ++d) fields[d] = (int32_t)(((int64_t)col.h_dict[d] - ps.base) / ps.scale);
      HIP_TRY(hipMalloc((void**)&col.d_plane_fields, fields.size() * 4));
      HIP_TRY(hipMemcpy(col.d_plane_fields, fields.data(), fields.size() * 4, hipMemcpyHostToDevice));
    }
    DevColumn dc;
    memset(&dc, 0, sizeof(dc));
    dc.fwd = col.d_fwd; dc.dict = col.d_plane_fields ? col.d_plane_fields : col.d_dict; dc.bits = col.bits; dc.cardinality = col.cardinality; dc.dict_bytes = col.cardinality * 4;
    const int in_slot = ((256 * col.bits + 16) + 15) & ~15;
    const int waves = 4;
    const size_t lds = (size_t)waves * (size_t)(in_slot + 64 * w * 4 + 16);
    const int blocks = (int)std::max<long long>(1, std::min<long long>(((long long)seg->num_tiles + waves - 1) / waves, (long long)seg->num_cus * 2));
    set_dynamic_lds(materialize_plane_kernel, lds);
    materialize_plane_kernel<<<dim3((unsigned)blocks), dim3(waves * 64), lds, seg->plane_stream>>>(dc, plane, w, col.d_plane_fields ? 0 : (int32_t)col.plane_base, seg->num_docs,
                                                                                                    seg->num_tiles, in_slot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(col.plane_event, seg->plane_stream));
    col.d_plane = plane;
    col.plane_bytes = bytes;
    col.plane_state = 1;
    g_planes.total_bytes += bytes;
    seg->plane_bytes += bytes;
    seg->device_bytes += bytes;
    g_planes.resident.emplace_back(seg, column);
    seg->plane_epoch.fetch_add(1, std::memory_order_release);
  }
  if (col.plane_state == 1) {
    const hipError_t built = g_engine.plane_async ? hipEventQuery(col.plane_event) : hipEventSynchronize(col.plane_event);
    if (built == hipErrorNotReady) return PG_OK;
    if (built != hipSuccess) return fail(PG_ERR_DEVICE, "value plane of %s: %s", col.name.c_str(), hipGetErrorString(built));
    if (col.d_plane_fields) { (void)hipFree(col.d_plane_fields); col.d_plane_fields = nullptr; }
    col.plane_ready = true;
    col.plane_state = 2;
    seg->plane_epoch.fetch_add(1, std::memory_order_release);
  }
  col.plane_users++;
  col.plane_last_use = ++g_planes.tick;
  *ready = true;
  return PG_OK;
}

void release_plane(pg_segment* seg, int column) {
  ColumnDev& col = seg->cols[(size_t)column];
  if (__atomic_load_n(&col.plane_fwd_published, __ATOMIC_ACQUIRE) != 0) return;      // (never counted: acquire_plane)
  std::lock_guard<std::mutex> lk(g_planes.mu);
  if (col.plane_users > 0) col.plane_users--;
}

// The planes a query holds, released on every way out of execute_impl.
struct PlaneHold {
  pg_segment* seg;
  std::vector<int> columns;
  PlaneHold(pg_segment* s, std::vector<int> c) : seg(s), columns(std::move(c)) {}
  PlaneHold(const PlaneHold&) = delete;
  PlaneHold& operator=(const PlaneHold&) = delete;
  PlaneHold(PlaneHold&& o) noexcept : seg(o.seg), columns(std::move(o.columns)) { o.columns.clear(); }
  ~PlaneHold() { for (int c : columns) release_plane(seg, c); }
};

// ---- query lowering ----
struct Lowered {
  PlanParams plan;                    // host-side plan (columns, leaves, nodes) ...
  ScanParams sp;                      // ... flattened into the kernel parameter block by flatten_plan()
  std::vector<int> col_of_slot;       // segment column index * 2 + (1 if the slot streams the value plane)
  std::vector<char> plane_cols;       // per segment column: aggregations read it through its value plane
  int num_scan_leaves = 0;
  int max_bits = 1;
  // index_and_kernel ran for the inverted-index children of the root AND: every match lies in one of the listed tiles
  const uint32_t* tile_list = nullptr;
  const uint32_t* tile_count = nullptr;
  const unsigned long long* d_cardinality = nullptr;
  bool index_and_is_whole_filter = false;      // the filter is exactly that AND: its cardinality answers COUNT(*)
  // ... and its bitmap is stored sparsely (tiles without a match are not written): a reader that does not go by the tile list calls
  // complete_index_and_bitmap first
  unsigned long long* and_bitmap = nullptr;
  const WindowInfo* and_info = nullptr;
  long long and_words = 0;
  // numEntriesScannedInFilter: how this query's count is obtained (pg_filter_stats.h)
  fstats::Plan stats_plan = fstats::Plan::kZero;
  int stats_scan_leaves = 0;
  bool stats_chain_flagged = false;            // the chain's scan leaves carry kNodeCountEntries
  bool stats_leap2_flagged = false;            // the root AND of two scan leaves carries kNodeLeapfrog2
  bool plane_pending = false;                  // a value plane this query wanted is still being built (or had no room): the lowering is not the one to keep
  bool cardinality_only_hint = false;          // in: the query is COUNT(*) only, so an index-only filter needs neither bitmap nor tile list
  // index_and_finalize_kernel (window masks -> the ascending tile list, its length, the cardinality) is launched only when somebody reads its
  // output: scan_sparse_kernel walks the window masks themselves and COUNT(*) takes the cardinality from index_and_kernel's own counter
  bool finalize_pending = false;
  unsigned finalize_windows = 0;
  // index_and_kernel itself is launched when the planner knows what follows it (launch_index_and): a COUNT(*) wants its cardinality counters,
  // an aggregation over a handful of survivors per window is done INSIDE it (gathered: no bitmap, no second kernel), everything else its
  // bitmap and window masks.  Lowering only prepares the kernel's arguments.
  bool and_pending = false, and_cardinality_only = false, gathered = false;
  // dictId-set leaves (IN lists): where the words were uploaded for THIS context, and the host's copy -- a batch's shared launch reads them
  // from the batch's own blob instead (enqueue_deferred), so such an item is not tied to a context
  struct SetLeaf { const uint32_t* ctx_words; const uint32_t* host_words; uint32_t bytes; };
  std::vector<SetLeaf> set_leaves;
  bool cardinality_atomic = false;             // index_and_kernel added its figures to the context's counter lines (launch_index_and; read_index_and_shards)
  int and_num_cus = 256;                       // the segment's CUs (pg_segment.num_cus): index_and_kernel's persistent grid is sized by them
  IndexAndParams and_params;
  double and_expected_docs = 0;                // the planner's estimate of the AND's cardinality (independent postings)
  FsmSide* side = nullptr;                     // in: the transducer pass wants the leaves' bitmaps (ScanParams.leaf_out)
  uint32_t* sp_leaf_out[kMaxLeaves] = {};      // out: ScanParams.leaf_out, by LEAF node ordinal
};

int slot_for(Lowered* lw, const pg_segment* seg, int column, bool plane = false) {
  const int id = column * 2 + (plane ? 1 : 0);
  for (size_t i = 0; i < lw->col_of_slot.size(); ++i) if (lw->col_of_slot[i] == id) return (int)i;
  if ((int)lw->col_of_slot.size() >= kMaxCols) return -1;
  const ColumnDev& c = seg->cols[column];
  DevColumn& d = lw->plan.cols[lw->col_of_slot.size()];
  memset(&d, 0, sizeof(d));
  if (plane && c.vkind != kValI32) {
    // wide plane of a LONG / DOUBLE dictionary column: the values themselves, big-endian 8 bytes per doc -- the image of a raw column
    d.fwd = c.d_plane;
    d.dict = nullptr;
    d.bits = 32;
    d.is_raw = 1;
    d.is_plane = 0;
    d.vkind = c.vkind;
  } else if (plane) {
    d.fwd = c.d_plane;
    d.dict = nullptr;
    d.bits = c.plane_bits;
    d.is_raw = c.plane_bits == 32;     // 32-bit planes are big-endian int32 values: the raw-column path
    d.is_plane = c.plane_bits == 32 ? 0 : 1;
    d.cardinality = 0;
    d.dict_bytes = 0;
  } else {
    d.fwd = c.d_fwd;
    d.dict = c.vkind == kValI32 ? c.d_dict : reinterpret_cast<const int32_t*>(c.d_dict64);
    d.bits = c.encoding == PG_FWD_RAW_FIXED_BYTE ? 32 : c.bits;
    d.is_raw = c.encoding == PG_FWD_RAW_FIXED_BYTE;
    d.cardinality = c.cardinality;
    d.dict_bytes = c.cardinality * (c.vkind == kValI32 ? 4 : 8);
    d.vkind = c.vkind;
  }
  if (!d.is_raw) lw->max_bits = std::max(lw->max_bits, d.bits);
  lw->col_of_slot.push_back(id);
  lw->plan.num_cols = (int)lw->col_of_slot.size();
  return lw->plan.num_cols - 1;
}

struct SeqNode { int src; int op; int num_children; int flags; };

// Re-orders the children of a root AND so that inverted-index (bitmap) leaves come first, rewrites that AND as a chain
// of binary ANDs flagged kNodeExitIfZero (AND is commutative and associative: same docId set), and reports where the
// bitmap prefix ends.  Everything else keeps its postfix order.
constexpr int kSeqIndexAnd = -2;       // SeqNode.src of the leaf that stands for all inverted-index children of the root AND

bool is_inverted_leaf(const pg_query* q, int node) {
  if (q->filter[node].op != PG_FILTER_LEAF) return false;
  const int pi = q->filter[node].predicate;
  if (pi < 0 || pi >= q->num_predicates) return false;
  const pg_predicate& pr = q->predicates[pi];
  return pr.eval == PG_EVAL_INVERTED && (pr.kind == PG_PRED_DICT_RANGE || pr.kind == PG_PRED_DICT_SET);
}

void build_sequence(const pg_query* q, std::vector<SeqNode>* seq, int* lazy_node, int* num_bitmap_prefix, std::vector<int>* and_members) {
  const int n = q->num_filter_nodes;
  *lazy_node = -1;
  *num_bitmap_prefix = 0;
  seq->clear();
  and_members->clear();
  if (n == 0) return;
  std::vector<int> start((size_t)n, 0);
  for (int i = 0; i < n; ++i) {
    const pg_filter_node& fn = q->filter[i];
    if (fn.op == PG_FILTER_LEAF) start[(size_t)i] = i;
    else {
      int k = fn.op == PG_FILTER_NOT ? 1 : fn.num_children;
      int idx = i - 1;
      for (int c = 0; c < k && idx >= 0; ++c) idx = start[(size_t)idx] - 1;
      start[(size_t)i] = idx + 1;
    }
  }
  const pg_filter_node& root = q->filter[n - 1];
  auto identity = [&](int from, int to) { for (int i = from; i <= to; ++i) seq->push_back(SeqNode{i, q->filter[i].op, q->filter[i].num_children, 0}); };
  if (root.op != PG_FILTER_AND || root.num_children < 2 || start[(size_t)n - 1] != 0) {
    if (n == 1 && is_inverted_leaf(q, 0)) {
      // a single inverted-index leaf: the same window-by-window kernel expands it and lists the tiles that hold a match
      and_members->push_back(q->filter[0].predicate);
      seq->push_back(SeqNode{kSeqIndexAnd, PG_FILTER_LEAF, 0, kNodeExitIfZero});
      return;
    }
    identity(0, n - 1);
    if (root.op == PG_FILTER_LEAF) seq->back().flags |= kNodeExitIfZero;
    return;
  }
  // children of the root, in query order
  std::vector<std::pair<int, int>> children;   // [first, last] node index of each child subtree
  int idx = n - 2;
  for (int c = 0; c < root.num_children; ++c) { children.push_back({start[(size_t)idx], idx}); idx = start[(size_t)idx] - 1; }
  std::reverse(children.begin(), children.end());
  auto is_bitmap_leaf = [&](const std::pair<int, int>& ch) {
    if (ch.first != ch.second || q->filter[ch.first].op != PG_FILTER_LEAF) return false;
    const int pi = q->filter[ch.first].predicate;
    if (pi < 0 || pi >= q->num_predicates) return false;
    const pg_predicate& pr = q->predicates[pi];
    // index-driven leaves go first, like the reference's priorities (sorted 0 < bitmap 100 < scan 500, FilterOperatorUtils.java:205-251)
    return pr.kind == PG_PRED_DOC_RANGE || pr.kind == PG_PRED_IS_NULL || (pr.eval == PG_EVAL_INVERTED && (pr.kind == PG_PRED_DICT_RANGE || pr.kind == PG_PRED_DICT_SET));
  };
  std::stable_partition(children.begin(), children.end(), is_bitmap_leaf);
  // The inverted-index children are and-ed by ONE kernel, container by container (AndDocIdSet.java:127-165 and-s the index-based
  // children first): they become a single leaf at the head of the chain.
  std::vector<std::pair<int, int>> rest;
  for (const auto& ch : children) {
    if (ch.first == ch.second && is_inverted_leaf(q, ch.first) && (int)and_members->size() < kMaxAndChildren) and_members->push_back(q->filter[ch.first].predicate);
    else rest.push_back(ch);
  }
  const int head = and_members->empty() ? 0 : 1;
  for (const auto& ch : rest) *num_bitmap_prefix += is_bitmap_leaf(ch) ? 1 : 0;
  *num_bitmap_prefix += head;
  for (size_t c = 0; c < rest.size() + (size_t)head; ++c) {
    if (head && c == 0) seq->push_back(SeqNode{kSeqIndexAnd, PG_FILTER_LEAF, 0, 0});
    else identity(rest[c - (size_t)head].first, rest[c - (size_t)head].second);
    if (c == 0) seq->back().flags |= kNodeExitIfZero;
    else seq->push_back(SeqNode{-1, PG_FILTER_AND, 2, kNodeExitIfZero});
    if ((int)c + 1 == *num_bitmap_prefix) *lazy_node = (int)seq->size() - 1;
  }
}

// kNodeCountEntries leaves: the device counter the lane-private kernels add their entries to, zeroed in stream order.
// Timed runs (PG_CFG_TIME_KERNELS): ev[0] opens the query's device work.  It is recorded by the first thing that enqueues work BEFORE
// the scan kernel (index AND, posting expansion, set uploads, table initialisation); a query that runs only its scan kernel opens with ev[1].
hipError_t mark_pre_work(ExecCtx* ctx) {
  ctx->pre_enqueued = true;          // (pg_execute_batch: such a query runs on its own context, not in the shared launch)
  if (!(g_engine.flags & PG_CFG_TIME_KERNELS) || ctx->pre_started) return hipSuccess;
  ctx->pre_started = true;
  return hipEventRecord(ctx->ev[0], ctx->stream);
}

pg_status arm_filter_entries(ExecCtx* ctx, unsigned long long** out_counter) {
  if (!ctx->d_filter_entries) HIP_TRY(hipMalloc((void**)&ctx->d_filter_entries, 8));
  if (!ctx->h_filter_entries) HIP_TRY(hipHostMalloc((void**)&ctx->h_filter_entries, 8, hipHostMallocDefault));
  HIP_TRY(mark_pre_work(ctx));
  HIP_TRY(hipMemsetAsync(ctx->d_filter_entries, 0, 8, ctx->stream));
  *out_counter = ctx->d_filter_entries;
  return PG_OK;
}

// Zeros in every tile index_and_kernel did not store: for the kernels that read the whole bitmap instead of the tile list.
// kNodeLeapfrog2: the per-tile bytes, the per-1024-tile summaries, and the (zeroed) entries counter the kernels and the chain add to
pg_status arm_leap_tables(const pg_segment* seg, ExecCtx* ctx, uint8_t** out_tables, unsigned long long** out_counter /* nullptr: the scan kernel's record carries the count */) {
  const size_t tiles = (size_t)std::max(seg->num_tiles, 1);
  if (ctx->leap_capacity < tiles) {
    if (ctx->d_leap_tables) (void)hipFree(ctx->d_leap_tables);
    ctx->d_leap_tables = nullptr; ctx->d_leap_blocks = nullptr; ctx->leap_capacity = 0;
    const size_t table_bytes = (tiles + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc((void**)&ctx->d_leap_tables, table_bytes + ((tiles + 1023) / 1024) * sizeof(Leap2Summary)));
    ctx->d_leap_blocks = reinterpret_cast<Leap2Summary*>(ctx->d_leap_tables + table_bytes);
    ctx->leap_capacity = tiles;
  }
  *out_tables = ctx->d_leap_tables;
  return out_counter ? arm_filter_entries(ctx, out_counter) : PG_OK;
}

// behind the scan kernel: the entry states of the tiles, chained in tile order, decide which tiles' corrections count.  The result goes
// into the context's pinned record (host_seq != 0: aggregation queries) or onto the device counter (group-by queries).
pg_status launch_leap_chain(const pg_segment* seg, ExecCtx* ctx, unsigned long long host_seq) {
  const long long tiles = ((long long)seg->num_docs + 2047) / 2048;
  const int blocks = (int)std::max<long long>(1, (tiles + 1023) / 1024);
  leapfrog2_chain_kernel<<<dim3((unsigned)blocks), dim3(1024), 0, ctx->stream>>>(ctx->d_leap_tables, tiles, ctx->d_leap_blocks, ctx->d_done + (kFoldShards + 1) * kFoldStride,
                                                                                  host_seq ? ctx->h_record_dev : nullptr, host_seq, host_seq ? nullptr : ctx->d_filter_entries);
  HIP_TRY(hipGetLastError());
  return PG_OK;
}

// The aggregation INSIDE index_and_kernel: ScanParams.agg_cols are the columns it reads; no bitmap is stored.
static void arm_index_gather(Lowered* lw, const ScanParams* gather_from) {
  IndexAndParams& ap = lw->and_params;
  ap.gather_cols = gather_from->num_agg_cols;
  for (int a = 0; a < gather_from->num_agg_cols && a < kMaxAndGather; ++a) ap.gather_col[a] = gather_from->agg_cols[a];
  ap.out = nullptr;                              // nobody reads a bitmap
  lw->gathered = true;
}
// pg_execute_batch: an index-led item whose whole device work is index_and_kernel publishing the query's record -- COUNT(*) over the whole
// filter, or the gathered aggregation -- can share index_and_batch_kernel's launch when nothing of it lives in the lowering context:
// no child expanded densely ahead of the kernel (such a child is a bitmap of the context, and a launch of its own before this one).
static bool index_and_shares_a_launch(const Lowered& lw) {
  if (!lw.and_pending || lw.finalize_windows == 0) return false;
  for (int c = 0; c < lw.and_params.num_children; ++c) if (lw.and_params.child[c].dense != nullptr) return false;
  return true;
}

// index_and_kernel, launched when its consumer is known.  `gather_from`: the aggregation runs inside the kernel (ScanParams.agg_cols are the
// columns it reads; no bitmap is stored); else the kernel leaves its bitmap / window masks.  COUNT(*) over the whole filter and the gathered
// aggregation come back on the context's counter lines (IndexAndParams.shards): the caller copies them to ctx->h_and_shards, reads them
// with read_index_and_shards and zeroes them again behind the answer.
pg_status launch_index_and(Lowered* lw, ExecCtx* ctx, const ScanParams* gather_from) {
  if (!lw->and_pending) return PG_OK;
  lw->and_pending = false;
  IndexAndParams& ap = lw->and_params;
  const unsigned num_windows = lw->finalize_windows;
  // One wavefront per workgroup; a wave takes windows key, key + grid, ... with the next window's directory lookups in flight
  // (pg_index_and.h).  PINOT_GPU_INDEX_AND_WAVES: 0 (default) = a persistent grid of as many waves as are resident; n > 0 = of n waves per
  // CU; -k = k windows per wave (grid = windows / k workgroups handed out by the dispatcher as slots free up; -1: a wave per window, rounds
  // 2-5).  Measured on C5 at 1 B rows (profiles/r6/c5_index_and_*.jsonl): the kernel is bound by instruction issue, not by waiting (SQ
  // counters: the waves' active-instruction cycles per SIMD add up to the kernel's duration), so the grid's shape moves it by a few
  // percent only -- after the scalar-instruction diet the resident grid is ahead (COUNT 39.5 vs 43.3 us, gathered SUM 58.1 vs 58.5-59.6);
  // 8 or 12 waves per CU lose 20 - 40 %.
  const int per_cu = g_engine.index_and_waves > 0 ? g_engine.index_and_waves : waves_index_and();
  const unsigned grid = g_engine.index_and_waves < 0 ? (num_windows + (unsigned)(-g_engine.index_and_waves) - 1) / (unsigned)(-g_engine.index_and_waves)
                                                      : (unsigned)std::min<long long>(num_windows, (long long)lw->and_num_cus * per_cu);
  memset(&ap.pub, 0, sizeof(ap.pub));
  ap.shards = nullptr;
  ap.num_windows = (int32_t)num_windows;
  if (gather_from != nullptr) arm_index_gather(lw, gather_from);
  if (lw->and_cardinality_only || gather_from != nullptr) {
    // the wavefronts add what they found to counter lines the host keeps at zero between queries
    if (ctx->and_counter_dirty) HIP_TRY(hipMemsetAsync(ctx->d_and_counters + 2, 0, kAndShardBytes, ctx->stream));      // (a query that failed before it read them)
    ap.shards = ctx->d_and_counters + 2;
    ctx->and_counter_dirty = true;
    lw->cardinality_atomic = true;
  }
  if (num_windows) {
    launch_index_and_kernel((int)grid, ctx->stream, ap, num_windows);
    HIP_TRY(hipGetLastError());
    // (index_and_finalize_kernel: only when the tile list is read -- complete_index_list)
    lw->finalize_pending = !lw->and_cardinality_only && gather_from == nullptr;
  } else if (!lw->and_cardinality_only && gather_from == nullptr) {
    HIP_TRY(hipMemsetAsync(ctx->d_and_counters, 0, 16, ctx->stream));
  }
  return PG_OK;
}
// The counter lines of a COUNT(*) / gathering index_and_kernel, copied to ctx->h_and_shards and complete (the stream was waited for): zeroes
// the device's lines again behind the answer (nobody waits for that) and folds the 64 lines into one record.
pg_status read_index_and_shards(ExecCtx* ctx, int gather_cols, BlockPartial* g) {
  HIP_TRY(hipMemsetAsync(ctx->d_and_counters + 2, 0, kAndShardBytes, ctx->stream));
  ctx->and_counter_dirty = false;
  memset(g, 0, sizeof(*g));
  for (int a = 0; a < kMaxAggCols; ++a) { g->kmin[a] = 0x7FFFFFFF; g->kmax[a] = (int32_t)0x80000000; }
  const unsigned long long* lines = ctx->h_and_shards;
  unsigned long long kmin_inv[kMaxAndGather] = {0ull, 0ull}, kmax[kMaxAndGather] = {0ull, 0ull};
  for (int sh = 0; sh < kAndCardinalityShards; ++sh) {
    const unsigned long long* line = lines + (size_t)sh * 16;
    g->count += line[0];
    for (int a = 0; a < gather_cols && a < kMaxAndGather; ++a) {
      g->sum[a] += (long long)line[1 + 3 * a];
      kmin_inv[a] = std::max(kmin_inv[a], line[2 + 3 * a]);
      kmax[a] = std::max(kmax[a], line[3 + 3 * a]);
    }
  }
  for (int a = 0; a < gather_cols && a < kMaxAndGather; ++a) {
    if (g->count == 0ull) continue;                 // (identities stay)
    g->kmin[a] = (int32_t)(0xFFFFFFFFull - kmin_inv[a]);
    g->kmax[a] = (int32_t)kmax[a];
  }
  return PG_OK;
}

// The tile list of an index-led filter, for the kernels that read one (everything but scan_sparse_kernel): index_and_finalize_kernel behind
// index_and_kernel, on the query's stream, the first time the list is asked for.
pg_status complete_index_list(Lowered* lw, ExecCtx* ctx) {
  { const pg_status ls = launch_index_and(lw, ctx, nullptr); if (ls != PG_OK) return ls; }
  if (!lw->finalize_pending) return PG_OK;
  lw->finalize_pending = false;
  index_and_finalize_kernel<<<dim3((lw->finalize_windows + 255) / 256), dim3(256), 0, ctx->stream>>>(ctx->d_window_info, (int)lw->finalize_windows, ctx->d_tile_list,
                                                                                                    reinterpret_cast<uint32_t*>(ctx->d_and_counters + 1), ctx->d_and_counters);
  HIP_TRY(hipGetLastError());
  return PG_OK;
}

pg_status complete_index_and_bitmap(Lowered* lw, ExecCtx* ctx) {
  { const pg_status ls = launch_index_and(lw, ctx, nullptr); if (ls != PG_OK) return ls; }
  if (!lw->and_bitmap) return PG_OK;
  HIP_TRY(mark_pre_work(ctx));
  index_and_zero_unlisted_kernel<<<dim3(2048), dim3(256), 0, ctx->stream>>>(lw->and_info, lw->and_bitmap, lw->and_words);
  HIP_TRY(hipGetLastError());
  lw->and_bitmap = nullptr;
  return PG_OK;
}

pg_status lower_filter(pg_segment* seg, ExecCtx* ctx, const pg_query* q, Lowered* lw) {
  PlanParams& sp = lw->plan;
  if (q->num_filter_nodes < 0 || q->num_filter_nodes > kMaxNodes) return fail(PG_ERR_UNSUPPORTED, "filter tree has %d nodes (max %d)", q->num_filter_nodes, kMaxNodes);
  if (q->num_filter_nodes > 0 && (!q->filter || !q->predicates)) return fail(PG_ERR_INVALID_ARGUMENT, "filter nodes without predicates");
  for (int n = 0; n < q->num_filter_nodes; ++n) {
    const int op = q->filter[n].op;
    if (op < PG_FILTER_LEAF || op > PG_FILTER_NOT) return fail(PG_ERR_INVALID_ARGUMENT, "unknown filter op %d", op);
    if ((op == PG_FILTER_AND || op == PG_FILTER_OR) && q->filter[n].num_children < 1) return fail(PG_ERR_INVALID_ARGUMENT, "malformed filter tree (node %d)", n);
  }
  std::vector<SeqNode> seq;
  std::vector<int> and_members;
  int lazy_node = -1, num_bitmap_prefix = 0;
  build_sequence(q, &seq, &lazy_node, &num_bitmap_prefix, &and_members);
  for (int pi : and_members) if (pi < 0 || pi >= q->num_predicates) return fail(PG_ERR_INVALID_ARGUMENT, "bad predicate index");
  // InvertedIndexFilterOperator.getTrues as a dense doc-order bitmap: OR of the postings of every matching dictId (the general path:
  // leaves under OR / NOT, and members of the index AND with more than kMaxAndPostings postings)
  auto expand_dense = [&](const ColumnDev& col, const pg_predicate& pr, unsigned long long* bm) -> pg_status {
    const long long words = (long long)seg->num_tiles * kMaxTileSteps;
    const unsigned num_windows = (unsigned)((words + 1023) / 1024);
    bool first_posting = true;
    HIP_TRY(mark_pre_work(ctx));
    for (int d = 0; d < col.cardinality; ++d) {
      bool in;
      if (pr.kind == PG_PRED_DICT_RANGE) in = d >= pr.lo && d < pr.hi;
      else in = (d >> 5) < pr.num_set_words && ((pr.set_words[d >> 5] >> (d & 31)) & 1u);
      if (!in) continue;
      const int64_t first = col.posting_first[d], cnt = col.posting_first[d + 1] - first;
      if (cnt <= 0 && !first_posting) continue;
      // the first posting stores every window (zeros where it has no container); later ones OR
      roaring_expand_kernel<<<dim3(num_windows), dim3(kBlockThreads), 0, ctx->stream>>>(col.d_inv, col.d_dir, (int)first, (int)cnt, bm, words, first_posting ? 0 : 1);
      first_posting = false;
    }
    if (first_posting) fill_words_kernel<<<dim3(256), dim3(256), 0, ctx->stream>>>(bm, words, 0ull);   // no dictId matched
    HIP_TRY(hipGetLastError());
    return PG_OK;
  };
  if ((int)seq.size() > kMaxNodes) return fail(PG_ERR_UNSUPPORTED, "filter tree has %d nodes (max %d)", (int)seq.size(), kMaxNodes);
  int depth = 0, max_depth = 0;
  size_t bitmap_idx = 1;   // bitmap 0 is reserved for pg_filter_bitmap output
  size_t set_idx = 0;
  for (int n = 0; n < (int)seq.size(); ++n) {
    pg_filter_node fn;
    memset(&fn, 0, sizeof(fn));
    if (seq[(size_t)n].src >= 0) fn = q->filter[seq[(size_t)n].src];
    fn.op = seq[(size_t)n].op;
    fn.num_children = seq[(size_t)n].num_children;
    PlanNode& dn = sp.nodes[n];
    dn.op = fn.op;
    dn.leaf = -1;
    dn.num_children = fn.num_children;
    dn.flags = seq[(size_t)n].flags;
    if (lw->side != nullptr) {
      // every leaf's mask of every tile feeds the walk: no leaf may end a tile early; a leaf that is one of the walk's inputs stores its mask
      dn.flags &= ~kNodeExitIfZero;
      if (fn.op == PG_FILTER_LEAF) {
        int ordinal = 0;
        for (int m = 0; m < n; ++m) ordinal += seq[(size_t)m].op == PG_FILTER_LEAF ? 1 : 0;
        if (seq[(size_t)n].src >= 0 && ordinal < kMaxLeaves) {
          const std::vector<int>& inputs = lw->side->fsm->input_predicate;
          for (size_t i = 0; i < inputs.size(); ++i) {
            if (inputs[i] != fn.predicate || lw->side->mapped[i]) continue;
            lw->sp_leaf_out[ordinal] = lw->side->bitmap[i];
            lw->side->mapped[i] = true;
          }
        }
      }
    }
    if (lw->stats_plan == fstats::Plan::kLeap2) {
      // `a AND b`, two scan leaves: both masks of EVERY tile feed the leap-frog entry count, so no leaf may end a tile early
      // (whichever leaf is scanning when a tile is entered keeps looking at its docs even where the other one matches nothing)
      if (fn.op == PG_FILTER_LEAF) dn.flags &= ~kNodeExitIfZero;
      else if (fn.op == PG_FILTER_AND && n == (int)seq.size() - 1 && seq.size() == 3) { dn.flags |= kNodeLeapfrog2; lw->stats_leap2_flagged = true; }
    }
    if (lw->stats_plan == fstats::Plan::kChain && fn.op == PG_FILTER_LEAF && seq[(size_t)n].src >= 0 && fn.predicate >= 0 && fn.predicate < q->num_predicates &&
        fstats::classify(q->predicates[fn.predicate]) == fstats::LeafClass::kScan && n > 0) {
      // a scan-based child of the root AND, behind the index-based ones: ScanBasedDocIdIterator.applyAnd looks at every doc still standing
      dn.flags |= kNodeCountEntries;
      lw->stats_chain_flagged = true;
    }
    if (fn.op == PG_FILTER_LEAF && seq[(size_t)n].src == kSeqIndexAnd) {
      // ---- the inverted-index children of the root AND, intersected container by container (index_and_kernel) ----
      if (sp.num_leaves >= kMaxLeaves) return fail(PG_ERR_UNSUPPORTED, "more than %d filter leaves", kMaxLeaves);
      DevLeaf& L = sp.leaves[sp.num_leaves];
      memset(&L, 0, sizeof(L));
      dn.leaf = sp.num_leaves++;
      struct Member { AndChild child; double estimate; std::vector<std::pair<int32_t, int32_t>> postings; };
      std::vector<Member> members;
      bool empty = false;
      for (int pi : and_members) {
        const pg_predicate& pr = q->predicates[pi];
        if (pr.column < 0 || pr.column >= (int)seg->cols.size()) return fail(PG_ERR_INVALID_ARGUMENT, "predicate column %d out of range", pr.column);
        const ColumnDev& col = seg->cols[pr.column];
        if (col.encoding != PG_FWD_FIXED_BIT_DICT) return fail(PG_ERR_INVALID_ARGUMENT, "dictionary predicate on raw column %s", col.name.c_str());
        if (!col.d_inv) return fail(PG_ERR_INVALID_ARGUMENT, "column %s has no inverted index", col.name.c_str());
        if (pr.kind == PG_PRED_DICT_SET && (pr.num_set_words < 0 || (pr.num_set_words > 0 && !pr.set_words))) return fail(PG_ERR_INVALID_ARGUMENT, "bad dictId set");
        Member mb;
        memset(&mb.child, 0, sizeof(mb.child));
        mb.child.inv = col.d_inv; mb.child.dir = col.d_dir; mb.child.exclusive = pr.exclusive ? 1 : 0;
        double docs = 0;
        int postings = 0;
        // (a range walks its own dictIds only, a set the dictIds its words can name)
        const bool is_range = pr.kind == PG_PRED_DICT_RANGE;
        const int d_begin = is_range ? (int)std::max<int64_t>(pr.lo, 0) : 0;
        const int d_end = is_range ? (int)std::min<int64_t>(pr.hi, col.cardinality) : (int)std::min<int64_t>((int64_t)pr.num_set_words * 32, col.cardinality);
        for (int d = d_begin; d < d_end; ++d) {
          if (!is_range && !((pr.set_words[d >> 5] >> (d & 31)) & 1u)) continue;
          const int64_t first = col.posting_first[d], cnt = col.posting_first[d + 1] - first;
          if (cnt <= 0) continue;
          docs += (double)col.posting_docs[(size_t)d];
          if (postings < kMaxChildPostings) mb.postings.emplace_back((int32_t)first, (int32_t)cnt);
          postings++;
        }
        if (postings == 0 && !pr.exclusive) { empty = true; break; }          // nothing matches this child: the AND is empty
        if (postings == 0 && pr.exclusive) continue;                          // NOT (nothing) = everything: the child drops out
        mb.estimate = pr.exclusive ? (double)seg->num_docs - docs : docs;
        size_t inline_so_far = 0;
        for (const Member& m : members) inline_so_far += m.postings.size();
        if (postings > kMaxChildPostings || inline_so_far + (size_t)postings > (size_t)kMaxAndPostings) {
          // too many postings for one lookup per lane: this child is expanded densely first
          pg_status st = ensure_bitmap(seg, ctx, bitmap_idx);
          if (st != PG_OK) return st;
          unsigned long long* bm = ctx->d_bitmaps[bitmap_idx++];
          st = expand_dense(col, pr, bm);
          if (st != PG_OK) return st;
          mb.child.inv = nullptr; mb.child.dir = nullptr; mb.child.dense = bm;
          mb.postings.clear();
        }
        members.push_back(std::move(mb));
      }
      if (empty) { L.kind = kLeafMatchNone; depth++; max_depth = std::max(max_depth, depth); continue; }
      if (members.empty()) { L.kind = kLeafMatchAll; depth++; max_depth = std::max(max_depth, depth); continue; }
      // smallest first, like AndDocIdSet sorts its bitmaps: an empty window ends the work for the later children
      std::stable_sort(members.begin(), members.end(), [](const Member& a, const Member& b) { return a.estimate < b.estimate; });
      const bool cardinality_only = lw->cardinality_only_hint && seq.size() == 1;     // FastFilteredCount: neither the bitmap nor the tile list is read
      unsigned long long* bm = nullptr;
      if (!cardinality_only) {
        pg_status st = ensure_bitmap(seg, ctx, bitmap_idx);
        if (st != PG_OK) return st;
        bm = ctx->d_bitmaps[bitmap_idx++];
      }
      const long long num_words = (long long)seg->num_tiles * kMaxTileSteps;
      const unsigned num_windows = (unsigned)((num_words + 1023) / 1024);
      if (ctx->tile_list_capacity < (size_t)seg->num_tiles + 64 || ctx->window_info_capacity < (size_t)num_windows + 1) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_tile_list) (void)hipFree(ctx->d_tile_list);
        if (ctx->d_window_info) (void)hipFree(ctx->d_window_info);
        ctx->d_tile_list = nullptr; ctx->d_window_info = nullptr;
        ctx->tile_list_capacity = (size_t)seg->num_tiles + 64;
        ctx->window_info_capacity = (size_t)num_windows + 1;
        HIP_TRY(hipMalloc((void**)&ctx->d_tile_list, ctx->tile_list_capacity * 4));
        HIP_TRY(hipMalloc((void**)&ctx->d_window_info, ctx->window_info_capacity * sizeof(WindowInfo)));
      }
      if (!ctx->d_and_counters) {
        // [0] cardinality and [1] tile count: index_and_finalize_kernel's outputs; behind them index_and_kernel's counter lines
        HIP_TRY(hipMalloc((void**)&ctx->d_and_counters, 16 + kAndShardBytes));
        HIP_TRY(hipMemsetAsync(ctx->d_and_counters, 0, 16 + kAndShardBytes, ctx->stream));
        HIP_TRY(hipHostMalloc((void**)&ctx->h_and_shards, kAndShardBytes, hipHostMallocDefault));
        ctx->and_counter_dirty = false;
      }
      IndexAndParams ap;
      memset(&ap, 0, sizeof(ap));
      ap.num_children = (int32_t)members.size();
      ap.num_docs = seg->num_docs;
      ap.num_words = num_words;
      ap.out = bm;
      ap.sparse_out = 1;
      ap.window_info = ctx->d_window_info;
      for (size_t c = 0; c < members.size(); ++c) {
        ap.child[c] = members[c].child;
        ap.child[c].posting_begin = ap.num_postings;
        for (const auto& ps : members[c].postings) {
          ap.first[ap.num_postings] = ps.first; ap.count[ap.num_postings] = ps.second; ap.posting_child[ap.num_postings] = (uint8_t)c;
          ap.num_postings++;
        }
        ap.child[c].posting_end = ap.num_postings;
      }
      unsigned long long* d_cardinality = ctx->d_and_counters;      // (index_and_finalize_kernel's; a cardinality-only or gathering launch publishes a record instead)
      uint32_t* d_tile_count = reinterpret_cast<uint32_t*>(ctx->d_and_counters + 1);
      HIP_TRY(mark_pre_work(ctx));
      // (the launch itself: launch_index_and, once the planner knows what reads the kernel's output)
      lw->and_params = ap;
      lw->and_num_cus = seg->num_cus;
      lw->and_pending = true;
      lw->and_cardinality_only = cardinality_only;
      lw->finalize_windows = num_windows;
      {
        double expected = (double)seg->num_docs;
        for (const Member& mb : members) expected *= std::min(1.0, std::max(0.0, (double)mb.estimate) / std::max(1.0, (double)seg->num_docs));
        lw->and_expected_docs = expected;
      }
      lw->d_cardinality = d_cardinality;
      lw->index_and_is_whole_filter = seq.size() == 1;
      if (cardinality_only) { L.kind = kLeafMatchAll; depth++; max_depth = std::max(max_depth, depth); continue; }   // never evaluated: execute_impl answers from the cardinality
      L.kind = kLeafBitmap;
      L.bitmap = bm;
      sp.num_bitmap_leaves++;
      lw->tile_list = ctx->d_tile_list;
      lw->tile_count = d_tile_count;
      lw->and_bitmap = bm;
      lw->and_info = ctx->d_window_info;
      lw->and_words = num_words;
      depth++;
      max_depth = std::max(max_depth, depth);
      continue;
    }
    if (fn.op == PG_FILTER_LEAF) {
      if (fn.predicate < 0 || fn.predicate >= q->num_predicates) return fail(PG_ERR_INVALID_ARGUMENT, "filter node %d: bad predicate index", n);
      if (sp.num_leaves >= kMaxLeaves) return fail(PG_ERR_UNSUPPORTED, "more than %d filter leaves", kMaxLeaves);
      const pg_predicate& pr = q->predicates[fn.predicate];
      DevLeaf& L = sp.leaves[sp.num_leaves];
      memset(&L, 0, sizeof(L));
      dn.leaf = sp.num_leaves++;
      L.exclusive = pr.exclusive ? 1 : 0;
      L.col = 0;
      if (pr.kind == PG_PRED_MATCH_ALL) { L.kind = kLeafMatchAll; }
      else if (pr.kind == PG_PRED_MATCH_NONE) { L.kind = kLeafMatchNone; }
      else if (pr.kind == PG_PRED_IS_NULL) {
        // FilterPlanNode.java:294-310: the null bitmap as a BitmapBasedFilterOperator; no null vector -> Empty (IS NULL) / MatchAll (IS NOT NULL)
        if (pr.column < 0 || pr.column >= (int)seg->cols.size()) return fail(PG_ERR_INVALID_ARGUMENT, "predicate column %d out of range", pr.column);
        const ColumnDev& col = seg->cols[pr.column];
        if (!col.d_null_bitmap) { L.kind = kLeafMatchNone; }
        else { L.kind = kLeafBitmap; L.bitmap = col.d_null_bitmap; sp.num_bitmap_leaves++; }
      }
      else if (pr.kind == PG_PRED_DOC_RANGE) {
        // SortedIndexBasedFilterOperator: one inclusive docId range; nothing is scanned
        const int64_t lo = std::max<int64_t>(pr.lo, 0), hi = std::min<int64_t>(pr.hi, (int64_t)seg->num_docs - 1);
        if (lo > hi) L.kind = kLeafMatchNone;
        else { L.kind = kLeafDocRange; L.lo = (int32_t)lo; L.span = (uint32_t)(hi - lo); }
      }
      else {
        if (pr.column < 0 || pr.column >= (int)seg->cols.size()) return fail(PG_ERR_INVALID_ARGUMENT, "predicate column %d out of range", pr.column);
        const ColumnDev& col = seg->cols[pr.column];
        const bool is_dict = col.encoding == PG_FWD_FIXED_BIT_DICT;
        if ((pr.kind == PG_PRED_DICT_RANGE || pr.kind == PG_PRED_DICT_SET) && !is_dict)
          return fail(PG_ERR_INVALID_ARGUMENT, "dictionary predicate on raw column %s", col.name.c_str());
        if (pr.kind == PG_PRED_RAW_RANGE && is_dict) return fail(PG_ERR_INVALID_ARGUMENT, "raw predicate on dictionary column %s", col.name.c_str());
        if (pr.eval == PG_EVAL_INVERTED) {
          if (!col.d_inv) return fail(PG_ERR_INVALID_ARGUMENT, "column %s has no inverted index", col.name.c_str());
          if (pr.kind != PG_PRED_DICT_RANGE && pr.kind != PG_PRED_DICT_SET) return fail(PG_ERR_INVALID_ARGUMENT, "inverted-index leaf needs a dictionary predicate");
          // InvertedIndexFilterOperator.getTrues: OR of the postings of every matching dictId.
          if (pr.kind == PG_PRED_DICT_SET && (pr.num_set_words < 0 || (pr.num_set_words > 0 && !pr.set_words))) return fail(PG_ERR_INVALID_ARGUMENT, "bad dictId set");
          pg_status st = ensure_bitmap(seg, ctx, bitmap_idx);
          if (st != PG_OK) return st;
          unsigned long long* bm = ctx->d_bitmaps[bitmap_idx++];
          st = expand_dense(col, pr, bm);
          if (st != PG_OK) return st;
          L.kind = kLeafBitmap;
          L.bitmap = bm;
          sp.num_bitmap_leaves++;
        } else if (pr.kind == PG_PRED_DICT_RANGE) {
          int64_t lo = std::max<int64_t>(pr.lo, 0), hi = std::min<int64_t>(pr.hi, col.cardinality);
          if (lo >= hi) { L.kind = kLeafMatchNone; }
          else if (!lw->plane_cols.empty() && lw->plane_cols[(size_t)pr.column] && col.vkind == kValI32) {
            // The column is also summed through its value plane: evaluate the range on the plane so only one
            // stream is read.  The dictionary is sorted, so dictIds [lo, hi) <=> values [dict[lo], dict[hi-1]].
            int s = slot_for(lw, seg, pr.column, true);
            if (s < 0) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
            sp.cols[s].in_filter = 1;
            const int64_t vlo = col.h_dict[(size_t)lo], vhi = col.h_dict[(size_t)hi - 1];
            if (sp.cols[s].is_raw) { L.kind = kLeafRawRange; L.lo = (int32_t)vlo; L.span = (uint32_t)(vhi - vlo); }
            else { L.kind = kLeafDictRange; L.lo = (int32_t)((vlo - col.plane_base) / col.plane_scale); L.span = (uint32_t)((vhi - vlo) / col.plane_scale + 1); }
            L.col = s;
            lw->num_scan_leaves++;
          } else {
            int s = slot_for(lw, seg, pr.column);
            if (s < 0) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
            sp.cols[s].in_filter = 1;
            L.kind = kLeafDictRange; L.col = s; L.lo = (int32_t)lo; L.span = (uint32_t)(hi - lo);
            lw->num_scan_leaves++;
          }
        } else if (pr.kind == PG_PRED_DICT_SET) {
          if (pr.num_set_words < 0 || (pr.num_set_words > 0 && !pr.set_words)) return fail(PG_ERR_INVALID_ARGUMENT, "bad dictId set");
          int s = slot_for(lw, seg, pr.column);
          if (s < 0) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
          sp.cols[s].in_filter = 1;
          size_t bytes = (size_t)pr.num_set_words * 4;
          pg_status st = ensure_set(ctx, set_idx, bytes);
          if (st != PG_OK) return st;
          if (bytes) {
            // (the upload alone does not tie the query to this context: pg_execute_batch's shared launch carries the words itself)
            const bool tied = ctx->pre_enqueued;
            HIP_TRY(mark_pre_work(ctx));
            ctx->pre_enqueued = tied;
            HIP_TRY(hipMemcpyAsync(ctx->d_sets[set_idx], pr.set_words, bytes, hipMemcpyHostToDevice, ctx->stream));
            lw->set_leaves.push_back(Lowered::SetLeaf{ctx->d_sets[set_idx], pr.set_words, (uint32_t)bytes});
          }
          // the host words may go out of scope as soon as pg_execute returns; the copy is ordered before the kernel
          // on the same stream and the caller's buffer is read synchronously for pageable memory.
          L.kind = kLeafDictSet; L.col = s; L.set_words = ctx->d_sets[set_idx]; L.set_bytes = (int32_t)bytes;
          set_idx++;
          lw->num_scan_leaves++;
        } else if (pr.kind == PG_PRED_RAW_RANGE && col.stored_type == PG_TYPE_LONG) {
          // LongRawValueBasedRangePredicateEvaluator (RangePredicateEvaluatorFactory.java:411-446): inclusive int64 bounds
          if (pr.lo > pr.hi) { L.kind = kLeafMatchNone; }
          else {
            int s = slot_for(lw, seg, pr.column);
            if (s < 0) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
            sp.cols[s].in_filter = 1;
            const uint64_t span = (uint64_t)pr.hi - (uint64_t)pr.lo;
            L.kind = kLeafRawRange64; L.col = s;
            L.lo = (int32_t)(uint32_t)(uint64_t)pr.lo; L.lo_hi = (int32_t)(uint32_t)((uint64_t)pr.lo >> 32);
            L.span = (uint32_t)span; L.span_hi = (uint32_t)(span >> 32);
            lw->num_scan_leaves++;
          }
        } else if (pr.kind == PG_PRED_RAW_RANGE && col.stored_type != PG_TYPE_INT) {
          // Float / DoubleRawValueBasedRangePredicateEvaluator: lo / hi are the bit patterns of the inclusive double bounds.
          // value >= lo && value <= hi  <=>  key(value) in [key(lo'), key(hi')] with a zero lower bound taken as -0.0 and a zero
          // upper bound as +0.0 (primitive compares treat the zeros as equal; NaN values match nothing because their keys lie
          // outside [key(-inf), key(+inf)]).
          double dlo, dhi;
          memcpy(&dlo, &pr.lo, 8); memcpy(&dhi, &pr.hi, 8);
          if (!(dlo <= dhi)) { L.kind = kLeafMatchNone; }
          else {
            int s = slot_for(lw, seg, pr.column);
            if (s < 0) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
            sp.cols[s].in_filter = 1;
            if (dlo == 0.0) dlo = -0.0;
            if (dhi == 0.0) dhi = 0.0;
            auto key = [](double v) { long long bb; memcpy(&bb, &v, 8); return bb ^ ((bb >> 63) & 0x7FFFFFFFFFFFFFFFll); };
            const long long klo = key(dlo), khi = key(dhi);
            const uint64_t span = (uint64_t)khi - (uint64_t)klo;
            L.kind = col.stored_type == PG_TYPE_FLOAT ? kLeafRawRangeF32 : kLeafRawRangeF64; L.col = s;
            L.lo = (int32_t)(uint32_t)(uint64_t)klo; L.lo_hi = (int32_t)(uint32_t)((uint64_t)klo >> 32);
            L.span = (uint32_t)span; L.span_hi = (uint32_t)(span >> 32);
            lw->num_scan_leaves++;
          }
        } else if (pr.kind == PG_PRED_RAW_RANGE) {
          int64_t lo = std::max<int64_t>(pr.lo, std::numeric_limits<int32_t>::min());
          int64_t hi = std::min<int64_t>(pr.hi, std::numeric_limits<int32_t>::max());
          if (lo > hi) { L.kind = kLeafMatchNone; }
          else {
            int s = slot_for(lw, seg, pr.column);
            if (s < 0) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
            sp.cols[s].in_filter = 1;
            L.kind = kLeafRawRange; L.col = s; L.lo = (int32_t)lo; L.span = (uint32_t)(hi - lo);
            lw->num_scan_leaves++;
          }
        } else {
          return fail(PG_ERR_INVALID_ARGUMENT, "unknown predicate kind %d", pr.kind);
        }
      }
      depth++;
    } else if (fn.op == PG_FILTER_NOT) {
      if (depth < 1) return fail(PG_ERR_INVALID_ARGUMENT, "malformed filter tree (NOT without operand)");
    } else if (fn.op == PG_FILTER_AND || fn.op == PG_FILTER_OR) {
      if (fn.num_children < 1 || depth < fn.num_children) return fail(PG_ERR_INVALID_ARGUMENT, "malformed filter tree (node %d)", n);
      depth -= fn.num_children - 1;
    } else {
      return fail(PG_ERR_INVALID_ARGUMENT, "unknown filter op %d", fn.op);
    }
    max_depth = std::max(max_depth, depth);
  }
  if (!seq.empty() && depth != 1) return fail(PG_ERR_INVALID_ARGUMENT, "malformed filter tree (%d roots)", depth);
  if (max_depth > kStackDepth) return fail(PG_ERR_UNSUPPORTED, "filter tree deeper than %d", kStackDepth);
  sp.num_nodes = (int)seq.size();
  // index-driven query: postings first, the scan columns only for tiles the postings did not eliminate
  sp.lazy_columns = (num_bitmap_prefix > 0 && lw->num_scan_leaves > 0 && lazy_node >= 0) ? 1 : 0;
  sp.lazy_node = lazy_node;
  return PG_OK;
}

struct Geometry {
  int blocks = 1;
  int threads = kBlockThreads;
  size_t lds = 0;
  bool table_in_lds = false;
};

constexpr size_t kLdsBudget = 156 * 1024;     // of the 160 KiB per CU; leaves room for the runtime's own use

// Lays out the per-wave LDS region (one staging slot per packed column, 256 bytes per bitmap leaf, the gather queue),
// then picks tile size (32 or 16 steps) and workgroup size so that the most wavefronts stay resident per CU within the
// LDS budget, and sizes the grid to exactly what is co-resident (a persistent grid: a grid larger than residency runs
// in rounds and leaves the chip partly empty during the last one).
// Plan -> kernel parameter block: self-contained node records, statically indexed staging / aggregation descriptors.
void flatten_plan(Lowered* lw) {
  const PlanParams& pl = lw->plan;
  ScanParams& sp = lw->sp;
  sp.num_cols = pl.num_cols;
  sp.num_leaves = pl.num_leaves;
  sp.num_nodes = pl.num_nodes;
  sp.num_agg_cols = pl.num_agg_cols;
  sp.num_bitmap_leaves = 0;
  sp.lazy_columns = pl.lazy_columns;
  sp.lazy_node = pl.lazy_node;
  sp.num_stage = 0;
  for (int c = 0; c < pl.num_cols; ++c) {
    if (pl.cols[c].is_raw) continue;
    DevStage& st = sp.stage[sp.num_stage++];
    st.fwd = pl.cols[c].fwd; st.bits = pl.cols[c].bits; st.slot_off = pl.cols[c].slot_off; st.in_filter = pl.cols[c].in_filter; st.pad = 0;
  }
  for (int n = 0; n < pl.num_nodes; ++n) {
    DevNode& dn = sp.nodes[n];
    memset(&dn, 0, sizeof(dn));
    dn.op = pl.nodes[n].op; dn.flags = pl.nodes[n].flags; dn.num_children = pl.nodes[n].num_children;
    if (pl.nodes[n].op != PG_FILTER_LEAF) continue;
    const DevLeaf& L = pl.leaves[pl.nodes[n].leaf];
    dn.kind = L.kind; dn.exclusive = L.exclusive; dn.lo = L.lo; dn.span = L.span; dn.set_bytes = L.set_bytes; dn.set_words = L.set_words;
    dn.lds_off = L.lds_off;
    if (L.kind == kLeafBitmap) dn.set_words = reinterpret_cast<const uint32_t*>(L.bitmap);   // scan_private_kernel reads the bitmap as dwords
    if (L.kind == kLeafDictRange || L.kind == kLeafDictSet || L.kind == kLeafRawRange || L.kind >= kLeafRawRange64) {
      const DevColumn& c = pl.cols[L.col];
      dn.bits = c.bits; dn.slot_off = c.slot_off; dn.fwd = c.fwd;
    }
    if (L.kind >= kLeafRawRange64) { dn.lo_hi = L.lo_hi; dn.set_bytes = (int32_t)L.span_hi; }
  }
  for (int l = 0; l < pl.num_leaves; ++l) {
    if (pl.leaves[l].kind != kLeafBitmap) continue;
    sp.bitmaps[sp.num_bitmap_leaves] = pl.leaves[l].bitmap;
    sp.bitmap_lds_off[sp.num_bitmap_leaves] = pl.leaves[l].lds_off;
    sp.num_bitmap_leaves++;
  }
  for (int a = 0; a < pl.num_agg_cols; ++a) {
    const DevColumn& c = pl.cols[pl.agg_cols[a].col];
    DevAggCol& ac = sp.agg_cols[a];
    ac.need_sum = pl.agg_cols[a].need_sum; ac.need_minmax = pl.agg_cols[a].need_minmax;
    ac.bits = c.bits; ac.slot_off = c.slot_off; ac.is_raw = c.is_raw; ac.is_plane = c.is_plane; ac.dict_bytes = c.dict_bytes; ac.vkind = c.vkind;
    ac.fwd = c.fwd; ac.dict = c.dict;
  }
}

void finish_geometry(const pg_segment* seg, Lowered* lw, size_t table_bytes, bool need_queue, int max_block_waves, int wave_cap, Geometry* g) {
  ScanParams& sp = lw->sp;
  PlanParams& pl = lw->plan;
  sp.num_docs = seg->num_docs;
  sp.double_buffer = g_engine.double_buffer ? 1 : 0;
  sp.queue_cap = 256;
  const bool table_fits_lds = table_bytes > 0 && table_bytes <= 96 * 1024;
  int best_steps = 32, best_waves = 1, best_resident = -1;
  bool best_table = false;
  for (int steps : {32, 16}) {
    if (g_engine.tile_steps != 0 && steps != g_engine.tile_steps) continue;
    int stage = 0;
    for (int c = 0; c < pl.num_cols; ++c) if (!pl.cols[c].is_raw) stage += ((8 * pl.cols[c].bits * steps + 16) + 15) & ~15;
    stage = std::max(stage, 16);
    const size_t wave_lds = std::max<size_t>((size_t)(sp.double_buffer ? 2 : 1) * stage + 512 * pl.num_bitmap_leaves + (need_queue ? sp.queue_cap * 4 : 0), 128);
    for (int pass = 0; pass < 2; ++pass) {
      const bool in_lds = pass == 0 ? table_fits_lds : false;
      if (pass == 1 && table_fits_lds && best_table) break;     // an LDS-table configuration exists: keep it
      const size_t fixed = in_lds ? table_bytes : 0;
      for (int w = max_block_waves; w >= 1; w >>= 1) {
        const size_t lds_w = (size_t)w * wave_lds + fixed;
        if (lds_w > kLdsBudget) continue;
        const int resident = std::min(wave_cap, (int)(kLdsBudget / lds_w) * w);
        // prefer more resident wavefronts; on ties the larger tile, then the larger workgroup (fewer table copies)
        if (resident > best_resident) { best_resident = resident; best_steps = steps; best_waves = w; best_table = in_lds; }
      }
      if (!table_fits_lds) break;
    }
  }
  if (best_resident < 0) { best_steps = 16; best_waves = 1; best_table = false; }
  // final layout for the chosen tile size
  const int steps = best_steps;
  sp.tile_steps = steps;
  sp.num_tiles = (int)(((long long)seg->num_docs + 64 * steps - 1) / (64 * steps));
  int off = 0;
  for (int c = 0; c < pl.num_cols; ++c) {
    pl.cols[c].slot_off = off;
    if (!pl.cols[c].is_raw) off += ((8 * pl.cols[c].bits * steps + 16) + 15) & ~15;
  }
  sp.stage_bytes = std::max(off, 16);
  off = (sp.double_buffer ? 2 : 1) * sp.stage_bytes;
  sp.bitmap_off = off;
  int boff = 0;
  for (int l = 0; l < pl.num_leaves; ++l) if (pl.leaves[l].kind == kLeafBitmap) { pl.leaves[l].lds_off = boff; boff += 256; }
  sp.bitmap_bytes = boff;
  off += 2 * boff;                            // the posting words of the next tile are always prefetched
  sp.queue_off = off;
  if (need_queue) off += sp.queue_cap * 4;
  sp.wave_lds_bytes = std::max(off, 128);
  g->table_in_lds = best_table;
  g->threads = best_waves * 64;
  g->lds = (size_t)best_waves * sp.wave_lds_bytes + (best_table ? table_bytes : 0);
  g->lds = std::max(g->lds, sizeof(BlockPartial) * (size_t)best_waves + 16);      // the waves' records + the fold flag (publish_block_partial)
  int bpc = std::max(1, std::min(wave_cap / best_waves, (int)(kLdsBudget / g->lds)));
  if (g_engine.blocks_per_cu > 0) bpc = g_engine.blocks_per_cu;
  const long long want = ((long long)sp.num_tiles + best_waves - 1) / best_waves;
  const long long cap = (long long)seg->num_cus * bpc;
  g->blocks = (int)std::max<long long>(1, std::min(want, cap));
  flatten_plan(lw);
}

// MIN / MAX key of a column in the 32-bit domain or of any dictionary column (dictId) -> the value as the reference's
// holder sees it ((double) of the typed minimum / maximum).
double agg_value_double(const ColumnDev& col, int32_t key, bool plane) {
  if (col.encoding == PG_FWD_RAW_FIXED_BYTE) return (double)key;
  if (plane) return (double)(col.value_base + col.plane_base + col.plane_scale * (int64_t)key);   // 32-bit planes: base 0, scale 1, key = value
  return col.h_dict_f64[(size_t)key];
}
// An integer sum known exactly (128 bits): sum_i64 is it modulo 2^64, `sum` its correctly rounded double; exact while it
// fits int64 (the reference's double accumulation agrees bit for bit below 2^53 and to rounding above).
void set_integer_sum(pg_agg_value* v, __int128 t) {
  v->sum_i64 = (int64_t)(unsigned long long)(unsigned __int128)t;
  v->sum_exact = ((__int128)v->sum_i64 == t) ? 1 : 0;
  v->sum = (double)t;
}
// what count * base adds back to a sum accumulated in the 32-bit domain
int64_t sum_base(const ColumnDev& col, bool plane) { return col.value_base + (plane ? col.plane_base : 0); }
int64_t sum_scale(const ColumnDev& col, bool plane) { return plane ? col.plane_scale : 1; }
// 64-bit MIN / MAX key of a raw LONG / FLOAT / DOUBLE column (f64_order_key is its own inverse)
double key64_to_double(const ColumnDev& col, long long key) {
  if (col.vkind == kValI64) return (double)key;
  const long long b = key ^ ((key >> 63) & 0x7FFFFFFFFFFFFFFFll);
  double v;
  memcpy(&v, &b, 8);
  return v;
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  std::this_thread::yield();
#endif
}

// Bounded spin on a pinned sequence number (a query is tens of microseconds to a millisecond of device time: polling beats the
// stream-synchronise wake-up by ~10 us), then the blocking wait: a combine pool may have twice as many such threads as the host has
// cores, none of them may hold a core for longer than its kernel plausibly runs.  `docs`: what the launch scans -- the bound is
// 300 us + 1 ns per 1000 docs (a 1 B-row scan is ~0.6 ms).
template <typename Done>
pg_status wait_polled(hipStream_t stream, long long docs, Done&& done) {
  long long spins = 0;
  std::chrono::steady_clock::time_point until;
  while (!done()) {
    cpu_relax();
    if ((++spins & 0x3FF) != 0) continue;
    const auto now = std::chrono::steady_clock::now();
    if (spins == 0x400) { until = now + std::chrono::nanoseconds(300'000 + docs / 1000); continue; }
    if (now >= until || hipStreamQuery(stream) != hipErrorNotReady) { HIP_TRY(hipStreamSynchronize(stream)); break; }      // finished (or failed) without publishing: the stream's verdict
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return PG_OK;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char* pg_last_error(void) { return g_error.c_str(); }
const char* pg_version(void) { return "pinot_amd 0.1 (gfx950)"; }

pg_status pg_init(const pg_config* config) {
  std::lock_guard<std::mutex> lk(g_engine.mu);
  if (config && config->abi_version != PG_ABI_VERSION) return fail(PG_ERR_INVALID_ARGUMENT, "ABI version mismatch: got %d, built %d", config->abi_version, PG_ABI_VERSION);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) return fail(PG_ERR_DEVICE, "no HIP device available: %s", hipGetErrorString(e));
  g_engine.physical_devices = count;
  const char* ald = getenv("PINOT_GPU_ALIAS_DEVICES");
  g_engine.logical_devices = (ald && atoi(ald) > count) ? std::min(atoi(ald), 64) : count;
  int dev = config ? config->device_id : 0;
  if (dev < 0 || dev >= g_engine.logical_devices) return fail(PG_ERR_INVALID_ARGUMENT, "device %d out of range (have %d)", dev, g_engine.logical_devices);
  g_engine.device = dev;
  g_engine.blocks_per_cu = config ? config->blocks_per_cu : 0;
  g_engine.flags = config ? config->flags : 0;
  const char* nodma = getenv("PINOT_GPU_NO_DMA");
  g_engine.use_dma = !(nodma && nodma[0] == '1');
  const char* vp = getenv("PINOT_GPU_VALUE_PLANE");
  g_engine.value_plane = vp ? atoi(vp) : -1;
  const char* db = getenv("PINOT_GPU_DOUBLE_BUFFER");
  g_engine.double_buffer = db && db[0] == '1';
  const char* drv = getenv("PINOT_GPU_DIRECT_RESULT");
  g_engine.direct_result = !(drv && drv[0] == '0');
  const char* ffz = getenv("PINOT_GPU_FOLD_FINALIZE");
  g_engine.fold_finalize = ffz ? (ffz[0] == '0' ? 0 : 1) : -1;
  const char* foc = getenv("PINOT_GPU_FOLD_ONE_COUNTER");
  g_engine.fold_one_counter = foc ? (foc[0] == '0' ? 0 : 1) : 1;
  const char* prs = getenv("PINOT_GPU_POLL_RESULT");
  g_engine.poll_result = !(prs && prs[0] == '0');
  const char* lsk = getenv("PINOT_GPU_LANE_SKIP");
  g_engine.lane_skip = !(lsk && lsk[0] == '0');
  const char* bbc = getenv("PINOT_GPU_BATCH_BLOCKS_PER_CU");
  g_engine.batch_blocks_per_cu_forced = bbc != nullptr;
  g_engine.batch_blocks_per_cu = (bbc && atoi(bbc) > 0) ? atoi(bbc) : 4;      // measured on 64 x 10 M rows: 2 / 4 / 8 / 16 / 32 / 64 -> 0.77 / 0.55 / 0.57 / 0.59 / 0.61 / 0.65 ms
  const char* ssp = getenv("PINOT_GPU_SCAN_SPARSE");
  g_engine.scan_sparse = !(ssp && ssp[0] == '0');
  const char* ssm = getenv("PINOT_GPU_SCAN_SIMPLE");
  g_engine.scan_simple = !(ssm && ssm[0] == '0');
  const char* srw = getenv("PINOT_GPU_SCAN_RAW");
  g_engine.scan_raw = !(srw && srw[0] == '0');
  const char* bla = getenv("PINOT_GPU_BATCH_LAUNCH");
  g_engine.batch_launch = !(bla && bla[0] == '0');
  auto env_on = [](const char* name) { const char* v = getenv(name); return !(v && v[0] == '0'); };
  g_engine.lean_batch = env_on("PINOT_GPU_LEAN_BATCH");
  g_engine.set_lds = env_on("PINOT_GPU_SET_LDS");
  g_engine.partition_two_level = env_on("PINOT_GPU_PARTITION_TWO_LEVEL");
  g_engine.fsm_perm = env_on("PINOT_GPU_FSM_PERM");
  g_engine.fsm_stats = env_on("PINOT_GPU_FSM_STATS");
  g_engine.fsm_fused = env_on("PINOT_GPU_FSM_FUSED");
  g_engine.fsm_episodes = env_on("PINOT_GPU_FSM_EPISODES");
  g_engine.index_gather = env_on("PINOT_GPU_INDEX_GATHER");
  g_engine.batch_index = env_on("PINOT_GPU_BATCH_INDEX");
  { const char* iaw = getenv("PINOT_GPU_INDEX_AND_WAVES"); g_engine.index_and_waves = iaw ? std::max(-64, std::min(32, atoi(iaw))) : 0; }
  g_engine.group_one_launch = env_on("PINOT_GPU_GROUP_ONE_LAUNCH");
  g_engine.plan_cache = env_on("PINOT_GPU_PLAN_CACHE");
  g_engine.group_publish = env_on("PINOT_GPU_GROUP_PUBLISH");
  const char* bmo = getenv("PINOT_GPU_BATCH_MORE");
  g_engine.batch_more = !(bmo && bmo[0] == '0');
  const char* bgr = getenv("PINOT_GPU_BATCH_GROUP");
  g_engine.batch_group = !(bgr && bgr[0] == '0');
  const char* bhi = getenv("PINOT_GPU_BATCH_HIST");
  g_engine.batch_hist = !(bhi && bhi[0] == '0');
  const char* lp2 = getenv("PINOT_GPU_LEAP2");
  g_engine.leap2 = !(lp2 && lp2[0] == '0');
  const char* spl = getenv("PINOT_GPU_SPARSE_LANES");
  g_engine.sparse_lanes = spl ? std::max(0, std::min(64, atoi(spl))) : 32;
  const char* pgv = getenv("PINOT_GPU_PLANE_GCD");
  g_engine.plane_gcd = !(pgv && pgv[0] == '0');
  const char* spv = getenv("PINOT_GPU_SCAN_PRIVATE");
  g_engine.scan_private = !(spv && spv[0] == '0');
  const char* gpv = getenv("PINOT_GPU_GROUP_PRIVATE");
  g_engine.group_private = !(gpv && gpv[0] == '0');
  const char* grv = getenv("PINOT_GPU_GROUP_REPLICAS");
  g_engine.group_log_replicas = grv ? std::max(0, std::min(4, atoi(grv))) : 3;
  const char* stp = getenv("PINOT_GPU_SCAN_TYPED_PRIVATE");
  g_engine.scan_typed_private = !(stp && stp[0] == '0');
  const char* gpt = getenv("PINOT_GPU_GROUP_PARTITION");
  g_engine.group_partition = !(gpt && gpt[0] == '0');
  g_engine.partition_min_docs = (gpt && gpt[0] == 'f') ? 0 : (1ll << 22);
  const char* sns = getenv("PINOT_GPU_SCAN_NARROW_SINGLE");
  g_engine.scan_narrow_single = !(sns && sns[0] == '0');
  const char* snw = getenv("PINOT_GPU_SCAN_NARROW");
  g_engine.scan_narrow = !(snw && snw[0] == '0');
  const char* gtb = getenv("PINOT_GPU_GROUP_TABLE_BYTES");
  g_engine.group_table_bytes = (gtb && atoll(gtb) > 0) ? (unsigned long long)atoll(gtb) : (64ull << 30);
  const char* gpk = getenv("PINOT_GPU_GROUP_PACK");
  g_engine.group_pack = !(gpk && gpk[0] == '0');
  const char* gw = getenv("PINOT_GPU_GROUP_WAVES");
  g_engine.group_waves = (gw && atoi(gw) > 0) ? atoi(gw) : 0;
  const char* hs = getenv("PINOT_GPU_HIST");
  g_engine.hist = hs ? atoi(hs) : -1;
  const char* hbl = getenv("PINOT_GPU_HIST_BLOCKS");
  g_engine.hist_blocks = (hbl && atoi(hbl) > 0) ? atoi(hbl) : 0;
  const char* hg = getenv("PINOT_GPU_HIST_GUARD");
  g_engine.hist_guard = hg && hg[0] == '1';
  const char* hb = getenv("PINOT_GPU_HIST_BITS");
  g_engine.hist_bits = (hb && (atoi(hb) == 8 || atoi(hb) == 16)) ? atoi(hb) : 0;
  const char* ts = getenv("PINOT_GPU_TILE_STEPS");
  g_engine.tile_steps = (ts && (atoi(ts) == 16 || atoi(ts) == 32)) ? atoi(ts) : 0;
  // (pg_init may be called again with another environment -- tools/ab_r3.py, tests: every switch goes back to its default first)
  g_engine.raw64_coalesced = true; g_engine.wide_plane = -1; g_engine.partition_stats_cache = true; g_engine.partition_packed = true;
  g_engine.plane_async = true; g_engine.staged_h2d = true; g_engine.exact_stats_docs = 64ll << 20;
  const char* r64 = getenv("PINOT_GPU_RAW64_COALESCED");
  if (r64) g_engine.raw64_coalesced = atoi(r64) != 0;
  const char* wpl = getenv("PINOT_GPU_WIDE_PLANE");
  if (wpl) g_engine.wide_plane = atoi(wpl);
  const char* psc = getenv("PINOT_GPU_PARTITION_STATS_CACHE");
  if (psc) g_engine.partition_stats_cache = atoi(psc) != 0;
  const char* ppk = getenv("PINOT_GPU_PARTITION_PACKED");
  if (ppk) g_engine.partition_packed = atoi(ppk) != 0;
  const char* pas = getenv("PINOT_GPU_PLANE_ASYNC");
  if (pas) g_engine.plane_async = atoi(pas) != 0;
  const char* sh2d = getenv("PINOT_GPU_STAGED_H2D");
  if (sh2d) g_engine.staged_h2d = atoi(sh2d) != 0;
  {
    // value-plane budget: pg_config, then the environment; default a quarter of the device's memory
    uint64_t budget = config ? config->plane_budget_bytes : 0;
    const char* pb = getenv("PINOT_GPU_PLANE_BUDGET_BYTES");
    if (pb) budget = strtoull(pb, nullptr, 10);
    if (budget == 0) {
      size_t free_bytes = 0, total_bytes = 0;
      (void)hipSetDevice(phys_device(dev));
      if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) budget = total_bytes / 4; else budget = 64ull << 30;
    }
    std::lock_guard<std::mutex> lk(g_planes.mu);
    g_planes.budget_bytes = budget;
  }
  const char* esd = getenv("PINOT_GPU_EXACT_FILTER_STATS_DOCS");
  if (esd) g_engine.exact_stats_docs = atoll(esd);
  const char* bpc = getenv("PINOT_GPU_BLOCKS_PER_CU");
  if (bpc && atoi(bpc) > 0) g_engine.blocks_per_cu = atoi(bpc);
  g_engine.epoch.fetch_add(1, std::memory_order_release);      // (what segments remember of earlier lowerings was made under the previous settings)
  g_engine.initialized = true;
  return PG_OK;
}

pg_status pg_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_engine.mu);
  g_engine.initialized = false;
  { std::lock_guard<std::mutex> sl(g_stage.mu); destroy_stage_pool_locked(); }
  return PG_OK;
}

pg_status pg_device_info(int32_t device_id, char* arch_name, int32_t arch_name_len, int32_t* num_cus, uint64_t* hbm_bytes) {
  if (!device_id_ok(device_id)) return fail(PG_ERR_INVALID_ARGUMENT, "device id %d out of range", device_id);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, phys_device(device_id)));
  if (arch_name && arch_name_len > 0) { strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1); arch_name[arch_name_len - 1] = 0; }
  if (num_cus) *num_cus = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return PG_OK;
}

pg_status pg_device_count(int32_t* out_devices, int32_t* out_physical) {
  if (!g_engine.initialized) return fail(PG_ERR_NOT_INITIALIZED, "pg_init has not been called");
  if (out_devices) *out_devices = g_engine.logical_devices;
  if (out_physical) *out_physical = g_engine.physical_devices;
  return PG_OK;
}

pg_status pg_measure_stream_read(int32_t device_id, uint64_t bytes, int32_t launches, double* out_gbps) {
  if (!out_gbps || bytes < (1u << 20) || launches < 1) return fail(PG_ERR_INVALID_ARGUMENT, "bad stream probe arguments");
  if (!device_id_ok(device_id)) return fail(PG_ERR_INVALID_ARGUMENT, "device id %d out of range", device_id);
  HIP_TRY(hipSetDevice(phys_device(device_id)));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, phys_device(device_id)));
  uint8_t* buf = nullptr;
  unsigned long long* sink = nullptr;
  HIP_TRY(hipMalloc((void**)&buf, bytes));
  hipError_t e = hipMalloc((void**)&sink, 8);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e == hipSuccess) e = hipMemset(buf, 0, bytes);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  float best = 0.f;
  if (e == hipSuccess) {
    const unsigned blocks = (unsigned)prop.multiProcessorCount * 8u;        // 2048 threads per CU: a persistent grid, like the scan kernels
    for (int i = 0; i < launches + 2 && e == hipSuccess; ++i) {
      (void)hipEventRecord(e0, 0);
      stream_read_probe_kernel<<<dim3(blocks), dim3(kBlockThreads), 0, 0>>>(reinterpret_cast<const uint4*>(buf), (size_t)(bytes / 16), sink);
      (void)hipEventRecord(e1, 0);
      e = hipEventSynchronize(e1);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (i >= 2 && ms > 0.f) best = std::max(best, (float)((double)bytes / (ms * 1e-3) / 1e9));
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(buf);
  if (sink) (void)hipFree(sink);
  if (e != hipSuccess) return fail(PG_ERR_DEVICE, "stream probe: %s", hipGetErrorString(e));
  *out_gbps = best;
  return PG_OK;
}

pg_status pg_segment_open(const pg_segment_desc* desc, pg_segment** out_segment) {
  if (!g_engine.initialized) return fail(PG_ERR_NOT_INITIALIZED, "pg_init has not been called");
  if (!desc || !out_segment) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  if (desc->num_docs < 0 || desc->num_columns < 0 || (desc->num_columns > 0 && !desc->columns)) return fail(PG_ERR_INVALID_ARGUMENT, "bad segment descriptor");
  pg_segment* seg = new pg_segment();
  seg->device = desc->device_id >= 0 ? desc->device_id : g_engine.device;
  if (seg->device >= g_engine.logical_devices) { const int d = seg->device; delete seg; return fail(PG_ERR_INVALID_ARGUMENT, "pg_segment_desc.device_id %d out of range (have %d)", d, g_engine.logical_devices); }
  seg->num_docs = desc->num_docs;
  seg->num_tiles = (int)(((long long)desc->num_docs + kMaxTileDocs - 1) / kMaxTileDocs);   // 2048-doc tiles (buffer padding unit)
  seg->name = desc->name ? desc->name : "";
  pg_status st = PG_OK;
  auto bail = [&](pg_status s) { free_segment(seg); return s; };
  {
    hipError_t e = hipSetDevice(phys_device(seg->device));
    if (e != hipSuccess) return bail(fail(PG_ERR_DEVICE, "hipSetDevice(%d): %s", seg->device, hipGetErrorString(e)));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, phys_device(seg->device));
    if (e != hipSuccess) return bail(fail(PG_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e)));
    seg->num_cus = prop.multiProcessorCount;
    // Test switch: every grid of this segment is sized as if the device had this many CUs.  With 1, a 100 000-doc segment is ~50 tiles for
    // ~32 waves -- every kernel's tile loop runs more than once per wave on the suite's SMALL segments (a kernel that was
    // wrong from a wave's second tile on passed every test below 4.5 M docs on the full grid).
    const char* tc = getenv("PINOT_GPU_TEST_CUS");
    if (tc && atoi(tc) > 0) seg->num_cus = atoi(tc);
  }
  seg->cols.resize((size_t)desc->num_columns);
  for (int i = 0; i < desc->num_columns; ++i) {
    const pg_column_desc& cd = desc->columns[i];
    ColumnDev& col = seg->cols[(size_t)i];
    col.name = cd.name ? cd.name : "";
    col.stored_type = cd.stored_type;
    col.encoding = cd.fwd_encoding;
    col.bits = cd.bits_per_value;
    col.cardinality = cd.cardinality;
    if (cd.stored_type < PG_TYPE_INT || cd.stored_type > PG_TYPE_DOUBLE) return bail(fail(PG_ERR_UNSUPPORTED, "column %s: stored type %d is not offloaded", col.name.c_str(), cd.stored_type));
    const int value_bytes = (cd.stored_type == PG_TYPE_INT || cd.stored_type == PG_TYPE_FLOAT) ? 4 : 8;
    if (!cd.fwd_data && desc->num_docs > 0) return bail(fail(PG_ERR_INVALID_ARGUMENT, "column %s: missing forward index", col.name.c_str()));
    const uint8_t* fwd = (const uint8_t*)cd.fwd_data;
    if (cd.fwd_encoding == PG_FWD_FIXED_BIT_DICT) {
      if (cd.bits_per_value < 1 || cd.bits_per_value > 31) return bail(fail(PG_ERR_INVALID_ARGUMENT, "column %s: bits_per_value %d", col.name.c_str(), cd.bits_per_value));
      // FixedBitIntReaderWriter precondition (segl/io/util/FixedBitIntReaderWriter.java:30-33), in 64-bit
      const uint64_t expect = ((uint64_t)desc->num_docs * (uint64_t)cd.bits_per_value + 7) / 8;
      if (cd.fwd_size != expect) return bail(fail(PG_ERR_INVALID_ARGUMENT, "column %s: forward index is %llu bytes, expected %llu", col.name.c_str(), (unsigned long long)cd.fwd_size, (unsigned long long)expect));
      // BaseImmutableDictionary precondition (BaseImmutableDictionary.java:51-53)
      if (cd.cardinality < 1 || !cd.dict_data || cd.dict_size != (uint64_t)cd.cardinality * (uint64_t)value_bytes) return bail(fail(PG_ERR_INVALID_ARGUMENT, "column %s: dictionary buffer size mismatch", col.name.c_str()));
      col.fwd_alloc_bytes = (size_t)std::max(seg->num_tiles, 1) * 256 * (size_t)cd.bits_per_value + 64;
      hipError_t e = hipMalloc((void**)&col.d_fwd_alloc, col.fwd_alloc_bytes);
      if (e != hipSuccess) return bail(fail(PG_ERR_OUT_OF_MEMORY, "column %s: hipMalloc(%zu): %s", col.name.c_str(), col.fwd_alloc_bytes, hipGetErrorString(e)));
      col.d_fwd = col.d_fwd_alloc;
      // zero the padding past the file bytes so tail tiles decode deterministic (masked) values
      size_t tail = col.fwd_alloc_bytes - (size_t)cd.fwd_size;
      e = hipMemset(col.d_fwd_alloc + cd.fwd_size, 0, tail);
      if (e == hipSuccess && cd.fwd_size) e = h2d_copy(col.d_fwd_alloc, fwd, (size_t)cd.fwd_size, phys_device(seg->device));
      if (e != hipSuccess) return bail(fail(PG_ERR_DEVICE, "column %s: H2D copy: %s", col.name.c_str(), hipGetErrorString(e)));
      // Dictionary values -> host order (IntDictionary / LongDictionary / FloatDictionary / DoubleDictionary: C big-endian
      // fixed-width values, ascending; FixedByteValueReaderWriter.getInt/getLong/getFloat/getDouble)
      const size_t C = (size_t)cd.cardinality;
      const uint8_t* dp = (const uint8_t*)cd.dict_data;
      col.h_dict_f64.resize(C);
      std::vector<unsigned long long> wide;
      if (cd.stored_type == PG_TYPE_INT) {
        col.h_dict.resize(C); col.h_dict_i64.resize(C);
        for (size_t d = 0; d < C; ++d) { col.h_dict[d] = (int32_t)be32(dp + 4 * d); col.h_dict_i64[d] = col.h_dict[d]; col.h_dict_f64[d] = (double)col.h_dict[d]; }
      } else if (cd.stored_type == PG_TYPE_LONG) {
        col.h_dict_i64.resize(C);
        for (size_t d = 0; d < C; ++d) {
          col.h_dict_i64[d] = (int64_t)(((uint64_t)be32(dp + 8 * d) << 32) | (uint64_t)be32(dp + 8 * d + 4));
          col.h_dict_f64[d] = (double)col.h_dict_i64[d];
        }
        const uint64_t range = (uint64_t)col.h_dict_i64[C - 1] - (uint64_t)col.h_dict_i64[0];
        if (col.h_dict_i64[C - 1] >= col.h_dict_i64[0] && range < (1ull << 31)) {
          // offset dictionary: the whole 32-bit machinery applies to (value - min)
          col.value_base = col.h_dict_i64[0];
          col.h_dict.resize(C);
          for (size_t d = 0; d < C; ++d) col.h_dict[d] = (int32_t)(col.h_dict_i64[d] - col.value_base);
        } else {
          col.vkind = kValI64;
          wide.resize(C);
          for (size_t d = 0; d < C; ++d) wide[d] = (unsigned long long)col.h_dict_i64[d];
        }
      } else {
        col.vkind = kValF64;
        wide.resize(C);
        for (size_t d = 0; d < C; ++d) {
          double v;
          if (cd.stored_type == PG_TYPE_FLOAT) { const uint32_t b = be32(dp + 4 * d); float f; memcpy(&f, &b, 4); v = (double)f; }
          else { const uint64_t b = ((uint64_t)be32(dp + 8 * d) << 32) | (uint64_t)be32(dp + 8 * d + 4); memcpy(&v, &b, 8); }
          col.h_dict_f64[d] = v;
          memcpy(&wide[d], &v, 8);
        }
      }
      if (col.vkind == kValI32) {
        const PlaneShape ps = compute_plane_shape(col);
        col.shape_base = ps.base; col.shape_scale = ps.scale; col.shape_bits = ps.bits; col.shape_is_fwd = ps.is_fwd;
        e = hipMalloc((void**)&col.d_dict, C * 4);
        if (e == hipSuccess) e = hipMemcpy(col.d_dict, col.h_dict.data(), C * 4, hipMemcpyHostToDevice);
      } else {
        e = hipMalloc((void**)&col.d_dict64, C * 8);
        if (e == hipSuccess) e = hipMemcpy(col.d_dict64, wide.data(), C * 8, hipMemcpyHostToDevice);
      }
      if (e != hipSuccess) return bail(fail(PG_ERR_DEVICE, "column %s: dictionary upload: %s", col.name.c_str(), hipGetErrorString(e)));
      seg->device_bytes += col.fwd_alloc_bytes + C * (col.vkind == kValI32 ? 4 : 8);
    } else if (cd.fwd_encoding == PG_FWD_RAW_FIXED_BYTE) {
      // BaseChunkForwardIndexReader header (BaseChunkForwardIndexReader.java:61-111)
      if (cd.fwd_size < 16) return bail(fail(PG_ERR_INVALID_ARGUMENT, "column %s: raw forward index too small", col.name.c_str()));
      int off = 0;
      const int version = (int)be32(fwd + off); off += 4;
      const int num_chunks = (int)be32(fwd + off); off += 4;
      off += 4;  // numDocsPerChunk
      const int entry_size = (int)be32(fwd + off); off += 4;
      int data_header_start = off;
      int compression = 2;
      if (version > 1) {
        if (cd.fwd_size < 28) return bail(fail(PG_ERR_INVALID_ARGUMENT, "column %s: raw forward index header truncated", col.name.c_str()));
        off += 4;  // totalDocs
        compression = (int)be32(fwd + off); off += 4;
        data_header_start = (int)be32(fwd + off);
      }
      if (compression != 0) return bail(fail(PG_ERR_UNSUPPORTED, "column %s: only PASS_THROUGH raw chunks are offloaded (compressionType=%d)", col.name.c_str(), compression));
      if (entry_size != value_bytes) return bail(fail(PG_ERR_UNSUPPORTED, "column %s: raw entry size %d for stored type %d", col.name.c_str(), entry_size, cd.stored_type));
      col.vkind = cd.stored_type == PG_TYPE_INT ? kValI32 : (cd.stored_type == PG_TYPE_LONG ? kValI64 : (cd.stored_type == PG_TYPE_FLOAT ? kValF32 : kValF64));
      const uint64_t raw_start = (uint64_t)data_header_start + (uint64_t)num_chunks * (version <= 2 ? 4 : 8);
      if (raw_start + (uint64_t)desc->num_docs * (uint64_t)value_bytes > cd.fwd_size) return bail(fail(PG_ERR_INVALID_ARGUMENT, "column %s: raw forward index shorter than numDocs", col.name.c_str()));
      // padded so that whole 2048-doc tiles can be read past numDocs (the lane-private kernels never clamp docIds)
      // The file is placed so that the VALUES (not the header) start on a 256-byte boundary: a lane's 32 docs are then whole,
      // aligned 16-byte loads and a tile's rows whole cache lines (the header is 28 + 4 * numChunks bytes: any multiple of 4).
      const size_t lead = (256 - (size_t)(raw_start % 256)) % 256;
      col.fwd_alloc_bytes = lead + std::max<size_t>((size_t)cd.fwd_size, (size_t)raw_start + (size_t)std::max(seg->num_tiles, 1) * 2048 * (size_t)value_bytes) + 64;
      hipError_t e = hipMalloc((void**)&col.d_fwd_alloc, col.fwd_alloc_bytes);
      if (e != hipSuccess) return bail(fail(PG_ERR_OUT_OF_MEMORY, "column %s: hipMalloc(%zu): %s", col.name.c_str(), col.fwd_alloc_bytes, hipGetErrorString(e)));
      e = hipMemset(col.d_fwd_alloc, 0, col.fwd_alloc_bytes);
      if (e == hipSuccess) e = h2d_copy(col.d_fwd_alloc + lead, fwd, (size_t)cd.fwd_size, phys_device(seg->device));
      if (e != hipSuccess) return bail(fail(PG_ERR_DEVICE, "column %s: H2D copy: %s", col.name.c_str(), hipGetErrorString(e)));
      col.d_fwd = col.d_fwd_alloc + lead + raw_start;
      col.bits = 32;
      col.cardinality = 0;
      seg->device_bytes += col.fwd_alloc_bytes;
    } else {
      return bail(fail(PG_ERR_UNSUPPORTED, "column %s: unknown forward-index encoding %d", col.name.c_str(), cd.fwd_encoding));
    }
    if (cd.inv_data && cd.inv_size && cd.fwd_encoding == PG_FWD_FIXED_BIT_DICT) {
      // BitmapInvertedIndexReader (BitmapInvertedIndexReader.java:45-62)
      const uint8_t* inv = (const uint8_t*)cd.inv_data;
      const uint64_t offsets_end = ((uint64_t)cd.cardinality + 1) * 4;
      if (cd.inv_size < offsets_end) return bail(fail(PG_ERR_INVALID_ARGUMENT, "column %s: inverted index too small", col.name.c_str()));
      const uint64_t first_offset = be32(inv);
      col.posting_first.assign((size_t)cd.cardinality + 1, 0);
      for (int d = 0; d < cd.cardinality; ++d) {
        col.posting_first[(size_t)d] = (int64_t)col.h_dir.size();
        const uint64_t o0 = be32(inv + 4 * (size_t)d), o1 = be32(inv + 4 * (size_t)(d + 1));
        if (o1 < o0 || offsets_end + (o1 - first_offset) > cd.inv_size) return bail(fail(PG_ERR_INVALID_ARGUMENT, "column %s: inverted index offsets out of range", col.name.c_str()));
        st = parse_roaring(inv, offsets_end + (o0 - first_offset), o1 - o0, &col.h_dir);
        if (st != PG_OK) return bail(st);
      }
      col.posting_first[(size_t)cd.cardinality] = (int64_t)col.h_dir.size();
      col.posting_docs.assign((size_t)cd.cardinality, 0);
      for (int d = 0; d < cd.cardinality; ++d)
        for (int64_t k = col.posting_first[(size_t)d]; k < col.posting_first[(size_t)d + 1]; ++k) col.posting_docs[(size_t)d] += (int64_t)col.h_dir[(size_t)k].cardinality;
      hipError_t e = hipMalloc((void**)&col.d_inv, (size_t)cd.inv_size + 64);
      if (e == hipSuccess) e = h2d_copy(col.d_inv, inv, (size_t)cd.inv_size, phys_device(seg->device));
      if (e == hipSuccess && !col.h_dir.empty()) {
        e = hipMalloc((void**)&col.d_dir, col.h_dir.size() * sizeof(DevContainer));
        if (e == hipSuccess) e = hipMemcpy(col.d_dir, col.h_dir.data(), col.h_dir.size() * sizeof(DevContainer), hipMemcpyHostToDevice);
      }
      if (e != hipSuccess) return bail(fail(PG_ERR_DEVICE, "column %s: inverted index upload: %s", col.name.c_str(), hipGetErrorString(e)));
      col.inv_size = cd.inv_size;
      seg->device_bytes += cd.inv_size + col.h_dir.size() * sizeof(DevContainer);
    }
    if (cd.null_data && cd.null_size) {
      // NullValueVectorReaderImpl.getNullBitmap (NullValueVectorReaderImpl.java:44-46): the whole buffer is one RoaringBitmap.  It is
      // expanded once into a doc-order bitmap that IS_NULL leaves and the null-skipping aggregation lanes read like a posting.
      std::vector<DevContainer> dir;
      st = parse_roaring((const uint8_t*)cd.null_data, 0, cd.null_size, &dir);
      if (st != PG_OK) return bail(st);
      const long long words = (long long)std::max(seg->num_tiles, 1) * kMaxTileSteps;
      int64_t nulls = 0;
      std::vector<DevContainer> kept;
      for (const DevContainer& dc : dir) {
        if ((long long)dc.key * 1024 >= words) continue;   // containers beyond numDocs cannot hold a docId of this segment
        kept.push_back(dc);
        nulls += dc.cardinality;
      }
      if (nulls > 0) {
        uint8_t* d_bytes = nullptr;
        DevContainer* d_cont = nullptr;
        hipError_t e = hipMalloc((void**)&d_bytes, (size_t)cd.null_size + 16);
        if (e == hipSuccess) e = hipMemcpy(d_bytes, cd.null_data, (size_t)cd.null_size, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&d_cont, kept.size() * sizeof(DevContainer));
        if (e == hipSuccess) e = hipMemcpy(d_cont, kept.data(), kept.size() * sizeof(DevContainer), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&col.d_null_bitmap, (size_t)words * 8);
        if (e == hipSuccess) {
          roaring_expand_kernel<<<dim3((unsigned)((words + 1023) / 1024)), dim3(kBlockThreads), 0, 0>>>(d_bytes, d_cont, 0, (int)kept.size(), col.d_null_bitmap, words, 0);
          e = hipDeviceSynchronize();
        }
        if (d_bytes) (void)hipFree(d_bytes);
        if (d_cont) (void)hipFree(d_cont);
        if (e != hipSuccess) return bail(fail(PG_ERR_DEVICE, "column %s: null value vector upload: %s", col.name.c_str(), hipGetErrorString(e)));
        col.num_nulls = nulls;
        seg->device_bytes += (uint64_t)words * 8;
      }
    }
  }
  seg->num_user_cols = (int)seg->cols.size();
  seg->cols.reserve(seg->cols.size() * 3);      // hidden images are appended (at most two per column); the vector never reallocates after open
  // Raw INT / LONG columns as group keys: the value range (one pass over the resident column), and -- when it fits the int dictId
  // domain -- a placeholder for the column's KEY IMAGE, the fixed-bit stream of (value - min) that ensure_key_image builds the first
  // time the column is grouped by (pg_kernels.h, build_raw_key_image_kernel).  Nullable ones are built right away: their null-key image
  // below is made from it.
  {
    long long* d_mm = nullptr;
    for (int i = 0; i < seg->num_user_cols; ++i) {
      ColumnDev& col = seg->cols[(size_t)i];
      if (col.encoding != PG_FWD_RAW_FIXED_BYTE || seg->num_docs <= 0) continue;
      auto rank_placeholder = [&] {
        ColumnDev image;
        image.name = col.name + "$rankimage";
        image.stored_type = col.stored_type; image.encoding = PG_FWD_FIXED_BIT_DICT; image.vkind = kValI32;
        image.cardinality = 0; image.bits = 0; image.rank_image = true;
        image.key_image_of = i;
        col.keyimage_column = (int)seg->cols.size();
        seg->cols.push_back(std::move(image));
      };
      if (col.vkind != kValI32 && col.vkind != kValI64) { rank_placeholder(); continue; }      // FLOAT / DOUBLE: keyed by value through a rank image (pg_rank_image.h)
      long long mm[2] = {0x7FFFFFFFFFFFFFFFll, (long long)0x8000000000000000ull};
      hipError_t e = d_mm ? hipSuccess : hipMalloc((void**)&d_mm, 16);
      if (e == hipSuccess) e = hipMemcpy(d_mm, mm, 16, hipMemcpyHostToDevice);
      if (e == hipSuccess) {
        raw_min_max_kernel<<<dim3((unsigned)std::max(1, std::min(seg->num_docs / 1024 + 1, seg->num_cus * 8))), dim3(256), 0, 0>>>(col.d_fwd, col.vkind == kValI32 ? 4 : 8, seg->num_docs, d_mm);
        e = hipMemcpy(mm, d_mm, 16, hipMemcpyDeviceToHost);
      }
      if (e != hipSuccess) { if (d_mm) (void)hipFree(d_mm); return bail(fail(PG_ERR_DEVICE, "column %s: value range: %s", col.name.c_str(), hipGetErrorString(e))); }
      col.raw_min = mm[0]; col.raw_max = mm[1];
      if ((unsigned long long)mm[1] - (unsigned long long)mm[0] >= 0x7FFFFFFEull) { rank_placeholder(); continue; }      // max - min + 1 is not an int (with room for a null digit): a rank image instead
      ColumnDev image;
      image.name = col.name + "$keyimage";
      image.stored_type = col.stored_type; image.encoding = PG_FWD_FIXED_BIT_DICT; image.vkind = kValI32;
      image.cardinality = (int)(mm[1] - mm[0] + 1);
      image.bits = 1;
      while (image.bits < 31 && (1ll << image.bits) < (long long)image.cardinality) ++image.bits;      // PinotDataBitSet.getNumBitsPerValue(cardinality - 1)
      image.key_image_of = i; image.key_base = mm[0];
      col.keyimage_column = (int)seg->cols.size();
      seg->cols.push_back(std::move(image));
    }
    if (d_mm) (void)hipFree(d_mm);
    for (int i = 0; i < seg->num_user_cols; ++i) {
      if (seg->cols[(size_t)i].keyimage_column < 0 || !seg->cols[(size_t)i].d_null_bitmap) continue;
      if (seg->cols[(size_t)seg->cols[(size_t)i].keyimage_column].rank_image) continue;      // (no null-key image of a rank image: such a key under null handling stays with the CPU plan)
      const pg_status kst = ensure_key_image(seg, i, nullptr);
      if (kst != PG_OK) return bail(kst);
    }
  }
  // GROUP BY under enableNullHandling treats NULL as a key of its own (DefaultGroupByExecutor.java:106-121: the no-dictionary key
  // generators).  A nullable dictionary column therefore gets a second forward index whose dictId is `cardinality` wherever the doc is
  // null: the group-by kernels then need no notion of null (execute_null_handling points the key at this image).  A nullable raw
  // INT / LONG column gets the same image, made from its key image.
  for (int i = 0; i < seg->num_user_cols; ++i) {
    if (!seg->cols[(size_t)i].d_null_bitmap) continue;
    const int src = seg->cols[(size_t)i].encoding == PG_FWD_FIXED_BIT_DICT ? i : seg->cols[(size_t)i].keyimage_column;
    if (src < 0 || seg->cols[(size_t)src].rank_image) continue;
    int bits_out = 1;
    while (bits_out < 31 && (1ll << bits_out) <= (long long)seg->cols[(size_t)src].cardinality) ++bits_out;      // PinotDataBitSet.getNumBitsPerValue(cardinality)
    ColumnDev image;
    const ColumnDev& col = seg->cols[(size_t)src];
    image.key_image_of = col.key_image_of; image.key_base = col.key_base;      // (null-key image of a key image: still min + digit, digit == cardinality - 1 is NULL)
    image.name = seg->cols[(size_t)i].name + "$nullkey";
    image.stored_type = col.stored_type; image.encoding = col.encoding; image.vkind = col.vkind; image.value_base = col.value_base;
    image.bits = bits_out;
    image.cardinality = col.cardinality + 1;
    image.h_dict = col.h_dict; image.h_dict.push_back(0);
    image.h_dict_f64 = col.h_dict_f64; image.h_dict_f64.push_back(std::numeric_limits<double>::quiet_NaN());
    image.h_dict_i64 = col.h_dict_i64; if (!image.h_dict_i64.empty()) image.h_dict_i64.push_back(0);
    image.d_dict = col.d_dict; image.d_dict64 = col.d_dict64; image.borrows_dictionary = true;      // never gathered: the image is only ever a group key
    image.fwd_alloc_bytes = (size_t)std::max(seg->num_tiles, 1) * 256 * (size_t)bits_out + 64;
    hipError_t e = hipMalloc((void**)&image.d_fwd_alloc, image.fwd_alloc_bytes);
    if (e == hipSuccess) e = hipMemset(image.d_fwd_alloc, 0, image.fwd_alloc_bytes);
    if (e == hipSuccess) {
      build_nullkey_fwd_kernel<<<dim3((unsigned)std::max(1, std::min(seg->num_tiles / 4 + 1, seg->num_cus * 8))), dim3(256), 0, 0>>>(col.d_fwd, col.bits, seg->cols[(size_t)i].d_null_bitmap, image.d_fwd_alloc,
                                                                                                                                     bits_out, (uint32_t)col.cardinality, seg->num_tiles);
      e = hipDeviceSynchronize();
    }
    if (e != hipSuccess) { if (image.d_fwd_alloc) (void)hipFree(image.d_fwd_alloc); return bail(fail(PG_ERR_DEVICE, "column %s: null-key image: %s", col.name.c_str(), hipGetErrorString(e))); }
    image.d_fwd = image.d_fwd_alloc;
    seg->device_bytes += image.fwd_alloc_bytes;
    seg->cols[(size_t)i].nullkey_column = (int)seg->cols.size();
    seg->cols.push_back(std::move(image));
  }
  // Everything enqueued above (null-stream memsets of the padding, the posting expansion) has run before the first query can: the
  // queries' streams are non-blocking, they do not wait for the null stream by themselves.
  {
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return bail(fail(PG_ERR_DEVICE, "opening segment %s: %s", seg->name.c_str(), hipGetErrorString(e)));
  }
  *out_segment = seg;
  return PG_OK;
}

pg_status pg_segment_close(pg_segment* segment) {
  if (!segment) return fail(PG_ERR_INVALID_ARGUMENT, "null segment");
  (void)hipSetDevice(phys_device(segment->device));
  free_segment(segment);
  return PG_OK;
}

pg_status pg_segment_num_docs(const pg_segment* segment, int32_t* out_num_docs) {
  if (!segment || !out_num_docs) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  *out_num_docs = segment->num_docs;
  return PG_OK;
}

pg_status pg_segment_device_bytes(const pg_segment* segment, uint64_t* out_bytes) {
  if (!segment || !out_bytes) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  *out_bytes = segment->device_bytes.load(std::memory_order_relaxed);
  return PG_OK;
}

pg_status pg_group_key_info(const pg_segment* segment, int32_t column, int64_t* out_base, int32_t* out_is_offset, int32_t* out_null_entry) {
  if (!segment || !out_base || !out_is_offset || !out_null_entry) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  if (column < 0 || column >= segment->num_user_cols) return fail(PG_ERR_INVALID_ARGUMENT, "column %d out of range", column);
  const ColumnDev& col = segment->cols[(size_t)column];
  *out_base = 0; *out_is_offset = 0; *out_null_entry = col.cardinality;
  if (col.encoding == PG_FWD_FIXED_BIT_DICT) return PG_OK;
  if (col.keyimage_column < 0) return fail(PG_ERR_UNSUPPORTED, "group-by on raw column %s: no key image", col.name.c_str());
  const ColumnDev& image = segment->cols[(size_t)col.keyimage_column];
  if (image.rank_image) {
    // keyed by value through the column's own dictionary: an entry is a RANK, the values come from pg_group_key_values
    // (the NULL entry is the cardinality: known once the dictionary exists -- built here when no query has grouped by the column yet)
    const pg_status rst = ensure_key_image(const_cast<pg_segment*>(segment), column, nullptr);
    if (rst != PG_OK) return rst;
    *out_base = 0; *out_is_offset = 2; *out_null_entry = image.cardinality;
    return PG_OK;
  }
  *out_base = col.raw_min; *out_is_offset = 1; *out_null_entry = image.cardinality;
  return PG_OK;
}

pg_status pg_group_key_values(pg_segment* segment, int32_t column, int64_t* out_value_bits, int32_t capacity, int32_t* out_count) {
  if (!segment || !out_count) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  if (column < 0 || column >= segment->num_user_cols) return fail(PG_ERR_INVALID_ARGUMENT, "column %d out of range", column);
  ColumnDev& col = segment->cols[(size_t)column];
  if (col.keyimage_column < 0 || !segment->cols[(size_t)col.keyimage_column].rank_image)
    return fail(PG_ERR_INVALID_ARGUMENT, "column %s is not keyed through a rank image (pg_group_key_info: is_offset != 2)", col.name.c_str());
  const pg_status st = ensure_key_image(segment, column, nullptr);
  if (st != PG_OK) return st;
  const ColumnDev& image = segment->cols[(size_t)col.keyimage_column];
  *out_count = image.cardinality;
  if (!out_value_bits) return PG_OK;                                   // (sizing call)
  if (capacity < image.cardinality) return fail(PG_ERR_INVALID_ARGUMENT, "room for %d values, the column has %d distinct ones", capacity, image.cardinality);
  const bool floating = col.vkind == kValF32 || col.vkind == kValF64;
  for (int d = 0; d < image.cardinality; ++d) out_value_bits[d] = rank_image_value_bits(image.h_rank_keys[(size_t)d], floating);
  return PG_OK;
}

void pg_result_free(pg_result* r) {
  if (!r) return;
  free(r->aggregations);
  free(r->group_ids);
  free(r->group_ids64);
  free(r->group_key_dict_ids);
  free(r->group_aggregations);
  memset(r, 0, sizeof(*r));
}

constexpr int32_t kQueryHashHolder = 1 << 30;      // internal pg_query.flags bit (execute_null_handling -> execute_impl)

// ---- plan-time eligibility (pg_query_check) ----
// Every reason pg_execute can answer PG_ERR_UNSUPPORTED for, decided from the query and the segment's metadata alone: no context, no
// allocation, no launch.  pg_execute runs it first, so both entry points reject the same queries; the same conditions further down the
// execution path are safety nets behind it.  Where the exact resource use depends on run-time state (which summed columns read a
// value plane decides how many column streams a query stages) the check takes the upper bound: it may decline a query whose streams
// would just have fitted, never the other way round.  InstancePlanMakerImplV2.makeSegmentPlanNode (:270-289) is where the caller asks.
// Group-by key spaces beyond an int: the reference's LongMapBasedHolder (the raw key fits a long) and ArrayMapBasedHolder (it does not)
// -- DictionaryBasedGroupKeyGenerator.java:162-176.  Here: a hashed table in HBM (GroupParams.hash_*), sized at plan time to at least
// twice the keys that can exist (min(numDocs, product)), so that it cannot fill up at run time.
struct HashPlan {
  int kind = 0;                    // 0: int raw keys (direct-indexed table); 1: long raw keys; 2: beyond a long (chained tables)
  int levels = 0;                  // first tables of a key beyond a long: table l takes over at column split[l]
  int split[kMaxHashLevels] = {};
  long long slots = 0, slots_lvl[kMaxHashLevels] = {};
  unsigned long long mult[kMaxGroupCols] = {};
  long long level_slots() const { long long t = 0; for (int l = 0; l < levels; ++l) t += slots_lvl[l]; return t; }
  int segment_begin(int l) const { return l == 0 ? 0 : split[l - 1]; }        // key l (l in [0, levels]) covers columns [segment_begin(l), segment_end(l, n))
  int segment_end(int l, int n) const { return l < levels ? split[l] : n; }
};
static pg_status plan_hash_holder(const pg_segment* seg, const std::vector<int>& cards, HashPlan* hp) {
  *hp = HashPlan();
  // the product of up to kMaxGroupCols 31-bit cardinalities does not fit 128 bits: saturate far above anything compared against
  const unsigned __int128 kSat = (unsigned __int128)1 << 100;
  auto sat_mul = [&](unsigned __int128 a, unsigned __int128 b) { return (a >= kSat || b >= kSat || a * b >= kSat) ? kSat : a * b; };
  unsigned __int128 prod = 1;
  for (int c : cards) prod = sat_mul(prod, (unsigned __int128)std::max(c, 1));
  if (prod <= (unsigned __int128)kMaxGroupSlots) return PG_OK;
  if ((long long)seg->num_docs > (1ll << 29))
    return fail(PG_ERR_UNSUPPORTED, "group-by with raw keys beyond an int on a segment of more than 2^29 docs (the hashed table would need more than 2^30 slots)");
  auto pow2_at_least = [](unsigned __int128 v) { long long p = 1 << 16; while ((unsigned __int128)p < v) p <<= 1; return p; };
  const int n = (int)cards.size();
  const unsigned __int128 kLongMax = (unsigned __int128)0x7FFFFFFFFFFFFFFFull;
  hp->kind = prod > kLongMax ? 2 : 1;
  hp->slots = pow2_at_least(2 * std::min<unsigned __int128>((unsigned __int128)seg->num_docs, prod));
  // Columns are taken in order while the key so far still fits a long; when the next column would push it beyond, the key so far goes
  // through a first table and its slot number (< 2^30) stands for it from there on (slot * card always fits: 2^30 * 2^31).
  unsigned __int128 bound = 1;             // exclusive upper bound of the running key
  unsigned __int128 seen = 1;              // distinct keys the columns so far can form
  for (int c = 0; c < n; ++c) {
    const unsigned __int128 card = (unsigned __int128)std::max(cards[(size_t)c], 1);
    if (bound * card > kLongMax) {
      if (hp->levels == kMaxHashLevels) return fail(PG_ERR_UNSUPPORTED, "group-by raw key needs more than %d chained tables", kMaxHashLevels);
      const long long s = pow2_at_least(2 * std::min<unsigned __int128>((unsigned __int128)seg->num_docs, seen));
      hp->split[hp->levels] = c; hp->slots_lvl[hp->levels] = s; hp->levels++;
      bound = (unsigned __int128)s;
    }
    hp->mult[c] = (unsigned long long)bound;
    bound *= card;
    seen = sat_mul(seen, card);
  }
  return PG_OK;
}

static pg_status check_query_plan(const pg_segment* seg, const pg_query* q, int extra_and_leaves) {
  const int num_cols_total = (int)seg->cols.size();
  if (q->num_filter_nodes < 0 || q->num_filter_nodes > kMaxNodes) return fail(PG_ERR_UNSUPPORTED, "filter tree has %d nodes (max %d)", q->num_filter_nodes, kMaxNodes);
  if (q->num_filter_nodes > 0 && (!q->filter || !q->predicates)) return fail(PG_ERR_INVALID_ARGUMENT, "filter nodes without predicates");
  std::vector<int> filter_cols;           // columns whose stream a scan leaf stages
  {
    int depth = 0;
    for (int n = 0; n < q->num_filter_nodes; ++n) {
      const pg_filter_node& fn = q->filter[n];
      if (fn.op == PG_FILTER_LEAF) {
        if (fn.predicate < 0 || fn.predicate >= q->num_predicates) return fail(PG_ERR_INVALID_ARGUMENT, "filter node %d: bad predicate index", n);
        depth++;
      } else if (fn.op == PG_FILTER_NOT) {
        if (depth < 1) return fail(PG_ERR_INVALID_ARGUMENT, "malformed filter tree (node %d)", n);
      } else if (fn.op == PG_FILTER_AND || fn.op == PG_FILTER_OR) {
        if (fn.num_children < 1 || depth < fn.num_children) return fail(PG_ERR_INVALID_ARGUMENT, "malformed filter tree (node %d)", n);
        depth -= fn.num_children - 1;
      } else return fail(PG_ERR_INVALID_ARGUMENT, "unknown filter op %d", fn.op);
    }
    if (q->num_filter_nodes > 0 && depth != 1) return fail(PG_ERR_INVALID_ARGUMENT, "malformed filter tree (%d roots)", depth);
    // the same re-ordering lower_filter applies: the root AND as a chain, its inverted-index children as one leaf
    std::vector<SeqNode> seq;
    std::vector<int> and_members;
    int lazy_node = -1, num_bitmap_prefix = 0;
    build_sequence(q, &seq, &lazy_node, &num_bitmap_prefix, &and_members);
    if ((int)seq.size() + 2 * extra_and_leaves > kMaxNodes) return fail(PG_ERR_UNSUPPORTED, "filter tree has %d nodes (max %d)", (int)seq.size() + 2 * extra_and_leaves, kMaxNodes);
    int leaves = extra_and_leaves, max_depth = extra_and_leaves > 0 ? 2 : 0;
    depth = 0;
    for (const SeqNode& sn : seq) {
      if (sn.op == PG_FILTER_LEAF) { leaves++; depth++; }
      else if (sn.op != PG_FILTER_NOT) depth -= sn.num_children - 1;
      max_depth = std::max(max_depth, depth);
      if (sn.op == PG_FILTER_LEAF && sn.src >= 0) {
        const pg_predicate& pr = q->predicates[q->filter[sn.src].predicate];
        const bool stages = pr.kind == PG_PRED_RAW_RANGE || ((pr.kind == PG_PRED_DICT_RANGE || pr.kind == PG_PRED_DICT_SET) && pr.eval != PG_EVAL_INVERTED);
        if (stages) {
          if (pr.column < 0 || pr.column >= num_cols_total) return fail(PG_ERR_INVALID_ARGUMENT, "predicate column %d out of range", pr.column);
          if (std::find(filter_cols.begin(), filter_cols.end(), pr.column) == filter_cols.end()) filter_cols.push_back(pr.column);
        }
      }
    }
    if (leaves > kMaxLeaves) return fail(PG_ERR_UNSUPPORTED, "more than %d filter leaves", kMaxLeaves);
    if (max_depth + (extra_and_leaves > 0 ? 1 : 0) > kStackDepth) return fail(PG_ERR_UNSUPPORTED, "filter tree deeper than %d", kStackDepth);
  }
  const int na = q->num_aggregations, ng = q->num_group_by;
  if (na < 0 || ng < 0 || (na > 0 && !q->aggregations) || (ng > 0 && !q->group_by_columns)) return fail(PG_ERR_INVALID_ARGUMENT, "bad aggregation / group-by lists");
  if (ng > kMaxGroupCols) return fail(PG_ERR_UNSUPPORTED, "more than %d group-by columns", kMaxGroupCols);
  std::vector<int> key_cols, agg_cols, key_cards;
  long long product = 1;
  for (int g = 0; g < ng; ++g) {
    int c = q->group_by_columns[g];
    if (c < 0 || c >= num_cols_total) return fail(PG_ERR_INVALID_ARGUMENT, "group-by column %d out of range", c);
    if (seg->cols[(size_t)c].encoding != PG_FWD_FIXED_BIT_DICT) {
      // a raw INT / LONG column is grouped by through its key image (cardinality and width are known since open; nothing is built here)
      if (seg->cols[(size_t)c].keyimage_column < 0)
        return fail(PG_ERR_UNSUPPORTED, "group-by on raw column %s: no key image", seg->cols[(size_t)c].name.c_str());
      if (seg->cols[(size_t)seg->cols[(size_t)c].keyimage_column].rank_image) {
        // a rank image's cardinality is what the plan is priced with: the column's dictionary is built here, the first time a query that
        // groups by it is checked (once per column and segment; the one thing pg_query_check ever launches)
        // A build that fails -- no HBM for its transient keys, a device error -- is a reason to keep the CPU plan, not to fail a query at PLAN
        // time: the JNI layer throws on every status but OK / UNSUPPORTED and GpuPlanMaker.makeSegmentPlanNode lets that through (advisor, round 5).
        const pg_status rst = ensure_key_image(const_cast<pg_segment*>(seg), c, nullptr);
        if (rst != PG_OK) {
          (void)hipGetLastError();
          return fail(PG_ERR_UNSUPPORTED, "group-by on raw column %s: its dictionary and rank image could not be built (status %d: %s) -- CPU plan", seg->cols[(size_t)c].name.c_str(), (int)rst,
                      pg_last_error());
        }
      }
      c = seg->cols[(size_t)c].keyimage_column;
    }
    const ColumnDev& col = seg->cols[(size_t)c];
    key_cards.push_back(col.cardinality);
    if (std::find(key_cols.begin(), key_cols.end(), c) == key_cols.end()) key_cols.push_back(c);
  }
  HashPlan hash_plan;
  if (ng > 0) { const pg_status hst = plan_hash_holder(seg, key_cards, &hash_plan); if (hst != PG_OK) return hst; }
  if (hash_plan.kind == 0) for (int card : key_cards) product *= std::max(card, 1);
  else {
    // Long / ArrayMap holders run in the lane-private group-by kernel only: 32-bit-domain aggregations, lane-private filter leaves
    product = hash_plan.slots;
    if (q->flags & kQueryHashHolder) return fail(PG_ERR_UNSUPPORTED, "group-by with raw keys beyond an int under null handling (plan-time fallback)");
    for (int n = 0; n < q->num_filter_nodes; ++n) {
      if (q->filter[n].op != PG_FILTER_LEAF) continue;
      const pg_predicate& pr = q->predicates[q->filter[n].predicate];
      if (pr.kind == PG_PRED_RAW_RANGE && pr.column >= 0 && pr.column < num_cols_total && seg->cols[(size_t)pr.column].stored_type != PG_TYPE_INT)
        return fail(PG_ERR_UNSUPPORTED, "group-by with raw keys beyond an int under a raw 8-byte range predicate (plan-time fallback)");
    }
  }
  std::vector<std::pair<int, int>> group_aggs;     // distinct (column, SUM | MIN | MAX)
  for (int a = 0; a < na; ++a) {
    const pg_aggregation& ag = q->aggregations[a];
    if (ag.function < PG_AGG_COUNT || ag.function > PG_AGG_AVG) return fail(PG_ERR_UNSUPPORTED, "aggregation function %d", ag.function);
    if (ag.function == PG_AGG_COUNT) continue;
    if (ag.column < 0 || ag.column >= num_cols_total) return fail(PG_ERR_INVALID_ARGUMENT, "aggregation column %d out of range", ag.column);
    if (std::find(agg_cols.begin(), agg_cols.end(), ag.column) == agg_cols.end()) agg_cols.push_back(ag.column);
    if (ng > 0) {
      const ColumnDev& col = seg->cols[(size_t)ag.column];
      const int kind = (ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG) ? 0 : (ag.function == PG_AGG_MIN ? 1 : 2);
      if (std::find(group_aggs.begin(), group_aggs.end(), std::make_pair(ag.column, kind)) == group_aggs.end()) group_aggs.emplace_back(ag.column, kind);
      if (col.encoding == PG_FWD_RAW_FIXED_BYTE && col.vkind != kValI32) {
        // group_typed_direct_kernel takes it, with the lane-private filter: a range leaf on a raw 8-byte column is not in that filter
        for (int n = 0; n < q->num_filter_nodes; ++n) {
          if (q->filter[n].op != PG_FILTER_LEAF) continue;
          const pg_predicate& pr = q->predicates[q->filter[n].predicate];
          if (pr.kind == PG_PRED_RAW_RANGE && pr.column >= 0 && pr.column < num_cols_total && seg->cols[(size_t)pr.column].stored_type != PG_TYPE_INT)
            return fail(PG_ERR_UNSUPPORTED, "group-by aggregation of a raw 8-byte column under a raw 8-byte range predicate (plan-time fallback)");
        }
      }
      // (an 8-byte input under a hashed holder -- RAW, or the SUM of a dictionary column with 8-byte values -- takes group_typed_direct_kernel<.., kHash>: round 6b)
      if (kind == 0 && col.vkind == kValI64 && !col.h_dict_i64.empty()) {
        const double max_abs = std::max(std::fabs((double)col.h_dict_i64.front()), std::fabs((double)col.h_dict_i64.back()));
        if ((double)seg->num_docs * max_abs >= 9.2e18) return fail(PG_ERR_UNSUPPORTED, "group-by SUM of LONG column %s could overflow int64", col.name.c_str());
      }
    }
  }
  if (ng == 0 && (int)agg_cols.size() > kMaxAggCols) return fail(PG_ERR_UNSUPPORTED, "more than %d aggregated columns", kMaxAggCols);
  if ((int)group_aggs.size() > kMaxGroupAggs) return fail(PG_ERR_UNSUPPORTED, "more than %d distinct group-by aggregations", kMaxGroupAggs);
  if (ng > 0 && ((unsigned long long)product * (1ull + group_aggs.size() + (hash_plan.kind ? 1 : 0)) + (unsigned long long)hash_plan.level_slots()) * 8ull > g_engine.group_table_bytes)
    return fail(PG_ERR_UNSUPPORTED, "group-by table of %lld slots x %zu words exceeds the %llu-byte budget (PINOT_GPU_GROUP_TABLE_BYTES)", product, 1 + group_aggs.size(),
                (unsigned long long)g_engine.group_table_bytes);
  // Column streams, as slot_for hands them out: (column, read through its value plane?).  A column summed through its plane is read
  // through the plane by everything that can be (its other aggregations, a dictId-range leaf on it); set leaves and group keys read
  // the dictIds.  Which summed columns have a plane is want_value_plane's decision; the histogram path, which reads the dictIds
  // instead, is not anticipated here (run-time tiers): at worst one stream too many is counted.
  std::vector<char> plane((size_t)std::max(num_cols_total, 1), 0);
  for (int a = 0; a < na; ++a) {
    const pg_aggregation& ag = q->aggregations[a];
    if ((ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG) && (want_value_plane(seg->cols[(size_t)ag.column]) || want_wide_plane(seg, q, ag.column))) plane[(size_t)ag.column] = 1;
  }
  std::vector<int> streams;               // column * 2 + plane
  auto use = [&](int column, bool through_plane) {
    const int key = column * 2 + (through_plane ? 1 : 0);
    if (std::find(streams.begin(), streams.end(), key) == streams.end()) streams.push_back(key);
  };
  for (int n = 0; n < q->num_filter_nodes; ++n) {
    if (q->filter[n].op != PG_FILTER_LEAF) continue;
    const pg_predicate& pr = q->predicates[q->filter[n].predicate];
    if (pr.column < 0 || pr.column >= num_cols_total) continue;
    if (pr.kind == PG_PRED_RAW_RANGE) use(pr.column, false);
    else if (pr.kind == PG_PRED_DICT_RANGE && pr.eval != PG_EVAL_INVERTED) use(pr.column, plane[(size_t)pr.column] != 0);
    else if (pr.kind == PG_PRED_DICT_SET && pr.eval != PG_EVAL_INVERTED) use(pr.column, false);
  }
  for (int c : key_cols) use(c, false);
  for (int c : agg_cols) use(c, plane[(size_t)c] != 0);
  const int bound = (int)streams.size();
  if (bound > kMaxCols) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
  return PG_OK;
}

// ExecutionStatistics.numEntriesScannedInFilter from the plan pg_filter_stats.h chose; `counted`: the kernel that ran carried the
// kNodeCountEntries counter (its value has been copied to ctx->h_filter_entries and the stream is idle).
static void finish_filter_stats(fstats::Plan stats_plan, int stats_scan_leaves, const pg_segment* seg, int64_t counted_entries, bool counted, pg_result* out) {
  const int64_t upper_bound = (int64_t)stats_scan_leaves * seg->num_docs;      // every scan leaf looking at every doc
  switch (stats_plan) {
    case fstats::Plan::kZero: out->stats.num_entries_scanned_in_filter = 0; out->filter_entries_exact = 1; break;
    case fstats::Plan::kPerLeaf: out->stats.num_entries_scanned_in_filter = upper_bound; out->filter_entries_exact = 1; break;
    case fstats::Plan::kLeap2:
      // one entry per doc for whichever leaf is scanning there, plus the device's count of the docs where the other leaf was asked
      if (counted) { out->stats.num_entries_scanned_in_filter = (int64_t)seg->num_docs + counted_entries; out->filter_entries_exact = 1; break; }
      out->stats.num_entries_scanned_in_filter = upper_bound; out->filter_entries_exact = 0; break;
    case fstats::Plan::kChain:
      if (counted) { out->stats.num_entries_scanned_in_filter = counted_entries; out->filter_entries_exact = 1; break; }
      [[fallthrough]];                                                            // an LDS-staged kernel ran: replayed by pg_execute
    default: out->stats.num_entries_scanned_in_filter = upper_bound; out->filter_entries_exact = 0; break;
  }
}

static void finish_filter_stats(const Lowered& lw, const pg_segment* seg, int64_t counted_entries, bool counted, pg_result* out) {
  finish_filter_stats(lw.stats_plan, lw.stats_scan_leaves, seg, counted_entries, counted, out);
}

// pg_execute_batch: a query whose whole device work is ONE launch of scan_private_kernel -- nothing ahead of it on the stream, no index
// phase, no entry counter, no histogram -- is not launched by execute_impl but handed back as its kernel parameters plus the
// conversion of the folded record into a pg_result; the batch puts many of them into one launch (scan_private_batch_kernel).
static const pg_status kDeferred = static_cast<pg_status>(100);      // (internal: never leaves the library)
static const pg_status kRunAlone = static_cast<pg_status>(101);      // (internal: a batch item whose shared launch could not answer it -- pg_execute_batch runs it by itself)
// A query lowered for the shared launch of pg_execute_batch: the kernel's parameter block (the launch fills in where this item's records
// go), the workgroups it would get on its own, and the conversion of its folded record into the reference's holder types.  Immutable
// once built: the segment's plan_cache hands the same item to later batches.
struct OwnedQuery {                               // a deep copy of a pg_query (the caller's arrays are only valid during its call)
  pg_query q;
  std::vector<pg_filter_node> filter;
  std::vector<pg_predicate> predicates;
  std::vector<std::vector<uint32_t>> set_words;
  std::vector<pg_aggregation> aggregations;
  std::vector<int32_t> group_by;
};
struct LoweredItem {
  ScanParams sp;
  int blocks = 0;
  bool one_slot = true;
  std::function<void(const BlockPartial&, pg_result*)> convert;
  std::vector<int> plane_columns;                 // value planes sp reads: held (PlaneHold) by every batch that launches this item
  std::vector<Lowered::SetLeaf> sets;             // dictId-set leaves of sp: nodes whose set_words == ctx_words read the batch's copy of host_words (which the item's query owns)
  // lean_kind 6 (group_lds_batch_kernel): the item is a GroupParams; its table slice is count[G] | acc[NA][G], zero-identity keys
  std::shared_ptr<GroupParams> gp;
  int group_threads = 0;
  size_t group_lds = 0, group_table_words = 0;
  std::function<void(const unsigned long long*, pg_result*)> convert_group;      // the item's slice (on the host) -> the result
  // lean_kind 12 (index_and_batch_kernel): the item is an IndexAndParams whose kernel publishes the query's record (no ctx-owned memory in it)
  std::shared_ptr<IndexAndParams> and_params;
  uint32_t and_windows = 0;
  size_t hist_lds = 0;                            // lean_kind 3..5: the item's histogram (the launch's dynamic LDS is the largest item's)
  int hist_cw = 0, hist_col = -1;                 //   counter width; the summed column (a wrapped counter moves it to the guarded tier)
  // the cache's side (empty key: not cacheable)
  std::string key;                                // query_key of the query
  uint64_t engine_epoch = 0, plane_epoch = 0;
  std::shared_ptr<const OwnedQuery> query;        // `convert` reads the query through this copy
};
struct Deferred {
  std::shared_ptr<const LoweredItem> item;
  std::unique_ptr<PlaneHold> planes;           // value planes the kernel reads stay held until the batch has run
  bool cacheable = false;                      // out of execute_impl: nothing about this lowering was provisional
  bool single = false;                         // in: pg_execute's own call -- only a group-by of the LDS-table form is deferred (at any size), scans run as they always did
};

// The query's content as bytes: two queries with equal keys lower to the same item on the same segment.
static bool query_key(const pg_query* q, std::string* key) {
  key->clear();
  if (q->num_filter_nodes < 0 || q->num_predicates < 0 || q->num_aggregations < 0 || q->num_group_by < 0) return false;
  if ((q->num_filter_nodes > 0 && !q->filter) || (q->num_predicates > 0 && !q->predicates) || (q->num_aggregations > 0 && !q->aggregations) ||
      (q->num_group_by > 0 && !q->group_by_columns)) return false;
  const int32_t head[6] = {q->num_filter_nodes, q->num_predicates, q->num_aggregations, q->num_group_by, q->num_groups_limit, q->flags};
  key->append(reinterpret_cast<const char*>(head), sizeof(head));
  key->append(reinterpret_cast<const char*>(q->filter), sizeof(pg_filter_node) * (size_t)q->num_filter_nodes);
  for (int i = 0; i < q->num_predicates; ++i) {
    const pg_predicate& pr = q->predicates[i];
    if (pr.num_set_words < 0 || (pr.num_set_words > 0 && !pr.set_words)) { key->clear(); return false; }      // (an empty key = not cacheable)
    const int64_t fields[8] = {pr.kind, pr.column, pr.eval, pr.exclusive, pr.lo, pr.hi, pr.num_set_words, pr.reserved};
    key->append(reinterpret_cast<const char*>(fields), sizeof(fields));
    key->append(reinterpret_cast<const char*>(pr.set_words), sizeof(uint32_t) * (size_t)pr.num_set_words);
  }
  key->append(reinterpret_cast<const char*>(q->aggregations), sizeof(pg_aggregation) * (size_t)q->num_aggregations);
  key->append(reinterpret_cast<const char*>(q->group_by_columns), sizeof(int32_t) * (size_t)q->num_group_by);
  return true;
}

static std::shared_ptr<const OwnedQuery> own_query(const pg_query* q) {
  auto o = std::make_shared<OwnedQuery>();
  o->q = *q;
  if (q->num_filter_nodes > 0) o->filter.assign(q->filter, q->filter + q->num_filter_nodes);
  if (q->num_predicates > 0) o->predicates.assign(q->predicates, q->predicates + q->num_predicates);
  o->set_words.resize(o->predicates.size());
  for (size_t i = 0; i < o->predicates.size(); ++i) {
    pg_predicate& pr = o->predicates[i];
    if (pr.num_set_words > 0) { o->set_words[i].assign(pr.set_words, pr.set_words + pr.num_set_words); pr.set_words = o->set_words[i].data(); }
    else pr.set_words = nullptr;
  }
  if (q->num_aggregations > 0) o->aggregations.assign(q->aggregations, q->aggregations + q->num_aggregations);
  o->q.filter = o->filter.empty() ? nullptr : o->filter.data();
  o->q.predicates = o->predicates.empty() ? nullptr : o->predicates.data();
  o->q.aggregations = o->aggregations.empty() ? nullptr : o->aggregations.data();
  if (q->num_group_by > 0) o->group_by.assign(q->group_by_columns, q->group_by_columns + q->num_group_by);
  o->q.group_by_columns = o->group_by.empty() ? nullptr : o->group_by.data();
  return o;
}

static pg_status prepare_fsm_side(pg_segment* seg, ExecCtx* ctx, const pg::fstats::Fsm& fsm, FsmSide* side);
static pg_status device_fsm_filter_stats(pg_segment* seg, ExecCtx* ctx, const pg_query* q, const FsmSide& side, pg_result* out);
// PINOT_GPU_EXEC_TRACE=1: the host phases of every pg_execute on stderr -- eligibility check | context + lowering | launches enqueued | wait |
// result conversion (what a query's host clock is made of beside its kernels; execute_impl marks the boundaries it passes)
struct ExecTrace {
  std::chrono::steady_clock::time_point t[6];
  bool seen[6];
};
static thread_local ExecTrace t_exec_trace;
static bool exec_trace_on() { static const bool on = getenv("PINOT_GPU_EXEC_TRACE") != nullptr; return on; }
static inline void exec_mark(int i) {
  if (!exec_trace_on()) return;
  t_exec_trace.t[i] = std::chrono::steady_clock::now();
  t_exec_trace.seen[i] = true;
}

static pg_status execute_impl(pg_segment* seg, const pg_query* q, pg_result* out, unsigned long long* d_out_bitmap_request,
                              uint64_t* host_bitmap, int64_t host_bitmap_words, int64_t* out_cardinality, bool allow_metadata_plan = true,
                              Deferred* defer = nullptr, FsmSide* side = nullptr) {
  if (!g_engine.initialized) return fail(PG_ERR_NOT_INITIALIZED, "pg_init has not been called");
  if (!seg || !q) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  // d_out_bitmap_request: like host_bitmap, but the filter's doc-order bitmap is copied device to device into the caller's
  // buffer ((num_docs + 63) / 64 words, any stream-ordered device memory) and never visits the host.
  {
    pg_query shape = *q;                      // pg_filter_bitmap evaluates the filter only
    if (host_bitmap != nullptr || d_out_bitmap_request != nullptr) { shape.num_aggregations = 0; shape.num_group_by = 0; }
    const pg_status eligible = check_query_plan(seg, &shape, 0);
    if (eligible != PG_OK) return eligible;
  }
  exec_mark(1);
  HIP_TRY(hipSetDevice(phys_device(seg->device)));
  ExecCtx* ctx = nullptr;
  pg_status st = acquire_ctx(seg, &ctx);
  if (st != PG_OK) return st;
  CtxGuard guard{seg, ctx};

  const bool want_bitmap = host_bitmap != nullptr || d_out_bitmap_request != nullptr;
  const int na = want_bitmap ? 0 : q->num_aggregations;
  const int ng = want_bitmap ? 0 : q->num_group_by;
  if (na < 0 || ng < 0 || (na > 0 && !q->aggregations) || (ng > 0 && !q->group_by_columns)) return fail(PG_ERR_INVALID_ARGUMENT, "bad aggregation / group-by lists");
  if (ng > kMaxGroupCols) return fail(PG_ERR_UNSUPPORTED, "more than %d group-by columns", kMaxGroupCols);
  const bool timed = (g_engine.flags & PG_CFG_TIME_KERNELS) != 0;

  // NonScanBasedAggregationOperator (core/plan/AggregationPlanNode.java:98-115,159-190; core/operator/query/
  // NonScanBasedAggregationOperator.java:83-105): the filter matches everything and every function is COUNT, or MIN / MAX of a
  // dictionary column -> the answer comes from the segment metadata and the dictionary ends; nothing is scanned.
  if (ng == 0 && !want_bitmap && out && na > 0 && allow_metadata_plan) {
    bool match_all = q->num_filter_nodes == 0;
    if (q->num_filter_nodes == 1 && q->filter && q->predicates && q->filter[0].op == PG_FILTER_LEAF && q->filter[0].predicate >= 0 &&
        q->filter[0].predicate < q->num_predicates) {
      const pg_predicate& pr = q->predicates[q->filter[0].predicate];
      match_all = (pr.kind == PG_PRED_MATCH_ALL && !pr.exclusive) || (pr.kind == PG_PRED_MATCH_NONE && pr.exclusive);
    }
    bool fit = match_all;
    for (int a = 0; a < na && fit; ++a) {
      const pg_aggregation& ag = q->aggregations[a];
      if (ag.function == PG_AGG_COUNT) continue;
      fit = (ag.function == PG_AGG_MIN || ag.function == PG_AGG_MAX) && ag.column >= 0 && ag.column < (int)seg->cols.size() &&
            seg->cols[(size_t)ag.column].encoding == PG_FWD_FIXED_BIT_DICT;
    }
    if (fit) {
      memset(out, 0, sizeof(*out));
      out->num_aggregations = na;
      out->aggregations = (pg_agg_value*)calloc((size_t)na, sizeof(pg_agg_value));
      for (int a = 0; a < na; ++a) {
        const pg_aggregation& ag = q->aggregations[a];
        pg_agg_value& v = out->aggregations[a];
        v.count = seg->num_docs;
        v.min = std::numeric_limits<double>::infinity();
        v.max = -std::numeric_limits<double>::infinity();
        if (ag.function == PG_AGG_MIN) v.min = seg->cols[(size_t)ag.column].h_dict_f64.front();
        if (ag.function == PG_AGG_MAX) v.max = seg->cols[(size_t)ag.column].h_dict_f64.back();
      }
      out->stats.num_docs_scanned = seg->num_docs;          // NonScanBasedAggregationOperator.getExecutionStatistics: (totalDocs, 0, 0, totalDocs)
      out->stats.num_entries_scanned_in_filter = 0;
      out->filter_entries_exact = 1;
      out->stats.num_entries_scanned_post_filter = 0;
      out->stats.num_total_docs = seg->num_docs;
      if (out_cardinality) *out_cardinality = seg->num_docs;
      return PG_OK;
    }
  }

  Lowered lw;
  memset(&lw.sp, 0, sizeof(lw.sp));
  memset(&lw.plan, 0, sizeof(lw.plan));
  lw.plan.lazy_node = -1;
  const int num_cols_total = (int)seg->cols.size();
  // Columns that are summed are read through their value plane (built on first use); decided before the filter is
  // lowered so that a range predicate on the same column can be evaluated on the plane too.
  lw.plane_cols.assign((size_t)std::max(num_cols_total, 1), 0);
  PlaneHold planes(seg, {});
  // At most one summed column goes through the LDS histogram instead (scan_hist_kernel); it needs the lane-private kernel, so the
  // shapes that kernel does not take are ruled out here, before a plane is (not) built.
  int hist_col = -1;
  if (ng == 0 && !want_bitmap && g_engine.scan_private && !(g_engine.flags & PG_CFG_PROFILE_WAVES)) {
    bool shape_ok = true;
    int only_col = -1;                  // the histogram kernel aggregates ONE column (SUM / AVG / MIN / MAX of it, and COUNT)
    for (int a = 0; a < na && shape_ok; ++a) {
      const pg_aggregation& ag = q->aggregations[a];
      if (ag.function == PG_AGG_COUNT) continue;
      shape_ok = ag.column >= 0 && ag.column < num_cols_total && seg->cols[(size_t)ag.column].encoding == PG_FWD_FIXED_BIT_DICT &&
                 seg->cols[(size_t)ag.column].vkind == kValI32 && (only_col < 0 || only_col == ag.column);
      only_col = ag.column;
    }
    for (int i = 0; i < q->num_predicates && shape_ok && q->predicates; ++i) {
      const pg_predicate& pr = q->predicates[i];
      if (pr.kind == PG_PRED_RAW_RANGE) shape_ok = pr.column >= 0 && pr.column < num_cols_total && seg->cols[(size_t)pr.column].stored_type == PG_TYPE_INT;
    }
    for (int a = 0; a < na && shape_ok && hist_col < 0; ++a) {
      const pg_aggregation& ag = q->aggregations[a];
      if ((ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG) && want_hist(seg->cols[(size_t)ag.column])) hist_col = ag.column;
    }
  }
  for (int a = 0; a < na; ++a) {
    const pg_aggregation& ag = q->aggregations[a];
    if ((ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG) && ag.column >= 0 && ag.column < num_cols_total && ag.column != hist_col &&
        (want_value_plane(seg->cols[(size_t)ag.column]) || (!want_bitmap && want_wide_plane(seg, q, ag.column)))) {
      if (lw.plane_cols[(size_t)ag.column]) continue;
      bool ready = false;
      st = acquire_plane(seg, ag.column, &ready);
      if (st != PG_OK) return st;
      if (ready) { planes.columns.push_back(ag.column); lw.plane_cols[(size_t)ag.column] = 1; }
      else lw.plane_pending = true;
    }
  }
  ctx->pre_enqueued = false;
  ctx->pre_started = false;      // ev[0] is recorded by the first piece of work that precedes the scan kernel (mark_pre_work)
  ctx->ev_last = 3;
  if (out && !want_bitmap) lw.stats_plan = fstats::choose_plan(q, &lw.stats_scan_leaves);
  if (lw.stats_plan == fstats::Plan::kLeap2 && (!g_engine.leap2 || (q->flags & PG_QUERY_STATS_UPPER_BOUND_OK))) lw.stats_plan = fstats::Plan::kReplay;      // (kReplay with nobody replaying: the upper bound)
  // the caller takes the upper bound (PG_QUERY_STATS_UPPER_BOUND_OK): a leap-frogging filter has no pass behind its kernel, so in a batch it shares the launch like any other
  const bool stats_is_final = lw.stats_plan == fstats::Plan::kZero || lw.stats_plan == fstats::Plan::kPerLeaf ||
                              (lw.stats_plan == fstats::Plan::kReplay && (q->flags & PG_QUERY_STATS_UPPER_BOUND_OK) != 0);
  lw.cardinality_only_hint = ng == 0 && out && !want_bitmap && na > 0;
  for (int a = 0; a < na; ++a) lw.cardinality_only_hint = lw.cardinality_only_hint && q->aggregations[a].function == PG_AGG_COUNT;
  lw.side = nullptr;
  if (out && !want_bitmap && lw.stats_plan == fstats::Plan::kReplay && side != nullptr && side->fsm != nullptr) {
    // numEntriesScannedInFilter is a statistic: a pass that cannot get its scratch leaves the query's answer standing with
    // filter_entries_exact = 0 (pg_execute's host replay still applies at its sizes) -- it never fails the query
    if (prepare_fsm_side(seg, ctx, *side->fsm, side) == PG_OK) lw.side = side;
    else (void)hipGetLastError();
  }
  st = lower_filter(seg, ctx, q, &lw);
  if (st != PG_OK) return st;
  exec_mark(2);
  ScanParams& sp = lw.sp;
  PlanParams& pl = lw.plan;

  // distinct projected columns (ExecutionStatistics numEntriesScannedPostFilter = numDocsScanned * numProjectedColumns)
  std::vector<int> projected;
  auto add_projected = [&](int c) { if (std::find(projected.begin(), projected.end(), c) == projected.end()) projected.push_back(c); };

  if (out) memset(out, 0, sizeof(*out));

  if (ng == 0 && out && !want_bitmap && lw.index_and_is_whole_filter && na > 0) {
    // FastFilteredCountOperator (core/plan/AggregationPlanNode.java:98-115, core/operator/query/FastFilteredCountOperator.java:66-72): COUNT(*)
    // over a filter that the indexes answer alone is the cardinality of the and-ed bitmaps -- index_and_kernel counted it, nothing is scanned.
    bool only_count = true;
    for (int a = 0; a < na; ++a) only_count &= q->aggregations[a].function == PG_AGG_COUNT;
    if (only_count && defer != nullptr && !defer->single && g_engine.batch_index && g_engine.direct_result && lw.and_cardinality_only && index_and_shares_a_launch(lw)) {
      // pg_execute_batch: the item is index_and_kernel publishing the cardinality -- it shares index_and_batch_kernel's launch (lean_kind 12)
      auto item = std::make_shared<LoweredItem>();
      memset(&item->sp, 0, sizeof(item->sp));
      item->sp.lean_kind = 12;
      item->and_params = std::make_shared<IndexAndParams>(lw.and_params);
      item->and_windows = lw.finalize_windows;
      item->blocks = (int)std::min<long long>(((long long)lw.finalize_windows + index_and_batch_block_waves() - 1) / index_and_batch_block_waves(), (long long)lw.and_num_cus * index_and_batch_blocks_per_cu());
      const int total_docs = seg->num_docs;
      item->convert = [na, total_docs](const BlockPartial& fp, pg_result* o) {
        const int64_t card = (int64_t)fp.count;
        o->num_aggregations = na;
        o->aggregations = (pg_agg_value*)calloc((size_t)na, sizeof(pg_agg_value));
        for (int a = 0; a < na; ++a) {
          o->aggregations[a].count = card;
          o->aggregations[a].min = std::numeric_limits<double>::infinity();
          o->aggregations[a].max = -std::numeric_limits<double>::infinity();
        }
        o->dominant_kernel = PG_KERNEL_INDEX_AND;
        o->stats.num_docs_scanned = card;
        o->stats.num_entries_scanned_in_filter = 0;     // bitmaps only: nothing is scanned
        o->filter_entries_exact = 1;
        o->stats.num_entries_scanned_post_filter = 0;
        o->stats.num_total_docs = total_docs;
      };
      defer->item = std::move(item);
      defer->cacheable = !lw.plane_pending;
      defer->planes.reset(new PlaneHold(std::move(planes)));
      return kDeferred;
    }
    if (only_count) {
      st = launch_index_and(&lw, ctx, nullptr);
      if (st != PG_OK) return st;
      unsigned long long* h_card = &ctx->h_partial->count;
      if (lw.cardinality_atomic) HIP_TRY(hipMemcpyAsync(ctx->h_and_shards, ctx->d_and_counters + 2, kAndShardBytes, hipMemcpyDeviceToHost, ctx->stream));
      else HIP_TRY(hipMemcpyAsync(h_card, lw.d_cardinality, 8, hipMemcpyDeviceToHost, ctx->stream));
      if (timed) { HIP_TRY(mark_pre_work(ctx)); HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream)); }
      exec_mark(3);
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      exec_mark(4);
      if (lw.cardinality_atomic) {
        BlockPartial g;
        st = read_index_and_shards(ctx, 0, &g);
        if (st != PG_OK) return st;
        *h_card = g.count;
      }
      const int64_t card = (int64_t)*h_card;
      out->num_aggregations = na;
      out->aggregations = (pg_agg_value*)calloc((size_t)na, sizeof(pg_agg_value));
      for (int a = 0; a < na; ++a) {
        out->aggregations[a].count = card;
        out->aggregations[a].min = std::numeric_limits<double>::infinity();
        out->aggregations[a].max = -std::numeric_limits<double>::infinity();
      }
      out->dominant_kernel = PG_KERNEL_INDEX_AND;
      out->stats.num_docs_scanned = card;
      out->stats.num_entries_scanned_in_filter = 0;     // bitmaps only: nothing is scanned
      out->filter_entries_exact = 1;
      out->stats.num_entries_scanned_post_filter = 0;
      out->stats.num_total_docs = seg->num_docs;
      if (out_cardinality) *out_cardinality = card;
      if (timed) {
        float ms_all = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms_all, ctx->ev[0], ctx->ev[3]));
        out->device_ms = ms_all;
        out->dominant_kernel_ms = ms_all;        // the index AND is the query
      }
      return PG_OK;
    }
  }
  if (ng == 0) {
    // ---------------- aggregation only ----------------
    std::vector<int> agg_slot_of((size_t)std::max(na, 1), -1);
    for (int a = 0; a < na; ++a) {
      const pg_aggregation& ag = q->aggregations[a];
      if (ag.function == PG_AGG_COUNT) continue;
      if (ag.function < PG_AGG_COUNT || ag.function > PG_AGG_AVG) return fail(PG_ERR_UNSUPPORTED, "aggregation function %d", ag.function);
      if (ag.column < 0 || ag.column >= num_cols_total) return fail(PG_ERR_INVALID_ARGUMENT, "aggregation column %d out of range", ag.column);
      add_projected(ag.column);
      int s = slot_for(&lw, seg, ag.column, lw.plane_cols[(size_t)ag.column] != 0);
      if (s < 0) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
      pl.cols[s].in_agg = 1;
      int ac = -1;
      for (int i = 0; i < pl.num_agg_cols; ++i) if (pl.agg_cols[i].col == s) ac = i;
      if (ac < 0) {
        if (pl.num_agg_cols >= kMaxAggCols) return fail(PG_ERR_UNSUPPORTED, "more than %d aggregated columns", kMaxAggCols);
        ac = pl.num_agg_cols++;
        pl.agg_cols[ac] = PlanAggCol{s, 0, 0, 0};
      }
      if (ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG) pl.agg_cols[ac].need_sum = 1;
      if (ag.function == PG_AGG_MIN || ag.function == PG_AGG_MAX) pl.agg_cols[ac].need_minmax = 1;
      agg_slot_of[(size_t)a] = ac;
    }
    bool need_queue = false, typed = false;
    for (int i = 0; i < pl.num_agg_cols; ++i) {
      const DevColumn& c = pl.cols[pl.agg_cols[i].col];
      need_queue |= pl.agg_cols[i].need_sum && !c.is_raw && !c.is_plane && c.vkind == kValI32;
      typed |= c.vkind != kValI32 && (pl.agg_cols[i].need_sum || c.is_raw);     // 8-byte / floating-point values are read
    }
    // The lane-private kernel (no LDS, plain global loads) takes every query whose leaves and aggregations it implements:
    // scan / set / bitmap leaves and raw INT ranges; COUNT, and SUM through a value plane / MIN / MAX on dictionary columns.
    // (the per-wave phase counters of PG_CFG_PROFILE_WAVES exist in the LDS-staged kernel only)
    bool use_private = g_engine.scan_private && !typed && !(g_engine.flags & PG_CFG_PROFILE_WAVES);
    for (int l = 0; l < pl.num_leaves && use_private; ++l) use_private = pl.leaves[l].kind <= kLeafBitmap || pl.leaves[l].kind == kLeafDocRange;
    int hist_slot = -1;
    for (int i = 0; i < pl.num_agg_cols && hist_col >= 0; ++i) if (lw.col_of_slot[(size_t)pl.agg_cols[i].col] == hist_col * 2) hist_slot = i;
    for (int i = 0; i < pl.num_agg_cols && use_private; ++i) {
      const DevColumn& c = pl.cols[pl.agg_cols[i].col];
      use_private = !c.is_raw && c.vkind == kValI32 && c.bits <= 31 && (!pl.agg_cols[i].need_sum || c.is_plane || i == hist_slot);
    }
    const bool use_hist = use_private && hist_slot == 0 && pl.num_agg_cols == 1;      // the histogram kernel aggregates one column
    if (hist_slot >= 0 && !use_hist) use_private = false;                           // (rare: the gather path of the LDS-staged kernel)
    const int hist_cw = use_hist ? hist_counter_bits(seg->cols[(size_t)hist_col]) : 0;
    const int hist_tier = use_hist ? std::max(__atomic_load_n(&seg->cols[(size_t)hist_col].hist_tier, __ATOMIC_RELAXED), g_engine.hist_guard ? 1 : 0) : 0;
    const bool hist_guarded = hist_tier >= 1 && hist_cw < 32;
    // Raw columns and 8-byte dictionaries: the same lane-private layout, read with 16-byte loads (scan_private_typed_kernel).
    // PINOT_GPU_SCAN_TYPED_PRIVATE=0 keeps them in the LDS-staged kernel.
    bool use_private_typed = g_engine.scan_private && g_engine.scan_typed_private && !use_private && pl.num_agg_cols > 0 && !(g_engine.flags & PG_CFG_PROFILE_WAVES);
    for (int l = 0; l < pl.num_leaves && use_private_typed; ++l) use_private_typed = pl.leaves[l].kind <= kLeafBitmap || pl.leaves[l].kind == kLeafDocRange;
    for (int i = 0; i < pl.num_agg_cols && use_private_typed; ++i) {
      const DevColumn& c = pl.cols[pl.agg_cols[i].col];
      use_private_typed = c.is_raw || (c.vkind != kValI32 && c.bits <= 31);      // every slot raw, or an 8-byte dictionary
    }
    Geometry geo;
    const int agg_wave_cap = waves_scan_agg(pl.num_agg_cols <= 1, typed);
    finish_geometry(seg, &lw, 0, need_queue, kBlockThreads / 64, agg_wave_cap, &geo);
    int blocks = geo.blocks;
    const size_t lds = geo.lds;
    size_t hist_lds = 0, hist_set_off = 0;
    if (use_hist) {
      // one histogram per workgroup of 16 wavefronts; as many workgroups per CU as LDS and registers admit
      const int per_word = 32 / hist_cw;
      hist_lds = (((size_t)(seg->cols[(size_t)hist_col].cardinality + per_word - 1) / per_word * 4) + 15) & ~(size_t)15;
      hist_lds = std::max(hist_lds, sizeof(BlockPartial) * (kHistBlockThreads / 64) + 16);      // the reduction records (+ the fold flag) reuse the counters' LDS
      // the filter's dictId sets (IN lists) behind the counters: ScanParams.set_leaves_in_lds = 1 + the area's byte offset (scan_hist_body)
      hist_set_off = 0;
      if (g_engine.set_lds && hist_lds + (size_t)kSetLdsWords * 4 <= 150 * 1024)
        for (int nd = 0; nd < sp.num_nodes; ++nd) if (sp.nodes[nd].op == PG_FILTER_LEAF && sp.nodes[nd].kind == kLeafDictSet) hist_set_off = hist_lds;
      if (hist_set_off != 0) hist_lds += (size_t)kSetLdsWords * 4;
      const size_t per_block = hist_lds + 256;
      int bpc = std::max(1, std::min(waves_scan_hist(hist_cw, hist_guarded) / (kHistBlockThreads / 64), (int)((160 * 1024 - 2048) / per_block)));
      if (g_engine.blocks_per_cu > 0) bpc = g_engine.blocks_per_cu;
      const long long tiles2k = ((long long)seg->num_docs + 2047) / 2048;
      const int wpb = kHistBlockThreads / 64;
      blocks = (int)std::max<long long>(1, std::min<long long>((tiles2k + wpb - 1) / wpb, (long long)seg->num_cus * bpc));
      if (g_engine.hist_blocks > 0) blocks = std::min(blocks, g_engine.hist_blocks);
      geo.threads = kHistBlockThreads;
      sp.hist_slot = hist_slot;
      sp.hist_bins = seg->cols[(size_t)hist_col].cardinality;
    } else if (use_private || use_private_typed) {
      const int cap = use_private_typed ? waves_scan_private_typed(pl.num_agg_cols) : waves_scan_private(pl.num_agg_cols);
      int bpc = std::max(1, cap / (kBlockThreads / 64));
      if (g_engine.blocks_per_cu > 0) bpc = g_engine.blocks_per_cu;
      const long long tiles2k = ((long long)seg->num_docs + 2047) / 2048;
      blocks = (int)std::max<long long>(1, std::min<long long>((tiles2k + 3) / 4, (long long)seg->num_cus * bpc));
      geo.threads = kBlockThreads;
    } else if ((size_t)sp.wave_lds_bytes > kLdsBudget) return fail(PG_ERR_UNSUPPORTED, "query needs %d bytes of LDS per wavefront", sp.wave_lds_bytes);
    // Nothing but a filter over narrow dictionary columns (COUNT(*) and / or the bitmap): scan_narrow_kernel, four tiles per wave and iteration
    // (a leap-frogging `a AND b` is counted by eval_filter_private's hook: the narrow kernels have their own evaluator and do not carry it)
    bool use_narrow = g_engine.scan_narrow && use_private && !use_hist && pl.num_agg_cols == 0 && lw.tile_list == nullptr && sp.num_nodes > 0 && !(out && lw.stats_leap2_flagged);
    if (use_narrow) {
      int depth = 0, max_depth = 0;
      for (int n = 0; n < sp.num_nodes && use_narrow; ++n) {
        const DevNode& dn = sp.nodes[n];
        if (dn.op == PG_FILTER_LEAF) {
          use_narrow = dn.kind == kLeafMatchAll || dn.kind == kLeafMatchNone || ((dn.kind == kLeafDictRange || (dn.kind == kLeafDictSet && g_engine.set_lds)) && dn.bits >= 1 && dn.bits <= kNarrowMaxBits);      // (a set of a narrow column: eight words in LDS, pg_scan_narrow.h)
          depth++;
        } else if (dn.op != PG_FILTER_NOT) depth -= dn.num_children - 1;
        max_depth = std::max(max_depth, depth);
      }
      use_narrow = use_narrow && max_depth <= kNarrowStack;
    }
    const bool narrow_single = use_narrow && g_engine.scan_narrow_single && sp.num_nodes == 1 && sp.nodes[0].kind == kLeafDictRange;
    // The whole filter is ONE bitmap that index_and_kernel made (its tiles listed), and every aggregated column is read as bit-packed
    // fields: eight tiles per wave and iteration, only the matching docs' values are touched (scan_sparse_kernel)
    const bool use_sparse = g_engine.scan_sparse && use_private && !use_hist && !use_narrow && !want_bitmap && pl.num_agg_cols > 0 && lw.tile_list != nullptr &&
                            sp.num_nodes == 1 && sp.nodes[0].kind == kLeafBitmap && sp.nodes[0].exclusive == 0 && !(g_engine.flags & PG_CFG_PROFILE_WAVES);
    if (use_sparse) {
      const long long tiles2k = ((long long)seg->num_docs + 2047) / 2048;
      int bpc = std::max(1, waves_scan_sparse(pl.num_agg_cols <= 1) / (kBlockThreads / 64));
      if (g_engine.blocks_per_cu > 0) bpc = g_engine.blocks_per_cu;
      blocks = (int)std::max<long long>(1, std::min<long long>((tiles2k + 4 * kSparseTiles - 1) / (4 * kSparseTiles), (long long)seg->num_cus * bpc));
    }
    // One dictionary-range leaf (or no filter) in front of at most one aggregated packed column, both of at most kSimpleMaxBits bits:
    // scan_simple_kernel -- the same per-tile code with none of the general machinery, at twice the waves per SIMD (pg_scan_simple.h).
    // (the one-stream shape `SUM(v) WHERE v in range` keeps scan_private_kernel's fused decode)
    bool use_simple = g_engine.scan_simple && use_private && !use_hist && !use_narrow && !use_sparse && !want_bitmap && lw.tile_list == nullptr && sp.num_nodes <= 1 &&
                      pl.num_agg_cols <= 1 && !(g_engine.flags & PG_CFG_PROFILE_WAVES) && !(out && (lw.stats_leap2_flagged || lw.stats_chain_flagged));
    bool simple_set = false;
    if (use_simple && sp.num_nodes == 1) {
      const DevNode& dn = sp.nodes[0];
      // (a dictId SET over a column of at most 16 bits -- its words fit the LDS area -- takes scan_simple_set_kernel: round 6b)
      simple_set = dn.op == PG_FILTER_LEAF && dn.kind == kLeafDictSet && g_engine.set_lds && dn.bits <= 16;
      use_simple = dn.op == PG_FILTER_LEAF && (dn.kind == kLeafDictRange || simple_set) && dn.bits >= 1 && dn.bits <= kSimpleMaxBits && (dn.flags & (kNodeCountEntries | kNodeLeapfrog2)) == 0;
    }
    if (use_simple && pl.num_agg_cols == 1) {
      const DevAggCol& ac = sp.agg_cols[0];
      use_simple = ac.bits >= 1 && ac.bits <= kSimpleMaxBits && !ac.is_raw;
      if (use_simple && sp.num_nodes == 1 && sp.nodes[0].fwd == ac.fwd && sp.nodes[0].bits == ac.bits && ac.need_sum != 0 && ac.need_minmax == 0 && sp.nodes[0].exclusive == 0) use_simple = false;
    }
    // A segment whose tiles all fit the chip at once (one tile per wave: a 10 M-row segment at five waves per SIMD) is latency from end
    // to end -- launch, one round of loads, the hand-off of the workgroups' records to the fold.  Ten waves per workgroup there: 2.5x
    // fewer records, and a folding workgroup of 640 threads takes them in ONE round of loads (256 threads took five for 1221 records).
    int lean_threads = kBlockThreads;
    auto lean_geometry = [&](int wave_cap) {
      const long long tiles2k = ((long long)seg->num_docs + 2047) / 2048;
      static const bool wide_ok = getenv("PINOT_GPU_WIDE_BLOCKS") && getenv("PINOT_GPU_WIDE_BLOCKS")[0] == '1';      // (measured slower at every size: off unless asked for)
      const int wide_waves = kWideBlockThreads / 64;
      const bool wide = wide_ok && wave_cap >= 2 * wide_waves && tiles2k > 64 && tiles2k <= (long long)seg->num_cus * wave_cap && g_engine.blocks_per_cu <= 0;
      lean_threads = wide ? kWideBlockThreads : kBlockThreads;
      const int wpb = lean_threads / 64;
      int bpc = std::max(1, wave_cap / wpb);
      // Up to two rounds of resident waves' worth of tiles (~20 M rows), the hand-off of the workgroups' records to the fold weighs more
      // than a second round of loads: two workgroups per CU (profiles/r4/c1_probe_blocks_per_cu.jsonl: 10 M rows, 1221 workgroups
      // 21.4 us, 512 workgroups 17.8 us; the scan alone 12.9 us)
      static const int small_bpc = getenv("PINOT_GPU_SMALL_BLOCKS_PER_CU") ? atoi(getenv("PINOT_GPU_SMALL_BLOCKS_PER_CU")) : 2;
      if (!wide && small_bpc > 0 && tiles2k <= 2ll * seg->num_cus * wave_cap) bpc = std::min(bpc, small_bpc);
      if (g_engine.blocks_per_cu > 0) bpc = g_engine.blocks_per_cu;
      blocks = (int)std::max<long long>(1, std::min<long long>((tiles2k + wpb - 1) / wpb, (long long)seg->num_cus * bpc));
    };
    if (use_simple) lean_geometry(waves_scan_simple());
    // The same idea for raw INT columns (BASELINE.json configs[0]'s scan pair): one raw-range leaf (or no filter) in front of at most one
    // aggregated raw INT column -- scan_raw_kernel, five waves per SIMD, coalesced reads (pg_scan_raw.h).
    bool use_raw = g_engine.scan_raw && (use_private || use_private_typed) && !use_hist && !use_narrow && !use_sparse && !use_simple && !want_bitmap && lw.tile_list == nullptr &&
                   lw.side == nullptr && sp.num_nodes <= 1 && pl.num_agg_cols <= 1 && sp.num_nodes + pl.num_agg_cols >= 1 && !(g_engine.flags & PG_CFG_PROFILE_WAVES) &&
                   !(out && (lw.stats_leap2_flagged || lw.stats_chain_flagged));
    if (use_raw && sp.num_nodes == 1) {
      const DevNode& dn = sp.nodes[0];
      use_raw = dn.op == PG_FILTER_LEAF && dn.kind == kLeafRawRange && dn.fwd != nullptr && (dn.flags & (kNodeCountEntries | kNodeLeapfrog2)) == 0;
    }
    if (use_raw && pl.num_agg_cols == 1) {
      const DevAggCol& ac = sp.agg_cols[0];
      use_raw = ac.is_raw != 0 && ac.vkind == kValI32 && ac.is_plane == 0 && ac.bits == 32;
    }
    if (use_raw) lean_geometry(waves_scan_raw());
    if (use_narrow) {
      const int per_wave = narrow_single ? kNarrowSingleTiles : kNarrowTiles;
      const long long quads = (((long long)seg->num_docs + 2047) / 2048 + per_wave - 1) / per_wave;
      int bpc = std::max(1, waves_scan_narrow(narrow_single) / (kBlockThreads / 64));
      if (g_engine.blocks_per_cu > 0) bpc = g_engine.blocks_per_cu;
      blocks = (int)std::max<long long>(1, std::min<long long>((quads + 3) / 4, (long long)seg->num_cus * bpc));
    }
    sp.speculate = 1;
    sp.profile = (g_engine.flags & PG_CFG_PROFILE_WAVES) ? 1 : 0;
    st = ensure_partials(ctx, blocks);
    if (st != PG_OK) return st;
    sp.partials = ctx->d_partials;
    sp.out_bitmap = nullptr;
    if (want_bitmap) {
      st = ensure_bitmap(seg, ctx, 0);
      if (st != PG_OK) return st;
      sp.out_bitmap = ctx->d_bitmaps[0];
    }
    sp.sparse_windows = nullptr; sp.sparse_num_windows = 0;
    bool defer_index_and = false;
    const unsigned long long seq = ++ctx->seq;      // the sequence number the query's folded record is published under (a gathering index_and_kernel's, or the scan kernel's)
    if (use_sparse) {
      // A handful of survivors per window (the planner's estimate from the postings' sizes: independent predicates): index_and_kernel reads
      // their values itself -- ONE launch for `SUM(v) WHERE p = 3 AND q = 5 AND r = 7` (BASELINE.json configs[4]); else scan_sparse_kernel
      // walks the window masks behind it (no list, no index_and_finalize_kernel between the two kernels).  PINOT_GPU_INDEX_GATHER=0: never.
      const bool gather = g_engine.index_gather && lw.and_pending && lw.index_and_is_whole_filter && out && pl.num_agg_cols <= kMaxAndGather &&
                          lw.and_expected_docs <= 4.0 * (double)lw.finalize_windows;
      defer_index_and = gather && defer != nullptr && !defer->single && g_engine.batch_index && g_engine.direct_result && !want_bitmap && index_and_shares_a_launch(lw);
      if (defer_index_and) arm_index_gather(&lw, &sp);      // (launched by the batch: index_and_batch_kernel, lean_kind 12)
      else st = launch_index_and(&lw, ctx, gather ? &sp : nullptr);
      if (st != PG_OK) return st;
      sp.tile_list = lw.tile_list; sp.tile_count = lw.tile_count;      // (not read by the kernel; "listed" is what the planner and the statistics go by)
      sp.sparse_windows = lw.and_info; sp.sparse_num_windows = (int32_t)lw.finalize_windows;
    } else if (!want_bitmap && (use_hist || use_private || use_private_typed)) {
      st = complete_index_list(&lw, ctx); if (st != PG_OK) return st;
      sp.tile_list = lw.tile_list; sp.tile_count = lw.tile_count;
    }
    else { st = complete_index_and_bitmap(&lw, ctx); if (st != PG_OK) return st; }
    const bool count_leap2 = out && lw.stats_leap2_flagged && !use_narrow && (use_hist || use_private || use_private_typed);
    const bool count_entries = (out && lw.stats_chain_flagged && (use_hist || use_private || use_private_typed)) || count_leap2;
    if (lw.side != nullptr) {
      // the transducer pass behind this query: a kernel that evaluates the filter with eval_filter_private over every tile leaves the leaves' bitmaps behind
      const bool wrote = !(use_narrow && narrow_single) && !use_sparse && !use_simple && !use_raw && (use_hist || use_private || use_private_typed) && sp.tile_list == nullptr;
      for (int l = 0; l < kMaxLeaves; ++l) sp.leaf_out[l] = wrote ? lw.sp_leaf_out[l] : nullptr;
      sp.leaf_out_enabled = wrote ? 1 : 0;
      lw.side->kernel_wrote = wrote;
    }
    // The transducer walked INSIDE the general lane-private kernel (scan_private_fsm_kernel): machines of at most four states over at most
    // four inputs, every input a leaf of its own in the lowered filter.  No leaf bitmap is written or read back; the tiles' tables are
    // joined by fsm_chain_kernel / fsm_finish_kernel behind the scan, on the query's stream.  PINOT_GPU_FSM_FUSED=0: the separate pass.
    bool fuse_fsm = false;
    sp.fsm_tables = nullptr; sp.fsm_states = 0; sp.fsm_inputs = 0;
    // (NOT in scan_narrow_kernel: measured on `COUNT(*) WHERE p = 3 AND q = 5 AND r = 7` over 4 / 6 / 8-bit columns, 1 B rows -- the walk takes the
    //  kernel from 164 to 199 registers, three to two waves per SIMD, 0.53 -> 0.85 ms, more than the 0.24 ms pass it replaces: 0.767 -> 0.868 ms
    //  for the query; the same query through this kernel: 1.06 ms.  profiles/r5/fsm_walk_inside_the_scan_ab.txt)
    if (lw.side != nullptr && sp.leaf_out_enabled != 0 && g_engine.fsm_fused && g_engine.fsm_perm && use_private && !use_hist && !use_narrow && !use_sparse && !use_simple && !use_raw && out) {
      const fstats::Fsm& f = *lw.side->fsm;
      int max_inc = 0;
      for (uint8_t d : f.delta) max_inc = std::max(max_inc, (int)(d >> 4));
      bool fits = f.num_states <= 4 && f.num_inputs <= 4 && max_inc <= 7 && !f.has_episodes();      // (episodes need the second walk of the pass)
      for (int i = 0; i < f.num_inputs && fits; ++i) fits = lw.side->mapped[i];
      if (fits) {
        fuse_fsm = true;
        sp.fsm_tables = lw.side->tables; sp.fsm_states = f.num_states; sp.fsm_inputs = f.num_inputs;
        for (int l = 0; l < kMaxLeaves; ++l) {
          sp.fsm_input_of_leaf[l] = -1;
          for (int i = 0; i < f.num_inputs; ++i) if (sp.leaf_out[l] != nullptr && sp.leaf_out[l] == lw.side->bitmap[i]) sp.fsm_input_of_leaf[l] = (int8_t)i;
          sp.leaf_out[l] = nullptr;
        }
        sp.leaf_out_enabled = 0;
        memset(sp.fsm_delta, 0, sizeof(sp.fsm_delta));
        for (int st8 = 0; st8 < f.num_states; ++st8)
          for (int in = 0; in < (1 << f.num_inputs); ++in) sp.fsm_delta[(st8 << 4) | in] = f.delta[(size_t)((st8 << f.num_inputs) | in)];
        if (!ctx->h_filter_entries) HIP_TRY(hipHostMalloc((void**)&ctx->h_filter_entries, 8, hipHostMallocDefault));
      }
    }
    sp.filter_entries = nullptr;
    sp.leap_tables = nullptr;
    if (count_leap2) { st = arm_leap_tables(seg, ctx, &sp.leap_tables, nullptr); if (st != PG_OK) return st; }
    else if (lw.stats_leap2_flagged) for (int n = 0; n < sp.num_nodes; ++n) sp.nodes[n].flags &= ~kNodeLeapfrog2;      // a kernel without the count runs this query
    sp.raw64_coalesced = g_engine.raw64_coalesced ? 1 : 0;
    // (the entries counted by the kernel travel in its record: BlockPartial.entries -- no counter to zero, no copy command)
    // The folded record -> the reference's holder types.  Everything is captured by value: pg_execute_batch calls it after this function
    // has returned (the query, the segment and the context's pinned counter outlive the batch).
    const int kernel_id = lw.gathered ? PG_KERNEL_INDEX_AND : use_hist ? PG_KERNEL_SCAN_HIST : use_narrow ? PG_KERNEL_SCAN_NARROW : use_sparse ? PG_KERNEL_SCAN_SPARSE : use_simple ? PG_KERNEL_SCAN_SIMPLE : use_raw ? PG_KERNEL_SCAN_RAW : use_private ? PG_KERNEL_SCAN_PRIVATE : (use_private_typed ? PG_KERNEL_SCAN_PRIVATE_TYPED : PG_KERNEL_SCAN_AGG);
    const HostRecord* host_record = ctx->h_record;
    const int profile_waves = blocks * (geo.threads / 64);
    const size_t num_projected = projected.size();
    auto convert = [q, seg, na, agg_slot_of, lw, kernel_id, host_record, count_entries, count_leap2, profile_waves, num_projected](const BlockPartial& fp, pg_result* out) {
      out->num_aggregations = na;
      out->aggregations = (pg_agg_value*)calloc((size_t)std::max(na, 1), sizeof(pg_agg_value));
      for (int a = 0; a < na; ++a) {
        const pg_aggregation& ag = q->aggregations[a];
        pg_agg_value& v = out->aggregations[a];
        v.count = (int64_t)fp.count;
        v.min = std::numeric_limits<double>::infinity();
        v.max = -std::numeric_limits<double>::infinity();
        if (ag.function == PG_AGG_COUNT) continue;
        const int ac = agg_slot_of[(size_t)a];
        const ColumnDev& col = seg->cols[(size_t)ag.column];
        const bool plane = lw.plane_cols[(size_t)ag.column] != 0;
        const bool raw = col.encoding == PG_FWD_RAW_FIXED_BYTE;
        if (ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG) {
          if (col.vkind == kValF64 || col.vkind == kValF32) {
            // SumAggregationFunction on FLOAT / DOUBLE: a double sum; the addition order differs from the reference's
            // doc order, so the last bits may (tests/test_gpu_typed.py states the tolerance)
            v.sum = fp.fsum[ac];
            v.sum_i64 = 0;
            v.sum_exact = 0;
          } else if (col.vkind == kValI64) {
            // LONG values: the int64 sum is exact unless it wrapped, which the double image of the same sum reveals (a wrap moves
            // it by a multiple of 2^64).  Wrapped: report the double sum, like the reference's double accumulation, inexact.
            v.sum_i64 = fp.sum[ac];
            const bool wrapped = std::fabs(fp.fsum[ac] - (double)fp.sum[ac]) > 4.6e18;
            v.sum_exact = wrapped ? 0 : 1;
            v.sum = wrapped ? fp.fsum[ac] : (double)v.sum_i64;
          } else {
            // value plane: sum(value) = count * base + sum(value - base); offset dictionaries add count * value_base
            set_integer_sum(&v, (__int128)fp.sum[ac] * (__int128)sum_scale(col, plane) + (__int128)fp.count * (__int128)sum_base(col, plane));
          }
        } else if (fp.count > 0) {
          if (raw && col.vkind != kValI32) {
            double mn = key64_to_double(col, fp.kmin64[ac]), mx = key64_to_double(col, fp.kmax64[ac]);
            if (mx != mx) mn = mx;      // Math.min / Math.max propagate NaN
            if (ag.function == PG_AGG_MIN) v.min = mn;
            if (ag.function == PG_AGG_MAX) v.max = mx;
          } else {
            if (ag.function == PG_AGG_MIN) v.min = agg_value_double(col, fp.kmin[ac], plane);
            if (ag.function == PG_AGG_MAX) v.max = agg_value_double(col, fp.kmax[ac], plane);
          }
        }
      }
      out->dominant_kernel = kernel_id;
      for (int c = 0; c < 4; ++c) out->profile_cycles[c] = fp.cyc[c];
      out->profile_waves = profile_waves;
      out->stats.num_docs_scanned = (int64_t)fp.count;
      // (the scan kernels count into their records; a leap-frogging a AND b adds what leapfrog2_chain_kernel left in the pinned record)
      finish_filter_stats(lw, seg, (int64_t)fp.entries + (count_leap2 ? host_record->leap_correction : 0), count_entries, out);
      out->stats.num_entries_scanned_post_filter = (int64_t)fp.count * (int64_t)num_projected;
      out->stats.num_total_docs = seg->num_docs;
    };
    // The workgroups' records are folded by the scan kernel's last workgroup, straight into the pinned host record.
    // Measured (profiles/r3, tools/ab_r3.py): the fold costs the kernel 5.5 us of tail on a 1024-workgroup grid where the finalize launch
    // costs 8.8 us (boundary + a one-workgroup kernel) -- 13 % of a 10 M-row query's device time.  Round 4: at EVERY size (1 B-row scans
    // were left with the separate launch so that the scan kernel's profiler duration was "the scan and nothing else"; the query paid
    // ~15 us for that, and its roofline fraction is a statement about the query, not about its largest kernel).
    const bool folded = g_engine.fold_finalize >= 0 ? g_engine.fold_finalize != 0 : true;
    sp.done_counter = folded ? ctx->d_done : nullptr;
    sp.host_out = g_engine.direct_result ? ctx->h_record_dev : nullptr;
    sp.host_seq = seq;
    // fields a fold reduces: the aggregation slots in use (the histogram kernel keeps its checksum in slot 1), typed extras only for the typed kernels
    sp.fold_slots = use_hist ? 2 : pl.num_agg_cols;
    sp.fold_typed = (!use_raw && (use_private_typed || (!use_hist && !use_narrow && !use_private && typed))) ? 1 : 0;
    sp.lane_skip = g_engine.lane_skip ? 1 : 0;
    sp.set_leaves_in_lds = 0;
    if (g_engine.set_lds) for (int nd = 0; nd < sp.num_nodes; ++nd) if (sp.nodes[nd].op == PG_FILTER_LEAF && sp.nodes[nd].kind == kLeafDictSet) sp.set_leaves_in_lds = 1;
    if (use_hist) sp.set_leaves_in_lds = hist_set_off != 0 ? 1 + (int32_t)hist_set_off : 0;      // (the histogram kernel keeps the area in its dynamic LDS, behind the counters)
    sp.sparse_lanes = g_engine.sparse_lanes;
    sp.fold_one_counter = g_engine.fold_one_counter;
    if (defer_index_and) {
      auto item = std::make_shared<LoweredItem>();
      memset(&item->sp, 0, sizeof(item->sp));
      item->sp.lean_kind = 12;
      item->and_params = std::make_shared<IndexAndParams>(lw.and_params);
      item->and_windows = lw.finalize_windows;
      item->blocks = (int)std::min<long long>(((long long)lw.finalize_windows + index_and_batch_block_waves() - 1) / index_and_batch_block_waves(), (long long)lw.and_num_cus * index_and_batch_blocks_per_cu());
      item->one_slot = pl.num_agg_cols <= 1;
      item->convert = convert;
      item->plane_columns = planes.columns;
      defer->item = std::move(item);
      defer->cacheable = !lw.plane_pending;
      defer->planes.reset(new PlaneHold(std::move(planes)));
      return kDeferred;
    }
    if (defer != nullptr) {
      // (the shared launch is for the many small segments of a server: a segment that fills the chip on its own -- more tiles than a few
      //  rounds of resident waves -- runs the kernel the planner picked for it, concurrently with the others, on a worker thread's stream:
      //  eight 1 B-row items 4.72 ms in one launch, 4.45 ms as eight overlapping launches)
      // kinds of shared launch (ScanParams.lean_kind; pg_execute_batch groups a batch's deferred items by device and kind, one launch each):
      //   0 scan_private_batch_kernel (the general body)   1 / 2 scan_lean_batch_kernel (scan_simple / scan_raw shape)
      //   3 / 4 / 5 scan_hist_batch_kernel<8 | 16 | 32> (SUM through the LDS histogram: dictionaries without structure, plain counters)
      //   6 group_lds_batch_kernel (group-bys of the LDS-table form: lowered in the group-by branch below)
      //   7 / 8 scan_narrow_batch_kernel<general | single leaf> (COUNT under filters over columns of at most 8 bits)
      //   9 / 10 / 11 scan_typed_batch_kernel<1 | 2 | kMaxAggCols> (raw and 8-byte aggregated columns)
      const bool hist_item = use_hist && !hist_guarded && g_engine.batch_hist;
      const bool narrow_item = use_narrow && g_engine.batch_more;
      const bool typed_item = use_private_typed && !use_raw && !use_sparse && g_engine.batch_more;
      if (!defer->single && ((use_private && !use_hist && !use_narrow) || use_raw || hist_item || narrow_item || typed_item) && !use_sparse && !want_bitmap && out && sp.tile_list == nullptr && !count_entries && !ctx->pre_enqueued && g_engine.direct_result &&
          lw.side == nullptr && ((long long)seg->num_docs + 2047) / 2048 <= kBatchMaxTiles &&
          stats_is_final) {
        // items of scan_simple_kernel's / scan_raw_kernel's shape share a launch of their own kind (scan_lean_batch_kernel), the rest the general one
        const bool lean_batch = g_engine.lean_batch;
        sp.lean_kind = hist_item ? (hist_cw == 8 ? 3 : (hist_cw == 16 ? 4 : 5)) : narrow_item ? (narrow_single ? 8 : 7) : typed_item ? (pl.num_agg_cols <= 1 ? 9 : (pl.num_agg_cols == 2 ? 10 : 11))
                       : use_simple ? (simple_set ? 13 : 1) : (use_raw ? 2 : 0);      // (13: scan_lean_batch_kernel<13>, the simple body with its one set leaf in LDS)
        if (!lean_batch && (sp.lean_kind == 1 || sp.lean_kind == 13)) sp.lean_kind = 0;      // (a raw-shaped item has no general form when its column is aggregated: it stays lean)
        if (sp.lean_kind == 2 && !lean_batch && use_private) sp.lean_kind = 0;
        auto item = std::make_shared<LoweredItem>();
        item->sp = sp;
        item->blocks = blocks;
        item->one_slot = pl.num_agg_cols <= 1;
        item->convert = convert;
        item->plane_columns = planes.columns;
        item->sets = lw.set_leaves;
        if (hist_item) { item->hist_lds = hist_lds; item->hist_cw = hist_cw; item->hist_col = hist_col; }
        defer->item = std::move(item);
        defer->cacheable = !lw.plane_pending;
        defer->planes.reset(new PlaneHold(std::move(planes)));
        return kDeferred;
      }
    }
    // HIP events (PG_CFG_TIME_KERNELS): [ev_first, ev_last] brackets the query's device work, [ev[1], ev[2]] the scan kernel.  Each
    // record is a packet of its own on the queue, so a query that runs nothing but the scan kernel records just the two.
    // (the chain kernel / copy commands behind the scan kernel; a kernel that leaves leaf bitmaps behind for the transducer pass must have
    //  RETIRED before that pass reads them -- its plain stores are only ordered by the end of the kernel, not by the pinned record's seq)
    if (lw.gathered) {
      // index_and_kernel did the aggregation: its counter lines are the query's record
      // (timed runs: ev[0] in front of the kernel and ev[3] behind the copy, like the COUNT(*) path -- two more records between the
      //  kernel and its copy were two more packets on the queue, ~5 us that only a timed run paid)
      HIP_TRY(hipMemcpyAsync(ctx->h_and_shards, ctx->d_and_counters + 2, kAndShardBytes, hipMemcpyDeviceToHost, ctx->stream));
      if (timed) HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
      ctx->ev_last = 3;
      exec_mark(3);
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      exec_mark(4);
      BlockPartial g;
      st = read_index_and_shards(ctx, pl.num_agg_cols, &g);
      if (st != PG_OK) return st;
      *ctx->h_partial = g;
    } else {
    const bool post_work = !g_engine.direct_result || count_leap2 || want_bitmap || sp.leaf_out_enabled || fuse_fsm;
    if (timed) HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    // the single-aggregated-column instantiation needs a third fewer registers (one more wavefront per SIMD)
    const bool one = pl.num_agg_cols <= 1;
    if (use_hist) launch_scan_hist(hist_cw, hist_guarded, blocks, hist_lds, ctx->stream, sp);
    else if (use_narrow) launch_scan_narrow(narrow_single, blocks, ctx->stream, sp);
    else if (use_sparse) launch_scan_sparse(one, blocks, ctx->stream, sp);
    else if (use_simple) launch_scan_simple(blocks, lean_threads, ctx->stream, sp, simple_set && use_simple);
    else if (use_raw) launch_scan_raw(blocks, lean_threads, ctx->stream, sp);
    else if (use_private && fuse_fsm) launch_scan_private_fsm(pl.num_agg_cols, blocks, ctx->stream, sp);
    else if (use_private) launch_scan_private(pl.num_agg_cols, blocks, ctx->stream, sp);
    else if (use_private_typed) launch_scan_private_typed(pl.num_agg_cols, blocks, ctx->stream, sp);
    else launch_scan_agg(g_engine.use_dma, one, typed, blocks, geo.threads, lds, ctx->stream, sp);
    HIP_TRY(hipGetLastError());
    if (timed) HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    if (!folded) {
      finalize_partials_kernel<<<dim3(1), dim3(kBlockThreads), 0, ctx->stream>>>(ctx->d_partials, blocks, g_engine.direct_result ? ctx->h_record_dev : nullptr, seq,
                                                                                  sp.fold_slots, sp.fold_typed, sp.profile);
      HIP_TRY(hipGetLastError());
    }
    if (!g_engine.direct_result) HIP_TRY(hipMemcpyAsync(ctx->h_partial, ctx->d_partials + blocks, sizeof(BlockPartial), hipMemcpyDeviceToHost, ctx->stream));
    if (count_leap2) { st = launch_leap_chain(seg, ctx, seq); if (st != PG_OK) return st; }
    if (fuse_fsm) {
      // the tiles' tables -> the count (the same two kernels that end the separate pass), then eight bytes to the context's pinned counter
      const FsmSide& fs = *lw.side;
      fsm_chain_kernel<<<dim3((unsigned)fs.num_chunks), dim3(1024), 0, ctx->stream>>>(fs.tables, fs.num_tiles, sp.fsm_states, fs.chunks);
      HIP_TRY(hipGetLastError());
      fsm_finish_kernel<<<dim3(1), dim3(1024), (size_t)fs.num_chunks * (size_t)sp.fsm_states * 4, ctx->stream>>>(fs.chunks, (int)fs.num_chunks, sp.fsm_states, fs.entries);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(ctx->h_filter_entries, fs.entries, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (want_bitmap) {
      const int64_t need = ((int64_t)seg->num_docs + 63) / 64;
      if (d_out_bitmap_request) {
        if (need) HIP_TRY(hipMemcpyAsync(d_out_bitmap_request, ctx->d_bitmaps[0], (size_t)need * 8, hipMemcpyDeviceToDevice, ctx->stream));
      } else {
        if (host_bitmap_words < need) return fail(PG_ERR_INVALID_ARGUMENT, "bitmap buffer has %lld words, need %lld", (long long)host_bitmap_words, (long long)need);
        if (need) HIP_TRY(hipMemcpyAsync(host_bitmap, ctx->d_bitmaps[0], (size_t)need * 8, hipMemcpyDeviceToHost, ctx->stream));
      }
    }
    if (timed && (post_work || !folded)) HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->ev_last = (post_work || !folded) ? 3 : 2;
    exec_mark(3);
    if (g_engine.poll_result && !post_work && !timed) {
      // nothing follows the kernel on the stream: the record's sequence number is the completion signal
      volatile unsigned long long* flag = &ctx->h_record->seq;
      st = wait_polled(ctx->stream, (long long)seg->num_docs, [&] { return *flag == seq; });
      if (st != PG_OK) return st;
      if (*flag != seq) return fail(PG_ERR_INTERNAL, "the scan kernel finished without publishing its record");
    } else {
      HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    exec_mark(4);
    if (count_leap2 && ctx->h_record->leap_seq != seq) return fail(PG_ERR_INTERNAL, "the leap-frog chain kernel did not publish its result");
    if (g_engine.direct_result && ctx->h_record->seq != seq) return fail(PG_ERR_INTERNAL, "the scan kernel's record carries sequence %llu, expected %llu", ctx->h_record->seq, seq);
    }
    const BlockPartial& fp = *ctx->h_partial;
    // (the in-kernel fold orders the workgroups' records against their arrival counter through write-through stores, not through a
    //  release / acquire pair: every record carries its launch's stamp and a foreign one is an error, never an answer)
    if (fp.flags & kPartialStale) return fail(PG_ERR_INTERNAL, "the scan kernel's fold read a record that was not written by this launch (sequence %llu)", seq);
    // plain narrow counters: the counters must add up to the matches (a wrapped counter always leaves the total short)
    const bool hist_wrapped = use_hist && !hist_guarded && hist_cw < 32 && (unsigned long long)fp.sum[1] != fp.count;
    if (use_hist && (hist_wrapped || (fp.flags & kPartialHistAlarm))) {
      // Skewed dictIds: the histogram's sum is not used.  From now on the column runs in the next tier -- guarded counters, which count
      // hot dictIds exactly through guard-bit claims, then the value plane / gather path -- and this query is answered again.
      __atomic_store_n(&seg->cols[(size_t)hist_col].hist_tier, hist_wrapped ? 1 : 2, __ATOMIC_RELAXED);
      seg->plane_epoch.fetch_add(1, std::memory_order_acq_rel);      // (what the plan cache holds for this segment was lowered for the plain tier)
      release_ctx(seg, ctx);
      guard.ctx = nullptr;
      return execute_impl(seg, q, out, d_out_bitmap_request, host_bitmap, host_bitmap_words, out_cardinality, allow_metadata_plan, nullptr, side);
    }
    if (out_cardinality) *out_cardinality = (int64_t)fp.count;
    if (out) convert(fp, out);
    if (out && fuse_fsm) {
      // (the stream was synchronised: the walk's count is in the pinned counter)
      out->stats.num_entries_scanned_in_filter = (int64_t)*ctx->h_filter_entries;
      out->filter_entries_exact = 1;
      lw.side->fused = true;
    }
  } else {
    // ---------------- group-by (ArrayBasedHolder) ----------------
    GroupParams gp;
    memset(&gp, 0, sizeof(gp));
    int group_slot[kMaxGroupCols] = {}, group_mult[kMaxGroupCols] = {};
    PlanGroupAgg plan_aggs[kMaxGroupAggs];
    long long product = 1;
    std::vector<int> cards;
    bool no_dict_keys = false;            // a key is a raw column read through its key image, or the null-key image of one
    for (int g = 0; g < ng; ++g) {
      int c = q->group_by_columns[g];
      if (c < 0 || c >= num_cols_total) return fail(PG_ERR_INVALID_ARGUMENT, "group-by column %d out of range", c);
      add_projected(seg->cols[(size_t)c].key_image_of >= 0 ? seg->cols[(size_t)c].key_image_of : c);      // numEntriesScannedPostFilter counts the caller's column
      if (seg->cols[(size_t)c].encoding != PG_FWD_FIXED_BIT_DICT) {
        // NoDictionarySingle/MultiColumnGroupKeyGenerator: the raw INT / LONG column through its key image (value - min as the dictId)
        st = ensure_key_image(seg, c, &c);
        if (st != PG_OK) return st;
      }
      const ColumnDev& col = seg->cols[(size_t)c];
      no_dict_keys |= col.key_image_of >= 0;
      int s = slot_for(&lw, seg, c);
      if (s < 0) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
      pl.cols[s].in_agg = 1;
      group_slot[g] = s;
      cards.push_back(col.cardinality);
    }
    // DictionaryBasedGroupKeyGenerator.java:164-184: up to arrayBasedThreshold (10 000) the raw key IS the group id (ArrayBasedHolder);
    // above it the reference hashes raw keys (IntMapBasedHolder) -- here the table stays direct-indexed, in HBM, one slot per raw
    // key, and only the groups that exist come back.  2^24 slots keep the 24-bit key multiplies exact and the table <= 1.2 GB.
    // Beyond an int (Long / ArrayMap holders) the table is hashed: `product` is then its number of slots (plan_hash_holder).
    HashPlan hash_plan;
    st = plan_hash_holder(seg, cards, &hash_plan);
    if (st != PG_OK) return st;
    if (hash_plan.kind == 0) {
      for (int g = 0; g < ng; ++g) { group_mult[g] = (int32_t)product; product *= cards[(size_t)g]; }
    } else {
      product = hash_plan.slots;
      if (q->flags & kQueryHashHolder) return fail(PG_ERR_UNSUPPORTED, "group-by with raw keys beyond an int under null handling (plan-time fallback)");
    }
    // (kQueryHashHolder: the no-dictionary key generators of null handling hand out group ids by first appearance up to numGroupsLimit
    //  whatever the key space: the compaction path below is the one that honours the limit)
    // A raw key column always runs the no-dictionary generators (DefaultGroupByExecutor.java:106-121): _globalGroupIdUpperBound =
    // numGroupsLimit whatever the key space (NoDictionarySingleColumnGroupKeyGenerator.java:73-79), ids by first appearance.
    const bool first_appearance = (q->flags & kQueryHashHolder) != 0 || (no_dict_keys && hash_plan.kind == 0 && (long long)(q->num_groups_limit > 0 ? q->num_groups_limit : 100000) < product);
    const bool map_based = product > 10000 || first_appearance || hash_plan.kind != 0;
    bool typed_direct = false;            // an aggregation input is a raw LONG / FLOAT / DOUBLE column: group_typed_direct_kernel
    gp.num_group_cols = ng;
    gp.num_groups = (int32_t)product;
    gp.dense_ok = 1;
    std::vector<int> dev_agg_of((size_t)std::max(na, 1), -1);
    for (int a = 0; a < na; ++a) {
      const pg_aggregation& ag = q->aggregations[a];
      if (ag.function == PG_AGG_COUNT) continue;
      if (ag.function < PG_AGG_COUNT || ag.function > PG_AGG_AVG) return fail(PG_ERR_UNSUPPORTED, "aggregation function %d", ag.function);
      if (ag.column < 0 || ag.column >= num_cols_total) return fail(PG_ERR_INVALID_ARGUMENT, "aggregation column %d out of range", ag.column);
      add_projected(ag.column);
      int s = slot_for(&lw, seg, ag.column, lw.plane_cols[(size_t)ag.column] != 0);
      if (s < 0) return fail(PG_ERR_UNSUPPORTED, "query references more than %d columns", kMaxCols);
      pl.cols[s].in_agg = 1;
      const int kind = (ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG) ? kGroupSum : (ag.function == PG_AGG_MIN ? kGroupMin : kGroupMax);
      int da = -1;
      for (int i = 0; i < gp.num_group_aggs; ++i) if (plan_aggs[i].col == s && plan_aggs[i].kind == kind) da = i;
      if (da < 0) {
        if (gp.num_group_aggs >= kMaxGroupAggs) return fail(PG_ERR_UNSUPPORTED, "more than %d distinct group-by aggregations", kMaxGroupAggs);
        da = gp.num_group_aggs++;
        plan_aggs[da] = PlanGroupAgg{s, kind};
      }
      dev_agg_of[(size_t)a] = da;
    }
    gp.wide_keys = product > (1ll << 24) ? 1 : 0;
    // (hashed holders: one more word per slot for its key, and the first table of an ArrayMap-range key)
    const size_t table_words = (size_t)gp.num_groups * (size_t)(1 + gp.num_group_aggs + (hash_plan.kind ? 1 : 0)) + (size_t)hash_plan.level_slots();
    if ((unsigned long long)table_words * 8ull > g_engine.group_table_bytes)
      return fail(PG_ERR_UNSUPPORTED, "group-by table of %lld slots x %d words exceeds the %llu-byte budget (PINOT_GPU_GROUP_TABLE_BYTES)", product, 1 + gp.num_group_aggs,
                  (unsigned long long)g_engine.group_table_bytes);
    st = ensure_table(ctx, table_words, map_based ? 0 : table_words);
    if (st != PG_OK) return st;
    struct TableTrim {            // a table of the upper IntMapBasedHolder range goes back to the allocator with the query
      ExecCtx* c;
      ~TableTrim() {
        if (c->table_capacity * 8 > kGroupTableKeepBytes) { (void)hipFree(c->d_table); c->d_table = nullptr; c->table_capacity = 0; }
      }
    } table_trim{ctx};
    gp.table_count = ctx->d_table;
    gp.table_acc = reinterpret_cast<long long*>(ctx->d_table + gp.num_groups);
    gp.hash_kind = hash_plan.kind;
    gp.hash_levels = hash_plan.levels;
    if (hash_plan.kind != 0) {
      gp.hash_mask = (unsigned long long)gp.num_groups - 1ull;
      gp.hash_keys = ctx->d_table + (size_t)gp.num_groups * (size_t)(1 + gp.num_group_aggs);
      unsigned long long* next_table = gp.hash_keys + gp.num_groups;
      for (int l = 0; l < hash_plan.levels; ++l) {
        gp.hash_split[l] = hash_plan.split[l];
        gp.hash_mask_lvl[l] = (unsigned long long)hash_plan.slots_lvl[l] - 1ull;
        gp.hash_keys_lvl[l] = next_table;
        next_table += hash_plan.slots_lvl[l];
      }
      for (int g = 0; g < ng; ++g) gp.key_mult[g] = hash_plan.mult[g];
    }
    const size_t table_bytes = table_words * 8;
    Geometry geo;
    const int group_wave_cap = waves_scan_group();
    finish_geometry(seg, &lw, table_bytes, false, g_engine.group_waves > 0 ? std::min(g_engine.group_waves, kGroupBlockThreads / 64) : kGroupBlockThreads / 64, group_wave_cap, &geo);
    if ((size_t)sp.wave_lds_bytes > kLdsBudget) return fail(PG_ERR_UNSUPPORTED, "query needs %d bytes of LDS per wavefront", sp.wave_lds_bytes);
    gp.use_lds_table = geo.table_in_lds ? 1 : 0;
    const int blocks = geo.blocks;
    const size_t lds = geo.lds;
    sp.speculate = 1;
    for (int g = 0; g < ng; ++g) {
      const DevColumn& c = pl.cols[group_slot[g]];
      gp.group_keys[g] = DevGroupKey{c.bits, c.slot_off, group_mult[g], 0, c.fwd};
    }
    for (int a = 0; a < gp.num_group_aggs; ++a) {
      const DevColumn& c = pl.cols[plan_aggs[a].col];
      DevGroupAgg& ga = gp.group_aggs[a];
      ga.kind = plan_aggs[a].kind; ga.bits = c.bits; ga.slot_off = c.slot_off; ga.is_raw = c.is_raw; ga.is_plane = c.is_plane;
      ga.dict_bytes = c.dict_bytes; ga.fwd = c.fwd; ga.dict = c.dict;
      // MIN / MAX run on dictIds whatever the value type; only a SUM reads 8-byte / floating-point dictionary entries
      ga.vkind = (ga.kind == kGroupSum) ? c.vkind : kValI32;
      if (c.is_raw && c.vkind != kValI32) { typed_direct = true; ga.vkind = c.vkind; }      // group_typed_direct_kernel: the value type decides the accumulator
      // (the SUM of a dictionary column with 8-byte values under a Long / ArrayMap holder: the staged kernel that gathers such values has no
      //  hashed table, group_typed_direct_kernel<.., kHash> gathers them too)
      if (hash_plan.kind != 0 && ga.vkind != kValI32) typed_direct = true;
      if (ga.vkind != kValI32) gp.dense_ok = 0;
      if (ga.vkind == kValI64 && !c.is_raw) {
        // the table slot is one wrapping int64: refuse (plan-time fallback) when numDocs * max|value| could overflow it
        const ColumnDev& sc = seg->cols[(size_t)(lw.col_of_slot[(size_t)plan_aggs[a].col] / 2)];
        const double max_abs = std::max(std::fabs((double)sc.h_dict_i64.front()), std::fabs((double)sc.h_dict_i64.back()));
        if ((double)seg->num_docs * max_abs >= 9.2e18) return fail(PG_ERR_UNSUPPORTED, "group-by SUM of LONG column %s could overflow int64", sc.name.c_str());
      }
    }
    // Without a filter every tile is aggregated in full: the lane-private kernel decodes straight from HBM and needs LDS only
    // for the table (group_private_kernel).
    // every leaf kind the lane-private filter implements (scan / set / bitmap / docId-range leaves, raw INT ranges)
    bool private_leaves = true;
    for (int l = 0; l < pl.num_leaves; ++l) private_leaves &= pl.leaves[l].kind <= kLeafBitmap || pl.leaves[l].kind == kLeafDocRange;
    if (typed_direct && !private_leaves) return fail(PG_ERR_UNSUPPORTED, "group-by aggregation of a raw 8-byte column under a raw 8-byte range predicate (plan-time fallback)");
    const bool use_private = (g_engine.group_private || hash_plan.kind != 0) && private_leaves && gp.dense_ok && !want_bitmap && !typed_direct;
    if (hash_plan.kind != 0 && !use_private && !typed_direct) return fail(PG_ERR_UNSUPPORTED, "group-by with raw keys beyond an int: only 32-bit-domain or raw 8-byte aggregations under lane-private filter leaves");
    int pblocks = blocks, pthreads = geo.threads;
    size_t plds = lds;
    if (use_private) {
      const int private_wave_cap = waves_group_private();
      const bool in_lds = table_bytes <= 96 * 1024;
      gp.use_lds_table = in_lds ? 1 : 0;
      int waves = in_lds ? kGroupBlockThreads / 64 : kBlockThreads / 64;
      if (g_engine.group_waves > 0) waves = std::min(waves, g_engine.group_waves);
      while (waves > private_wave_cap) waves >>= 1;
      pthreads = waves * 64;
      plds = in_lds ? table_bytes : 0;
      int bpc = std::max(1, private_wave_cap / waves);
      if (in_lds) bpc = std::max(1, std::min(bpc, (int)(kLdsBudget / std::max<size_t>(plds, 1))));
      if (g_engine.blocks_per_cu > 0) bpc = g_engine.blocks_per_cu;
      const long long tiles2k = ((long long)seg->num_docs + 2047) / 2048;
      pblocks = (int)std::max<long long>(1, std::min<long long>((tiles2k + waves - 1) / waves, (long long)seg->num_cus * bpc));
    }
    // Count packing: when the first summed value plane is narrow enough, its 64-bit LDS slot carries (count << shift) | sum
    // and the separate count atomic disappears.  Safe while a workgroup sees fewer than 2^cbits docs:
    // sum < 2^cbits * 2^w = 2^shift and count < 2^cbits, so cbits + shift <= 64 never carries into or out of the count.
    gp.packed_agg = -1;
    gp.packed_shift = 0;
    if (gp.use_lds_table && g_engine.group_pack) {
      const long long waves_total = use_private ? (long long)pblocks * (pthreads / 64) : (long long)blocks * (geo.threads / 64);
      const long long tiles_total = use_private ? ((long long)seg->num_docs + 2047) / 2048 : (long long)sp.num_tiles;
      const long long tiles_per_wave = (tiles_total + waves_total - 1) / waves_total;
      const long long docs_per_block = use_private ? tiles_per_wave * (pthreads / 64) * 2048 : tiles_per_wave * (geo.threads / 64) * 64 * sp.tile_steps;
      int cbits = 1;
      while ((1ll << cbits) <= docs_per_block) ++cbits;
      for (int a = 0; a < gp.num_group_aggs; ++a) {
        const DevGroupAgg& ga = gp.group_aggs[a];
        if (ga.kind == kGroupSum && ga.is_plane && !ga.is_raw && 2 * cbits + ga.bits <= 64) { gp.packed_agg = a; gp.packed_shift = std::max(32, cbits + ga.bits); break; }
      }
    }
    // group_private_kernel's LDS table in as many bank-interleaved copies as the workgroup's share of the CU's LDS holds (pg_kernels.h,
    // lds_group_table_bytes): C3's 1000 groups x (SUM + MAX) are 12 KB a copy, eight copies for the one 16-wave workgroup of a CU.
    gp.lds_log_replicas = 0;
    // (a filter with dictId-set leaves: kSetLdsWords words of the workgroup's LDS hold the sets, behind the table -- pg_kernels.h stage_filter_sets)
    sp.set_leaves_in_lds = 0;
    if (g_engine.set_lds) for (int nd = 0; nd < sp.num_nodes; ++nd) if (sp.nodes[nd].op == PG_FILTER_LEAF && sp.nodes[nd].kind == kLeafDictSet) sp.set_leaves_in_lds = 1;
    const size_t set_area = (use_private && sp.set_leaves_in_lds != 0) ? (size_t)kSetLdsWords * 4 : 0;
    if (use_private && gp.use_lds_table) {
      const int resident = std::max(1, std::min(waves_group_private() / std::max(1, pthreads / 64), g_engine.blocks_per_cu > 0 ? g_engine.blocks_per_cu : 1 << 30));
      const size_t budget = kLdsBudget / (size_t)resident - set_area;
      int log_r = 0;
      while (log_r < g_engine.group_log_replicas && (size_t)lds_group_table_bytes(gp, log_r + 1) <= budget) ++log_r;
      gp.lds_log_replicas = log_r;
      plds = lds_group_table_bytes(gp, log_r);
    }
    gp.set_lds_off = -1;
    if (set_area != 0 && plds + set_area <= kLdsBudget) { gp.set_lds_off = (int32_t)((plds + 15) & ~(size_t)15); plds = (size_t)gp.set_lds_off + set_area; }
    gp.scan = sp;
    gp.scan.partials = nullptr;
    gp.scan.out_bitmap = nullptr;
    if (lw.tile_list != nullptr) { st = complete_index_list(&lw, ctx); if (st != PG_OK) return st; }
    gp.scan.tile_list = lw.tile_list;            // read by group_private_kernel only
    gp.scan.tile_count = lw.tile_count;
    gp.scan.filter_entries = nullptr;
    // pg_execute_batch: a group-by of the LDS-table form over a small segment shares ONE launch with the batch's other such items
    // (group_lds_batch_kernel; ScanParams.lean_kind 6) -- no table init, no compaction launches: the item's slice of the batch's table is
    // all-zero before the launch (zero-identity keys), comes back whole in the batch's one copy, and the host keeps the slots whose
    // count is not zero.  What GroupByCombineOperator.java:102-165 gets from one task per segment.
    if (defer != nullptr && g_engine.batch_group && use_private && gp.use_lds_table && hash_plan.kind == 0 && !first_appearance && !typed_direct && !want_bitmap && out &&
        lw.tile_list == nullptr && lw.side == nullptr && !lw.stats_leap2_flagged && !lw.stats_chain_flagged && !ctx->pre_enqueued &&
        stats_is_final &&
        (long long)(q->num_groups_limit > 0 ? q->num_groups_limit : 100000) >= product && (defer->single || ((long long)seg->num_docs + 2047) / 2048 <= kBatchMaxTiles)) {
      auto item = std::make_shared<LoweredItem>();
      item->gp = std::make_shared<GroupParams>(gp);
      item->gp->zero_identity = 1;
      item->gp->scan.lean_kind = 6;
      item->sp.lean_kind = 6;
      item->blocks = pblocks;
      item->group_threads = pthreads;
      item->group_lds = plds;
      item->group_table_words = ((size_t)gp.num_groups * (size_t)(1 + gp.num_group_aggs) + 31) & ~(size_t)31;      // (slices start on 256-byte boundaries)
      item->plane_columns = planes.columns;
      item->sets = lw.set_leaves;
      const int G = gp.num_groups, NA = gp.num_group_aggs;
      int agg_kind[kMaxGroupAggs] = {};
      for (int a = 0; a < NA; ++a) agg_kind[a] = gp.group_aggs[a].kind;
      std::array<int, kMaxGroupAggs> kinds{};
      for (int a = 0; a < NA; ++a) kinds[(size_t)a] = agg_kind[a];
      const size_t num_projected = projected.size();
      // (captured by value and kept with the cached item: what the conversion needs of the lowering -- the planes its aggregations read through
      //  and the statistics plan -- not the whole Lowered with its index-AND parameter block: advisor, round 5)
      const std::vector<char> plane_cols = lw.plane_cols;
      const fstats::Plan stats_plan = lw.stats_plan;
      const int stats_scan_leaves = lw.stats_scan_leaves;
      item->convert_group = [q, seg, na, ng, cards, dev_agg_of, plane_cols, stats_plan, stats_scan_leaves, G, kinds, num_projected, no_dict_keys](const unsigned long long* table, pg_result* out) {
        int num_present = 0;
        for (int g = 0; g < G; ++g) num_present += table[g] != 0ull ? 1 : 0;
        out->num_aggregations = na;
        out->dominant_kernel = PG_KERNEL_GROUP_PRIVATE;
        out->num_groups = num_present;
        out->group_id_upper_bound = no_dict_keys ? (q->num_groups_limit > 0 ? q->num_groups_limit : 100000) : G;
        out->group_ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)std::max(num_present, 1));
        out->group_aggregations = (pg_agg_value*)calloc((size_t)std::max(num_present, 1) * (size_t)std::max(na, 1), sizeof(pg_agg_value));
        out->group_key_dict_ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)std::max(num_present, 1) * (size_t)ng);
        out->group_key_kind = 0;
        long long docs = 0;
        int k = 0;
        for (int g = 0; g < G; ++g) {
          const unsigned long long group_docs = table[g];
          if (group_docs == 0ull) continue;
          docs += (long long)group_docs;
          out->group_ids[k] = g;
          long long raw = g;
          for (int c = 0; c < ng; ++c) { out->group_key_dict_ids[(size_t)k * (size_t)ng + (size_t)c] = (int32_t)(raw % cards[(size_t)c]); raw /= cards[(size_t)c]; }
          for (int a = 0; a < na; ++a) {
            const pg_aggregation& ag = q->aggregations[a];
            pg_agg_value& v = out->group_aggregations[(size_t)k * (size_t)na + (size_t)a];
            v.count = (int64_t)group_docs;
            v.min = std::numeric_limits<double>::infinity();
            v.max = -std::numeric_limits<double>::infinity();
            if (ag.function == PG_AGG_COUNT) continue;
            const int da = dev_agg_of[(size_t)a];
            long long acc = (long long)table[(size_t)G * (size_t)(1 + da) + (size_t)g];
            // zero-identity keys of the shared launch (group_private_body's flush): MIN travelled as 2^31 - v, MAX as v + 2^31 + 1
            if (kinds[(size_t)da] == kGroupMin) acc = 0x80000000ll - acc;
            else if (kinds[(size_t)da] == kGroupMax) acc = acc - 0x80000001ll;
            const ColumnDev& col = seg->cols[(size_t)ag.column];
            const bool plane = plane_cols[(size_t)ag.column] != 0;
            if (ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG) set_integer_sum(&v, (__int128)acc * (__int128)sum_scale(col, plane) + (__int128)group_docs * (__int128)sum_base(col, plane));
            else if (ag.function == PG_AGG_MIN) v.min = agg_value_double(col, (int32_t)acc, plane);
            else v.max = agg_value_double(col, (int32_t)acc, plane);
          }
          ++k;
        }
        out->stats.num_docs_scanned = docs;
        finish_filter_stats(stats_plan, stats_scan_leaves, seg, 0, false, out);
        out->stats.num_entries_scanned_post_filter = docs * (int64_t)num_projected;
        out->stats.num_total_docs = seg->num_docs;
      };
      defer->item = std::move(item);
      defer->cacheable = !lw.plane_pending;
      defer->planes.reset(new PlaneHold(std::move(planes)));
      return kDeferred;
    }
    HIP_TRY(mark_pre_work(ctx));
    init_group_table_kernel<<<dim3((unsigned)std::max<long long>(64, std::min<long long>(product >> 12, (long long)seg->num_cus * 16))), dim3(256), 0, ctx->stream>>>(gp);
    HIP_TRY(hipGetLastError());
    if (timed) HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    // Key spaces above the LDS table: partition the docs by key range first, then aggregate every partition in LDS
    // (pg_group_partition.h) instead of one global atomic per doc and aggregation.
    const int partition_entry_bytes = 4 + 8 * gp.num_group_aggs;
    const int partition_shift = partition_entry_bytes <= 20 ? 12 : 11;
    const long long fine_partitions = (product + (1ll << partition_shift) - 1) >> partition_shift;
    // More fine partitions than one scatter pass can address (key spaces of 2 M .. 2^31 raw keys): two levels -- pass A scatters by
    // coarse partition (2^k fine ones each), group_repartition_*_kernel scatter every coarse partition's records by fine partition
    // (pg_group_partition.h).  PINOT_GPU_PARTITION_TWO_LEVEL=0: such key spaces keep the direct HBM atomics.
    const bool two_level_on = g_engine.partition_two_level;
    int log2_fine_per_coarse = 0;
    while (((fine_partitions + (1ll << log2_fine_per_coarse) - 1) >> log2_fine_per_coarse) > kMaxPartitions) ++log2_fine_per_coarse;
    const bool two_level = log2_fine_per_coarse > 0;
    const long long num_partitions = (fine_partitions + (1ll << log2_fine_per_coarse) - 1) >> log2_fine_per_coarse;       // what pass A scatters into
    const int scatter_shift = partition_shift + log2_fine_per_coarse;
    const bool use_partition = hash_plan.kind == 0 && !typed_direct && map_based && g_engine.group_partition && g_engine.group_private && private_leaves && gp.dense_ok && !want_bitmap &&
                               gp.num_group_aggs <= kMaxPartitionAggs && (!two_level || (two_level_on && (1 << log2_fine_per_coarse) <= kMaxFinePerCoarse)) &&
                               (long long)seg->num_docs >= g_engine.partition_min_docs;
    static const bool partition_trace = getenv("PINOT_GPU_PARTITION_TRACE") != nullptr;
    if (partition_trace)
      fprintf(stderr, "group-by plan: product %lld, fine partitions %lld (shift %d), 2^%d per coarse -> %lld scatter partitions; partition %d (hash %d typed_direct %d map_based %d "
                      "private_leaves %d dense_ok %d aggs %d docs %d)\n", product, fine_partitions, partition_shift, log2_fine_per_coarse, num_partitions, (int)use_partition,
              hash_plan.kind, (int)typed_direct, (int)map_based, (int)private_leaves, gp.dense_ok, gp.num_group_aggs, seg->num_docs);
    if (use_partition || !(use_private || typed_direct)) { st = complete_index_and_bitmap(&lw, ctx); if (st != PG_OK) return st; }
    if (lw.side != nullptr) {
      const bool wrote = ((use_private && !use_partition) || typed_direct) && gp.scan.tile_list == nullptr;
      for (int l = 0; l < kMaxLeaves; ++l) gp.scan.leaf_out[l] = wrote ? lw.sp_leaf_out[l] : nullptr;
      gp.scan.leaf_out_enabled = wrote ? 1 : 0;
      lw.side->kernel_wrote = wrote;
    }
    const bool count_leap2 = out && lw.stats_leap2_flagged && ((use_private && !use_partition) || typed_direct);
    const bool count_entries = (out && lw.stats_chain_flagged && ((use_private && !use_partition) || typed_direct)) || count_leap2;
    gp.scan.leap_tables = nullptr;
    if (count_leap2) { st = arm_leap_tables(seg, ctx, &gp.scan.leap_tables, &gp.scan.filter_entries); if (st != PG_OK) return st; }
    else if (lw.stats_leap2_flagged) for (int n = 0; n < gp.scan.num_nodes; ++n) gp.scan.nodes[n].flags &= ~kNodeLeapfrog2;
    if (count_entries && !count_leap2) { st = arm_filter_entries(ctx, &gp.scan.filter_entries); if (st != PG_OK) return st; }
    if (use_partition) {
      const int P = (int)num_partitions;
      const size_t N = (size_t)std::max(seg->num_tiles, 1) * 2048;
      const size_t max_work = N / (1u << 16) + (size_t)P + 1;
      auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
      const size_t off_upper = 0, off_cursor = align(off_upper + (size_t)P * 4), off_offsets = align(off_cursor + (size_t)P * 4);
      const size_t off_work = align(off_offsets + (size_t)(P + 1) * 4), off_key = align(off_work + max_work * sizeof(PartitionWork));
      // one packed dword per doc when the only aggregation input is an unsigned field that fits beside the slot (pg_group_partition.h)
      int packed_bits = 0;
      if (g_engine.partition_packed) {
        if (gp.num_group_aggs == 0) packed_bits = 1;
        else if (gp.num_group_aggs == 1) {
          const DevGroupAgg& ga = gp.group_aggs[0];
          const bool unsigned_field = !ga.is_raw && ga.vkind == kValI32 && (ga.kind != kGroupSum || ga.is_plane);
          if (unsigned_field && ga.bits + scatter_shift <= 32) packed_bits = ga.bits;
        }
      }
      const size_t off_val = align(off_key + N * 4), level1_bytes = off_val + (packed_bits > 0 ? 0 : (size_t)gp.num_group_aggs * align(N * 4));
      // two levels: the second record buffer(s), the fine partitions' count / offset / cursor arrays, pass B's device-built work list and its
      // length, and the re-scatter's own chunk list
      const int num_vals = packed_bits > 0 ? 0 : gp.num_group_aggs;
      const size_t fine_slots = (size_t)P << log2_fine_per_coarse;
      const size_t max_work2 = N / (1u << 16) + fine_slots + 1, max_chunks = N / kRepartitionChunk + (size_t)P + 1;
      const size_t off_key2 = align(level1_bytes), off_val2 = align(off_key2 + N * 4);
      const size_t off_fine = align(off_val2 + (size_t)num_vals * align(N * 4));                       // count | cursor | work_count (zeroed together), then offsets
      const size_t off_fine_offsets = align(off_fine + 2 * fine_slots * 4 + 64);
      const size_t off_work2 = align(off_fine_offsets + fine_slots * 4), off_chunks = align(off_work2 + max_work2 * sizeof(PartitionWork));
      const size_t bytes = two_level ? align(off_chunks + max_chunks * sizeof(PartitionWork)) : level1_bytes;
      if (ctx->partition_capacity < bytes) {
        if (ctx->d_partition) (void)hipFree(ctx->d_partition);
        ctx->d_partition = nullptr; ctx->partition_capacity = 0;
        if (hipMalloc((void**)&ctx->d_partition, bytes) != hipSuccess) return fail(PG_ERR_OUT_OF_MEMORY, "partitioned group-by needs %zu bytes of record buffers", bytes);
        ctx->partition_capacity = bytes;
      }
      PartitionParams pp;
      memset(&pp, 0, sizeof(pp));
      pp.gp = gp;
      pp.shift = scatter_shift;                     // (two levels: pass 0 and pass A work on coarse partitions)
      pp.num_partitions = P;
      pp.packed_bits = packed_bits;
      pp.upper = reinterpret_cast<uint32_t*>(ctx->d_partition + off_upper);
      pp.cursor = reinterpret_cast<uint32_t*>(ctx->d_partition + off_cursor);
      pp.offsets = reinterpret_cast<const uint32_t*>(ctx->d_partition + off_offsets);
      pp.work = reinterpret_cast<const PartitionWork*>(ctx->d_partition + off_work);
      pp.part_key = reinterpret_cast<uint32_t*>(ctx->d_partition + off_key);
      for (int a = 0; a < gp.num_group_aggs; ++a) pp.part_val[a] = reinterpret_cast<uint32_t*>(ctx->d_partition + off_val + (size_t)a * align(N * 4));
      HIP_TRY(hipMemsetAsync(ctx->d_partition, 0, off_offsets, ctx->stream));          // upper and cursor
      const long long tiles2k = ((long long)seg->num_docs + 2047) / 2048;
      const int hist_blocks = (int)std::max<long long>(1, std::min<long long>((tiles2k + 3) / 4, (long long)seg->num_cus * 6));
      std::vector<int> stats_key{scatter_shift};
      for (int g = 0; g < ng; ++g) stats_key.push_back(q->group_by_columns[g]);
      std::vector<uint32_t> upper;
      if (g_engine.partition_stats_cache) {
        std::lock_guard<std::mutex> lk(seg->partition_stats_mu);
        for (const auto& e : seg->partition_stats) if (e.first == stats_key) upper = e.second;
      }
      if ((int)upper.size() != P) {
        upper.assign((size_t)P, 0u);
        launch_group_partition_histogram(hist_blocks, ctx->stream, pp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(upper.data(), pp.upper, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (g_engine.partition_stats_cache) {
          std::lock_guard<std::mutex> lk(seg->partition_stats_mu);
          if (seg->partition_stats.size() < 64) seg->partition_stats.emplace_back(stats_key, upper);
        }
      }
      std::vector<uint32_t> offsets((size_t)P + 1, 0u);
      std::vector<PartitionWork> work;
      // chunk size: about two rounds of resident pass-B workgroups over the whole input, so that the flush of the touched slots
      // (one global atomic per slot, accumulator and chunk) stays small next to the records a chunk aggregates
      unsigned long long total_upper = 0;
      for (int p = 0; p < P; ++p) total_upper += upper[(size_t)p];
      const unsigned long long want_chunk = (total_upper + (unsigned long long)seg->num_cus * 6 - 1) / ((unsigned long long)seg->num_cus * 6);
      const uint32_t chunk = (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>((want_chunk + 1023) & ~1023ull, 1u << 16), (unsigned long long)kPartitionChunk);
      for (int p = 0; p < P; ++p) {
        offsets[(size_t)p + 1] = offsets[(size_t)p] + upper[(size_t)p];
        for (uint32_t s0 = 0; s0 < upper[(size_t)p]; s0 += chunk) work.push_back(PartitionWork{p, s0, std::min<uint32_t>(chunk, upper[(size_t)p] - s0), 0u});
      }
      HIP_TRY(hipMemcpyAsync(ctx->d_partition + off_offsets, offsets.data(), offsets.size() * 4, hipMemcpyHostToDevice, ctx->stream));
      if (!work.empty()) HIP_TRY(hipMemcpyAsync(ctx->d_partition + off_work, work.data(), work.size() * sizeof(PartitionWork), hipMemcpyHostToDevice, ctx->stream));
      const int scatter_bpc = packed_bits > 0 ? blocks_per_cu_group_partition_scatter_packed(P) : std::max(1, waves_group_partition_scatter() / 4);
      const int scatter_blocks = (int)std::max<long long>(1, std::min<long long>((tiles2k + 3) / 4, (long long)seg->num_cus * scatter_bpc));
      launch_group_partition_scatter(scatter_blocks, ctx->stream, pp);
      HIP_TRY(hipGetLastError());
      std::vector<PartitionWork> chunks;
      if (two_level && !work.empty()) {
        RepartitionParams rp;
        memset(&rp, 0, sizeof(rp));
        rp.src_key = pp.part_key;
        rp.dst_key = reinterpret_cast<uint32_t*>(ctx->d_partition + off_key2);
        for (int a = 0; a < num_vals; ++a) { rp.src_val[a] = pp.part_val[a]; rp.dst_val[a] = reinterpret_cast<uint32_t*>(ctx->d_partition + off_val2 + (size_t)a * align(N * 4)); }
        rp.coarse_offsets = pp.offsets; rp.coarse_cursor = pp.cursor;
        rp.fine_count = reinterpret_cast<uint32_t*>(ctx->d_partition + off_fine);
        rp.fine_cursor = rp.fine_count + fine_slots;
        rp.work_count = rp.fine_cursor + fine_slots;
        rp.fine_offsets = reinterpret_cast<uint32_t*>(ctx->d_partition + off_fine_offsets);
        rp.work = reinterpret_cast<PartitionWork*>(ctx->d_partition + off_work2);
        rp.chunks = reinterpret_cast<const PartitionWork*>(ctx->d_partition + off_chunks);
        rp.num_coarse = P; rp.log2_fine_per_coarse = log2_fine_per_coarse; rp.fine_shift = partition_shift; rp.packed_bits = packed_bits; rp.num_vals = num_vals;
        rp.aggregate_chunk = chunk;
        for (int p = 0; p < P; ++p)
          for (uint32_t s0 = 0; s0 < upper[(size_t)p]; s0 += kRepartitionChunk) chunks.push_back(PartitionWork{p, s0, std::min<uint32_t>(kRepartitionChunk, upper[(size_t)p] - s0), 0u});
        HIP_TRY(hipMemsetAsync(ctx->d_partition + off_fine, 0, 2 * fine_slots * 4 + 64, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->d_partition + off_chunks, chunks.data(), chunks.size() * sizeof(PartitionWork), hipMemcpyHostToDevice, ctx->stream));
        launch_group_repartition((int)chunks.size(), ctx->stream, rp);
        HIP_TRY(hipGetLastError());
        // pass B over the fine partitions: the second buffer, the device-built work list (as many workgroups as it can have entries)
        PartitionParams fine = pp;
        fine.shift = partition_shift;
        fine.num_partitions = (int32_t)fine_slots;
        fine.cursor = rp.fine_cursor; fine.offsets = rp.fine_offsets;
        fine.work = rp.work; fine.work_count = rp.work_count;
        fine.part_key = rp.dst_key;
        for (int a = 0; a < num_vals; ++a) fine.part_val[a] = rp.dst_val[a];
        launch_group_partition_aggregate((int)max_work2, ((size_t)partition_entry_bytes) << partition_shift, ctx->stream, fine);
      } else if (!work.empty()) {
        launch_group_partition_aggregate((int)work.size(), ((size_t)partition_entry_bytes) << partition_shift, ctx->stream, pp);
      }
      HIP_TRY(hipStreamSynchronize(ctx->stream));      // `offsets` / `work` / `chunks` are pageable host vectors: keep them alive until the copies ran
    }
    else if (typed_direct) {
      const long long tiles2k = ((long long)seg->num_docs + 2047) / 2048;
      launch_group_typed_direct((int)std::max<long long>(1, std::min<long long>((tiles2k + 3) / 4, (long long)seg->num_cus * 8)), ctx->stream, gp);
    }
    else if (use_private) launch_group_private(gp.use_lds_table != 0, pblocks, pthreads, plds, ctx->stream, gp);
    else launch_scan_group(g_engine.use_dma, gp.use_lds_table != 0, blocks, geo.threads, lds, ctx->stream, gp);
    HIP_TRY(hipGetLastError());
    if (timed) HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    if (count_leap2) { st = launch_leap_chain(seg, ctx, 0); if (st != PG_OK) return st; }
    if (count_entries) HIP_TRY(hipMemcpyAsync(ctx->h_filter_entries, ctx->d_filter_entries, 8, hipMemcpyDeviceToHost, ctx->stream));
    // The groups that exist, in ascending raw-key order: (raw key, doc count, accumulators[a * num_present + k]).
    std::vector<unsigned long long> hash_keys;                    // hashed holders: the keys of the present slots ...
    std::vector<std::vector<unsigned long long>> hash_keys_lvl;    // ... and, per chained first table, the key behind the slot number a later key starts with
    std::vector<int32_t> present_ids;
    std::vector<unsigned long long> present_counts;
    std::vector<long long> present_acc;
    const int32_t* ids_of = nullptr;                 // [num_present] where the conversion below reads: the vectors, or pinned staging
    const unsigned long long* counts_of = nullptr;
    const long long* acc_of = nullptr;
    int num_present = 0;
    long long docs = 0;
    if (!map_based) {
      HIP_TRY(hipMemcpyAsync(ctx->h_table, ctx->d_table, table_bytes, hipMemcpyDeviceToHost, ctx->stream));
      if (timed) HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      const unsigned long long* hc = ctx->h_table;
      const long long* ha = reinterpret_cast<const long long*>(ctx->h_table + gp.num_groups);
      for (int g = 0; g < gp.num_groups; ++g) if (hc[g]) { num_present++; docs += (long long)hc[g]; }
      present_ids.reserve((size_t)num_present); present_counts.reserve((size_t)num_present);
      present_acc.resize((size_t)num_present * (size_t)gp.num_group_aggs);
      int k = 0;
      for (int g = 0; g < gp.num_groups; ++g) {
        if (!hc[g]) continue;
        present_ids.push_back(g); present_counts.push_back(hc[g]);
        for (int a = 0; a < gp.num_group_aggs; ++a) present_acc[(size_t)a * (size_t)num_present + (size_t)k] = ha[(size_t)a * (size_t)gp.num_groups + (size_t)g];
        k++;
      }
    } else {
      // IntMapBasedHolder range: compact the HBM table on the device; honour numGroupsLimit the way the reference does
      // (_globalGroupIdUpperBound = min(product, numGroupsLimit), DictionaryBasedGroupKeyGenerator.java:176).
      const int limit = q->num_groups_limit > 0 ? q->num_groups_limit : 100000;
      const long long bound = hash_plan.kind != 0 ? (long long)limit : std::min<long long>(product, limit);      // LongMap / ArrayMap holders: _globalGroupIdUpperBound = numGroupsLimit (:150-163)
      const int num_chunks = (int)(((long long)gp.num_groups + kGroupChunk - 1) / kGroupChunk);
      DeviceScratch scratch(ctx);
      uint32_t* d_chunk_counts = (uint32_t*)scratch.alloc((size_t)num_chunks * 4);
      uint32_t* d_chunk_offsets = (uint32_t*)scratch.alloc((size_t)(num_chunks + 1) * 4);
      unsigned long long* d_total_docs = (unsigned long long*)scratch.alloc(8);
      if (!d_chunk_counts || !d_chunk_offsets || !d_total_docs) return fail(PG_ERR_OUT_OF_MEMORY, "group-by compaction scratch");
      uint32_t* d_first_doc = nullptr;
      uint32_t max_first_doc = 0xFFFFFFFFu;
      auto count_groups = [&](bool with_docs, uint32_t* total) -> pg_status {
        if (with_docs) HIP_TRY(hipMemsetAsync(d_total_docs, 0, 8, ctx->stream));
        group_chunk_count_kernel<<<dim3((unsigned)num_chunks), dim3(256), 0, ctx->stream>>>(gp.table_count, d_first_doc, max_first_doc, gp.num_groups, d_chunk_counts,
                                                                                             with_docs ? d_total_docs : nullptr);
        group_chunk_scan_kernel<<<dim3(1), dim3(1024), 0, ctx->stream>>>(d_chunk_counts, num_chunks, d_chunk_offsets);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(total, d_chunk_offsets + num_chunks, 4, hipMemcpyDeviceToHost, ctx->stream));
        return PG_OK;
      };
      uint32_t total = 0;
      unsigned long long total_docs = 0;
      st = count_groups(true, &total);
      if (st != PG_OK) return st;
      HIP_TRY(hipMemcpyAsync(&total_docs, d_total_docs, 8, hipMemcpyDeviceToHost, ctx->stream));
      if (timed) HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      docs = (long long)total_docs;
      out->num_groups_limit_reached = (long long)total >= (long long)limit ? 1 : 0;      // GroupByOperator.java:114-115
      if ((long long)total > bound) {
        // More groups than the reference would have created: it hands out group ids in order of first appearance (docId order)
        // and drops the docs of later keys, so the survivors are the `bound` groups whose first doc comes earliest.
        unsigned long long* d_filter = nullptr;
        if (q->num_filter_nodes > 0 && hash_plan.kind == 0) {
          d_filter = (unsigned long long*)scratch.alloc((((size_t)seg->num_docs + 63) / 64 + 1) * 8);
          if (!d_filter) return fail(PG_ERR_OUT_OF_MEMORY, "group-by first-doc pass: filter bitmap");
          pg_query fq = *q;
          fq.num_aggregations = 0; fq.num_group_by = 0;
          st = execute_impl(seg, &fq, nullptr, d_filter, nullptr, 0, nullptr);      // stays on the device
          if (st != PG_OK) return st;
        }
        d_first_doc = (uint32_t*)scratch.alloc((size_t)gp.num_groups * 4);
        if (!d_first_doc) return fail(PG_ERR_OUT_OF_MEMORY, "group-by first-doc pass: %d slots", gp.num_groups);
        HIP_TRY(hipMemsetAsync(d_first_doc, 0xFF, (size_t)gp.num_groups * 4, ctx->stream));
        // The survivors are the distinct keys of a PREFIX of the matching docs -- the reference stops admitting keys at the doc
        // where the bound-th one appears.  With far more groups than the limit that prefix is short, so the docs are visited in
        // growing prefixes until `bound` keys have a first doc, not all of them.
        uint32_t seen = 0;
        long long done_docs = 0;
        long long step_docs = std::max<long long>(1 << 16, 4 * bound);
        if (hash_plan.kind != 0) {
          // hashed holders: the group-by kernel once more, in its first-doc mode (same filter, same keys, every key already has its slot)
          GroupParams fg = gp;
          fg.first_doc = d_first_doc;
          fg.scan.filter_entries = nullptr; fg.scan.leap_tables = nullptr;
          for (int n = 0; n < fg.scan.num_nodes; ++n) fg.scan.nodes[n].flags &= ~(kNodeLeapfrog2 | kNodeCountEntries);
          // (plds: the set area, when the filter has dictId sets.  A query whose aggregation runs in group_typed_direct_kernel never sized a
          //  group_private_kernel launch: the pass takes that kernel's own geometry then)
          const long long fd_tiles = ((long long)seg->num_docs + 2047) / 2048;
          const int fd_threads = use_private ? pthreads : kBlockThreads;
          const int fd_blocks = use_private ? pblocks : (int)std::max<long long>(1, std::min<long long>((fd_tiles + 3) / 4, (long long)seg->num_cus * 8));
          if (!use_private) fg.set_lds_off = -1;
          launch_group_private(false, fd_blocks, fd_threads, use_private ? plds : 0, ctx->stream, fg);
          HIP_TRY(hipGetLastError());
          done_docs = seg->num_docs;
          max_first_doc = 0xFFFFFFFEu;
          st = count_groups(false, &seen);
          if (st != PG_OK) return st;
          HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        while (done_docs < (long long)seg->num_docs && (long long)seen < bound) {
          const long long hi = std::min<long long>((long long)seg->num_docs, done_docs + step_docs);
          const long long span = hi - done_docs;
          group_first_doc_kernel<<<dim3((unsigned)std::min<long long>((span + 255) / 256, (long long)seg->num_cus * 16)), dim3(256), 0, ctx->stream>>>(gp, d_filter, d_first_doc, done_docs, hi);
          HIP_TRY(hipGetLastError());
          done_docs = hi;
          step_docs *= 2;
          max_first_doc = 0xFFFFFFFEu;            // "has a first doc"
          st = count_groups(false, &seen);
          if (st != PG_OK) return st;
          HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        uint32_t* d_present_first = (uint32_t*)scratch.alloc((size_t)std::max<uint32_t>(seen, 1) * 4);
        if (!d_present_first) return fail(PG_ERR_OUT_OF_MEMORY, "group-by first-doc pass: %u groups", seen);
        group_compact_kernel<<<dim3((unsigned)num_chunks), dim3(256), 0, ctx->stream>>>(gp.table_count, gp.table_acc, 0, gp.num_groups, d_first_doc, 0xFFFFFFFEu, d_chunk_offsets, seen,
                                                                                         nullptr, nullptr, nullptr, d_present_first);
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> firsts((size_t)seen);
        HIP_TRY(hipMemcpyAsync(firsts.data(), d_present_first, (size_t)seen * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        std::nth_element(firsts.begin(), firsts.begin() + (bound - 1), firsts.end());
        max_first_doc = firsts[(size_t)bound - 1];      // first docs are distinct (a doc has one key): exactly `bound` groups pass
        st = count_groups(false, &total);
        if (st != PG_OK) return st;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
      }
      num_present = (int)total;
      if (num_present > 0) {
        const size_t ids_bytes = ((size_t)num_present * 4 + 255) & ~(size_t)255, counts_bytes = (size_t)num_present * 8;
        const size_t acc_bytes = (size_t)num_present * (size_t)gp.num_group_aggs * 8;
        st = ensure_host_groups(ctx, ids_bytes + counts_bytes + acc_bytes + 256);
        if (st != PG_OK) return st;
        int32_t* h_ids = reinterpret_cast<int32_t*>(ctx->h_groups);
        unsigned long long* h_counts = reinterpret_cast<unsigned long long*>(ctx->h_groups + ids_bytes);
        long long* h_acc = reinterpret_cast<long long*>(ctx->h_groups + ids_bytes + counts_bytes);
        ids_of = h_ids; counts_of = h_counts; acc_of = h_acc;
        int32_t* d_ids = (int32_t*)scratch.alloc((size_t)num_present * 4);
        unsigned long long* d_counts = (unsigned long long*)scratch.alloc((size_t)num_present * 8);
        long long* d_acc = (long long*)scratch.alloc(std::max<size_t>((size_t)num_present * (size_t)gp.num_group_aggs * 8, 8));
        if (!d_ids || !d_counts || !d_acc) return fail(PG_ERR_OUT_OF_MEMORY, "group-by result of %d groups", num_present);
        group_compact_kernel<<<dim3((unsigned)num_chunks), dim3(256), 0, ctx->stream>>>(gp.table_count, gp.table_acc, gp.num_group_aggs, gp.num_groups, d_first_doc, max_first_doc,
                                                                                         d_chunk_offsets, total, d_ids, d_counts, d_acc, nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_ids, d_ids, (size_t)num_present * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(h_counts, d_counts, counts_bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (gp.num_group_aggs > 0) HIP_TRY(hipMemcpyAsync(h_acc, d_acc, acc_bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (hash_plan.kind != 0) {
          // the slots that hold a group -> their 64-bit keys (and, for a key beyond a long, the first table's key behind its slot number)
          unsigned long long* d_keys = (unsigned long long*)scratch.alloc((size_t)num_present * 8 * (size_t)(1 + hash_plan.levels));
          if (!d_keys) return fail(PG_ERR_OUT_OF_MEMORY, "group-by keys of %d groups", num_present);
          hash_keys.resize((size_t)num_present);
          const unsigned gblocks = (unsigned)std::min<long long>(((long long)num_present + 255) / 256, (long long)seg->num_cus * 8);
          gather_u64_kernel<<<dim3(gblocks), dim3(256), 0, ctx->stream>>>(gp.hash_keys, d_ids, num_present, gp.hash_mask, d_keys);
          HIP_TRY(hipGetLastError());
          HIP_TRY(hipMemcpyAsync(hash_keys.data(), d_keys, (size_t)num_present * 8, hipMemcpyDeviceToHost, ctx->stream));
          // chained first tables, last one first: the low part of a key is the slot number of the table before it
          hash_keys_lvl.resize((size_t)hash_plan.levels);
          for (int l = hash_plan.levels - 1; l >= 0; --l) {
            unsigned long long* src = d_keys + (size_t)(hash_plan.levels - 1 - l) * (size_t)num_present;
            hash_keys_lvl[(size_t)l].resize((size_t)num_present);
            gather_u64_by_key_kernel<<<dim3(gblocks), dim3(256), 0, ctx->stream>>>(gp.hash_keys_lvl[l], src, num_present, gp.hash_mask_lvl[l], src + num_present);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(hash_keys_lvl[(size_t)l].data(), src + num_present, (size_t)num_present * 8, hipMemcpyDeviceToHost, ctx->stream));
          }
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
      }
    }
    if (!ids_of) { ids_of = present_ids.data(); counts_of = present_counts.data(); acc_of = present_acc.data(); }
    out->num_aggregations = na;
    out->dominant_kernel = use_partition ? PG_KERNEL_GROUP_PARTITION : (use_private ? PG_KERNEL_GROUP_PRIVATE : PG_KERNEL_SCAN_GROUP);
    out->num_groups = num_present;
    out->group_id_upper_bound = (hash_plan.kind != 0 || no_dict_keys) ? (q->num_groups_limit > 0 ? q->num_groups_limit : 100000) : gp.num_groups;      // hashed holders / no-dictionary generators: numGroupsLimit, like the reference (:150-163)
    out->group_ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)std::max(num_present, 1));
    out->group_aggregations = (pg_agg_value*)calloc((size_t)std::max(num_present, 1) * (size_t)std::max(na, 1), sizeof(pg_agg_value));
    // The keys as dictId tuples, for every kind of holder (what GroupKeyGenerator.getGroupKeys turns into values); rows in ascending
    // raw-key order.  Hashed holders come out of the table in slot order: `perm` sorts them.
    out->group_key_dict_ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)std::max(num_present, 1) * (size_t)ng);
    out->group_key_kind = hash_plan.kind;
    std::vector<int> perm;
    if (hash_plan.kind == 0) {
      for (int k = 0; k < num_present; ++k) {
        long long raw = ids_of[(size_t)k];
        for (int g = 0; g < ng; ++g) { out->group_key_dict_ids[(size_t)k * (size_t)ng + (size_t)g] = (int32_t)(raw % cards[(size_t)g]); raw /= cards[(size_t)g]; }
      }
    } else {
      std::vector<int32_t> tuples((size_t)num_present * (size_t)ng);
      for (int k = 0; k < num_present; ++k) {
        // key l covers the columns [segment_begin(l), segment_end(l)); above its lowest digit -- the slot number of table l - 1 -- it is
        // a mixed-radix number of those columns' dictIds
        for (int l = 0; l <= hash_plan.levels; ++l) {
          unsigned long long key = l == hash_plan.levels ? hash_keys[(size_t)k] : hash_keys_lvl[(size_t)l][(size_t)k];
          if (l > 0) key /= (unsigned long long)hash_plan.slots_lvl[l - 1];
          for (int g = hash_plan.segment_begin(l); g < hash_plan.segment_end(l, ng); ++g) {
            tuples[(size_t)k * (size_t)ng + (size_t)g] = (int32_t)(key % (unsigned long long)cards[(size_t)g]); key /= (unsigned long long)cards[(size_t)g];
          }
        }
      }
      perm.resize((size_t)num_present);
      for (int k = 0; k < num_present; ++k) perm[(size_t)k] = k;
      std::sort(perm.begin(), perm.end(), [&](int a, int b) {
        for (int g = ng - 1; g >= 0; --g) {      // the last column is the most significant digit of the raw key
          const int32_t x = tuples[(size_t)a * (size_t)ng + (size_t)g], y = tuples[(size_t)b * (size_t)ng + (size_t)g];
          if (x != y) return x < y;
        }
        return false;
      });
      if (hash_plan.kind == 1) out->group_ids64 = (int64_t*)malloc(sizeof(int64_t) * (size_t)std::max(num_present, 1));
      for (int k = 0; k < num_present; ++k) {
        const int src = perm[(size_t)k];
        memcpy(out->group_key_dict_ids + (size_t)k * (size_t)ng, tuples.data() + (size_t)src * (size_t)ng, sizeof(int32_t) * (size_t)ng);
        if (hash_plan.kind == 1) out->group_ids64[k] = (int64_t)hash_keys[(size_t)src];
      }
    }
    // Turning accumulators into the reference's holder values is independent per group: large results (the IntMapBasedHolder range
    // returns up to numGroupsLimit rows) are converted by a few host threads, each touching its own pages of the result.
    auto convert_groups = [&](int k_begin, int k_end) {
    for (int k = k_begin; k < k_end; ++k) {
      const int src = perm.empty() ? k : perm[(size_t)k];       // the row of the compacted table behind result row k
      const unsigned long long group_docs = counts_of[(size_t)src];
      out->group_ids[k] = perm.empty() ? ids_of[(size_t)src] : k;      // (hashed holders: a row number; the key is in group_ids64 / group_key_dict_ids)
      for (int a = 0; a < na; ++a) {
        const pg_aggregation& ag = q->aggregations[a];
        pg_agg_value& v = out->group_aggregations[(size_t)k * (size_t)na + (size_t)a];
        v.count = (int64_t)group_docs;
        v.min = std::numeric_limits<double>::infinity();
        v.max = -std::numeric_limits<double>::infinity();
        if (ag.function == PG_AGG_COUNT) continue;
        const long long acc = acc_of[(size_t)dev_agg_of[(size_t)a] * (size_t)num_present + (size_t)src];
        const ColumnDev& col = seg->cols[(size_t)ag.column];
        const bool plane = lw.plane_cols[(size_t)ag.column] != 0;
        if (ag.function == PG_AGG_SUM || ag.function == PG_AGG_AVG) {
          if (col.vkind == kValF64 || col.vkind == kValF32) {
            memcpy(&v.sum, &acc, 8);          // the slot accumulated doubles (ds_add_f64 / global_atomic_add_f64)
            v.sum_i64 = 0;
            v.sum_exact = 0;
          } else {
            set_integer_sum(&v, (__int128)acc * (__int128)sum_scale(col, plane) + (__int128)group_docs * (__int128)sum_base(col, plane));
          }
        }
        else if (col.encoding == PG_FWD_RAW_FIXED_BYTE && col.vkind != kValI32) {      // raw LONG value, or the order key of a raw FLOAT / DOUBLE value
          if (ag.function == PG_AGG_MIN) v.min = key64_to_double(col, acc); else v.max = key64_to_double(col, acc);
        }
        else if (ag.function == PG_AGG_MIN) v.min = agg_value_double(col, (int32_t)acc, plane);
        else v.max = agg_value_double(col, (int32_t)acc, plane);
      }
    }
    };
    const int convert_threads = num_present >= (1 << 16) ? (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (convert_threads <= 1) {
      convert_groups(0, num_present);
    } else {
      std::vector<std::thread> workers;
      for (int t = 0; t < convert_threads; ++t) {
        const int k0 = (int)((long long)num_present * t / convert_threads), k1 = (int)((long long)num_present * (t + 1) / convert_threads);
        workers.emplace_back(convert_groups, k0, k1);
      }
      for (auto& w : workers) w.join();
    }
    out->stats.num_docs_scanned = docs;
    finish_filter_stats(lw, seg, count_entries ? (int64_t)*ctx->h_filter_entries : 0, count_entries, out);
    out->stats.num_entries_scanned_post_filter = docs * (int64_t)projected.size();
    out->stats.num_total_docs = seg->num_docs;
  }
  if (timed && out && lw.gathered) {
    // index_and_kernel is the query: [ev[0], ev[3]] brackets the kernel and the copy of its counter lines
    float ms_all = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms_all, ctx->ev[0], ctx->ev[3]));
    out->device_ms = ms_all;
    out->dominant_kernel_ms = ms_all;
  } else if (timed && out) {
    float ms_all = 0.f, ms_scan = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms_scan, ctx->ev[1], ctx->ev[2]));
    if (!ctx->pre_started && ctx->ev_last == 2) ms_all = ms_scan;
    else HIP_TRY(hipEventElapsedTime(&ms_all, ctx->ev[ctx->pre_started ? 0 : 1], ctx->ev[ctx->ev_last]));
    out->device_ms = ms_all;
    out->dominant_kernel_ms = ms_scan;
    if (ctx->pre_started && lw.tile_list != nullptr) {
      // index-led query: when the index phase (index_and_kernel + its finalize) outlasts the scan of the listed tiles, IT is the dominant kernel
      float ms_index = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms_index, ctx->ev[0], ctx->ev[1]));
      if (ms_index > ms_scan) { out->dominant_kernel = PG_KERNEL_INDEX_AND; out->dominant_kernel_ms = ms_index; }
    }
  }
  if (out && lw.side != nullptr && !lw.side->fused && !out->filter_entries_exact) {
    // The transducer pass (pg_filter_fsm.h) behind the query's own kernels: they have retired (the result above was read), the leaves' bitmaps
    // they left are in this context's scratch, the context is still ours.  A pass that fails leaves the answer standing, inexact.
    if (device_fsm_filter_stats(seg, ctx, q, *lw.side, out) != PG_OK) { (void)hipGetLastError(); out->filter_entries_exact = 0; }
  }
  return PG_OK;
}

// ---- PG_QUERY_NULL_HANDLING (query option enableNullHandling=true) ----
// The reference gives every filter operator three docId sets -- trues, nulls, falses (BaseFilterOperator.java:85-113) -- and the
// aggregation functions skip the null docs of their own column (NullableSingleInputAggregationFunction.java:72-166).  Both are
// lowered HERE, on the host, into plain two-valued work for the kernels: the filter becomes an ordinary tree whose extra leaves
// read the columns' null bitmaps (IS_NULL postings expanded at open), and the aggregations are split into one lane per nullable
// column, each lane running the filter AND "column IS NOT NULL".  The kernels do not know about nulls.
struct FilterExpr {
  int op = PG_FILTER_LEAF;
  pg_predicate pred{};
  std::vector<FilterExpr> kids;
};

struct NullRewriter {
  const pg_segment* seg;
  bool has_nulls(int column) const { return column >= 0 && column < (int)seg->cols.size() && seg->cols[(size_t)column].d_null_bitmap != nullptr; }
  static bool column_leaf(const pg_predicate& p) {
    return p.kind == PG_PRED_DICT_RANGE || p.kind == PG_PRED_DICT_SET || p.kind == PG_PRED_RAW_RANGE || p.kind == PG_PRED_DOC_RANGE;
  }
  static FilterExpr leaf(const pg_predicate& p) { FilterExpr e; e.pred = p; return e; }
  static FilterExpr null_leaf(int column, bool is_not_null) {
    pg_predicate p{};
    p.kind = PG_PRED_IS_NULL; p.column = column; p.exclusive = is_not_null ? 1 : 0;
    return leaf(p);
  }
  static FilterExpr combine(int op, std::vector<FilterExpr> kids) {
    if (kids.size() == 1 && op != PG_FILTER_NOT) return std::move(kids[0]);
    FilterExpr e; e.op = op; e.kids = std::move(kids);
    return e;
  }
  static FilterExpr negate(FilterExpr e) { std::vector<FilterExpr> k; k.push_back(std::move(e)); return combine(PG_FILTER_NOT, std::move(k)); }

  // getTrues(): BaseColumnFilterOperator.java:45-53 (matches AND NOT nulls); And / Or: of the children's trues; Not: the child's falses
  FilterExpr trues(const FilterExpr& e) const {
    if (e.op == PG_FILTER_LEAF) {
      if (column_leaf(e.pred) && has_nulls(e.pred.column)) return combine(PG_FILTER_AND, {leaf(e.pred), null_leaf(e.pred.column, true)});
      return e;
    }
    if (e.op == PG_FILTER_NOT) return falses(e.kids[0]);
    std::vector<FilterExpr> k;
    for (const auto& c : e.kids) k.push_back(trues(c));
    return combine(e.op, std::move(k));
  }
  // trues OR nulls of a child, as And / OrFilterOperator.getFalses collect them (AndFilterOperator.java:62-80): only column leaves
  // have a null set (BaseColumnFilterOperator.getNulls :56-64); (matches AND NOT nulls) OR nulls == matches OR nulls
  FilterExpr trues_or_nulls(const FilterExpr& e) const {
    if (e.op == PG_FILTER_LEAF && column_leaf(e.pred) && has_nulls(e.pred.column))
      return combine(PG_FILTER_OR, {leaf(e.pred), null_leaf(e.pred.column, false)});
    return trues(e);
  }
  // getFalses(): leaf NOT (trues OR nulls) (BaseFilterOperator.java:96-113); And / Or: NOT AND / OR of (trues_i OR nulls_i); Not: child trues
  FilterExpr falses(const FilterExpr& e) const {
    if (e.op == PG_FILTER_LEAF) return negate(trues_or_nulls(e));
    if (e.op == PG_FILTER_NOT) return trues(e.kids[0]);
    std::vector<FilterExpr> k;
    for (const auto& c : e.kids) k.push_back(trues_or_nulls(c));
    return negate(combine(e.op, std::move(k)));
  }
};

struct FlatQuery {
  std::vector<pg_filter_node> nodes;
  std::vector<pg_predicate> preds;
  std::vector<pg_aggregation> aggs;
  pg_query q{};
  void emit(const FilterExpr& e) {
    pg_filter_node n{};
    n.op = e.op;
    if (e.op == PG_FILTER_LEAF) { n.predicate = (int32_t)preds.size(); preds.push_back(e.pred); }
    else { for (const auto& c : e.kids) emit(c); n.predicate = -1; n.num_children = (int32_t)e.kids.size(); }
    nodes.push_back(n);
  }
  void finish(const pg_query& base) {
    q = base;
    q.flags = base.flags & ~PG_QUERY_NULL_HANDLING;
    q.filter = nodes.empty() ? nullptr : nodes.data(); q.num_filter_nodes = (int32_t)nodes.size();
    q.predicates = preds.empty() ? nullptr : preds.data(); q.num_predicates = (int32_t)preds.size();
  }
};

static pg_status parse_filter(const pg_query* q, FilterExpr* root, bool* has_filter) {
  *has_filter = q->num_filter_nodes > 0;
  if (!*has_filter) return PG_OK;
  if (!q->filter || !q->predicates) return fail(PG_ERR_INVALID_ARGUMENT, "filter nodes without predicates");
  std::vector<FilterExpr> stack;
  for (int n = 0; n < q->num_filter_nodes; ++n) {
    const pg_filter_node& fn = q->filter[n];
    FilterExpr e;
    e.op = fn.op;
    if (fn.op == PG_FILTER_LEAF) {
      if (fn.predicate < 0 || fn.predicate >= q->num_predicates) return fail(PG_ERR_INVALID_ARGUMENT, "filter node %d: bad predicate index", n);
      e.pred = q->predicates[fn.predicate];
    } else if (fn.op == PG_FILTER_AND || fn.op == PG_FILTER_OR || fn.op == PG_FILTER_NOT) {
      const int k = fn.op == PG_FILTER_NOT ? 1 : fn.num_children;
      if (k < 1 || (int)stack.size() < k) return fail(PG_ERR_INVALID_ARGUMENT, "malformed filter tree (node %d)", n);
      e.kids.assign(std::make_move_iterator(stack.end() - k), std::make_move_iterator(stack.end()));
      stack.resize(stack.size() - (size_t)k);
    } else {
      return fail(PG_ERR_INVALID_ARGUMENT, "unknown filter op %d", fn.op);
    }
    stack.push_back(std::move(e));
  }
  if (stack.size() != 1) return fail(PG_ERR_INVALID_ARGUMENT, "malformed filter tree (%d roots)", (int)stack.size());
  *root = std::move(stack[0]);
  return PG_OK;
}

static pg_status execute_null_handling(pg_segment* seg, const pg_query* q, pg_result* out, uint64_t* host_bitmap, int64_t host_bitmap_words,
                                       int64_t* out_cardinality) {
  if (!seg || !q) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  NullRewriter rw{seg};
  FilterExpr root, filter_trues;
  bool has_filter = false;
  pg_status st = parse_filter(q, &root, &has_filter);
  if (st != PG_OK) return st;
  if (has_filter) filter_trues = rw.trues(root);
  FlatQuery base;
  if (has_filter) base.emit(filter_trues);
  base.finish(*q);
  if (host_bitmap) return execute_impl(seg, &base.q, nullptr, nullptr, host_bitmap, host_bitmap_words, out_cardinality);
  const int na = q->num_aggregations, ng = q->num_group_by;
  if (na < 0 || ng < 0 || (na > 0 && !q->aggregations) || (ng > 0 && !q->group_by_columns)) return fail(PG_ERR_INVALID_ARGUMENT, "bad aggregation / group-by lists");
  // lanes: the nullable columns that are aggregated, in first-use order (AggregationPlanNode.hasNullValues :130-152)
  std::vector<int> lane_cols;
  for (int a = 0; a < na; ++a) {
    const int c = q->aggregations[a].column;
    if (rw.has_nulls(c) && std::find(lane_cols.begin(), lane_cols.end(), c) == lane_cols.end()) lane_cols.push_back(c);
  }
  if (ng > 0) {
    // DefaultGroupByExecutor.java:106-121: under null handling the keys come from the no-dictionary generators -- NULL is a key value of
    // its own -- and every aggregation function skips the null docs of its own column (NullableSingleInputAggregationFunction
    // .aggregateGroupBySV :118-160).  Here: a nullable key column is read through its null-key image (dictId = cardinality where the doc
    // is null: one more digit value of the raw group id), and the aggregations are split into lanes like above -- the base lane over the
    // filter's trues (COUNT(*) and the columns without nulls; it also decides which groups exist), one lane per nullable aggregated
    // column over trues AND column IS NOT NULL, merged by group id.  A group none of whose docs has a value in a lane comes back with
    // count 0 for that function: the reference's holder stays null.
    bool nullable_keys = false;
    std::vector<int32_t> keys((size_t)ng);
    for (int g = 0; g < ng; ++g) {
      const int c = q->group_by_columns[g];
      keys[(size_t)g] = c;
      if (!rw.has_nulls(c)) continue;
      if (seg->cols[(size_t)c].nullkey_column < 0) return fail(PG_ERR_UNSUPPORTED, "GROUP BY over nullable raw column %s keeps the CPU plan", seg->cols[(size_t)c].name.c_str());
      keys[(size_t)g] = seg->cols[(size_t)c].nullkey_column;
      nullable_keys = true;
    }
    base.q.group_by_columns = keys.data();
    if (nullable_keys || !lane_cols.empty()) {
      // numGroupsLimit binds at any key-space size here (NoDictionary*GroupKeyGenerator: _globalGroupIdUpperBound = numGroupsLimit)
      long long product = 1;
      for (int g = 0; g < ng; ++g) product *= std::max(seg->cols[(size_t)keys[(size_t)g]].cardinality, 1);
      const long long limit = q->num_groups_limit > 0 ? q->num_groups_limit : 100000;
      if (limit < product) base.q.flags |= kQueryHashHolder;
    }
    if (lane_cols.empty()) return execute_impl(seg, &base.q, out, nullptr, nullptr, 0, nullptr);
    std::vector<int> base_pos;
    for (int a = 0; a < na; ++a) if (!rw.has_nulls(q->aggregations[a].column)) { base.aggs.push_back(q->aggregations[a]); base_pos.push_back(a); }
    const bool synthetic_count = base.aggs.empty();
    if (synthetic_count) base.aggs.push_back(pg_aggregation{PG_AGG_COUNT, -1});
    base.q.aggregations = base.aggs.data(); base.q.num_aggregations = (int32_t)base.aggs.size();
    pg_result part;
    memset(&part, 0, sizeof(part));
    st = execute_impl(seg, &base.q, &part, nullptr, nullptr, 0, nullptr, false);
    if (st != PG_OK) { pg_result_free(&part); return st; }
    memset(out, 0, sizeof(*out));
    const int num_groups = part.num_groups;
    const int base_na = part.num_aggregations;
    out->num_aggregations = na;
    out->num_groups = num_groups;
    out->group_id_upper_bound = part.group_id_upper_bound;
    out->num_groups_limit_reached = part.num_groups_limit_reached;
    out->stats = part.stats;
    out->filter_entries_exact = 0;
    out->device_ms = part.device_ms; out->dominant_kernel_ms = part.dominant_kernel_ms; out->dominant_kernel = part.dominant_kernel;
    out->group_ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)std::max(num_groups, 1));
    out->group_aggregations = (pg_agg_value*)calloc((size_t)std::max(num_groups, 1) * (size_t)std::max(na, 1), sizeof(pg_agg_value));
    for (int k = 0; k < num_groups; ++k) {
      out->group_ids[k] = part.group_ids[k];
      for (int a = 0; a < na; ++a) {                       // until a lane says otherwise: no value reached the holder
        pg_agg_value& v = out->group_aggregations[(size_t)k * (size_t)na + (size_t)a];
        v.min = std::numeric_limits<double>::infinity();
        v.max = -std::numeric_limits<double>::infinity();
      }
      if (!synthetic_count) for (size_t i = 0; i < base_pos.size(); ++i) out->group_aggregations[(size_t)k * (size_t)na + (size_t)base_pos[i]] = part.group_aggregations[(size_t)k * (size_t)base_na + i];
    }
    out->group_key_kind = part.group_key_kind;
    out->group_key_dict_ids = part.group_key_dict_ids;      // (digits of nullable keys run to cardinality inclusive: the last one is NULL)
    part.group_key_dict_ids = nullptr;
    pg_result_free(&part);
    // numEntriesScannedPostFilter = numDocsScanned * projected columns of the WHOLE query (GroupByOperator.java:160-166): keys + inputs
    std::vector<int> projected;
    for (int g = 0; g < ng; ++g) if (std::find(projected.begin(), projected.end(), q->group_by_columns[g]) == projected.end()) projected.push_back(q->group_by_columns[g]);
    for (int a = 0; a < na; ++a) {
      const pg_aggregation& ag = q->aggregations[a];
      const bool reads = ag.function != PG_AGG_COUNT || rw.has_nulls(ag.column);
      if (reads && ag.column >= 0 && std::find(projected.begin(), projected.end(), ag.column) == projected.end()) projected.push_back(ag.column);
    }
    out->stats.num_entries_scanned_post_filter = out->stats.num_docs_scanned * (int64_t)projected.size();
    for (int c : lane_cols) {
      FlatQuery lane;
      std::vector<FilterExpr> both;
      if (has_filter) both.push_back(filter_trues);
      both.push_back(NullRewriter::null_leaf(c, true));
      lane.emit(NullRewriter::combine(PG_FILTER_AND, std::move(both)));
      lane.finish(*q);
      lane.q.group_by_columns = keys.data();
      lane.q.num_groups_limit = kMaxGroupSlots;            // which groups exist is the base lane's decision: this lane reports all of its own
      std::vector<int> pos;
      for (int a = 0; a < na; ++a) if (q->aggregations[a].column == c) { lane.aggs.push_back(q->aggregations[a]); pos.push_back(a); }
      lane.q.aggregations = lane.aggs.data(); lane.q.num_aggregations = (int32_t)lane.aggs.size();
      memset(&part, 0, sizeof(part));
      st = execute_impl(seg, &lane.q, &part, nullptr, nullptr, 0, nullptr, false);
      if (st != PG_OK) { pg_result_free(&part); return st; }
      // both id lists ascend: one merge pass; lane groups the base lane did not admit (numGroupsLimit) are dropped
      int k = 0;
      for (int j = 0; j < part.num_groups; ++j) {
        while (k < num_groups && out->group_ids[k] < part.group_ids[j]) ++k;
        if (k == num_groups) break;
        if (out->group_ids[k] != part.group_ids[j]) continue;
        for (size_t i = 0; i < pos.size(); ++i) out->group_aggregations[(size_t)k * (size_t)na + (size_t)pos[i]] = part.group_aggregations[(size_t)j * (size_t)part.num_aggregations + i];
      }
      out->device_ms += part.device_ms;
      if (part.dominant_kernel_ms > out->dominant_kernel_ms) { out->dominant_kernel_ms = part.dominant_kernel_ms; out->dominant_kernel = part.dominant_kernel; }
      pg_result_free(&part);
    }
    return PG_OK;
  }
  if (lane_cols.empty()) return execute_impl(seg, &base.q, out, nullptr, nullptr, 0, nullptr);

  // Base lane: COUNT(*) and the aggregations of columns without nulls, over the filter itself.  It also yields numDocsScanned; the
  // metadata-only plan is off because the reference does not take it when any aggregated column has nulls (AggregationPlanNode.java:99-100).
  std::vector<int> base_pos;
  for (int a = 0; a < na; ++a) if (!rw.has_nulls(q->aggregations[a].column)) { base.aggs.push_back(q->aggregations[a]); base_pos.push_back(a); }
  const bool synthetic_count = base.aggs.empty();
  if (synthetic_count) base.aggs.push_back(pg_aggregation{PG_AGG_COUNT, -1});
  base.q.aggregations = base.aggs.data(); base.q.num_aggregations = (int32_t)base.aggs.size();
  pg_result part;
  memset(&part, 0, sizeof(part));
  st = execute_impl(seg, &base.q, &part, nullptr, nullptr, 0, nullptr, false);
  if (st != PG_OK) { pg_result_free(&part); return st; }
  memset(out, 0, sizeof(*out));
  out->num_aggregations = na;
  out->aggregations = (pg_agg_value*)calloc((size_t)std::max(na, 1), sizeof(pg_agg_value));
  out->stats = part.stats;
  out->device_ms = part.device_ms; out->dominant_kernel_ms = part.dominant_kernel_ms; out->dominant_kernel = part.dominant_kernel;
  if (!synthetic_count) for (size_t i = 0; i < base_pos.size(); ++i) out->aggregations[base_pos[i]] = part.aggregations[i];
  pg_result_free(&part);
  // numEntriesScannedPostFilter = numDocsScanned * projected columns of the WHOLE query (AggregationOperator.java:88-93)
  std::vector<int> projected;
  for (int a = 0; a < na; ++a) {
    const pg_aggregation& ag = q->aggregations[a];
    const bool reads = ag.function != PG_AGG_COUNT || rw.has_nulls(ag.column);   // COUNT(col) keeps its input expression under null handling
    if (reads && ag.column >= 0 && std::find(projected.begin(), projected.end(), ag.column) == projected.end()) projected.push_back(ag.column);
  }
  out->stats.num_entries_scanned_post_filter = out->stats.num_docs_scanned * (int64_t)projected.size();

  for (int c : lane_cols) {
    FlatQuery lane;
    std::vector<FilterExpr> both;
    if (has_filter) both.push_back(filter_trues);
    both.push_back(NullRewriter::null_leaf(c, true));
    lane.emit(NullRewriter::combine(PG_FILTER_AND, std::move(both)));
    lane.finish(*q);
    std::vector<int> pos;
    for (int a = 0; a < na; ++a) if (q->aggregations[a].column == c) { lane.aggs.push_back(q->aggregations[a]); pos.push_back(a); }
    lane.q.aggregations = lane.aggs.data(); lane.q.num_aggregations = (int32_t)lane.aggs.size();
    memset(&part, 0, sizeof(part));
    st = execute_impl(seg, &lane.q, &part, nullptr, nullptr, 0, nullptr, false);
    if (st != PG_OK) { pg_result_free(&part); return st; }
    for (size_t i = 0; i < pos.size(); ++i) out->aggregations[pos[i]] = part.aggregations[i];
    out->device_ms += part.device_ms;
    if (part.dominant_kernel_ms > out->dominant_kernel_ms) { out->dominant_kernel_ms = part.dominant_kernel_ms; out->dominant_kernel = part.dominant_kernel; }
    pg_result_free(&part);
  }
  return PG_OK;
}

// numEntriesScannedInFilter of a filter that leap-frogs (pg_filter_stats.h, Plan::kReplay): every leaf's docId set comes off the device
// as a bitmap, the reference's iterator tree is replayed over them.  Sequential in the docId space, hence the segment-size limit.
static pg_status replay_filter_stats(pg_segment* seg, const pg_query* q, pg_result* out) {
  const size_t words = ((size_t)seg->num_docs + 63) / 64;
  std::vector<fstats::Words> leaf_words((size_t)q->num_predicates);
  for (int i = 0; i < q->num_filter_nodes; ++i) {
    const pg_filter_node& n = q->filter[i];
    if (n.op != PG_FILTER_LEAF || leaf_words[(size_t)n.predicate]) continue;
    const fstats::LeafClass c = fstats::classify(q->predicates[n.predicate]);
    if (c == fstats::LeafClass::kMatchAll || c == fstats::LeafClass::kEmpty) continue;
    auto w = std::make_shared<std::vector<uint64_t>>(std::max<size_t>(words, 1), 0ull);
    pg_filter_node leaf = n;
    pg_query lq;
    memset(&lq, 0, sizeof(lq));
    lq.filter = &leaf; lq.num_filter_nodes = 1;
    lq.predicates = q->predicates; lq.num_predicates = q->num_predicates;
    pg_status st = execute_impl(seg, &lq, nullptr, nullptr, w->data(), (int64_t)w->size(), nullptr);
    if (st != PG_OK) return st;
    leaf_words[(size_t)n.predicate] = std::move(w);
  }
  out->stats.num_entries_scanned_in_filter = fstats::replay(q, seg->num_docs, leaf_words);
  out->filter_entries_exact = 1;
  return PG_OK;
}

// The same count ON THE DEVICE, at any segment size, for the root ANDs pg_filter_fsm.h can compile (scan leaves, index-based leaves, ORs
// of leaves -- `a AND b AND c`, `a AND (b OR c)`, the reference's golden filter): every leaf's docId set stays on the device as a
// doc-order bitmap, the transducer's tables are built tile by tile and chained (pg_fsm_kernels.h).  Nothing but the count comes back.
// Layout of the pass's scratch (ExecCtx.d_fsm_scratch): L bitmaps of whole tiles | delta | tile tables | chunk tables | the count --
// and, for a machine with a NOT child (Fsm::marks: the episodes of pg_fsm_kernels.h), from the next 256-byte boundary on:
// episode count + final-pending flags | marks of every episode stream | chunk states | tile states | the tiles' unpaired closes | the tiles' last opens.
// A second walk of the docs behind the tile pass: the episodes of NOT children (one pass per episode stream), and -- round 6c -- the COUNT of
// machines of 9 .. 16 states over at most four inputs without episodes (one pass, no marks): their tile pass builds functions only
// (fsm_tile_fns_kernel) instead of walking sixteen chains of table reads per doc (fsm_tiles_kernel<16, L>).
static inline bool needs_second_walk(const fstats::Fsm& fsm) { return fsm.has_episodes() || (fsm.num_states > 8 && fsm.num_inputs <= 4); }
static inline int second_walk_passes(const fstats::Fsm& fsm) { return fsm.has_episodes() ? fsm.num_episode_streams() : 1; }
struct FsmScratch {
  size_t bitmap_bytes = 0, delta_bytes = 0, tables_bytes = 0, chunk_bytes = 0, total = 0;
  size_t episode_base = 0, chunk_state_bytes = 0, tile_state_bytes = 0, tile_pos_bytes = 0;
  size_t front_bytes = 0;            // machines with episodes: fsm_tile_fns_kernel's lane fronts (4 / 8 / 16 bytes a lane for <= 4 / 8 / 16 states), behind the last opens
  long long tiles = 0, chunks = 0;
  FsmScratch(const pg_segment* seg, const fstats::Fsm& fsm) {
    tiles = std::max<long long>(1, ((long long)seg->num_docs + 2047) / 2048);
    chunks = (tiles + kFsmChunk - 1) / kFsmChunk;
    bitmap_bytes = (size_t)tiles * 256;
    delta_bytes = (((size_t)fsm.num_states << fsm.num_inputs) + 255) & ~(size_t)255;
    tables_bytes = (size_t)tiles * (size_t)fsm.num_states * 4;
    chunk_bytes = ((size_t)chunks * (size_t)fsm.num_states * 4 + 255) & ~(size_t)255;
    total = bitmap_bytes * (size_t)fsm.num_inputs + delta_bytes + tables_bytes + chunk_bytes + 256;
    if (needs_second_walk(fsm)) {
      episode_base = (total + 255) & ~(size_t)255;
      chunk_state_bytes = ((size_t)chunks + 255) & ~(size_t)255;
      tile_state_bytes = ((size_t)tiles + 255) & ~(size_t)255;
      tile_pos_bytes = ((size_t)tiles * 4 + 255) & ~(size_t)255;
      total = episode_base + 256 + delta_bytes * (size_t)second_walk_passes(fsm) + chunk_state_bytes + tile_state_bytes + 2 * tile_pos_bytes;
      front_bytes = (size_t)tiles * 64 * (fsm.num_states <= 4 ? 4 : (fsm.num_states <= 8 ? 8 : 16));
      total += front_bytes;
    }
  }
};
// The context's scratch, grown when needed, and where every input's bitmap goes.  (execute_impl, with the context it runs on.)
static pg_status prepare_fsm_side(pg_segment* seg, ExecCtx* ctx, const fstats::Fsm& fsm, FsmSide* side) {
  const FsmScratch lay(seg, fsm);
  // fsm_finish_kernel keeps one table per chunk in LDS (chunks * S * 4 bytes): 16 states near 2^31 docs pass 64 KB -- such a machine is not counted here
  if ((size_t)lay.chunks * (size_t)fsm.num_states * 4 > (60u << 10)) return fail(PG_ERR_UNSUPPORTED, "transducer pass: %lld chunks x %d states exceed the finish kernel's LDS", lay.chunks, fsm.num_states);
  if (ctx->fsm_scratch_bytes < lay.total) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->d_fsm_scratch) (void)hipFree(ctx->d_fsm_scratch);
    seg->device_bytes -= ctx->fsm_scratch_bytes;               // (the scratch is part of what pg_segment_device_bytes reports: the HBM budget of the caller sees it)
    ctx->d_fsm_scratch = nullptr; ctx->fsm_scratch_bytes = 0;
    HIP_TRY(hipMalloc((void**)&ctx->d_fsm_scratch, lay.total));
    ctx->fsm_scratch_bytes = lay.total;
    seg->device_bytes += lay.total;
  }
  if (!ctx->h_fsm_stage) HIP_TRY(hipHostMalloc((void**)&ctx->h_fsm_stage, kFsmStageBytes, hipHostMallocDefault));
  for (hipEvent_t& e : ctx->ev_pass) if (!e) HIP_TRY(hipEventCreate(&e));
  const fstats::Fsm* keep = side->fsm;
  *side = FsmSide();
  side->fsm = keep ? keep : &fsm;
  for (int i = 0; i < fsm.num_inputs; ++i) side->bitmap[i] = reinterpret_cast<uint32_t*>(ctx->d_fsm_scratch + lay.bitmap_bytes * (size_t)i);
  uint8_t* at = ctx->d_fsm_scratch + lay.bitmap_bytes * (size_t)fsm.num_inputs + lay.delta_bytes;
  side->tables = reinterpret_cast<uint32_t*>(at); at += lay.tables_bytes;
  side->chunks = reinterpret_cast<uint32_t*>(at); at += lay.chunk_bytes;
  side->entries = reinterpret_cast<unsigned long long*>(at);
  side->num_tiles = lay.tiles; side->num_chunks = lay.chunks;
  side->prepared = true;
  return PG_OK;
}

// Behind the query's own kernels, on the query's stream, with the context still held: kernels and copies are enqueued, ONE synchronisation
// at the end (rounds 4-5: the NULL stream, four synchronous copies, under a segment-wide mutex).  For timed runs the pass is bracketed by
// ctx->ev_pass and added to device_ms by the caller.
static pg_status device_fsm_filter_stats(pg_segment* seg, ExecCtx* ctx, const pg_query* q, const FsmSide& side, pg_result* out) {
  const fstats::Fsm& fsm = *side.fsm;
  hipStream_t stream = ctx->stream;
  const FsmScratch lay(seg, fsm);
  const long long tiles = lay.tiles, chunks = lay.chunks;
  const int L = fsm.num_inputs, S = fsm.num_states;
  static const bool trace = getenv("PINOT_GPU_FSM_TRACE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  const auto t_begin = now();
  uint8_t* d_base = ctx->d_fsm_scratch;
  if ((((size_t)fsm.num_states << fsm.num_inputs) * (size_t)(1 + second_walk_passes(fsm)) + 64) > kFsmStageBytes) return fail(PG_ERR_INTERNAL, "transducer tables exceed the pinned staging area");
  FsmParams fp;
  memset(&fp, 0, sizeof(fp));
  int scanned_again = 0;
  for (int i = 0; i < L; ++i) {
    fp.leaf[i] = side.bitmap[i];
    if (side.kernel_wrote && side.mapped[i]) continue;        // the query's own kernel left it behind
    // a leaf the kernel did not evaluate as a node of its own (the inverted-index children of the root AND are one merged leaf there),
    // or a kernel without the store: this leaf's docId set by a pass of its own
    ++scanned_again;
    HIP_TRY(hipMemset(reinterpret_cast<uint8_t*>(side.bitmap[i]) + (lay.bitmap_bytes - 256), 0, 256));      // the last tile's dwords past numDocs
    pg_filter_node leaf;
    memset(&leaf, 0, sizeof(leaf));
    leaf.op = PG_FILTER_LEAF; leaf.predicate = fsm.input_predicate[(size_t)i];
    pg_query lq;
    memset(&lq, 0, sizeof(lq));
    lq.filter = &leaf; lq.num_filter_nodes = 1;
    lq.predicates = q->predicates; lq.num_predicates = q->num_predicates;
    const pg_status st = execute_impl(seg, &lq, nullptr, reinterpret_cast<unsigned long long*>(side.bitmap[i]), nullptr, 0, nullptr);      // (returns with the copy done)
    if (st != PG_OK) return st;
  }
  const auto t_leaves = now();
  const bool timed_pass = (g_engine.flags & PG_CFG_TIME_KERNELS) != 0;
  if (timed_pass) HIP_TRY(hipEventRecord(ctx->ev_pass[0], stream));
  uint8_t* at = d_base + lay.bitmap_bytes * (size_t)L;
  uint8_t* d_delta = at; at += lay.delta_bytes;
  uint32_t* d_tables = reinterpret_cast<uint32_t*>(at); at += lay.tables_bytes;
  uint32_t* d_chunks = reinterpret_cast<uint32_t*>(at); at += lay.chunk_bytes;
  unsigned long long* d_entries = reinterpret_cast<unsigned long long*>(at);
  uint8_t* const h_delta = ctx->h_fsm_stage + 64;
  uint8_t* const h_marks = h_delta + ((size_t)S << L);
  memcpy(h_delta, fsm.delta.data(), (size_t)S << L);
  HIP_TRY(hipMemcpyAsync(d_delta, h_delta, (size_t)S << L, hipMemcpyHostToDevice, stream));
  fp.delta = d_delta; fp.tables = d_tables;
  fp.num_inputs = L; fp.num_states = S; fp.num_docs = seg->num_docs; fp.num_tiles = (int32_t)tiles;
  const unsigned blocks = (unsigned)std::max<long long>(1, std::min<long long>((tiles + 3) / 4, (long long)seg->num_cus * 8));
#define PG_FSM_LAUNCH(SM, LM) fsm_tiles_kernel<SM, LM><<<dim3(blocks), dim3(256), 0, stream>>>(fp)
// (a machine over two inputs has at most three states -- 350 000 random root ANDs over two predicates, tools/kernel_coverage.py's search:
//  the <8, 2>, <16, 2> table walks and the eight-state byte-function walk over two inputs were instantiations no query could reach;
//  the coverage gate of round 5 found them, they are gone: such a machine, should one ever exist, walks the three-input form)
#define PG_FSM_LAUNCH_L(SM) do { if (L <= 2 && SM <= 4) PG_FSM_LAUNCH((SM <= 4 ? SM : 4), 2); else if (L <= 3) PG_FSM_LAUNCH(SM, 3); else if (L <= 4) PG_FSM_LAUNCH(SM, 4); else if (L <= 6) PG_FSM_LAUNCH(SM, 6); else PG_FSM_LAUNCH(SM, 8); } while (0)
  const bool perm_walk = g_engine.fsm_perm;
  // (the byte-function walk keeps a lane's entries per entry state in ONE byte: 16 steps x two docs x at most 7 entries per doc.  A doc
  //  costs more than 7 only when the tree names the same predicate in many leaves -- such machines walk tables)
  int max_inc = 0;
  for (uint8_t d : fsm.delta) max_inc = std::max(max_inc, (int)(d >> 4));
  // (machines with episodes over at most four inputs: the tile pass builds the tiles' functions only and leaves every lane's front; the range
  //  kernel of the first episode stream counts the per-doc entries -- see fsm_tile_fns_kernel)
  // (count_pass: nine to sixteen states WITHOUT episodes -- the same two kernels, the range kernel as the counter: one pass with no marks)
  const bool count_pass = perm_walk && !fsm.has_episodes() && S > 8 && S <= 16 && L <= 4;
  const bool fns_pass = perm_walk && S <= 16 && L <= 4 && (fsm.has_episodes() || count_pass);
  if (fns_pass) {
    fp.lane_front = reinterpret_cast<uint32_t*>(d_base + lay.episode_base + 256 + lay.delta_bytes * (size_t)second_walk_passes(fsm) + lay.chunk_state_bytes + lay.tile_state_bytes + 2 * lay.tile_pos_bytes);
    if (S <= 4) { if (L <= 2) fsm_tile_fns_kernel<4, 2><<<dim3(blocks), dim3(256), 0, stream>>>(fp); else fsm_tile_fns_kernel<4, 4><<<dim3(blocks), dim3(256), 0, stream>>>(fp); }
    else if (S <= 8) { if (L <= 2) fsm_tile_fns_kernel<8, 2><<<dim3(blocks), dim3(256), 0, stream>>>(fp); else fsm_tile_fns_kernel<8, 4><<<dim3(blocks), dim3(256), 0, stream>>>(fp); }
    else if (L <= 3) fsm_tile_fns_kernel<16, 3><<<dim3(blocks), dim3(256), 0, stream>>>(fp);
    else fsm_tile_fns_kernel<16, 4><<<dim3(blocks), dim3(256), 0, stream>>>(fp);
  }
  else if (S <= 4 && L <= 4 && max_inc <= 7 && perm_walk) {
    if (L <= 2) fsm_tiles_perm_kernel<2><<<dim3(blocks), dim3(256), 0, stream>>>(fp);
    else if (L <= 3) fsm_tiles_perm_kernel<3><<<dim3(blocks), dim3(256), 0, stream>>>(fp);
    else fsm_tiles_perm_kernel<4><<<dim3(blocks), dim3(256), 0, stream>>>(fp);
  }
  else if (S <= 8 && L <= 4 && max_inc <= 7 && perm_walk) {
    if (L <= 3) fsm_tiles_perm8_kernel<3><<<dim3(blocks), dim3(256), 0, stream>>>(fp);
    else fsm_tiles_perm8_kernel<4><<<dim3(blocks), dim3(256), 0, stream>>>(fp);
  }
  else if (S <= 2) PG_FSM_LAUNCH_L(2);
  else if (S <= 4) PG_FSM_LAUNCH_L(4);
  else if (S <= 8) PG_FSM_LAUNCH_L(8);
  else PG_FSM_LAUNCH_L(16);
#undef PG_FSM_LAUNCH_L
#undef PG_FSM_LAUNCH
  HIP_TRY(hipGetLastError());
  if (trace) HIP_TRY(hipDeviceSynchronize());
  const auto t_tiles = now();
  fsm_chain_kernel<<<dim3((unsigned)chunks), dim3(1024), 0, stream>>>(d_tables, tiles, S, d_chunks);
  HIP_TRY(hipGetLastError());
  fsm_finish_kernel<<<dim3(1), dim3(1024), (size_t)chunks * (size_t)S * 4, stream>>>(d_chunks, (int)chunks, S, d_entries);
  HIP_TRY(hipGetLastError());
  unsigned long long entries = 0, episodes = 0;
  if (fsm.has_episodes() || count_pass) {
    // A NOT child over a scan leaf: what its leaf scans in 256-doc batches is charged per episode (pg_fsm_kernels.h "NOT children").  The
    // tables of the count, walked downwards, give every tile its entry state; a second walk of the docs pairs the opens and the closes.
    // (count_pass: no episodes -- one such walk with all-zero marks, for the per-doc entries alone)
    uint8_t* eb = d_base + lay.episode_base;
    unsigned long long* d_episodes = reinterpret_cast<unsigned long long*>(eb);
    const int streams = second_walk_passes(fsm);
    uint8_t* d_marks_base = eb + 256;
    uint8_t* d_chunk_state = d_marks_base + lay.delta_bytes * (size_t)streams;
    uint8_t* d_tile_state = d_chunk_state + lay.chunk_state_bytes;
    int32_t* d_first_close = reinterpret_cast<int32_t*>(d_tile_state + lay.tile_state_bytes);
    int32_t* d_last_open = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(d_first_close) + lay.tile_pos_bytes);
    HIP_TRY(hipMemsetAsync(eb, 0, 64, stream));
    fsm_chunk_states_kernel<<<dim3(1), dim3(1024), (size_t)chunks * (size_t)S * 4, stream>>>(d_chunks, (int)chunks, S, d_chunk_state);
    HIP_TRY(hipGetLastError());
    fsm_tile_states_kernel<<<dim3((unsigned)chunks), dim3(1024), 0, stream>>>(d_tables, tiles, S, d_chunk_state, d_tile_state);
    HIP_TRY(hipGetLastError());
    // every NOT child over a scan leaf has an episode stream of its own: the same machine and entry states, its own marks, its own
    // final-pending flag; the streams' entries add up in *d_episodes (the kernels add into it)
    for (int k = 0; k < streams; ++k) {
    uint8_t* const d_marks = d_marks_base + lay.delta_bytes * (size_t)k;
    uint8_t* const h_marks_k = h_marks + ((size_t)S << L) * (size_t)k;
    int32_t* const d_final_pending = reinterpret_cast<int32_t*>(eb + 8) + k;
    const uint32_t pending_states = count_pass ? 0u : fsm.stream_pending(k);
    if (count_pass) memset(h_marks_k, 0, (size_t)S << L);
    else memcpy(h_marks_k, fsm.stream_marks(k).data(), (size_t)S << L);
    HIP_TRY(hipMemcpyAsync(d_marks, h_marks_k, (size_t)S << L, hipMemcpyHostToDevice, stream));
    if (fns_pass) {
      // machines of at most sixteen states over at most four inputs: the tile pass (fsm_tile_fns_kernel) left every lane's front; one chain per
      // lane from the real entry state, a contiguous range of tiles per wavefront -- one record per RANGE for the finish kernel
      // (pg_fsm_kernels.h "Round 6", "Round 6c")
      const long long num_ranges = std::min<long long>(tiles, (long long)blocks * 4);
      FsmEpisodeRangeParams rp;
      memset(&rp, 0, sizeof(rp));
      for (int i = 0; i < L; ++i) rp.leaf[i] = fp.leaf[i];
      rp.delta = d_delta; rp.marks = d_marks; rp.tile_state = d_tile_state;
      rp.range_first_close = d_first_close; rp.range_last_open = d_last_open;
      rp.episode_entries = d_episodes; rp.final_pending = d_final_pending;
      rp.pending_states = pending_states;
      rp.num_inputs = L; rp.num_states = S; rp.num_docs = seg->num_docs; rp.num_tiles = (int32_t)tiles; rp.num_ranges = (int32_t)num_ranges;
      rp.count_entries = k == 0 ? 1 : 0;      // (the tile pass built functions only: the first stream's walk counts what the docs cost)
      rp.lane_front = fp.lane_front;
      const dim3 rgrid((unsigned)((num_ranges + 3) / 4));
      if (S > 8) fsm_episode_ranges_kernel<16, 4><<<rgrid, dim3(256), 0, stream>>>(rp);
      else if (S <= 4) { if (L <= 2) fsm_episode_ranges_kernel<4, 2><<<rgrid, dim3(256), 0, stream>>>(rp); else fsm_episode_ranges_kernel<4, 4><<<rgrid, dim3(256), 0, stream>>>(rp); }
      else { if (L <= 2) fsm_episode_ranges_kernel<8, 2><<<rgrid, dim3(256), 0, stream>>>(rp); else fsm_episode_ranges_kernel<8, 4><<<rgrid, dim3(256), 0, stream>>>(rp); }
      HIP_TRY(hipGetLastError());
      fsm_episode_finish_kernel<<<dim3(1), dim3(1024), 0, stream>>>(d_first_close, d_last_open, (int)num_ranges, seg->num_docs, d_final_pending, d_episodes);
      HIP_TRY(hipGetLastError());
    } else {
      FsmEpisodeParams ep;
      memset(&ep, 0, sizeof(ep));
      for (int i = 0; i < L; ++i) ep.leaf[i] = fp.leaf[i];
      ep.delta = d_delta; ep.marks = d_marks; ep.tile_state = d_tile_state;
      ep.tile_first_close = d_first_close; ep.tile_last_open = d_last_open;
      ep.episode_entries = d_episodes; ep.final_pending = d_final_pending;
      ep.pending_states = pending_states;
      ep.num_inputs = L; ep.num_states = S; ep.num_docs = seg->num_docs; ep.num_tiles = (int32_t)tiles;
      fsm_episode_tiles_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(ep);
      HIP_TRY(hipGetLastError());
      fsm_episode_finish_kernel<<<dim3(1), dim3(1024), 0, stream>>>(d_first_close, d_last_open, (int)tiles, seg->num_docs, d_final_pending, d_episodes);
      HIP_TRY(hipGetLastError());
    }
    }
    HIP_TRY(hipMemcpyAsync(ctx->h_fsm_stage + 8, d_episodes, 8, hipMemcpyDeviceToHost, stream));
  }
  HIP_TRY(hipMemcpyAsync(ctx->h_fsm_stage, d_entries, 8, hipMemcpyDeviceToHost, stream));
  if (timed_pass) HIP_TRY(hipEventRecord(ctx->ev_pass[1], stream));
  HIP_TRY(hipStreamSynchronize(stream));
  memcpy(&entries, ctx->h_fsm_stage, 8);
  if (fsm.has_episodes() || count_pass) memcpy(&episodes, ctx->h_fsm_stage + 8, 8);
  entries += episodes;
  if (timed_pass) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_pass[0], ctx->ev_pass[1]));
    out->device_ms += ms;
  }
  if (trace) fprintf(stderr, "fsm stats: %d inputs %d states %lld tiles: %d leaf bitmaps scanned again %.1f us, tables kernel %.1f us, chain+finish+copy %.1f us%s\n", L, S, tiles,
                     scanned_again, us(t_begin, t_leaves), us(t_leaves, t_tiles), us(t_tiles, now()), fsm.has_episodes() ? " (with the episodes of a NOT child)" : "");
  out->stats.num_entries_scanned_in_filter = (int64_t)entries;
  out->filter_entries_exact = 1;
  return PG_OK;
}

// pg_execute, or -- with `defer` -- its first half: kDeferred means the query was lowered but not launched (see Deferred).
static pg_status execute_one(pg_segment* segment, const pg_query* query, pg_result* out_result, Deferred* defer) {
  if (!out_result) return fail(PG_ERR_INVALID_ARGUMENT, "null result");
  memset(out_result, 0, sizeof(*out_result));     // before anything can fail: every error path below ends in pg_result_free(out_result)
  const bool null_handling = query && (query->flags & PG_QUERY_NULL_HANDLING);
  // A leap-frogging filter whose shape compiles into the transducer of pg_filter_fsm.h is counted on the device at any size
  // (PINOT_GPU_FSM_STATS=0: never): the query's own kernel leaves the leaves' bitmaps in the segment's scratch (one such query at a
  // time per segment), the pass follows.  Other shapes: the host's replay of the iterator tree up to
  // PINOT_GPU_EXACT_FILTER_STATS_DOCS docs, else the upper bound stands.
  const bool bound_ok = query && (query->flags & PG_QUERY_STATS_UPPER_BOUND_OK);      // the caller takes the upper bound: no pass, no replay
  const bool use_fsm = g_engine.fsm_stats && !bound_ok;
  fstats::Fsm fsm;
  FsmSide side;
  if (use_fsm && !null_handling && segment && query && query->num_filter_nodes >= 3 && query->filter && query->predicates) {
    int scan_leaves = 0;
    // (the machine only: execute_impl gives it the scratch of the context the query runs on and runs the pass behind the query's kernels, on
    //  their stream -- two such queries on one segment overlap like any others; rounds 4-5 held a segment-wide mutex across query and pass)
    if (fstats::choose_plan(query, &scan_leaves) == fstats::Plan::kReplay && fstats::compile_fsm(query, &fsm) && (g_engine.fsm_episodes || !fsm.has_episodes())) side.fsm = &fsm;
  }
  pg_status st = null_handling ? execute_null_handling(segment, query, out_result, nullptr, 0, nullptr)
                               : execute_impl(segment, query, out_result, nullptr, nullptr, 0, nullptr, true, defer, side.fsm != nullptr ? &side : nullptr);
  if (st == kDeferred) return st;
  // (enableNullHandling changes the iterator tree -- nulls are or-ed in, NOT takes the falses: the upper bound stands there)
  if (st == PG_OK && !null_handling && !bound_ok && !out_result->filter_entries_exact && (int64_t)segment->num_docs <= g_engine.exact_stats_docs)
    st = replay_filter_stats(segment, query, out_result);
  if (st != PG_OK) pg_result_free(out_result);
  return st;
}

// (pg_execute: below, behind the batch machinery -- a small group-by runs as a one-item launch of the batch's group-by kernel)

// ---- pg_execute_batch ----
namespace {

// Worker threads of the library: lowering (and, for queries that cannot share the batch launch, running) the items of a batch side by
// side.  Started on first use, parked on a condition variable in between.
struct WorkerPool {
  std::mutex mu;
  std::condition_variable wake;
  std::vector<std::thread> threads;
  std::function<void(int)> job;        // job(item)
  std::atomic<int> next{0}, remaining{0}, inside{0};
  int count = 0, grain = 1, helpers = 0;
  unsigned long long generation = 0;
  std::atomic<unsigned long long> generation_hint{0};      // == generation, readable without the mutex (the helpers' spin)
  bool stop = false;

  // Items are claimed `grain` at a time with ONE atomic add (a mutex hand-off per item cost more than lowering a query: 64 items of
  // ~2 us each took 80 us on 16 threads); whoever finishes the last item ends the run.
  void drain() {
    for (;;) {
      const int first = next.fetch_add(grain, std::memory_order_relaxed);
      if (first >= count) return;
      const int last = std::min(count, first + grain);
      for (int item = first; item < last; ++item) job(item);
      remaining.fetch_sub(last - first, std::memory_order_release);
    }
  }
  // A helper that has just worked stays on its core for a moment (being woken through the condition variable costs a run ~30 us of
  // latency, and a batch's items that miss the plan cache come in bursts); kSpinNs without a new run and the helper parks.  Round 4: 1 ms
  // -> 100 us -- with the plan cache a busy server's batches rarely need the helpers at all, and a millisecond of spinning per helper
  // after every miss was a core's worth of idle work per sixteen of them.
  static constexpr long long kSpinNs = 100'000;
  void worker(int index) {
    unsigned long long seen = 0;
    bool worked = false;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      if (worked && !stop && generation == seen) {
        lk.unlock();
        const auto until = std::chrono::steady_clock::now() + std::chrono::nanoseconds(kSpinNs);
        for (int i = 0; generation_hint.load(std::memory_order_relaxed) == seen; ++i) {
          cpu_relax();
          if ((i & 255) == 255 && std::chrono::steady_clock::now() >= until) break;
        }
        lk.lock();
      }
      worked = false;
      wake.wait(lk, [&] { return stop || generation != seen; });
      if (stop) return;
      seen = generation;
      if (index >= helpers) continue;
      worked = true;
      inside.fetch_add(1, std::memory_order_acquire);
      lk.unlock();
      drain();
      inside.fetch_sub(1, std::memory_order_release);
      lk.lock();
    }
  }
  // Runs job(0 .. n) on the pool and the calling thread; returns when all are done.  `items_per_claim`: how many items a thread takes
  // at a time (1 for items that wait for the device, more for items of a few microseconds).
  void run(int n, int max_threads, int items_per_claim, std::function<void(int)> fn) {
    if (n <= 0) return;
    {
      std::unique_lock<std::mutex> lk(mu);
      const int want = std::max(0, std::min(max_threads, (n + items_per_claim - 1) / items_per_claim) - 1);
      while ((int)threads.size() < want) { const int index = (int)threads.size(); threads.emplace_back([this, index] { worker(index); }); }
      // a helper woken late for the previous run may be inside drain() (it enters under this mutex and finds nothing left): let it leave
      while (inside.load(std::memory_order_acquire) != 0) cpu_relax();
      job = std::move(fn);
      count = n; grain = std::max(1, items_per_claim); helpers = want;
      next.store(0, std::memory_order_relaxed);
      remaining.store(n, std::memory_order_relaxed);
      ++generation;
      generation_hint.store(generation, std::memory_order_relaxed);
      if (want > 0) wake.notify_all();
    }
    drain();
    // the last items are in other threads' hands for microseconds (lowering) or for a kernel's duration (worker_threads mode): spin
    // briefly, then yield; no helper may still be inside drain() when the next run rewrites job / count
    long long spins = 0;
    while (remaining.load(std::memory_order_acquire) != 0 || inside.load(std::memory_order_acquire) != 0) {
      if (++spins > 2000) std::this_thread::yield();
    }
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    wake.notify_all();
    for (auto& t : threads) t.join();
  }
};
WorkerPool g_pool;
// The pool runs one job at a time.  A second pg_execute_batch does NOT wait for it (two queries of a server used to run one after the
// other, end to end): it lowers its items on its own thread -- microseconds each -- or, when its items run whole kernels inside their
// claim (group-bys, 1 B-row scans), on short-lived threads of its own.  The lock covers the lowering phase only; the shared launches
// and their waits run on per-call BatchCtx's outside it.
std::mutex g_pool_mu;
void run_items(int count, int threads, int items_per_claim, bool heavy, const std::function<void(int)>& fn) {
  {
    std::unique_lock<std::mutex> lk(g_pool_mu, std::try_to_lock);
    if (lk.owns_lock()) { g_pool.run(count, threads, items_per_claim, fn); return; }
  }
  const int extra = heavy ? std::min(threads, count) - 1 : 0;
  if (extra <= 0) { for (int i = 0; i < count; ++i) fn(i); return; }
  std::atomic<int> next{0};
  auto drain = [&] { for (;;) { const int i = next.fetch_add(1, std::memory_order_relaxed); if (i >= count) return; fn(i); } };
  std::vector<std::thread> own;
  for (int t = 0; t < extra; ++t) own.emplace_back(drain);
  drain();
  for (auto& t : own) t.join();
}

// Device-side state of one batch launch, reused from call to call.
struct BatchCtx {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};      // [0], [1]: around the kernel (timed runs); [2]: the group tables' copy has landed
  // pinned staging / device copy of what the launch reads: [first workgroup of every item | the items' kernel parameters], ONE
  // allocation each so that ONE copy command precedes the launch
  uint8_t* h_blob = nullptr; uint8_t* d_blob = nullptr;
  size_t items_offset = 0;
  ScanParams* h_items = nullptr; ScanParams* d_items = nullptr;
  uint32_t* h_first = nullptr; uint32_t* d_first = nullptr;
  HostRecord* h_records = nullptr; HostRecord* h_records_dev = nullptr;  // pinned, device-mapped: one folded record per item
  size_t sets_offset = 0, set_capacity = 0;    // the blob's last part: the items' dictId sets (IN lists), uploaded with the items
  uint32_t* d_done = nullptr;                                            // kFoldShards + 1 arrival counters per item
  BlockPartial* d_partials = nullptr;
  int item_capacity = 0;
  size_t partial_capacity = 0;
  unsigned long long seq = 0;
  // group-by launches (lean_kind 6): the items' table slices -- ONE device allocation, all-zero between launches, and its pinned host image
  unsigned long long* d_gtable = nullptr; unsigned long long* h_gtable = nullptr; unsigned long long* h_gtable_dev = nullptr;      // (the image is device-mapped: items publish their slices themselves)
  size_t gtable_capacity = 0;      // words
  bool gtable_dirty = false;       // a launch did not publish every item (or the copy form's memset was not enqueued): zero the whole table and the arrival counters before the next launch
};
constexpr size_t kBatchItemSlot = sizeof(GroupParams) > sizeof(ScanParams) ? sizeof(GroupParams) : sizeof(ScanParams);      // a slot of the blob holds an item of either kind
std::mutex g_batch_mu;
std::vector<BatchCtx*> g_batch_free;

void destroy_batch_ctx(BatchCtx* b) {
  if (!b) return;
  if (b->stream) (void)hipStreamSynchronize(b->stream);      // (the group-by tables are zeroed behind the answers)
  if (b->h_blob) (void)hipHostFree(b->h_blob);
  if (b->d_blob) (void)hipFree(b->d_blob);
  if (b->h_records) (void)hipHostFree(b->h_records);
  if (b->d_done) (void)hipFree(b->d_done);
  if (b->d_partials) (void)hipFree(b->d_partials);
  if (b->d_gtable) (void)hipFree(b->d_gtable);
  if (b->h_gtable) (void)hipHostFree(b->h_gtable);
  for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

pg_status ensure_batch_ctx(BatchCtx* b, int items, size_t partials, size_t set_bytes = 0 /* the items' dictId sets, behind the item slots */) {
  if (!b->stream) {
    HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    for (auto& e : b->ev) HIP_TRY(hipEventCreate(&e));
  }
  if (b->item_capacity < items || b->set_capacity < set_bytes) {
    const int cap = std::max({items, b->item_capacity, 64});
    const size_t set_cap = set_bytes > b->set_capacity ? std::max<size_t>(set_bytes * 2, 64 * 1024) : b->set_capacity;
    if (b->stream) HIP_TRY(hipStreamSynchronize(b->stream));      // (nothing of the last launch may still read the blob)
    if (b->h_blob) (void)hipHostFree(b->h_blob);
    if (b->d_blob) (void)hipFree(b->d_blob);
    if (b->h_records) (void)hipHostFree(b->h_records);
    if (b->d_done) (void)hipFree(b->d_done);
    b->h_blob = nullptr; b->d_blob = nullptr; b->h_items = nullptr; b->d_items = nullptr; b->h_first = nullptr; b->d_first = nullptr; b->h_records = nullptr; b->d_done = nullptr;
    b->item_capacity = 0;
    b->items_offset = (4 * (size_t)(cap + 1) + 255) & ~(size_t)255;
    b->sets_offset = b->items_offset + kBatchItemSlot * (size_t)cap;
    const size_t blob_bytes = b->sets_offset + set_cap;
    b->set_capacity = set_cap;
    HIP_TRY(hipHostMalloc((void**)&b->h_blob, blob_bytes, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&b->d_blob, blob_bytes));
    memset(b->h_blob, 0, blob_bytes);
    b->h_first = reinterpret_cast<uint32_t*>(b->h_blob); b->d_first = reinterpret_cast<uint32_t*>(b->d_blob);
    b->h_items = reinterpret_cast<ScanParams*>(b->h_blob + b->items_offset); b->d_items = reinterpret_cast<ScanParams*>(b->d_blob + b->items_offset);
    HIP_TRY(hipHostMalloc((void**)&b->h_records, sizeof(HostRecord) * (size_t)cap, hipHostMallocMapped));
    memset(b->h_records, 0, sizeof(HostRecord) * (size_t)cap);
    HIP_TRY(hipHostGetDevicePointer((void**)&b->h_records_dev, b->h_records, 0));
    HIP_TRY(hipMalloc((void**)&b->d_done, (size_t)cap * (kFoldShards + 1) * kFoldStride * 4));
    HIP_TRY(hipMemsetAsync(b->d_done, 0, (size_t)cap * (kFoldShards + 1) * kFoldStride * 4, b->stream));      // ordered before the batch's launches
    b->item_capacity = cap;
  }
  if (b->partial_capacity < partials) {
    if (b->d_partials) (void)hipFree(b->d_partials);
    b->d_partials = nullptr; b->partial_capacity = 0;
    HIP_TRY(hipMalloc((void**)&b->d_partials, sizeof(BlockPartial) * partials));
    b->partial_capacity = partials;
  }
  return PG_OK;
}

// The deferred items of one device: one launch, every item folding into its own pinned record.  Two halves, so that a batch spanning
// several devices has every device's launch in flight before it waits for any (segment s on device s mod N: BaseCombineOperator.java:85-142
// runs all segments of a query on one pool): enqueue_deferred copies the items and launches, finish_deferred waits and converts.
// The plan cache of a segment (pg_segment.plan_cache): at most kPlanCacheEntries items, most recently used first.
constexpr size_t kPlanCacheEntries = 4;

// A hit hands out the item with its value planes held again; an item lowered under other engine settings, or before a plane of the
// segment came or went, is dropped.
bool cached_item(pg_segment* seg, const std::string& key, Deferred* out) {
  std::shared_ptr<const LoweredItem> item;
  {
    std::lock_guard<std::mutex> lk(seg->plan_cache_mu);
    auto& cache = seg->plan_cache;
    for (size_t i = 0; i < cache.size(); ++i) {
      if (cache[i]->key != key) continue;
      item = cache[i];
      if (i != 0) std::rotate(cache.begin(), cache.begin() + (long)i, cache.begin() + (long)i + 1);
      break;
    }
  }
  if (!item) return false;
  bool fresh = item->engine_epoch == g_engine.epoch.load(std::memory_order_acquire);
  std::vector<int> held;
  for (size_t c = 0; c < item->plane_columns.size() && fresh; ++c) {
    bool ready = false;
    if (acquire_plane(seg, item->plane_columns[c], &ready) != PG_OK || !ready) { fresh = false; break; }
    held.push_back(item->plane_columns[c]);
  }
  // (the planes are held now: the epoch cannot move under this batch for the columns the item reads)
  fresh = fresh && item->plane_epoch == seg->plane_epoch.load(std::memory_order_acquire);
  std::unique_ptr<PlaneHold> hold(new PlaneHold(seg, std::move(held)));
  if (!fresh) {
    std::lock_guard<std::mutex> lk(seg->plan_cache_mu);
    auto& cache = seg->plan_cache;
    cache.erase(std::remove(cache.begin(), cache.end(), item), cache.end());
    return false;                                   // (`hold` releases what was acquired)
  }
  out->item = std::move(item);
  out->planes = std::move(hold);
  out->cacheable = true;
  return true;
}

void remember_item(pg_segment* seg, std::string&& key, const std::shared_ptr<const OwnedQuery>& query, Deferred* d) {
  // (the item was made by this thread a moment ago and nobody else has seen it yet)
  LoweredItem* item = std::const_pointer_cast<LoweredItem>(d->item).get();
  item->query = query;
  item->key = std::move(key);
  item->engine_epoch = g_engine.epoch.load(std::memory_order_acquire);
  item->plane_epoch = seg->plane_epoch.load(std::memory_order_acquire);
  std::lock_guard<std::mutex> lk(seg->plan_cache_mu);
  auto& cache = seg->plan_cache;
  for (size_t i = 0; i < cache.size(); ++i) if (cache[i]->key == item->key) { cache.erase(cache.begin() + (long)i); break; }
  cache.insert(cache.begin(), d->item);
  if (cache.size() > kPlanCacheEntries) cache.pop_back();
}

struct DeferredLaunch {
  BatchCtx* b = nullptr;
  int device = -1, n = 0, lean_kind = 0;       // lean_kind: ScanParams.lean_kind of every item (0: scan_private_batch_kernel, 1 / 2: scan_lean_batch_kernel)
  std::vector<int> items, blocks;
  std::vector<size_t> table_offsets;       // lean_kind 6: where every item's slice starts in the context's table (words)
  long long total_blocks = 0, docs = 0;
  size_t lds = 0;
  unsigned long long seq = 0;
  bool timed = false, launched = false, publish = false;      // publish: the group-by items write their slices and sequence numbers themselves
  std::chrono::steady_clock::time_point t0, t1;
  ~DeferredLaunch() { if (b) { std::lock_guard<std::mutex> lk(g_batch_mu); g_batch_free.push_back(b); } }
};

// The group-by items of one device (lean_kind 6): one launch of group_lds_batch_kernel over the items' GroupParams, each with a slice of the
// context's table; then ONE copy of all slices to the pinned host image and a memset that leaves the table all-zero for the next launch.
// The dictId sets (IN lists) of a deferred item ride in the batch's blob: copied behind the item slots, the item's leaves pointed at the copy
// (they were lowered against the words of a context the item is no longer tied to).  *set_off: bytes of the set area used so far.
static size_t item_set_bytes(const LoweredItem& d) {
  size_t bytes = 0;
  for (const auto& sl : d.sets) bytes += ((size_t)sl.bytes + 15) & ~(size_t)15;
  return bytes;
}
static void place_item_sets(BatchCtx* b, const LoweredItem& d, ScanParams* sp, size_t* set_off) {
  for (const auto& sl : d.sets) {
    memcpy(b->h_blob + b->sets_offset + *set_off, sl.host_words, sl.bytes);
    const uint32_t* d_words = reinterpret_cast<const uint32_t*>(b->d_blob + b->sets_offset + *set_off);
    for (int nd = 0; nd < sp->num_nodes; ++nd)
      if (sp->nodes[nd].op == PG_FILTER_LEAF && sp->nodes[nd].kind == kLeafDictSet && sp->nodes[nd].set_words == sl.ctx_words) sp->nodes[nd].set_words = d_words;
    *set_off += ((size_t)sl.bytes + 15) & ~(size_t)15;
  }
}
pg_status enqueue_group_launch(DeferredLaunch* L, BatchCtx* b, std::vector<Deferred>& defs, pg_segment* const* segments) {
  const std::vector<int>& items = L->items;
  const int n = L->n;
  long long total_tiles = 0;
  size_t launch_lds = 0, table_words = 0;
  int threads = 64;
  for (int i : items) {
    const LoweredItem& d = *defs[(size_t)i].item;
    total_tiles += ((long long)segments[i]->num_docs + 2047) / 2048;
    L->docs += (long long)segments[i]->num_docs;
    launch_lds = std::max(launch_lds, d.group_lds);
    threads = std::max(threads, d.group_threads);
    table_words += d.group_table_words;
  }
  L->lds = launch_lds;
  const int wpb = threads / 64;
  int bpc = std::max(1, std::min(waves_group_lds_batch() / wpb, (int)(kLdsBudget / std::max<size_t>(launch_lds, 1))));
  if (g_engine.blocks_per_cu > 0) bpc = g_engine.blocks_per_cu;
  const long long budget = (long long)segments[items[0]]->num_cus * bpc;
  std::vector<int>& blocks = L->blocks;
  blocks.assign((size_t)n, 0);
  long long total_blocks = 0;
  for (int k = 0; k < n; ++k) {
    const LoweredItem& d = *defs[(size_t)items[(size_t)k]].item;
    const long long tiles = ((long long)segments[items[(size_t)k]]->num_docs + 2047) / 2048;
    const long long share = total_tiles > 0 ? (tiles * budget + total_tiles - 1) / total_tiles : 1;
    blocks[(size_t)k] = (int)std::max<long long>(1, std::min<long long>({(long long)d.blocks, share, (tiles + wpb - 1) / wpb}));
    total_blocks += blocks[(size_t)k];
  }
  L->total_blocks = total_blocks;
  size_t set_bytes = 0, set_off = 0;
  for (int i : items) set_bytes += item_set_bytes(*defs[(size_t)i].item);
  pg_status st = ensure_batch_ctx(b, n, 0, set_bytes);
  if (st != PG_OK) return st;
  if (b->gtable_capacity < table_words) {
    if (b->d_gtable) (void)hipFree(b->d_gtable);
    if (b->h_gtable) (void)hipHostFree(b->h_gtable);
    b->d_gtable = nullptr; b->h_gtable = nullptr; b->gtable_capacity = 0;
    const size_t cap = std::max(table_words, (size_t)1 << 18);
    HIP_TRY(hipMalloc((void**)&b->d_gtable, cap * 8));
    HIP_TRY(hipHostMalloc((void**)&b->h_gtable, cap * 8, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&b->h_gtable_dev, b->h_gtable, 0));
    HIP_TRY(hipMemsetAsync(b->d_gtable, 0, cap * 8, b->stream));      // ordered before the launch below
    b->gtable_capacity = cap;
    b->gtable_dirty = false;
  }
  if (b->gtable_dirty) {
    HIP_TRY(hipMemsetAsync(b->d_gtable, 0, b->gtable_capacity * 8, b->stream));
    HIP_TRY(hipMemsetAsync(b->d_done, 0, (size_t)b->item_capacity * (kFoldShards + 1) * kFoldStride * 4, b->stream));
    b->gtable_dirty = false;
  }
  // Items publish themselves (GroupParams.host_table; PINOT_GPU_GROUP_PUBLISH=0: one copy of all slices and a memset behind the launch,
  // rounds 4-6a): the last workgroup of an item writes its slice to the pinned image and its sequence number to the item's record.
  const bool publish = g_engine.group_publish;
  L->publish = publish;
  GroupParams* h_items = reinterpret_cast<GroupParams*>(b->h_blob + b->items_offset);
  GroupParams* d_items = reinterpret_cast<GroupParams*>(b->d_blob + b->items_offset);
  size_t off = 0;
  uint32_t first = 0;
  L->seq = ++b->seq;
  L->table_offsets.assign((size_t)n, 0);
  for (int k = 0; k < n; ++k) {
    const LoweredItem& d = *defs[(size_t)items[(size_t)k]].item;
    GroupParams& gp = h_items[k];                 // (the pinned copy the device reads: filled in place)
    gp = *d.gp;
    place_item_sets(b, d, &gp.scan, &set_off);
    gp.table_count = b->d_gtable + off;
    gp.table_acc = reinterpret_cast<long long*>(b->d_gtable + off + (size_t)gp.num_groups);
    if (publish) {
      gp.host_table = b->h_gtable_dev + off;
      gp.scan.done_counter = b->d_done + (size_t)k * (kFoldShards + 1) * kFoldStride;
      gp.scan.host_out = b->h_records_dev + k;
      gp.scan.host_seq = L->seq;
    }
    L->table_offsets[(size_t)k] = off;
    off += d.group_table_words;
    b->h_first[k] = first;
    first += (uint32_t)blocks[(size_t)k];
  }
  b->h_first[n] = first;
  L->timed = (g_engine.flags & PG_CFG_TIME_KERNELS) != 0;
  L->t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipMemcpyAsync(b->d_blob, b->h_blob, b->items_offset + sizeof(GroupParams) * (size_t)n, hipMemcpyHostToDevice, b->stream));
  if (set_off) HIP_TRY(hipMemcpyAsync(b->d_blob + b->sets_offset, b->h_blob + b->sets_offset, set_off, hipMemcpyHostToDevice, b->stream));
  if (L->timed) HIP_TRY(hipEventRecord(b->ev[0], b->stream));
  b->gtable_dirty = true;
  launch_group_lds_batch((int)total_blocks, threads, launch_lds, b->stream, d_items, b->d_first, n);
  HIP_TRY(hipGetLastError());
  if (L->timed) HIP_TRY(hipEventRecord(b->ev[1], b->stream));
  if (publish) b->gtable_dirty = false;          // (the launch leaves table and counters as it found them; finish_group_launch says otherwise when an item stays silent)
  if (publish) (void)hipStreamQuery(b->stream);      // (nothing else of this call touches the stream before the items' numbers arrive: hand the commands to the device now)
  if (!publish) {
    HIP_TRY(hipMemcpyAsync(b->h_gtable, b->d_gtable, off * 8, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipEventRecord(b->ev[2], b->stream));
    HIP_TRY(hipMemsetAsync(b->d_gtable, 0, off * 8, b->stream));          // all-zero again: the next launch's precondition (nobody waits for it here)
    b->gtable_dirty = false;
  }
  L->t1 = std::chrono::steady_clock::now();
  L->launched = true;
  return PG_OK;
}

pg_status finish_group_launch(DeferredLaunch* L, std::vector<Deferred>& defs, pg_result* results, pg_status* statuses) {
  BatchCtx* b = L->b;
  const int n = L->n;
  static const bool trace = getenv("PINOT_GPU_BATCH_TRACE") != nullptr;
  HIP_TRY(hipSetDevice(phys_device(L->device)));
  float ms = 0.f;
  if (L->publish) {
    // Every item's last workgroup has written the item's slice to the pinned image and then the launch's sequence number to the item's
    // record: an item is converted as soon as ITS number is there.  The items of a launch run side by side (every item has its share of the
    // resident workgroups), so their numbers arrive together near the end of the kernel -- measured (profiles/r6/group_publish_trace.txt): what
    // this form saves is the copy command and the memset behind the kernel (~70 us at 64 x 1 000 groups) and the helpers' wake-up (they are
    // woken when the kernel starts, sleep through most of its expected duration and poll the rest: ~12 us of conversion per item then
    // starts the moment the numbers are there).
    const unsigned long long seq = L->seq;
    const int conv_threads = n <= 1 ? 1 : (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency() / 2));      // (64 threads, one round of conversions, measured slower: 0.72-0.75 ms against 0.69-0.70)
    // (a group-by streams at most ~1.1 G docs per ms: 60 % of that time is slept, not polled, when it is worth a sleep)
    const auto sleep_until = L->t1 + std::chrono::nanoseconds((long long)((double)L->docs / 1100.0 * 0.6));
    const bool sleep_first = n > 1 && (double)L->docs / 1100.0 * 0.6 > 200'000.0;      // (nanoseconds)
    std::atomic<int> unpublished{0};
    std::atomic<long long> convert_ns{0}, wait_ns{0};          // (PINOT_GPU_BATCH_TRACE: summed over the items)
    run_items(n, conv_threads, 1, false, [&](int k) {
      const int i = L->items[(size_t)k];
      volatile unsigned long long* const flag = &b->h_records[k].seq;
      if (sleep_first && *flag != seq && std::chrono::steady_clock::now() < sleep_until) std::this_thread::sleep_until(sleep_until);
      const auto tw0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
      (void)wait_polled(b->stream, L->docs, [&] { return *flag == seq; });
      const auto tw1 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
      if (*flag != seq) { unpublished.fetch_add(1); statuses[i] = fail(PG_ERR_INTERNAL, "batch item %d did not publish its group table", i); return; }
      defs[(size_t)i].item->convert_group(b->h_gtable + L->table_offsets[(size_t)k], &results[i]);
      statuses[i] = PG_OK;
      if (trace) { wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(tw1 - tw0).count(); convert_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw1).count(); }
    });
    if (unpublished.load() != 0) b->gtable_dirty = true;
    if (L->timed) { HIP_TRY(hipEventSynchronize(b->ev[1])); HIP_TRY(hipEventElapsedTime(&ms, b->ev[0], b->ev[1])); }
    for (int k = 0; k < n; ++k) {
      const int i = L->items[(size_t)k];
      if (statuses[i] != PG_OK) continue;
      const float share = L->total_blocks > 0 ? ms * (float)L->blocks[(size_t)k] / (float)L->total_blocks : 0.f;
      results[i].device_ms = share;
      results[i].dominant_kernel_ms = share;
    }
    if (trace) fprintf(stderr, "  deferred group-by launch on device %d: %d items %lld workgroups, lds %zu, enqueue %.1f us, items published and converted %.1f us behind it (waits %.1f us, conversions %.1f us summed over the items), kernel %.1f us\n", L->device, n,
                       L->total_blocks, L->lds, std::chrono::duration<double, std::micro>(L->t1 - L->t0).count(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - L->t1).count(),
                       (double)wait_ns.load() * 1e-3, (double)convert_ns.load() * 1e-3, ms * 1e3);
    return PG_OK;
  }
  HIP_TRY(hipEventSynchronize(b->ev[2]));
  const auto t2 = std::chrono::steady_clock::now();
  if (L->timed) HIP_TRY(hipEventElapsedTime(&ms, b->ev[0], b->ev[1]));
  // the items' conversions (a thousand groups each: ~10 us) side by side on the library's worker threads
  run_items(n, (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency() / 2)), 1, false, [&](int k) {
    const int i = L->items[(size_t)k];
    defs[(size_t)i].item->convert_group(b->h_gtable + L->table_offsets[(size_t)k], &results[i]);
    const float share = L->total_blocks > 0 ? ms * (float)L->blocks[(size_t)k] / (float)L->total_blocks : 0.f;
    results[i].device_ms = share;
    results[i].dominant_kernel_ms = share;
    statuses[i] = PG_OK;
  });
  if (trace) fprintf(stderr, "  deferred group-by launch on device %d: %d items %lld workgroups, lds %zu, enqueue %.1f us, wait %.1f us, kernel %.1f us, convert %.1f us\n", L->device, n, L->total_blocks, L->lds,
                     std::chrono::duration<double, std::micro>(L->t1 - L->t0).count(), std::chrono::duration<double, std::micro>(t2 - L->t1).count(), ms * 1e3,
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t2).count());
  return PG_OK;
}

// The index-led items of one device (lean_kind 12): one launch of index_and_batch_kernel -- the items' workgroups
// in proportion to their windows, about what is resident in total -- every item with its own records, arrival counters and pinned host record.
static_assert(sizeof(IndexAndParams) <= kBatchItemSlot, "an index-AND item takes a slot of the batch's parameter blob");
pg_status enqueue_index_and_launch(DeferredLaunch* L, BatchCtx* b, std::vector<Deferred>& defs, pg_segment* const* segments) {
  const std::vector<int>& items = L->items;
  const int n = L->n;
  long long total_windows = 0;
  for (int i : items) { total_windows += defs[(size_t)i].item->and_windows; L->docs += (long long)segments[i]->num_docs; }
  const long long budget = (long long)segments[items[0]]->num_cus * index_and_batch_blocks_per_cu();      // workgroups (of index_and_batch_block_waves() wavefronts) resident at once
  std::vector<int>& blocks = L->blocks;
  blocks.assign((size_t)n, 0);
  size_t partials = 0;
  long long total_blocks = 0;
  for (int k = 0; k < n; ++k) {
    const LoweredItem& d = *defs[(size_t)items[(size_t)k]].item;
    const long long share = total_windows > 0 ? ((long long)d.and_windows * budget + total_windows - 1) / total_windows : 1;
    blocks[(size_t)k] = (int)std::max<long long>(1, std::min<long long>((long long)d.blocks, share));
    partials += (size_t)blocks[(size_t)k] + (size_t)kFoldExtraRecords;
    total_blocks += blocks[(size_t)k];
  }
  L->total_blocks = total_blocks;
  pg_status st = ensure_batch_ctx(b, n, partials);
  if (st != PG_OK) return st;
  IndexAndParams* h_items = reinterpret_cast<IndexAndParams*>(b->h_blob + b->items_offset);
  const IndexAndParams* d_items = reinterpret_cast<const IndexAndParams*>(b->d_blob + b->items_offset);
  size_t off = 0;
  uint32_t first = 0;
  const unsigned long long seq = L->seq = ++b->seq;
  for (int k = 0; k < n; ++k) {
    const LoweredItem& d = *defs[(size_t)items[(size_t)k]].item;
    IndexAndParams& ap = h_items[k];               // (the pinned copy the device reads: filled in place)
    ap = *d.and_params;
    ap.num_windows = (int32_t)d.and_windows;
    ap.out = nullptr; ap.window_info = nullptr;     // record mode: neither a bitmap nor window masks
    ap.shards = nullptr;
    memset(&ap.pub, 0, sizeof(ap.pub));
    ap.pub.done_counter = b->d_done + (size_t)k * (kFoldShards + 1) * kFoldStride;
    ap.pub.partials = b->d_partials + off;
    off += (size_t)blocks[(size_t)k] + (size_t)kFoldExtraRecords;
    ap.pub.host_out = b->h_records_dev + k;
    ap.pub.host_seq = seq;
    ap.pub.fold_slots = ap.gather_cols;
    ap.pub.fold_one_counter = g_engine.fold_one_counter;
    b->h_first[k] = first;
    first += (uint32_t)blocks[(size_t)k];
  }
  b->h_first[n] = first;
  L->timed = (g_engine.flags & PG_CFG_TIME_KERNELS) != 0;
  L->t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipMemcpyAsync(b->d_blob, b->h_blob, b->items_offset + sizeof(IndexAndParams) * (size_t)n, hipMemcpyHostToDevice, b->stream));
  if (L->timed) HIP_TRY(hipEventRecord(b->ev[0], b->stream));
  launch_index_and_batch((int)total_blocks, b->stream, d_items, b->d_first, n);
  HIP_TRY(hipGetLastError());
  if (L->timed) HIP_TRY(hipEventRecord(b->ev[1], b->stream));
  L->t1 = std::chrono::steady_clock::now();
  L->launched = true;
  return PG_OK;
}

pg_status enqueue_deferred(DeferredLaunch* L, std::vector<Deferred>& defs, pg_segment* const* segments) {
  const int device = L->device;
  const std::vector<int>& items = L->items;
  HIP_TRY(hipSetDevice(phys_device(device)));
  BatchCtx* b = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_batch_mu);
    // The context released LAST is taken first (round 6b).  Taken from the front, a caller rotated through every context the process had ever
    // needed at once -- sixteen after sixteen threads had run group-bys side by side -- and each of them grew its blob, set area and group
    // table on ITS first use by the new shape: the first dozen calls of a shape took twice their time (PINOT_GPU_BATCH_TRACE,
    // profiles/r6/batch_ctx_pool_lifo.txt; DESIGN.md section 9's "C5x64 in a busy process").
    for (size_t i = g_batch_free.size(); i-- > 0;) if (g_batch_free[i]->device == device) { b = g_batch_free[i]; g_batch_free.erase(g_batch_free.begin() + (long)i); break; }
  }
  if (!b) { b = new BatchCtx(); b->device = device; }
  L->b = b;
  const int n = L->n = (int)items.size();
  if (L->lean_kind == 6) return enqueue_group_launch(L, b, defs, segments);
  if (L->lean_kind == 12) return enqueue_index_and_launch(L, b, defs, segments);
  // Workgroups per item in proportion to its tiles, about sixteen per CU in total (four waves each: ~4x what is resident, so that
  // the items' tails overlap other items' scans); never more than the item would get on its own.
  long long total_tiles = 0;
  for (int i : items) { total_tiles += ((long long)segments[i]->num_docs + 2047) / 2048; L->docs += (long long)segments[i]->num_docs; }
  // (the lean kernels hold five -- raw: four -- waves per SIMD: a workgroup per CU more than the general body's four)
  const bool hist_kind = L->lean_kind >= 3 && L->lean_kind <= 5;
  const int hist_cw = L->lean_kind == 3 ? 8 : (L->lean_kind == 4 ? 16 : 32);
  const int wpb = hist_kind ? kHistBlockThreads / 64 : kBlockThreads / 64;      // wavefronts of a workgroup
  const bool narrow_kind = L->lean_kind == 7 || L->lean_kind == 8, typed_kind = L->lean_kind >= 9 && L->lean_kind <= 11;
  const int typed_slots = L->lean_kind == 9 ? 1 : (L->lean_kind == 10 ? 2 : kMaxAggCols);
  // tiles a wave takes per iteration (the narrow kernels walk four / eight tiles at a time)
  const int tiles_per_wave = L->lean_kind == 7 ? kNarrowTiles : (L->lean_kind == 8 ? kNarrowSingleTiles : 1);
  size_t launch_lds = 0;
  for (int i : items) launch_lds = std::max(launch_lds, defs[(size_t)i].item->hist_lds);
  L->lds = launch_lds;
  const int lean_bpc = hist_kind ? std::max(1, std::min(waves_scan_hist_batch(hist_cw) / wpb, (int)((160 * 1024 - 2048) / (launch_lds + 256))))
                       : narrow_kind ? std::max(1, waves_scan_narrow_batch(L->lean_kind == 8) / wpb)
                       : typed_kind ? std::max(1, waves_scan_typed_batch(typed_slots) / wpb)
                       : (L->lean_kind != 0 ? std::max(1, waves_scan_lean_batch(L->lean_kind) / wpb) : 0);
  const bool bpc_forced = g_engine.batch_blocks_per_cu_forced;      // (read once per pg_init: bench sweeps re-initialise the engine with it)
  const long long budget = (long long)segments[items[0]]->num_cus * ((L->lean_kind != 0 && (!bpc_forced || hist_kind)) ? lean_bpc : g_engine.batch_blocks_per_cu);
  std::vector<int>& blocks = L->blocks;
  blocks.assign((size_t)n, 0);
  size_t partials = 0;
  long long total_blocks = 0;
  bool one_slot = true;
  for (int k = 0; k < n; ++k) {
    const LoweredItem& d = *defs[(size_t)items[(size_t)k]].item;
    const long long tiles = ((long long)segments[items[(size_t)k]]->num_docs + 2047) / 2048;
    const long long share = total_tiles > 0 ? (tiles * budget + total_tiles - 1) / total_tiles : 1;
    blocks[(size_t)k] = (int)std::max<long long>(1, std::min<long long>({(long long)d.blocks, share, (tiles + (long long)wpb * tiles_per_wave - 1) / ((long long)wpb * tiles_per_wave)}));
    partials += (size_t)blocks[(size_t)k] + (size_t)kFoldExtraRecords;
    total_blocks += blocks[(size_t)k];
    one_slot = one_slot && d.one_slot;
  }
  L->total_blocks = total_blocks;
  size_t set_bytes = 0;
  for (int i : items) set_bytes += item_set_bytes(*defs[(size_t)i].item);
  pg_status st = ensure_batch_ctx(b, n, partials, set_bytes);
  if (st != PG_OK) return st;
  size_t off = 0, set_off = 0;
  uint32_t first = 0;
  const unsigned long long seq = L->seq = ++b->seq;
  for (int k = 0; k < n; ++k) {
    ScanParams& sp = b->h_items[k];               // (the pinned copy the device reads: filled in place)
    sp = defs[(size_t)items[(size_t)k]].item->sp;
    place_item_sets(b, *defs[(size_t)items[(size_t)k]].item, &sp, &set_off);
    sp.partials = b->d_partials + off;
    off += (size_t)blocks[(size_t)k] + (size_t)kFoldExtraRecords;
    sp.done_counter = b->d_done + (size_t)k * (kFoldShards + 1) * kFoldStride;
    sp.host_out = b->h_records_dev + k;
    sp.host_seq = seq;
    b->h_first[k] = first;
    first += (uint32_t)blocks[(size_t)k];
  }
  b->h_first[n] = first;
  L->timed = (g_engine.flags & PG_CFG_TIME_KERNELS) != 0;
  L->t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipMemcpyAsync(b->d_blob, b->h_blob, b->items_offset + sizeof(ScanParams) * (size_t)n, hipMemcpyHostToDevice, b->stream));
  if (set_off) HIP_TRY(hipMemcpyAsync(b->d_blob + b->sets_offset, b->h_blob + b->sets_offset, set_off, hipMemcpyHostToDevice, b->stream));
  if (L->timed) HIP_TRY(hipEventRecord(b->ev[0], b->stream));
  if (hist_kind) launch_scan_hist_batch(hist_cw, (int)total_blocks, launch_lds, b->stream, b->d_items, b->d_first, n);
  else if (narrow_kind) launch_scan_narrow_batch(L->lean_kind == 8, (int)total_blocks, b->stream, b->d_items, b->d_first, n);
  else if (typed_kind) launch_scan_typed_batch(typed_slots, (int)total_blocks, b->stream, b->d_items, b->d_first, n);
  else if (L->lean_kind != 0) launch_scan_lean_batch(L->lean_kind, (int)total_blocks, b->stream, b->d_items, b->d_first, n);
  else launch_scan_private_batch(one_slot, (int)total_blocks, b->stream, b->d_items, b->d_first, n);
  HIP_TRY(hipGetLastError());
  if (L->timed) HIP_TRY(hipEventRecord(b->ev[1], b->stream));
  L->t1 = std::chrono::steady_clock::now();
  L->launched = true;
  return PG_OK;
}

pg_status finish_deferred(DeferredLaunch* L, std::vector<Deferred>& defs, pg_result* results, pg_status* statuses) {
  if (L->lean_kind == 6) return finish_group_launch(L, defs, results, statuses);
  BatchCtx* b = L->b;
  const int n = L->n;
  const unsigned long long seq = L->seq;
  static const bool trace = getenv("PINOT_GPU_BATCH_TRACE") != nullptr;
  HIP_TRY(hipSetDevice(phys_device(L->device)));
  if (g_engine.poll_result && !L->timed) {
    // every item publishes its own pinned record: their sequence numbers are the completion signal (as in pg_execute)
    int k = 0;
    const pg_status st = wait_polled(b->stream, L->docs, [&] { while (k < n && *(volatile unsigned long long*)&b->h_records[k].seq == seq) ++k; return k == n; });
    if (st != PG_OK) return st;
  } else {
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  const auto t2 = std::chrono::steady_clock::now();
  float ms = 0.f;
  if (L->timed) HIP_TRY(hipEventElapsedTime(&ms, b->ev[0], b->ev[1]));
  if (trace) fprintf(stderr, "  deferred launch on device %d: %d items %lld workgroups, sizeof(ScanParams) %zu, enqueue %.1f us, wait %.1f us, kernel %.1f us\n", L->device, n, L->total_blocks,
                     sizeof(ScanParams), std::chrono::duration<double, std::micro>(L->t1 - L->t0).count(), std::chrono::duration<double, std::micro>(t2 - L->t1).count(), ms * 1e3);
  for (int k = 0; k < n; ++k) {
    const int i = L->items[(size_t)k];
    if (b->h_records[k].seq != seq) { statuses[i] = fail(PG_ERR_INTERNAL, "batch item %d did not publish its record", i); continue; }
    if (b->h_records[k].partial.flags & kPartialStale) { statuses[i] = fail(PG_ERR_INTERNAL, "batch item %d: the fold read a record that was not written by this launch", i); continue; }
    const LoweredItem& it = *defs[(size_t)i].item;
    if (it.hist_cw != 0 && it.hist_cw < 32 && (unsigned long long)b->h_records[k].partial.sum[1] != b->h_records[k].partial.count) {
      // a plain narrow counter wrapped (skewed dictIds): the histogram's sum is not used -- pg_execute answers this item again and moves
      // the column to the guarded tier, where its items run launches of their own (scan_hist_kernel's header: exact or not used)
      statuses[i] = kRunAlone;
      continue;
    }
    defs[(size_t)i].item->convert(b->h_records[k].partial, &results[i]);
    // ONE launch serves all items of the device: each item is charged its share of the workgroups, so that summing device_ms over a
    // batch's results gives the launch's time once (pg_result.device_ms of a batch item is an apportioned figure, not a measurement of its own)
    const float share = L->total_blocks > 0 ? ms * (float)L->blocks[(size_t)k] / (float)L->total_blocks : 0.f;
    results[i].device_ms = share;
    results[i].dominant_kernel_ms = share;
    statuses[i] = PG_OK;
  }
  return PG_OK;
}

}  // namespace

// A group-by of the LDS-table form (key space up to the LDS table: the C3 shape) is ONE launch here too: the one-item form of the batch's
// group_lds_batch_kernel -- no init_group_table_kernel, no count / scan / compact launches and the two waits between them; the item's
// all-zero table slice comes back whole and the host keeps the slots whose count is not zero (§4.1k).  Its lowering is remembered in the
// segment's plan cache like a batch item's.  Everything else takes execute_one as before.  PINOT_GPU_GROUP_ONE_LAUNCH=0: never.
pg_status pg_execute(pg_segment* segment, const pg_query* query, pg_result* out_result) {
  // (PG_CFG_PROFILE_WAVES: the per-wave phase counters live in the kernels execute_one launches itself, and its HIP events bracket ALL of a
  //  query's launches -- the deferred form would leave profile_* empty and report the kernel bracket only)
  if (!(g_engine.group_one_launch && g_engine.batch_launch && g_engine.batch_group && segment && query && out_result && query->num_group_by > 0 &&
        !(query->flags & PG_QUERY_NULL_HANDLING) && !(g_engine.flags & PG_CFG_PROFILE_WAVES) && g_engine.initialized))
  {
    if (exec_trace_on()) { memset(t_exec_trace.seen, 0, sizeof(t_exec_trace.seen)); exec_mark(0); }
    const pg_status st = execute_one(segment, query, out_result, nullptr);
    if (exec_trace_on()) {
      exec_mark(5);
      const ExecTrace& x = t_exec_trace;
      auto us = [&](int a, int b) { return (x.seen[a] && x.seen[b]) ? std::chrono::duration<double, std::micro>(x.t[b] - x.t[a]).count() : -1.0; };
      fprintf(stderr, "exec trace: status %d, check %.1f us, context + lowering %.1f us, launches %.1f us, wait %.1f us, after the wait %.1f us, total %.1f us (-1: phase not passed)\n",
              (int)st, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(0, 5));
    }
    return st;
  }
  std::vector<Deferred> defs(1);
  defs[0].single = true;
  std::string key;
  pg_status st = kDeferred;
  const bool keyed = g_engine.plan_cache && query_key(query, &key);
  if (keyed && cached_item(segment, key, &defs[0]) && defs[0].item->gp != nullptr) {
    memset(out_result, 0, sizeof(*out_result));
  } else {
    defs[0] = Deferred();
    defs[0].single = true;
    std::shared_ptr<const OwnedQuery> copy;
    const pg_query* q = query;
    if (keyed && !key.empty()) { copy = own_query(query); q = &copy->q; }
    st = execute_one(segment, q, out_result, &defs[0]);
    if (st == kDeferred && copy && defs[0].cacheable) remember_item(segment, std::move(key), copy, &defs[0]);
    else if (st == kDeferred && copy) std::const_pointer_cast<LoweredItem>(defs[0].item)->query = copy;
    if (st != kDeferred) return st;                     // ran the usual way (or failed): nothing was deferred
  }
  DeferredLaunch L;
  L.device = segment->device; L.lean_kind = 6; L.items.push_back(0);
  pg_segment* const segs[1] = {segment};
  pg_status item_status = PG_ERR_INTERNAL;
  st = enqueue_deferred(&L, defs, segs);
  if (st == PG_OK) st = finish_deferred(&L, defs, out_result, &item_status);
  if (st == PG_OK) st = item_status;
  if (st != PG_OK) pg_result_free(out_result);
  return st;
}

pg_status pg_execute_batch(pg_segment* const* segments, const pg_query* const* queries, int32_t count, pg_result* results, pg_status* statuses) {
  if (!g_engine.initialized) return fail(PG_ERR_NOT_INITIALIZED, "pg_init has not been called");
  if (count < 0 || (count > 0 && (!segments || !queries || !results || !statuses))) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  for (int i = 0; i < count; ++i) { memset(&results[i], 0, sizeof(pg_result)); statuses[i] = PG_ERR_INVALID_ARGUMENT; }
  if (count == 0) return PG_OK;
  static const bool trace = getenv("PINOT_GPU_BATCH_TRACE") != nullptr;      // host phases of every call on stderr
  const auto t_begin = std::chrono::steady_clock::now();
  std::vector<Deferred> defs((size_t)count);
  std::vector<std::string> errors((size_t)count);
  // 1. every item is lowered (and, when it cannot share the launch, run on a context of its own) on the library's worker threads --
  // except the items whose segment has this very query in its plan cache: those are picked up on the calling thread (~0.2 us each)
  const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 2));
  std::vector<float> item_us(trace ? (size_t)count : 0);
  const bool plan_cache = g_engine.plan_cache;
  std::vector<int> todo;
  std::vector<std::string> keys(plan_cache && g_engine.batch_launch ? (size_t)count : 0);
  int cache_hits = 0;
  for (int i = 0; i < count; ++i) {
    if (!keys.empty() && segments[i] && queries[i] && !(queries[i]->flags & PG_QUERY_NULL_HANDLING) && query_key(queries[i], &keys[(size_t)i]) &&
        cached_item(segments[i], keys[(size_t)i], &defs[(size_t)i])) { statuses[i] = kDeferred; ++cache_hits; continue; }
    todo.push_back(i);
  }
  // (a deferred item is ~2 us of lowering: eight per claim, so 64 items wake at most seven helpers; an item that runs its own kernel
  // is claimed alone)
  // (an item too large for the shared launch runs its whole kernel inside its claim: such batches are claimed one item at a time)
  bool all_small = g_engine.batch_launch;
  for (int i : todo) all_small = all_small && segments[i] != nullptr && ((long long)segments[i]->num_docs + 2047) / 2048 <= kBatchMaxTiles;
  const int todo_count = (int)todo.size();
  if (todo_count > 0) run_items(todo_count, all_small ? threads : std::min(threads, todo_count), all_small ? 8 : 1, !all_small, [&](int t) {
    const int i = todo[(size_t)t];
    if (!segments[i] || !queries[i]) { statuses[i] = PG_ERR_INVALID_ARGUMENT; errors[(size_t)i] = "null segment or query"; return; }
    const auto t_item = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    // (a query that may enter the plan cache is lowered from a copy of its own: the item's conversion reads the query after this call has returned)
    const pg_query* q = queries[i];
    std::shared_ptr<const OwnedQuery> copy;
    if (!keys.empty() && !keys[(size_t)i].empty()) { copy = own_query(q); q = &copy->q; }
    statuses[i] = execute_one(segments[i], q, &results[i], g_engine.batch_launch ? &defs[(size_t)i] : nullptr);
    if (statuses[i] == kDeferred && copy && defs[(size_t)i].cacheable) remember_item(segments[i], std::move(keys[(size_t)i]), copy, &defs[(size_t)i]);
    else if (statuses[i] == kDeferred && copy) std::const_pointer_cast<LoweredItem>(defs[(size_t)i].item)->query = copy;
    if (trace) item_us[(size_t)i] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t_item).count();
    if (statuses[i] != PG_OK && statuses[i] != kDeferred) errors[(size_t)i] = g_error;      // (g_error is the worker's thread-local)
  });
  const auto t_lowered = std::chrono::steady_clock::now();
  // 2. the deferred items, one launch per device: every device's launch is enqueued before any of them is waited for
  std::vector<std::unique_ptr<DeferredLaunch>> launches;
  for (int i = 0; i < count; ++i) {
    if (statuses[i] != kDeferred) continue;
    DeferredLaunch* L = nullptr;
    const int kind = defs[(size_t)i].item->sp.lean_kind;
    for (auto& l : launches) if (l->device == segments[i]->device && l->lean_kind == kind) L = l.get();
    if (!L) { launches.emplace_back(new DeferredLaunch()); L = launches.back().get(); L->device = segments[i]->device; L->lean_kind = kind; }
    L->items.push_back(i);
  }
  auto fail_items = [&](DeferredLaunch* L, pg_status st) {
    for (int i : L->items) if (statuses[i] == kDeferred) { statuses[i] = st; errors[(size_t)i] = g_error; pg_result_free(&results[i]); }
  };
  for (auto& l : launches) { const pg_status st = enqueue_deferred(l.get(), defs, segments); if (st != PG_OK) fail_items(l.get(), st); }
  for (auto& l : launches) {
    if (!l->launched) continue;
    const pg_status st = finish_deferred(l.get(), defs, results, statuses);
    if (st != PG_OK) fail_items(l.get(), st);
    else for (int i : l->items) if (statuses[i] != PG_OK) errors[(size_t)i] = g_error;
  }
  for (int i = 0; i < count; ++i) {
    if (statuses[i] != kRunAlone) continue;
    defs[(size_t)i].planes.reset();
    statuses[i] = execute_one(segments[i], queries[i], &results[i], nullptr);
    if (statuses[i] != PG_OK) errors[(size_t)i] = g_error;
  }
  // pg_last_error of the caller: the first failed item's message
  for (int i = 0; i < count; ++i) if (statuses[i] != PG_OK) { g_error = "batch item " + std::to_string(i) + ": " + errors[(size_t)i]; break; }
  if (trace) {
    const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    float sum_us = 0.f, max_us = 0.f;
    for (float v : item_us) { sum_us += v; max_us = std::max(max_us, v); }
    fprintf(stderr, "pg_execute_batch: %d items (%d from the plan cache), lower %.1f us (items: sum %.1f us, max %.1f us, %d threads), launches %.1f us\n", count, cache_hits,
            us(t_begin, t_lowered), sum_us, max_us, threads, us(t_lowered, std::chrono::steady_clock::now()));
  }
  return PG_OK;
}

pg_status pg_segment_plane_bytes(const pg_segment* segment, uint64_t* out_bytes) {
  if (!segment || !out_bytes) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> lk(g_planes.mu);
  *out_bytes = segment->plane_bytes;
  return PG_OK;
}

pg_status pg_set_plane_budget(uint64_t budget_bytes, uint64_t* out_previous) {
  std::lock_guard<std::mutex> lk(g_planes.mu);
  if (out_previous) *out_previous = g_planes.budget_bytes;
  g_planes.budget_bytes = budget_bytes;
  return PG_OK;
}

pg_status pg_query_check(const pg_segment* segment, const pg_query* query) {
  if (!g_engine.initialized) return fail(PG_ERR_NOT_INITIALIZED, "pg_init has not been called");
  if (!segment || !query) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  if (!(query->flags & PG_QUERY_NULL_HANDLING)) return check_query_plan(segment, query, 0);
  // enableNullHandling: the query that runs is the rewritten one (execute_null_handling): the filter's getTrues() over three-valued
  // leaves, one more IS NOT NULL leaf for every aggregated column that has null docs
  NullRewriter rw{segment};
  FilterExpr root;
  bool has_filter = false;
  pg_status st = parse_filter(query, &root, &has_filter);
  if (st != PG_OK) return st;
  FlatQuery base;
  if (has_filter) base.emit(rw.trues(root));
  base.finish(*query);
  const int na = query->num_aggregations, ng = query->num_group_by;
  if (na < 0 || ng < 0 || (na > 0 && !query->aggregations) || (ng > 0 && !query->group_by_columns)) return fail(PG_ERR_INVALID_ARGUMENT, "bad aggregation / group-by lists");
  bool lanes = false;
  for (int a = 0; a < na; ++a) lanes |= rw.has_nulls(query->aggregations[a].column);
  std::vector<int32_t> keys;
  if (ng > 0) {
    // nullable keys are read through their null-key images (one more digit value each: the key space grows)
    for (int g = 0; g < ng; ++g) {
      const int c = query->group_by_columns[g];
      if (rw.has_nulls(c) && segment->cols[(size_t)c].nullkey_column < 0) return fail(PG_ERR_UNSUPPORTED, "GROUP BY over nullable raw column keeps the CPU plan");
      keys.push_back(rw.has_nulls(c) ? segment->cols[(size_t)c].nullkey_column : c);
    }
    base.q.group_by_columns = keys.data();
  }
  return check_query_plan(segment, &base.q, lanes ? 1 : 0);
}

pg_status pg_filter_bitmap(pg_segment* segment, const pg_query* query, uint64_t* out_words, int64_t num_words, int64_t* out_cardinality) {
  if (!out_words) return fail(PG_ERR_INVALID_ARGUMENT, "null bitmap buffer");
  if (query && (query->flags & PG_QUERY_NULL_HANDLING)) return execute_null_handling(segment, query, nullptr, out_words, num_words, out_cardinality);
  return execute_impl(segment, query, nullptr, nullptr, out_words, num_words, out_cardinality);
}

static pg_status gather_impl(pg_segment* seg, int32_t column, const int32_t* doc_ids, int32_t length, int32_t* out_dict, int32_t* out_int, double* out_double,
                             int64_t* out_long = nullptr) {
  if (!g_engine.initialized) return fail(PG_ERR_NOT_INITIALIZED, "pg_init has not been called");
  if (!seg || (length > 0 && !doc_ids)) return fail(PG_ERR_INVALID_ARGUMENT, "null argument");
  if (column < 0 || column >= (int)seg->cols.size()) return fail(PG_ERR_INVALID_ARGUMENT, "column %d out of range", column);
  if (length <= 0) return PG_OK;
  for (int32_t i = 0; i < length; ++i) if (doc_ids[i] < 0 || doc_ids[i] >= seg->num_docs) return fail(PG_ERR_INVALID_ARGUMENT, "docId %d out of range", doc_ids[i]);
  const ColumnDev& col = seg->cols[(size_t)column];
  if (out_dict && col.encoding != PG_FWD_FIXED_BIT_DICT) return fail(PG_ERR_INVALID_ARGUMENT, "column %s is not dictionary encoded", col.name.c_str());
  if (out_int && col.stored_type != PG_TYPE_INT) return fail(PG_ERR_INVALID_ARGUMENT, "column %s is not INT: use pg_read_long_values / pg_read_double_values", col.name.c_str());
  HIP_TRY(hipSetDevice(phys_device(seg->device)));
  ExecCtx* ctx = nullptr;
  pg_status st = acquire_ctx(seg, &ctx);
  if (st != PG_OK) return st;
  CtxGuard guard{seg, ctx};
  if (ctx->gather_capacity < (size_t)length) {
    if (ctx->d_gather_in) (void)hipFree(ctx->d_gather_in);
    if (ctx->d_gather_out) (void)hipFree(ctx->d_gather_out);
    ctx->d_gather_in = nullptr; ctx->d_gather_out = nullptr; ctx->gather_capacity = 0;
    size_t cap = std::max<size_t>((size_t)length, 16384);
    HIP_TRY(hipMalloc((void**)&ctx->d_gather_in, cap * 4));
    HIP_TRY(hipMalloc((void**)&ctx->d_gather_out, cap * 8));
    ctx->gather_capacity = cap;
  }
  HIP_TRY(hipMemcpyAsync(ctx->d_gather_in, doc_ids, (size_t)length * 4, hipMemcpyHostToDevice, ctx->stream));
  DevColumn dc;
  memset(&dc, 0, sizeof(dc));
  dc.fwd = col.d_fwd; dc.bits = col.bits; dc.is_raw = col.encoding == PG_FWD_RAW_FIXED_BYTE;
  dc.vkind = col.vkind;
  dc.dict = col.vkind == kValI32 ? col.d_dict : reinterpret_cast<const int32_t*>(col.d_dict64);
  dc.cardinality = col.cardinality; dc.dict_bytes = col.cardinality * (col.vkind == kValI32 ? 4 : 8);
  const unsigned blocks = (unsigned)((length + 255) / 256);
  gather_values_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(dc, (long long)col.value_base, ctx->d_gather_in, length,
      out_dict ? (int32_t*)ctx->d_gather_out : nullptr, out_int ? (int32_t*)ctx->d_gather_out : nullptr,
      out_long ? (long long*)ctx->d_gather_out : nullptr, out_double ? (double*)ctx->d_gather_out : nullptr);
  HIP_TRY(hipGetLastError());
  void* host_out = out_dict ? (void*)out_dict : (out_int ? (void*)out_int : (out_long ? (void*)out_long : (void*)out_double));
  HIP_TRY(hipMemcpyAsync(host_out, ctx->d_gather_out, (size_t)length * ((out_double || out_long) ? 8 : 4), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return PG_OK;
}

pg_status pg_read_dict_ids(pg_segment* segment, int32_t column, const int32_t* doc_ids, int32_t length, int32_t* out_dict_ids) {
  if (!out_dict_ids && length > 0) return fail(PG_ERR_INVALID_ARGUMENT, "null output");
  return gather_impl(segment, column, doc_ids, length, out_dict_ids, nullptr, nullptr);
}
pg_status pg_read_int_values(pg_segment* segment, int32_t column, const int32_t* doc_ids, int32_t length, int32_t* out_values) {
  if (!out_values && length > 0) return fail(PG_ERR_INVALID_ARGUMENT, "null output");
  return gather_impl(segment, column, doc_ids, length, nullptr, out_values, nullptr);
}
pg_status pg_read_long_values(pg_segment* segment, int32_t column, const int32_t* doc_ids, int32_t length, int64_t* out_values) {
  if (!out_values && length > 0) return fail(PG_ERR_INVALID_ARGUMENT, "null output");
  return gather_impl(segment, column, doc_ids, length, nullptr, nullptr, nullptr, out_values);
}
pg_status pg_read_double_values(pg_segment* segment, int32_t column, const int32_t* doc_ids, int32_t length, double* out_values) {
  if (!out_values && length > 0) return fail(PG_ERR_INVALID_ARGUMENT, "null output");
  return gather_impl(segment, column, doc_ids, length, nullptr, nullptr, out_values);
}

}  // extern "C"
