/*
 * pinot_gpu.h -- C ABI of the MI355X segment scan-filter-aggregate engine.
 *
 * This is the drop-in boundary: the entry points a JNI shim inside Pinot's
 * `PlanMaker.makeSegmentPlanNode` replacement binds to.
 * Plain C, plain pointers and sizes, no C++/torch types.
 *
 * Reference interfaces each entry point replaces (paths under /root/reference,
 *   core/ = pinot-core/src/main/java/org/apache/pinot/core/,
 *   segl/ = pinot-segment-local/src/main/java/org/apache/pinot/segment/local/,
 *   sspi/ = pinot-segment-spi/src/main/java/org/apache/pinot/segment/spi/):
 *
 *   pg_init / pg_shutdown      PlanMaker.init(PinotConfiguration)            core/plan/maker/PlanMaker.java:42
 *   pg_segment_open            ImmutableSegment load: DataSource per column   sspi/datasource/DataSource.java:46-132
 *                              (ForwardIndexReader + Dictionary + InvertedIndexReader buffers handed over once)
 *   pg_segment_close           IndexSegment.destroy()                         sspi/IndexSegment.java:142
 *   pg_execute                 PlanNode.run().nextBlock() of AggregationPlanNode / GroupByPlanNode:
 *                              core/plan/AggregationPlanNode.java:71-121, core/plan/GroupByPlanNode.java:49-74,
 *                              core/operator/query/AggregationOperator.java:64-93, GroupByOperator.java:101-140
 *   pg_filter_bitmap           BaseFilterOperator.nextBlock().getBlockDocIdSet() materialised as a docId bitmap
 *                              core/operator/filter/BaseFilterOperator.java, core/common/BlockDocIdSet.java:61-67
 *   pg_read_dict_ids           ForwardIndexReader.readDictIds(int[] docIds, int length, int[] dictIdBuffer, ctx)
 *                              sspi/index/reader/ForwardIndexReader.java, segl/.../FixedBitSVForwardIndexReaderV2.java:65-99
 *   pg_read_int_values /       DataFetcher.fetchIntValues / fetchLongValues / fetchDoubleValues (BlockValSet.getIntValuesSV /
 *   pg_read_long_values /      getLongValuesSV / getDoubleValuesSV)
 *   pg_read_double_values      core/common/DataFetcher.java:111-123,335-470, core/common/BlockValSet.java:65-93
 *
 * Inside pg_execute the plan rules of the reference are applied to the lowered query: a filter that matches everything plus
 * COUNT / dictionary-based MIN, MAX is answered from metadata (AggregationPlanNode.java:98-115, NonScanBasedAggregationOperator);
 * PG_PRED_DOC_RANGE leaves stand for SortedIndexBasedFilterOperator, PG_EVAL_INVERTED leaves for InvertedIndexFilterOperator.
 *   pg_last_error              exception message carried into BaseCombineOperator.wrapOperatorException
 *                              core/operator/combine/BaseCombineOperator.java:185-199
 *
 * Ownership: the caller owns every pointer inside pg_segment_desc / pg_query; pg_segment_open copies
 * the index buffers into HBM and never dereferences the host pointers afterwards.  Results are
 * engine-owned POD, released by pg_result_free.  All functions return a status code and never
 * throw or abort; pg_last_error() returns the thread-local message of the last failure.
 * pg_execute is re-entrant per segment handle (concurrent queries on one segment), pg_segment_open /
 * pg_segment_close must be serialised per segment by the caller (the combine Phaser already does,
 * core/operator/combine/BaseCombineOperator.java:86-92).
 */
#ifndef PINOT_GPU_H
#define PINOT_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 3   /* 2: pg_result.filter_entries_exact, pg_query_check, pg_config.plane_budget_bytes;
                            * 3: pg_execute_batch, pg_result.group_key_kind / group_ids64 / group_key_dict_ids (Long / ArrayMap holders) */

typedef enum pg_status {
  PG_OK = 0,
  PG_ERR_INVALID_ARGUMENT = 1,
  PG_ERR_UNSUPPORTED = 2,      /* query shape not offloadable: caller keeps the CPU plan (decided at plan time) */
  PG_ERR_DEVICE = 3,           /* HIP runtime error */
  PG_ERR_OUT_OF_MEMORY = 4,
  PG_ERR_NOT_INITIALIZED = 5,
  PG_ERR_INTERNAL = 6
} pg_status;

/* FieldSpec.DataType stored types supported for fixed-width single-value columns
 * (pinot-spi/src/main/java/org/apache/pinot/spi/data/FieldSpec.java DataType.getStoredType()). */
typedef enum pg_data_type {
  PG_TYPE_INT = 0,
  PG_TYPE_LONG = 1,
  PG_TYPE_FLOAT = 2,
  PG_TYPE_DOUBLE = 3
} pg_data_type;

/* Forward-index encodings (segl/segment/index/forward/ForwardIndexReaderFactory.java:75-117). */
typedef enum pg_fwd_encoding {
  PG_FWD_FIXED_BIT_DICT = 0,   /* FixedBitSVForwardIndexReaderV2: big-endian MSB-first bit-packed dictIds, no header */
  PG_FWD_RAW_FIXED_BYTE = 1    /* FixedByteChunkSVForwardIndexReader, PASS_THROUGH: chunk header + big-endian values */
} pg_fwd_encoding;

typedef struct pg_config {
  int32_t abi_version;         /* must be PG_ABI_VERSION */
  int32_t device_id;           /* default HIP device for segments whose desc says device_id = -1 */
  int32_t blocks_per_cu;       /* 0 = engine default; launch geometry knob for tuning */
  int32_t flags;               /* PG_CFG_* */
  uint64_t plane_budget_bytes; /* HBM the value planes of ALL open segments may hold together (see pg_segment_plane_bytes); 0 = a quarter
                                * of the device's memory.  Over budget, the least recently used planes that no query is reading go first. */
} pg_config;

#define PG_CFG_TIME_KERNELS 1  /* bracket kernels with HIP events on the launch stream; report pg_result.device_ms */
#define PG_CFG_PROFILE_WAVES 2 /* diagnostic: per-wave s_memtime phase counters in pg_result.profile_cycles */

typedef struct pg_column_desc {
  const char* name;
  int32_t stored_type;         /* pg_data_type of the column values (dictionary values or raw values) */
  int32_t fwd_encoding;        /* pg_fwd_encoding */
  int32_t bits_per_value;      /* FIXED_BIT_DICT: PinotDataBitSet.getNumBitsPerValue(cardinality - 1) */
  int32_t cardinality;         /* dictionary length (0 for raw columns) */
  const void* fwd_data;        /* whole forward-index buffer exactly as on disk */
  uint64_t fwd_size;
  const void* dict_data;       /* sorted dictionary, big-endian fixed-width values, no header (NULL for raw) */
  uint64_t dict_size;
  const void* inv_data;        /* optional bitmap inverted index (BitmapInvertedIndexWriter layout) or NULL */
  uint64_t inv_size;
  const void* null_data;       /* optional null value vector: the `<column>.bitmap.nullvalue` file = ONE serialized RoaringBitmap of the
                                * null docIds (NullValueVectorCreator.seal, segment/creator/impl/nullvalue/NullValueVectorCreator.java:69-78;
                                * read by NullValueVectorReaderImpl.getNullBitmap :44-46), or NULL when the column has no null docs (the
                                * creator writes no file for an empty bitmap).  The forward index holds the default null value there. */
  uint64_t null_size;
} pg_column_desc;

typedef struct pg_segment_desc {
  const char* name;
  uint64_t crc;
  int32_t num_docs;
  int32_t num_columns;
  const pg_column_desc* columns;
  int32_t device_id;           /* -1 = pg_config.device_id */
  int32_t reserved;
} pg_segment_desc;

typedef struct pg_segment pg_segment;   /* opaque, HBM-resident segment */

/* ---- query description: the filter is already lowered the way the reference's PredicateEvaluators
 * lower it (dictionary binary search happens on the caller side or in the host mirror). ---- */

typedef enum pg_predicate_kind {
  PG_PRED_MATCH_ALL = 0,       /* predicateEvaluator.isAlwaysTrue()  -> MatchAllFilterOperator */
  PG_PRED_MATCH_NONE = 1,      /* predicateEvaluator.isAlwaysFalse() -> EmptyFilterOperator */
  PG_PRED_DICT_RANGE = 2,      /* SortedDictionaryBasedRangePredicateEvaluator.applySV: lo <= dictId < hi; EQ is [d, d+1) */
  PG_PRED_DICT_SET = 3,        /* DictionaryBasedInPredicateEvaluator.applySV: bit dictId of set_words is set */
  PG_PRED_RAW_RANGE = 4,       /* IntRawValueBasedRangePredicateEvaluator.applySV: lo <= value <= hi (both inclusive) */
  PG_PRED_DOC_RANGE = 5,       /* SortedIndexBasedFilterOperator (core/operator/filter/SortedIndexBasedFilterOperator.java:60-85): the docId
                                * range [lo, hi] (both inclusive) that SortedIndexReader.getDocIds gives for the predicate's dictIds on a sorted
                                * column; `exclusive` inverts it over [0, numDocs).  No column is read.  `column` is ignored unless the query
                                * runs with PG_QUERY_NULL_HANDLING, where it names the sorted column (its null docs are excluded) or is -1. */
  PG_PRED_IS_NULL = 6          /* FilterPlanNode.java:294-307: BitmapBasedFilterOperator over the column's null bitmap; `exclusive` = IS NOT NULL.
                                * A column without a null vector matches nothing (IS NULL) / everything (IS NOT NULL).  No entries are scanned. */
} pg_predicate_kind;

typedef enum pg_leaf_eval {
  PG_EVAL_SCAN = 0,            /* ScanBasedFilterOperator */
  PG_EVAL_INVERTED = 1         /* InvertedIndexFilterOperator: OR of the postings of the matching dictIds */
} pg_leaf_eval;

typedef struct pg_predicate {
  int32_t kind;                /* pg_predicate_kind */
  int32_t column;              /* index into pg_segment_desc.columns */
  int32_t eval;                /* pg_leaf_eval */
  int32_t exclusive;           /* 1 = NOT_EQ / NOT_IN: matches when the inner predicate does not */
  int64_t lo;                  /* DICT_RANGE: startDictId ; RAW_RANGE: inclusive lower bound */
  int64_t hi;                  /* DICT_RANGE: endDictId (exclusive) ; RAW_RANGE: inclusive upper bound.
                                * RAW_RANGE on a raw FLOAT / DOUBLE column: lo / hi carry the IEEE-754 bit pattern of the inclusive
                                * bounds as doubles (Float / DoubleRawValueBasedRangePredicateEvaluator after Math.nextUp / nextDown
                                * made exclusive bounds inclusive; FLOAT bounds widened exactly). */
  const uint32_t* set_words;   /* DICT_SET: bitset over dictIds, bit d = (set_words[d >> 5] >> (d & 31)) & 1 */
  int32_t num_set_words;
  int32_t reserved;
} pg_predicate;

typedef enum pg_filter_op {
  PG_FILTER_LEAF = 0,
  PG_FILTER_AND = 1,
  PG_FILTER_OR = 2,
  PG_FILTER_NOT = 3
} pg_filter_op;

/* Filter tree flattened in postfix order (children before parent). */
typedef struct pg_filter_node {
  int32_t op;                  /* pg_filter_op */
  int32_t predicate;           /* LEAF: index into pg_query.predicates */
  int32_t num_children;        /* AND / OR: operand count (>= 2); NOT: 1 */
  int32_t reserved;
} pg_filter_node;

typedef enum pg_agg_function {
  PG_AGG_COUNT = 0,            /* core/query/aggregation/function/CountAggregationFunction.java */
  PG_AGG_SUM = 1,              /* SumAggregationFunction.java */
  PG_AGG_MIN = 2,              /* MinAggregationFunction.java */
  PG_AGG_MAX = 3,              /* MaxAggregationFunction.java */
  PG_AGG_AVG = 4               /* AvgAggregationFunction.java (AvgPair = sum, count) */
} pg_agg_function;

typedef struct pg_aggregation {
  int32_t function;            /* pg_agg_function */
  int32_t column;              /* -1 for COUNT(*); COUNT(column) differs from COUNT(*) only under PG_QUERY_NULL_HANDLING */
} pg_aggregation;

typedef struct pg_query {
  const pg_filter_node* filter;      /* NULL / 0 nodes = match all */
  int32_t num_filter_nodes;
  int32_t num_predicates;
  const pg_predicate* predicates;
  const pg_aggregation* aggregations;
  int32_t num_aggregations;
  int32_t num_group_by;              /* 0 = aggregation only */
  const int32_t* group_by_columns;   /* dictionary-encoded columns; group id = raw key = sum dictId_j * prod_{k<j} card_k
                                      * (DictionaryBasedGroupKeyGenerator.java:298-338,437-445).  Product of cardinalities <= 10 000: the
                                      * reference's ArrayBasedHolder.  Up to Integer.MAX_VALUE: its IntMapBasedHolder range -- the same raw keys come back
                                      * (the reference's insertion-order group ids are internal to its hash map), at most num_groups_limit of
                                      * them: the groups whose first doc comes earliest, exactly those IntGroupIdMap.getGroupId :1022-1047
                                      * would have admitted.  The table is direct-indexed in HBM, 8 * (1 + distinct aggregations) bytes per raw
                                      * key, at most PINOT_GPU_GROUP_TABLE_BYTES (default 64 GiB of the 288) per query.  Beyond an int (the
                                      * reference's LongMapBasedHolder / ArrayMapBasedHolder, :628-806, :808+): a HASHED table in HBM of at
                                      * least 2 * min(numDocs, product) slots (same budget; segments up to 2^29 docs; aggregations of INT-domain
                                      * columns), keys come back in pg_result.group_ids64 / group_key_dict_ids.  Beyond the budget:
                                      * PG_ERR_UNSUPPORTED at plan time. */
  int32_t num_groups_limit;          /* InstancePlanMakerImplV2 numGroupsLimit (default 100000); 0 = default */
  int32_t flags;                     /* PG_QUERY_* */
} pg_query;

#define PG_QUERY_DEFAULT 0
/* Query option enableNullHandling=true (QueryContext.isNullHandlingEnabled), for columns that carry a null vector:
 *  - a column leaf is true only where the column is not null, and is NULL where it is (BaseColumnFilterOperator.getTrues / getNulls,
 *    core/operator/filter/BaseColumnFilterOperator.java:45-64); NOT takes the child's FALSE set, i.e. NOT (trues OR nulls)
 *    (BaseFilterOperator.getFalses :96-113, And/OrFilterOperator.getFalses, NotFilterOperator.getTrues); AND / OR / NOT / IS_NULL /
 *    MATCH_ALL / MATCH_NONE nodes have no NULL set of their own.  A predicate that is always true on a column WITH nulls must be passed
 *    as {PG_PRED_IS_NULL, exclusive} the way FilterOperatorUtils.java:78-86 builds it;
 *  - SUM / MIN / MAX / AVG / COUNT(column) skip the docs where their column is null (NullableSingleInputAggregationFunction.java:72-166);
 *    pg_agg_value.count is the number of non-null docs aggregated, and count == 0 means the reference's holder stays null;
 *  - the metadata / dictionary fast path is taken only when no aggregated column has nulls (AggregationPlanNode.java:99-100);
 *  - GROUP BY: NULL is a key value of its own (the reference switches to its no-dictionary key generators, DefaultGroupByExecutor.java:
 *    106-121).  On the raw group-id scale a key column WITH null docs has cardinality + 1 digit values, the last one meaning NULL:
 *    raw id = sum digit_j * prod_{k<j} (cardinality_k + hasNulls_k); group_id_upper_bound is that product.  Every function skips the
 *    null docs of its own column per group (count == 0: the holder stays null).  numGroupsLimit binds at any key-space size, the
 *    first keys in docId order surviving.  Raw (no-dictionary) key columns: PG_ERR_UNSUPPORTED at plan time. */
#define PG_QUERY_NULL_HANDLING 1
/* The caller does not need ExecutionStatistics.numEntriesScannedInFilter (core/operator/ExecutionStatistics.java:25-64) to be the
 * reference's exact count when that costs work beyond the query's own kernel.  For a filter whose iterators leap-frog (a root AND over
 * scan-based leaves, OR / NOT children: AndDocIdIterator / OrDocIdIterator / NotDocIdIterator over SVScanDocIdIterator) the exact count
 * is a walk of its own -- a transducer pass on the device, a chain kernel behind the scan, or a replay on the host (DESIGN.md section 5:
 * up to several times the query for a NOT child).  With this flag such a filter runs nothing but the query: the statistic is the estimate
 * numDocs x scan leaves (an upper bound for AND / OR trees; a NOT child, whose leaf is pulled in 256-doc batches, can pass it) and
 * pg_result.filter_entries_exact = 0.  Everything else of the result is unchanged, and filters whose count
 * is a closed form or falls out of the kernel (no scan leaf, no AND above one, scan leaves behind index-based children) stay exact. */
#define PG_QUERY_STATS_UPPER_BOUND_OK 2

/* Intermediate result of one aggregation function, in the reference's holder types:
 * SUM/MIN/MAX -> Double, COUNT -> Long, AVG -> AvgPair(sum, count). */
typedef struct pg_agg_value {
  int64_t count;               /* COUNT result / AVG count / number of aggregated docs */
  double sum;                  /* SUM / AVG sum as the reference's double holder value */
  int64_t sum_i64;             /* exact integer sum for INT/LONG sources (valid when sum_exact != 0) */
  int32_t sum_exact;
  int32_t reserved;
  double min;                  /* +inf when no doc matched (MinAggregationFunction.DEFAULT_VALUE) */
  double max;                  /* -inf when no doc matched */
} pg_agg_value;

/* ExecutionStatistics (core/operator/ExecutionStatistics.java:25-64). */
/* Which kernel executed the scan (pg_result.dominant_kernel); names as rocprofv3 prints them. */
typedef enum pg_kernel_id {
  PG_KERNEL_SCAN_AGG = 0,          /* scan_agg_kernel: LDS-staged scan -> filter -> aggregate */
  PG_KERNEL_SCAN_PRIVATE = 1,      /* scan_private_kernel: lane-private decode straight from HBM */
  PG_KERNEL_SCAN_GROUP = 2,        /* scan_group_kernel: LDS-staged group-by */
  PG_KERNEL_GROUP_PRIVATE = 3,     /* group_private_kernel: lane-private group-by */
  PG_KERNEL_GROUP_PARTITION = 4,   /* group_partition_scatter_kernel (+ histogram / aggregate): key spaces above the LDS table */
  PG_KERNEL_SCAN_PRIVATE_TYPED = 5,/* scan_private_typed_kernel: lane-private scan, raw / 8-byte aggregated columns */
  PG_KERNEL_INDEX_AND = 7,         /* index_and_kernel: COUNT(*) over a filter the inverted indexes answer (FastFilteredCountOperator), or an
                                    * index-led aggregation whose index phase (index_and_kernel + its tile-list pass) outlasts the scan of the listed tiles */
  PG_KERNEL_SCAN_NARROW = 8,       /* scan_narrow_kernel: COUNT(*) / bitmap of a filter over dictionary columns of at most 8 bits, four tiles per wave */
  PG_KERNEL_SCAN_SPARSE = 9,       /* scan_sparse_kernel: aggregation of the docs a sparse docId bitmap names (index-led filters), eight tiles per wave */
  PG_KERNEL_SCAN_RAW = 11,         /* scan_raw_kernel: one raw INT range leaf + at most one aggregated raw INT column, five waves per SIMD, coalesced reads */
  PG_KERNEL_SCAN_SIMPLE = 10,      /* scan_simple_kernel: one dictionary-range leaf + at most one aggregated packed column, twice the waves per SIMD */
  PG_KERNEL_SCAN_HIST = 6          /* scan_hist_kernel: lane-private scan, SUM = sum_d matches[d] * dictionary[d] through an LDS histogram */
} pg_kernel_id;

typedef struct pg_stats {
  int64_t num_docs_scanned;
  int64_t num_entries_scanned_in_filter;
  int64_t num_entries_scanned_post_filter;
  int64_t num_total_docs;
} pg_stats;

typedef struct pg_result {
  pg_stats stats;
  int32_t num_aggregations;
  int32_t num_groups;              /* group-by: number of groups present (ArrayBasedHolder flags set) */
  pg_agg_value* aggregations;      /* [num_aggregations], aggregation-only queries */
  int32_t* group_ids;              /* [num_groups] ascending raw group ids (DictionaryBasedGroupKeyGenerator.java:306-324) */
  pg_agg_value* group_aggregations;/* [num_groups * num_aggregations], row-major by group */
  int32_t group_id_upper_bound;    /* product of group-by cardinalities */
  int32_t num_groups_limit_reached;/* GroupByOperator.java:114-115: the query created >= numGroupsLimit groups (later keys were dropped) */
  double device_ms;                /* HIP-event time of this query's kernels (PG_CFG_TIME_KERNELS), else 0 */
  double dominant_kernel_ms;       /* HIP-event time of the scan kernel alone */
  uint64_t profile_cycles[4];      /* PG_CFG_PROFILE_WAVES: shader cycles summed over wavefronts: memory wait, filter, aggregate, total */
  int32_t profile_waves;           /* number of wavefronts the sums cover */
  int32_t dominant_kernel;         /* pg_kernel_id of the kernel dominant_kernel_ms refers to */
  int32_t filter_entries_exact;    /* stats.num_entries_scanned_in_filter is the reference's count (AndDocIdSet / SVScanDocIdIterator accounting);
                                    * 0: an upper bound (numDocs per scan leaf) -- filters whose iterators leap-frog in a shape the device does
                                    * not count (it counts root ANDs of scan / index leaves, ORs of leaves, NOTs over a leaf or over an OR of leaves -- up to three scan leaves
                                    * under NOTs in all -- at any size, while the machine minimises to at most 16 states over 8 leaves: two NOTs beside a third child and
                                    * `a AND NOT (b OR c)` do), on segments above PINOT_GPU_EXACT_FILTER_STATS_DOCS
                                    * docs, and enableNullHandling queries */
  int32_t group_key_kind;          /* which of the reference's RawKeyHolders the key space calls for (DictionaryBasedGroupKeyGenerator.java:150-184):
                                    * 0 the raw key is an int (Array / IntMapBasedHolder): group_ids hold it;
                                    * 1 it is a long (LongMapBasedHolder, :628-700): group_ids64 hold it, group_ids are row numbers;
                                    * 2 it is beyond a long (ArrayMapBasedHolder, :808+): only the dictId tuples identify a group.
                                    * Kinds 1 and 2: group_id_upper_bound = numGroupsLimit (:150-163), groups admitted in order of first appearance. */
  void* internal;
  int64_t* group_ids64;            /* [num_groups] kind 1: raw key = sum dictId_j * prod_{k<j} cardinality_k */
  int32_t* group_key_dict_ids;     /* [num_groups * num_group_by] every kind: the dictIds of each group's key, group-by column order (under
                                    * PG_QUERY_NULL_HANDLING the digit of a nullable key runs to cardinality inclusive = NULL).  Rows come
                                    * in ascending raw-key order (the last group-by column is the most significant digit). */
} pg_result;

pg_status pg_init(const pg_config* config);
pg_status pg_shutdown(void);
const char* pg_last_error(void);
const char* pg_version(void);
/* Fills name (e.g. "gfx950"), CU count, HBM bytes of the configured device. */
pg_status pg_device_info(int32_t device_id, char* arch_name, int32_t arch_name_len, int32_t* num_cus,
                         uint64_t* hbm_bytes);

/* Device ids the library accepts (pg_config.device_id, pg_segment_desc.device_id): 0 .. *out_devices - 1.  Normally the number of HIP
 * devices.  With PINOT_GPU_ALIAS_DEVICES=N in the environment at pg_init (N above the physical count) N ids are accepted and id d runs
 * on HIP device d mod *out_physical, each id with its own batch contexts / streams / placement: a one-GPU box then executes the
 * multi-device paths of pg_execute_batch (what a server's `gpu.devices` list drives). */
pg_status pg_device_count(int32_t* out_devices, int32_t* out_physical);

/* Diagnostic: the box's empirical HBM read ceiling -- a pure 16 B/lane read-reduce kernel over `bytes` of device memory, best of
 * `launches` timed launches, in GB/s (BASELINE.md section 2 asks for it next to the 8 TB/s vendor figure).  Not on the query path. */
pg_status pg_measure_stream_read(int32_t device_id, uint64_t bytes, int32_t launches, double* out_gbps);

pg_status pg_segment_open(const pg_segment_desc* desc, pg_segment** out_segment);
pg_status pg_segment_close(pg_segment* segment);
pg_status pg_segment_num_docs(const pg_segment* segment, int32_t* out_num_docs);
pg_status pg_segment_device_bytes(const pg_segment* segment, uint64_t* out_bytes);   /* index buffers + value planes */
/* The part of pg_segment_device_bytes held by materialised value planes: plane[doc] = dictionary[dictId[doc]] re-packed, built on the
 * device -- beside the queries, on a stream of its own -- the first time a column with an irregular dictionary is summed, so that
 * SUM streams values instead of gathering them.  A query never waits for a build (until the plane is there it runs the dictionary
 * path, same result) and never fails for lack of one.  pg_set_plane_budget changes the process-wide budget at run time. */
pg_status pg_segment_plane_bytes(const pg_segment* segment, uint64_t* out_bytes);
pg_status pg_set_plane_budget(uint64_t budget_bytes, uint64_t* out_previous);

/* Plan-time eligibility: PG_OK when pg_execute would run this query on this segment, PG_ERR_UNSUPPORTED (pg_last_error says why) when
 * it would decline it -- more filter leaves / nodes / column streams / aggregations than the kernels take, key spaces beyond the
 * direct-indexed table, raw 8-byte aggregations or nullable columns under GROUP BY ... -- decided from the query and the segment's
 * metadata alone: nothing is allocated or launched.  pg_execute makes the same call first, so the two never disagree.  This is what
 * InstancePlanMakerImplV2.makeSegmentPlanNode (core/plan/maker/InstancePlanMakerImplV2.java:270-289) asks before it swaps the
 * operator: on anything but PG_OK the caller keeps the CPU plan. */
pg_status pg_query_check(const pg_segment* segment, const pg_query* query);

pg_status pg_execute(pg_segment* segment, const pg_query* query, pg_result* out_result);
void pg_result_free(pg_result* result);

/* How a group-by column's entries of pg_result.group_key_dict_ids turn into key values.  A dictionary column: *out_is_offset = 0, the
 * entry is a dictId (GroupKeyGenerator.getGroupKeys looks it up, DictionaryBasedGroupKeyGenerator.java:260-290).  A raw (no-dictionary)
 * INT / LONG column -- the reference groups it by VALUE with NoDictionarySingleColumnGroupKeyGenerator / NoDictionaryMultiColumnGroupKey
 * Generator (core/query/aggregation/groupby/DefaultGroupByExecutor.java:106-121) --: *out_is_offset = 1 and the key value is
 * *out_base + entry (the column is grouped by through the stream of value - min; *out_base = the column's smallest value).
 * *out_null_entry: the entry that means NULL under PG_QUERY_NULL_HANDLING -- `cardinality` of a dictionary column, max - min + 1 of a raw
 * one (a column without a null value vector never produces it).
 * A raw FLOAT / DOUBLE column, or a raw INT / LONG column whose values span more than an int (NoDictionarySingleColumnGroupKeyGenerator.java:
 * 100-135 keys all four stored types by value): *out_is_offset = 2 -- the column is grouped by through a dictionary of its own distinct
 * values that the device builds the first time a query groups by it; the entry is the value's RANK among them, pg_group_key_values
 * returns the values.  (Such a column is not a key under PG_QUERY_NULL_HANDLING: pg_query_check declines.) */
pg_status pg_group_key_info(const pg_segment* segment, int32_t column, int64_t* out_base, int32_t* out_is_offset, int32_t* out_null_entry);
/* The distinct values of a raw group-by column whose pg_group_key_info says *out_is_offset = 2, in ascending order (Double.compare's
 * order for FLOAT / DOUBLE: -0.0 below 0.0, one NaN above +Infinity): out_value_bits[rank] = the long value (INT / LONG columns) or the
 * IEEE-754 bits of the double (FLOAT values widened exactly, DOUBLE).  out_value_bits == NULL: only *out_count is set (sizing call).
 * Builds the column's dictionary when no query has yet (one sort + unique pass over the resident column). */
pg_status pg_group_key_values(pg_segment* segment, int32_t column, int64_t* out_value_bits, int32_t capacity, int32_t* out_count);

/* One query over MANY resident segments in one call: what BaseCombineOperator does with a thread pool (core/operator/combine/
 * BaseCombineOperator.java:85-142: numTasks worker threads, each pulling the next segment's operator and merging its block; CombinePlanNode.java:
 * 92-110 builds one PlanNode per segment).  queries[i] is the query as lowered FOR segments[i] (dictIds differ from segment to segment);
 * results[i] / statuses[i] are what pg_execute(segments[i], queries[i], &results[i]) would have produced, item by item -- an item that
 * fails does not stop the others (the combine operator collects the exception of one segment and still answers with the rest).
 *
 * A server holds hundreds of few-million-row segments per table; launched one by one each is a ~20 us kernel whose launch latency, ramp
 * and drain are most of its device time.  Items whose whole device work is one launch of the lane-private scan kernel (aggregations
 * without GROUP BY over scan / sorted leaves) are therefore put into ONE launch (scan_private_batch_kernel: every item owns a share of the
 * grid, folds its own records and publishes its own pinned result); every other item runs as a pg_execute of its own, concurrently, on
 * the library's worker threads and on streams of their own.  The call returns when every item has finished.  Returns PG_OK whenever the
 * arguments were usable; per-item outcomes are in statuses[], pg_last_error() names the first failed item.  Every results[i] must be
 * released with pg_result_free (a failed item's is already empty). */
pg_status pg_execute_batch(pg_segment* const* segments, const pg_query* const* queries, int32_t count, pg_result* results, pg_status* statuses);

/* Evaluates only the filter and returns the matching docIds as a dense bitmap:
 * bit (docId & 63) of out_words[docId >> 6]; num_words >= ceil(num_docs / 64).  out_cardinality may be NULL. */
pg_status pg_filter_bitmap(pg_segment* segment, const pg_query* query, uint64_t* out_words, int64_t num_words,
                           int64_t* out_cardinality);

/* BlockValSet-level SPI (ProjectionOperatorUtils hook): arbitrary ascending docIds, length <= any. */
pg_status pg_read_dict_ids(pg_segment* segment, int32_t column, const int32_t* doc_ids, int32_t length,
                           int32_t* out_dict_ids);
pg_status pg_read_int_values(pg_segment* segment, int32_t column, const int32_t* doc_ids, int32_t length,
                             int32_t* out_values);
pg_status pg_read_double_values(pg_segment* segment, int32_t column, const int32_t* doc_ids, int32_t length,
                                double* out_values);
/* BlockValSet.getLongValuesSV (DataFetcher.fetchLongValues, core/common/DataFetcher.java:121-123): INT / LONG columns return
 * the value, FLOAT / DOUBLE the Java (long) cast.  pg_read_int_values is defined for INT columns only. */
pg_status pg_read_long_values(pg_segment* segment, int32_t column, const int32_t* doc_ids, int32_t length,
                              int64_t* out_values);

#ifdef __cplusplus
}
#endif
#endif /* PINOT_GPU_H */
