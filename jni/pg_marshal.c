/* See pg_marshal.h.  Plain C, no JNI types: built into libpinot_gpu_marshal.so here and into the JNI library on a box with a JDK. */
#include "pg_marshal.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static _Thread_local char g_error[256];

static void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

const char* pgm_last_error(void) { return g_error; }

struct pgm_query {
  pg_query query;
  pg_filter_node* nodes;
  pg_predicate* predicates;
  uint32_t* set_words;
  pg_aggregation* aggregations;
  int32_t* group_by;
};

void pgm_query_free(pgm_query* q) {
  if (!q) return;
  free(q->nodes); free(q->predicates); free(q->set_words); free(q->aggregations); free(q->group_by);
  free(q);
}

const pg_query* pgm_query_get(const pgm_query* q) { return q ? &q->query : NULL; }

static void* copy_of(const void* src, size_t count, size_t size) {
  void* p = calloc(count ? count : 1, size);
  if (p && src && count) memcpy(p, src, count * size);
  return p;
}

pgm_query* pgm_query_build(const int32_t* filter_nodes, int32_t num_nodes, const int32_t* pred_ints, const int64_t* pred_longs,
                           int32_t num_preds, const int32_t* set_offsets, const uint32_t* set_words, int32_t num_set_words,
                           const int32_t* aggregations, int32_t num_aggs, const int32_t* group_by, int32_t num_group_by,
                           int32_t num_groups_limit, int32_t flags) {
  if (num_nodes < 0 || num_preds < 0 || num_set_words < 0 || num_aggs < 0 || num_group_by < 0) { set_error("negative array length"); return NULL; }
  if ((num_nodes && !filter_nodes) || (num_preds && (!pred_ints || !pred_longs || !set_offsets)) || (num_set_words && !set_words) ||
      (num_aggs && !aggregations) || (num_group_by && !group_by)) { set_error("null array"); return NULL; }
  for (int32_t p = 0; p < num_preds; p++) {
    if (set_offsets[p] < 0 || set_offsets[p + 1] < set_offsets[p] || set_offsets[p + 1] > num_set_words) { set_error("dictId-set offsets of predicate %d leave the word array", p); return NULL; }
  }
  pgm_query* q = (pgm_query*)calloc(1, sizeof(pgm_query));
  if (!q) { set_error("out of memory"); return NULL; }
  q->nodes = (pg_filter_node*)calloc(num_nodes ? (size_t)num_nodes : 1, sizeof(pg_filter_node));
  q->predicates = (pg_predicate*)calloc(num_preds ? (size_t)num_preds : 1, sizeof(pg_predicate));
  q->set_words = (uint32_t*)copy_of(set_words, (size_t)num_set_words, sizeof(uint32_t));
  q->aggregations = (pg_aggregation*)calloc(num_aggs ? (size_t)num_aggs : 1, sizeof(pg_aggregation));
  q->group_by = (int32_t*)copy_of(group_by, (size_t)num_group_by, sizeof(int32_t));
  if (!q->nodes || !q->predicates || !q->set_words || !q->aggregations || !q->group_by) { pgm_query_free(q); set_error("out of memory"); return NULL; }
  for (int32_t n = 0; n < num_nodes; n++) {
    q->nodes[n].op = filter_nodes[PGM_FILTER_NODE_INTS * n];
    q->nodes[n].predicate = filter_nodes[PGM_FILTER_NODE_INTS * n + 1];
    q->nodes[n].num_children = filter_nodes[PGM_FILTER_NODE_INTS * n + 2];
  }
  for (int32_t p = 0; p < num_preds; p++) {
    pg_predicate* d = &q->predicates[p];
    d->kind = pred_ints[PGM_PRED_INTS * p];
    d->column = pred_ints[PGM_PRED_INTS * p + 1];
    d->eval = pred_ints[PGM_PRED_INTS * p + 2];
    d->exclusive = pred_ints[PGM_PRED_INTS * p + 3];
    d->lo = pred_longs[PGM_PRED_LONGS * p];
    d->hi = pred_longs[PGM_PRED_LONGS * p + 1];
    d->num_set_words = set_offsets[p + 1] - set_offsets[p];
    d->set_words = d->num_set_words ? q->set_words + set_offsets[p] : NULL;
  }
  for (int32_t a = 0; a < num_aggs; a++) { q->aggregations[a].function = aggregations[PGM_AGG_INTS * a]; q->aggregations[a].column = aggregations[PGM_AGG_INTS * a + 1]; }
  q->query.filter = num_nodes ? q->nodes : NULL;
  q->query.num_filter_nodes = num_nodes;
  q->query.predicates = num_preds ? q->predicates : NULL;
  q->query.num_predicates = num_preds;
  q->query.aggregations = num_aggs ? q->aggregations : NULL;
  q->query.num_aggregations = num_aggs;
  q->query.group_by_columns = num_group_by ? q->group_by : NULL;
  q->query.num_group_by = num_group_by;
  q->query.num_groups_limit = num_groups_limit;
  q->query.flags = flags;
  return q;
}

struct pgm_segment {
  pg_segment_desc desc;
  pg_column_desc* columns;
  char* name;
  char** column_names;
  int32_t num_columns;
};

void pgm_segment_free(pgm_segment* s) {
  if (!s) return;
  if (s->column_names) for (int32_t c = 0; c < s->num_columns; c++) free(s->column_names[c]);
  free(s->column_names); free(s->columns); free(s->name);
  free(s);
}

const pg_segment_desc* pgm_segment_get(const pgm_segment* s) { return s ? &s->desc : NULL; }

static char* dup_string(const char* s) {
  size_t n = strlen(s ? s : "") + 1;
  char* p = (char*)malloc(n);
  if (p) memcpy(p, s ? s : "", n);
  return p;
}

pgm_segment* pgm_segment_build(const char* name, int64_t crc, int32_t device_id, int32_t num_docs, int32_t num_columns, const char* const* names,
                               const int32_t* col_ints, const int64_t* col_buffers) {
  if (num_docs < 0 || num_columns < 0 || (num_columns && (!names || !col_ints || !col_buffers))) { set_error("bad segment arguments"); return NULL; }
  pgm_segment* s = (pgm_segment*)calloc(1, sizeof(pgm_segment));
  if (!s) { set_error("out of memory"); return NULL; }
  s->num_columns = num_columns;
  s->name = dup_string(name);
  s->columns = (pg_column_desc*)calloc(num_columns ? (size_t)num_columns : 1, sizeof(pg_column_desc));
  s->column_names = (char**)calloc(num_columns ? (size_t)num_columns : 1, sizeof(char*));
  if (!s->name || !s->columns || !s->column_names) { pgm_segment_free(s); set_error("out of memory"); return NULL; }
  for (int32_t c = 0; c < num_columns; c++) {
    s->column_names[c] = dup_string(names[c]);
    if (!s->column_names[c]) { pgm_segment_free(s); set_error("out of memory"); return NULL; }
    pg_column_desc* d = &s->columns[c];
    const int32_t* ci = col_ints + PGM_COLUMN_INTS * (size_t)c;
    const int64_t* cb = col_buffers + PGM_COLUMN_BUFFERS * (size_t)c;
    d->name = s->column_names[c];
    d->stored_type = ci[0];
    d->fwd_encoding = ci[1];
    d->bits_per_value = ci[2];
    d->cardinality = ci[3];
    d->fwd_data = (const void*)(intptr_t)cb[0];  d->fwd_size = (uint64_t)cb[1];
    d->dict_data = (const void*)(intptr_t)cb[2]; d->dict_size = (uint64_t)cb[3];
    d->inv_data = (const void*)(intptr_t)cb[4];  d->inv_size = (uint64_t)cb[5];
    d->null_data = (const void*)(intptr_t)cb[6]; d->null_size = (uint64_t)cb[7];
  }
  s->desc.name = s->name;
  s->desc.crc = (uint64_t)crc;
  s->desc.device_id = device_id;
  s->desc.num_docs = num_docs;
  s->desc.num_columns = num_columns;
  s->desc.columns = s->columns;
  return s;
}

int64_t pgm_result_rows(const pg_result* r, int32_t is_group_by) { return !r ? 0 : (is_group_by ? (int64_t)r->num_groups : 1); }

void pgm_result_header(const pg_result* r, int32_t is_group_by, int64_t* h) {
  memset(h, 0, sizeof(int64_t) * PGM_HEADER_LEN);
  if (!r) return;
  h[PGM_H_NUM_DOCS_SCANNED] = r->stats.num_docs_scanned;
  h[PGM_H_ENTRIES_IN_FILTER] = r->stats.num_entries_scanned_in_filter;
  h[PGM_H_ENTRIES_POST_FILTER] = r->stats.num_entries_scanned_post_filter;
  h[PGM_H_TOTAL_DOCS] = r->stats.num_total_docs;
  h[PGM_H_FILTER_ENTRIES_EXACT] = r->filter_entries_exact;
  h[PGM_H_NUM_AGGREGATIONS] = r->num_aggregations;
  h[PGM_H_NUM_GROUPS] = r->num_groups;
  h[PGM_H_GROUP_ID_UPPER_BOUND] = r->group_id_upper_bound;
  h[PGM_H_NUM_GROUPS_LIMIT_REACHED] = r->num_groups_limit_reached;
  h[PGM_H_DOMINANT_KERNEL] = r->dominant_kernel;
  h[PGM_H_IS_GROUP_BY] = is_group_by ? 1 : 0;
  h[PGM_H_GROUP_KEY_KIND] = r->group_key_kind;
}

int64_t pgm_result_fill(const pg_result* r, int32_t is_group_by, int32_t* group_ids, int64_t* counts, double* sums, int64_t* sums_i64,
                        int32_t* sum_exact, double* mins, double* maxs) {
  if (!r) return 0;
  const int64_t rows = pgm_result_rows(r, is_group_by);
  const int32_t na = r->num_aggregations;
  const pg_agg_value* values = is_group_by ? r->group_aggregations : r->aggregations;
  for (int64_t row = 0; row < rows; row++) {
    if (is_group_by && group_ids) group_ids[row] = r->group_ids[row];
    for (int32_t a = 0; a < na; a++) {
      const pg_agg_value* v = &values[row * na + a];
      const int64_t at = row * na + a;
      if (counts) counts[at] = v->count;
      if (sums) sums[at] = v->sum;
      if (sums_i64) sums_i64[at] = v->sum_i64;
      if (sum_exact) sum_exact[at] = v->sum_exact;
      if (mins) mins[at] = v->min;
      if (maxs) maxs[at] = v->max;
    }
  }
  return rows;
}

int64_t pgm_result_fill_keys(const pg_result* r, int32_t num_group_by, int32_t* group_keys) {
  if (!r || !group_keys || num_group_by <= 0 || !r->group_key_dict_ids) return 0;
  const int64_t cells = (int64_t)r->num_groups * (int64_t)num_group_by;
  memcpy(group_keys, r->group_key_dict_ids, sizeof(int32_t) * (size_t)cells);
  return cells;
}
