// numEntriesScannedInFilter (ExecutionStatistics, core/operator/ExecutionStatistics.java:25-64) for filters whose count is not a
// closed form of the device's own counters.
//
// The reference counts, per scan-based leaf, the docs whose value the leaf's matcher looks at (SVScanDocIdIterator._numEntriesScanned);
// which docs those are follows from how the BlockDocIdIterator tree above the leaf drives it:
//   * no AND anywhere above a scan leaf: the leaf is iterated with next() to the end -> numDocs entries
//   * root AND of leaves with at least one index-based child: the scan leaves are and-ed, in list order, into the docIds the children
//     before them left (ScanBasedDocIdIterator.applyAnd, AndDocIdSet.java:127-165) -> the lane-private kernels count that themselves
//     (kNodeCountEntries)
//   * everything else leap-frogs (AndDocIdIterator.java:41-74 calling advance() on scan leaves, OR / NOT iterators in between): the
//     count is a property of the whole iterator tree walking the docId space in order.  This file replays that walk over the leaves'
//     match bitmaps (which the device produces): DocIdSetTree mirrors getTrues / getFalses, the Iterator classes mirror the
//     reference's iterators.  It is a sequential walk -- one step per advance / next call -- so the engine only takes it for segments
//     up to PINOT_GPU_EXACT_FILTER_STATS_DOCS docs and reports an upper bound (pg_result.filter_entries_exact = 0) beyond.
// Host-side C++ only (no device code): pg_engine.hip includes it.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/pinot_gpu.h"

namespace pg {
namespace fstats {

constexpr int32_t kEof = INT32_MAX;          // Constants.EOF (any value no docId takes)
constexpr int kScanBatch = 256;              // BlockDocIdIterator.OPTIMAL_ITERATOR_BATCH_SIZE

using Words = std::shared_ptr<std::vector<uint64_t>>;

inline int32_t next_set(const std::vector<uint64_t>& w, int32_t num_docs, int64_t from) {
  if (from >= num_docs) return kEof;
  size_t i = (size_t)(from >> 6);
  uint64_t cur = w[i] & (~0ull << (from & 63));
  while (cur == 0) { if (++i >= w.size()) return kEof; cur = w[i]; }
  const int64_t d = (int64_t)i * 64 + __builtin_ctzll(cur);
  return d < num_docs ? (int32_t)d : kEof;
}

struct Iterator {
  enum class Type { kOther, kSorted, kBitmap, kScan };
  virtual ~Iterator() = default;
  virtual int32_t next() = 0;
  virtual int32_t advance(int32_t target) = 0;
  virtual Type type() const { return Type::kOther; }
  virtual const Words* docs() const { return nullptr; }       // index-based iterators: their docId set
};
using IteratorPtr = std::unique_ptr<Iterator>;

struct EmptyIterator : Iterator {
  int32_t next() override { return kEof; }
  int32_t advance(int32_t) override { return kEof; }
};

struct MatchAllIterator : Iterator {                           // dociditerators/MatchAllDocIdIterator.java:32-52
  int32_t num_docs, next_doc = 0;
  explicit MatchAllIterator(int32_t n) : num_docs(n) {}
  int32_t next() override { return next_doc < num_docs ? next_doc++ : kEof; }
  int32_t advance(int32_t t) override { next_doc = t; return next(); }
};

// SortedDocIdIterator, BitmapDocIdIterator and RangelessBitmapDocIdIterator: a cursor over a docId set (advanceIfNeeded + next)
struct SetIterator : Iterator {
  Words words; int32_t num_docs; int64_t pos = 0; Type kind;
  SetIterator(Words w, int32_t n, Type k) : words(std::move(w)), num_docs(n), kind(k) {}
  int32_t next() override {
    const int32_t d = next_set(*words, num_docs, pos);
    pos = d == kEof ? (int64_t)num_docs : (int64_t)d + 1;
    return d;
  }
  int32_t advance(int32_t t) override { if (t > pos) pos = t; return next(); }
  Type type() const override { return kind; }
  const Words* docs() const override { return &words; }
};

// SVScanDocIdIterator (dociditerators/SVScanDocIdIterator.java:76-145) over the matcher's answers
struct ScanIterator : Iterator {
  Words matches; int32_t num_docs; int64_t* entries;
  int32_t next_doc = 0, first_mismatch = 0, cursor = 0;
  int32_t batch[kScanBatch];
  ScanIterator(Words m, int32_t n, int64_t* e) : matches(std::move(m)), num_docs(n), entries(e) {}
  bool match(int32_t d) const { return (((*matches)[(size_t)d >> 6] >> (d & 63)) & 1ull) != 0; }
  int32_t next() override {
    if (cursor >= first_mismatch) {
      int32_t limit, found = 0;
      do {
        limit = std::min(num_docs - next_doc, kScanBatch);
        if (limit > 0) {
          found = 0;
          for (int32_t i = 0; i < limit; ++i) if (match(next_doc + i)) batch[found++] = next_doc + i;
          next_doc += limit;
          *entries += limit;
        }
      } while (limit > 0 && found == 0);
      first_mismatch = found;
      cursor = 0;
      if (found == 0) return kEof;
    }
    return batch[cursor++];
  }
  // SVScanDocIdIterator.advance (:91-106) looks at doc t, t + 1, ... until one matches, one entry each: here the next set bit of the
  // leaf's bitmap is found a word at a time and the docs in between are charged in one addition.
  int32_t advance(int32_t t) override {
    next_doc = t;
    first_mismatch = 0;
    if (t >= num_docs) return kEof;
    const size_t nw = ((size_t)num_docs + 63) / 64;
    size_t w = (size_t)t >> 6;
    uint64_t bits = (*matches)[w] & (~0ull << (t & 63));
    while (bits == 0ull && ++w < nw) bits = (*matches)[w];
    const int64_t d = bits ? (int64_t)(w << 6) + __builtin_ctzll(bits) : (int64_t)num_docs;
    if (d >= num_docs) { *entries += num_docs - t; next_doc = num_docs; return kEof; }
    *entries += d - t + 1;
    next_doc = (int32_t)d + 1;
    return (int32_t)d;
  }
  Type type() const override { return Type::kScan; }
};

struct AndIterator : Iterator {                                // dociditerators/AndDocIdIterator.java:41-80
  std::vector<IteratorPtr> children; int32_t next_doc = 0;
  int32_t next() override {
    int32_t max_doc = next_doc;
    int max_index = -1, index = 0;
    const int n = (int)children.size();
    while (index < n) {
      if (index == max_index) { ++index; continue; }
      const int32_t d = children[(size_t)index]->advance(max_doc);
      if (d == kEof) return kEof;
      if (d == max_doc) ++index;
      else { max_doc = d; max_index = index; index = 0; }
    }
    next_doc = max_doc;
    return next_doc++;
  }
  int32_t advance(int32_t t) override { next_doc = t; return next(); }
};

struct OrIterator : Iterator {                                 // dociditerators/OrDocIdIterator.java:52-140
  std::vector<IteratorPtr> children; std::vector<int32_t> next_ids; int live = 0; int32_t previous = -1;
  void start() { next_ids.assign(children.size(), -1); live = (int)children.size(); }
  void drop_exhausted() {
    int i = 0;
    while (i < live) {
      if (next_ids[(size_t)i] == kEof) { --live; std::swap(children[(size_t)i], children[(size_t)live]); std::swap(next_ids[(size_t)i], next_ids[(size_t)live]); }
      else ++i;
    }
  }
  template <typename Step> int32_t step(Step&& pull) {
    int32_t best = kEof;
    bool exhausted = false;
    for (int i = 0; i < live; ++i) {
      int32_t d = next_ids[(size_t)i];
      if (pull(i, &d)) { next_ids[(size_t)i] = d; if (d == kEof) { exhausted = true; continue; } }
      best = std::min(best, d);
    }
    if (exhausted) drop_exhausted();
    if (best != kEof) previous = best;
    return best;
  }
  int32_t next() override {
    return step([&](int i, int32_t* d) { if (*d != previous) return false; *d = children[(size_t)i]->next(); return true; });
  }
  int32_t advance(int32_t t) override {
    return step([&](int i, int32_t* d) { if (*d >= t) return false; *d = children[(size_t)i]->advance(t); return true; });
  }
};

struct NotIterator : Iterator {                                // dociditerators/NotDocIdIterator.java:36-70
  IteratorPtr child; int32_t num_docs, next_doc = 0, next_non_matching;
  NotIterator(IteratorPtr c, int32_t n) : child(std::move(c)), num_docs(n) {
    const int32_t d = child->next();
    next_non_matching = d == kEof ? num_docs : d;
  }
  int32_t next() override {
    if (next_doc >= num_docs) return kEof;
    while (next_doc == next_non_matching) {
      ++next_doc;
      const int32_t d = child->next();
      next_non_matching = d == kEof ? num_docs : d;
    }
    return next_doc >= num_docs ? kEof : next_doc++;
  }
  int32_t advance(int32_t t) override {
    next_doc = t;
    if (t > next_non_matching) {
      const int32_t d = child->advance(t);
      next_non_matching = d == kEof ? num_docs : d;
    }
    return next();
  }
};

// ---- BlockDocIdSets: what getTrues / getFalses hand to DocIdSetOperator before any iterator exists ----
struct DocIdSet {
  enum class Kind { kEmpty, kMatchAll, kScan, kSorted, kBitmap, kAnd, kOr, kNot } kind = Kind::kEmpty;
  Words words;                                  // leaves: the docs that match
  std::vector<DocIdSet> children;
};

inline bool is_index(const Iterator& it) { return it.type() == Iterator::Type::kSorted || it.type() == Iterator::Type::kBitmap; }

IteratorPtr make_iterator(const DocIdSet& s, int32_t num_docs, int64_t* entries);

// AndDocIdSet.iterator(), docidsets/AndDocIdSet.java:73-172 (andScanReordering off)
inline IteratorPtr make_and_iterator(const DocIdSet& s, int32_t num_docs, int64_t* entries) {
  std::vector<IteratorPtr> all;
  int num_index = 0, num_scan = 0;
  for (const DocIdSet& c : s.children) {
    all.push_back(make_iterator(c, num_docs, entries));
    num_index += is_index(*all.back()) ? 1 : 0;
    num_scan += all.back()->type() == Iterator::Type::kScan ? 1 : 0;
  }
  if (!((num_index > 0 && num_scan > 0) || num_index > 1)) {
    auto out = std::make_unique<AndIterator>();
    out->children = std::move(all);
    return out;
  }
  const size_t nw = ((size_t)num_docs + 63) / 64;
  auto docs = std::make_shared<std::vector<uint64_t>>(std::max<size_t>(nw, 1), ~0ull);
  if (num_docs & 63) (*docs)[nw - 1] = (1ull << (num_docs & 63)) - 1ull;
  if (nw == 0) (*docs)[0] = 0;
  for (const auto& it : all) if (is_index(*it)) for (size_t w = 0; w < nw; ++w) (*docs)[w] &= (**it->docs())[w];
  for (const auto& it : all) {
    if (it->type() != Iterator::Type::kScan) continue;
    const ScanIterator& scan = static_cast<const ScanIterator&>(*it);
    int64_t standing = 0;
    for (size_t w = 0; w < nw; ++w) standing += __builtin_popcountll((*docs)[w]);
    *entries += standing;                         // applyAnd: one entry per docId of the bitmap (none when it is empty)
    for (size_t w = 0; w < nw; ++w) (*docs)[w] &= (*scan.matches)[w];
  }
  IteratorPtr merged = std::make_unique<SetIterator>(docs, num_docs, Iterator::Type::kBitmap);      // RangelessBitmapDocIdIterator
  std::vector<IteratorPtr> remaining;
  for (auto& it : all) if (!is_index(*it) && it->type() != Iterator::Type::kScan) remaining.push_back(std::move(it));
  if (remaining.empty()) return merged;
  auto out = std::make_unique<AndIterator>();
  out->children.push_back(std::move(merged));
  for (auto& it : remaining) out->children.push_back(std::move(it));
  return out;
}

inline IteratorPtr make_iterator(const DocIdSet& s, int32_t num_docs, int64_t* entries) {
  switch (s.kind) {
    case DocIdSet::Kind::kEmpty: return std::make_unique<EmptyIterator>();
    case DocIdSet::Kind::kMatchAll: return std::make_unique<MatchAllIterator>(num_docs);
    case DocIdSet::Kind::kScan: return std::make_unique<ScanIterator>(s.words, num_docs, entries);
    case DocIdSet::Kind::kSorted: return std::make_unique<SetIterator>(s.words, num_docs, Iterator::Type::kSorted);
    case DocIdSet::Kind::kBitmap: return std::make_unique<SetIterator>(s.words, num_docs, Iterator::Type::kBitmap);
    case DocIdSet::Kind::kAnd: return make_and_iterator(s, num_docs, entries);
    case DocIdSet::Kind::kOr: {
      // OrDocIdSet.iterator(), docidsets/OrDocIdSet.java:62-126: this fork never fills its list of bitmap-based children (:80-82), so
      // only two or more SORTED children are merged (into one BitmapDocIdIterator that leads the OrDocIdIterator); the bitmap
      // children are merged along with them here, which keeps the docId set right where the reference would lose them.
      std::vector<IteratorPtr> all;
      int num_sorted = 0;
      for (const DocIdSet& c : s.children) { all.push_back(make_iterator(c, num_docs, entries)); num_sorted += all.back()->type() == Iterator::Type::kSorted ? 1 : 0; }
      auto out = std::make_unique<OrIterator>();
      if (num_sorted > 1) {
        const size_t nw = ((size_t)num_docs + 63) / 64;
        auto docs = std::make_shared<std::vector<uint64_t>>(std::max<size_t>(nw, 1), 0ull);
        for (auto& it : all) if (is_index(*it)) { for (size_t w = 0; w < nw; ++w) (*docs)[w] |= (**it->docs())[w]; it.reset(); }
        out->children.push_back(std::make_unique<SetIterator>(docs, num_docs, Iterator::Type::kBitmap));
      }
      for (auto& it : all) if (it) out->children.push_back(std::move(it));
      if (out->children.size() == 1) return std::move(out->children[0]);
      out->start();
      return out;
    }
    default: return std::make_unique<NotIterator>(make_iterator(s.children.at(0), num_docs, entries), num_docs);
  }
}

// ---- getTrues / getFalses over the flattened filter (AndFilterOperator.java:52-88, OrFilterOperator.java:51-87,
//      NotFilterOperator.java:52-63, BaseFilterOperator.java:96-113; enableNullHandling off) ----
enum class LeafClass { kMatchAll, kEmpty, kScan, kSorted, kBitmap };

inline LeafClass classify(const pg_predicate& p) {
  if (p.kind == PG_PRED_MATCH_ALL || p.kind == PG_PRED_MATCH_NONE) return ((p.kind == PG_PRED_MATCH_ALL) != (p.exclusive != 0)) ? LeafClass::kMatchAll : LeafClass::kEmpty;
  if (p.kind == PG_PRED_DOC_RANGE) return LeafClass::kSorted;
  if (p.kind == PG_PRED_IS_NULL || p.eval == PG_EVAL_INVERTED) return LeafClass::kBitmap;
  return LeafClass::kScan;
}

struct TreeBuilder {
  const pg_query* q;
  const std::vector<Words>* leaf_words;         // per predicate index (null for constant leaves)
  std::vector<int> start;                       // first postfix position of every node's subtree

  explicit TreeBuilder(const pg_query* query, const std::vector<Words>* words) : q(query), leaf_words(words) {
    start.assign((size_t)q->num_filter_nodes, 0);
    for (int i = 0; i < q->num_filter_nodes; ++i) {
      int s = i;
      for (int c = 0; c < arity(i); ++c) s = start[(size_t)s - 1];
      start[(size_t)i] = s;
    }
  }
  int arity(int node) const { const pg_filter_node& n = q->filter[node]; return n.op == PG_FILTER_LEAF ? 0 : (n.op == PG_FILTER_NOT ? 1 : n.num_children); }
  std::vector<int> children_of(int node) const {
    std::vector<int> kids((size_t)arity(node));
    int end = node - 1;
    for (int c = (int)kids.size() - 1; c >= 0; --c) { kids[(size_t)c] = end; end = start[(size_t)end] - 1; }
    return kids;
  }
  DocIdSet leaf(int node) const {
    const pg_predicate& p = q->predicates[q->filter[node].predicate];
    DocIdSet s;
    switch (classify(p)) {
      case LeafClass::kMatchAll: s.kind = DocIdSet::Kind::kMatchAll; return s;
      case LeafClass::kEmpty: s.kind = DocIdSet::Kind::kEmpty; return s;
      case LeafClass::kScan: s.kind = DocIdSet::Kind::kScan; break;
      case LeafClass::kSorted: s.kind = DocIdSet::Kind::kSorted; break;
      case LeafClass::kBitmap: s.kind = DocIdSet::Kind::kBitmap; break;
    }
    s.words = (*leaf_words)[(size_t)q->filter[node].predicate];
    return s;
  }
  static DocIdSet constant(bool all) { DocIdSet s; s.kind = all ? DocIdSet::Kind::kMatchAll : DocIdSet::Kind::kEmpty; return s; }
  static DocIdSet negate(DocIdSet inner) { DocIdSet s; s.kind = DocIdSet::Kind::kNot; s.children.push_back(std::move(inner)); return s; }

  DocIdSet trues(int node) const {
    const pg_filter_node& n = q->filter[node];
    if (n.op == PG_FILTER_LEAF) return leaf(node);
    const std::vector<int> kids = children_of(node);
    if (n.op == PG_FILTER_NOT) return falses(kids[0]);
    DocIdSet s;
    s.kind = n.op == PG_FILTER_AND ? DocIdSet::Kind::kAnd : DocIdSet::Kind::kOr;
    for (int k : kids) s.children.push_back(trues(k));
    return s;
  }
  DocIdSet falses(int node) const {
    const pg_filter_node& n = q->filter[node];
    if (n.op == PG_FILTER_LEAF) {
      DocIdSet t = leaf(node);
      if (t.kind == DocIdSet::Kind::kMatchAll) return constant(false);
      if (t.kind == DocIdSet::Kind::kEmpty) return constant(true);
      return negate(std::move(t));
    }
    const std::vector<int> kids = children_of(node);
    if (n.op == PG_FILTER_NOT) return trues(kids[0]);
    const bool is_and = n.op == PG_FILTER_AND;
    DocIdSet inner;
    inner.kind = is_and ? DocIdSet::Kind::kAnd : DocIdSet::Kind::kOr;
    for (int k : kids) {
      DocIdSet t = trues(k);
      // AND: an empty child empties the AND, so its complement is everything; match-all children drop out.  OR: the mirror image.
      if (t.kind == (is_and ? DocIdSet::Kind::kEmpty : DocIdSet::Kind::kMatchAll)) return constant(is_and);
      if (t.kind == (is_and ? DocIdSet::Kind::kMatchAll : DocIdSet::Kind::kEmpty)) continue;
      inner.children.push_back(std::move(t));
    }
    if (inner.children.empty()) return constant(!is_and);
    if (inner.children.size() == 1) return negate(std::move(inner.children[0]));
    return negate(std::move(inner));
  }
};

// How the engine gets the count for a query.
enum class Plan {
  kZero,          // no filter, or no scan-based leaf
  kPerLeaf,       // no AND above any scan leaf: numDocs per scan leaf
  kChain,         // root AND of plain leaves with an index-based child: counted by the lane-private kernels (kNodeCountEntries)
  kReplay,        // the iterator walk below
  kLeap2          // root AND of exactly two scan leaves: the leap-frog of AndDocIdIterator over two SVScanDocIdIterators, counted by the
                  // lane-private kernels as a two-state carry chain (kNodeLeapfrog2, leapfrog2_tile / leapfrog2_chain_*_kernel in pg_kernels.h)
};

inline bool malformed(const pg_query* q) {
  int depth = 0;
  for (int i = 0; i < q->num_filter_nodes; ++i) {
    const pg_filter_node& n = q->filter[i];
    const int k = n.op == PG_FILTER_LEAF ? 0 : (n.op == PG_FILTER_NOT ? 1 : n.num_children);
    if (n.op == PG_FILTER_LEAF && (n.predicate < 0 || n.predicate >= q->num_predicates)) return true;
    if (n.op != PG_FILTER_LEAF && (k < 1 || depth < k)) return true;
    depth += 1 - k;
  }
  return q->num_filter_nodes > 0 && depth != 1;
}

inline Plan choose_plan(const pg_query* q, int* num_scan_leaves) {
  *num_scan_leaves = 0;
  if (q->num_filter_nodes == 0 || malformed(q)) return Plan::kZero;
  bool has_and = false, has_constant = false;
  for (int i = 0; i < q->num_filter_nodes; ++i) {
    const pg_filter_node& n = q->filter[i];
    if (n.op == PG_FILTER_AND) has_and = true;
    if (n.op != PG_FILTER_LEAF) continue;
    const LeafClass c = classify(q->predicates[n.predicate]);
    if (c == LeafClass::kScan) ++*num_scan_leaves;
    if (c == LeafClass::kMatchAll || c == LeafClass::kEmpty) has_constant = true;
  }
  if (*num_scan_leaves == 0) return Plan::kZero;
  if (has_constant) return Plan::kReplay;        // (FilterOperatorUtils removes constant children before the tree exists; a caller that
                                                  //  keeps them gets the iterators' answer for the tree as given)
  if (!has_and) return Plan::kPerLeaf;
  const pg_filter_node& root = q->filter[q->num_filter_nodes - 1];
  if (root.op == PG_FILTER_AND && root.num_children == q->num_filter_nodes - 1) {      // every child is a leaf
    int num_index = 0;
    for (int i = 0; i + 1 < q->num_filter_nodes; ++i) num_index += classify(q->predicates[q->filter[i].predicate]) != LeafClass::kScan ? 1 : 0;
    if (num_index > 0) return Plan::kChain;
    if (root.num_children == 2) return Plan::kLeap2;
  }
  return Plan::kReplay;
}

// The commonest replayed shape -- a root AND whose children are all scan leaves -- without iterator objects: the loop of
// AndDocIdIterator.next() (:41-74) over SVScanDocIdIterator.advance (:91-106), every advance a word-level search for the leaf's next
// match that charges the docs it passes.  Same count as the generic walk (tests/test_filter_stats_cpu.py compares both with the oracle).
inline int64_t replay_and_of_scans(const std::vector<const uint64_t*>& leaves, int32_t num_docs) {
  const int n = (int)leaves.size();
  const size_t nw = ((size_t)num_docs + 63) / 64;
  int64_t entries = 0;
  int32_t next_doc = 0;
  for (;;) {
    int32_t max_doc = next_doc;
    int max_index = -1, index = 0;
    while (index < n) {
      if (index == max_index) { ++index; continue; }
      if (max_doc >= num_docs) return entries;                 // advance past the end: EOF, nothing looked at
      const uint64_t* m = leaves[(size_t)index];
      size_t w = (size_t)max_doc >> 6;
      uint64_t bits = m[w] & (~0ull << (max_doc & 63));
      while (bits == 0ull && ++w < nw) bits = m[w];
      const int64_t d = bits ? (int64_t)(w << 6) + __builtin_ctzll(bits) : (int64_t)num_docs;
      if (d >= num_docs) return entries + (num_docs - max_doc);
      entries += d - max_doc + 1;
      if (d == max_doc) ++index;
      else { max_doc = (int32_t)d; max_index = index; index = 0; }
    }
    next_doc = max_doc + 1;                                    // a result doc; DocIdSetOperator asks for the next one
  }
}

// The same count in parallel.  The loop above is a state machine over the docs: exactly one leaf is SCANNING at any doc (it looks at
// every doc until one matches); at its match the other leaves are asked about that doc in child order, one entry each, until one says
// no -- that one scans on from the next doc -- and when all say yes the doc is a result and child 0 scans on.  A chunk of docs is
// therefore a function {scanning leaf at its first doc} -> {entries, scanning leaf after its last doc}: every chunk is simulated from
// each of the k possible states on its own thread, and the chunks' tables are chained in order afterwards.
struct ChunkOutcome { int64_t entries; int end_state; };

inline ChunkOutcome simulate_and_chunk(const std::vector<const uint64_t*>& leaves, int32_t begin, int32_t end, int state) {
  const int n = (int)leaves.size();
  int64_t entries = 0;
  int32_t d = begin;
  const size_t last_word = ((size_t)end + 63) / 64;
  while (d < end) {
    const uint64_t* m = leaves[(size_t)state];
    size_t w = (size_t)d >> 6;
    uint64_t bits = m[w] & (~0ull << (d & 63));
    while (bits == 0ull && ++w < last_word) bits = m[w];
    const int64_t p = bits ? (int64_t)(w << 6) + __builtin_ctzll(bits) : (int64_t)end;
    if (p >= end) { entries += end - d; break; }               // still scanning at the end of the chunk
    entries += p - d + 1;
    int next_state = 0;
    for (int i = 0; i < n; ++i) {
      if (i == state) continue;
      ++entries;
      if (!((leaves[(size_t)i][(size_t)p >> 6] >> (p & 63)) & 1ull)) { next_state = i; break; }
    }
    state = next_state;
    d = (int32_t)p + 1;
  }
  return ChunkOutcome{entries, state};
}

inline int64_t replay_and_of_scans_parallel(const std::vector<const uint64_t*>& leaves, int32_t num_docs, int32_t chunk_docs, int num_threads) {
  const int k = (int)leaves.size();
  chunk_docs = std::max(chunk_docs, 64);
  const int num_chunks = (int)(((int64_t)num_docs + chunk_docs - 1) / chunk_docs);
  if (num_chunks <= 1 || num_threads <= 1) return replay_and_of_scans(leaves, num_docs);
  std::vector<ChunkOutcome> table((size_t)num_chunks * (size_t)k);
  std::atomic<int> next_chunk{0};
  auto worker = [&] {
    for (int c = next_chunk.fetch_add(1); c < num_chunks; c = next_chunk.fetch_add(1)) {
      const int32_t begin = (int32_t)((int64_t)c * chunk_docs), end = (int32_t)std::min<int64_t>(num_docs, (int64_t)begin + chunk_docs);
      // chunk 0 starts with child 0 scanning: its other rows are never read
      for (int s = 0; s < (c == 0 ? 1 : k); ++s) table[(size_t)c * (size_t)k + (size_t)s] = simulate_and_chunk(leaves, begin, end, s);
    }
  };
  std::vector<std::thread> threads;
  for (int t = 1; t < std::min(num_threads, num_chunks); ++t) threads.emplace_back(worker);
  worker();
  for (auto& t : threads) t.join();
  int64_t entries = 0;
  int state = 0;
  for (int c = 0; c < num_chunks; ++c) {
    const ChunkOutcome& o = table[(size_t)c * (size_t)k + (size_t)state];
    entries += o.entries;
    state = o.end_state;
  }
  return entries;
}

// Plan::kLeap2 on the host, in the device's own structure (leapfrog2_tile / leapfrog2_chain_kernel in pg_kernels.h): 32-doc lanes whose
// "child 1 is scanning" state is the carry of one addition (generate a & ~b, propagate ~(a | b)), 64 lanes per tile chained through a
// 64-bit add of their generate / propagate ballots, tiles summarised for entry state 0 plus the correction for entry state 1, summaries
// chained in order.  Returns the EXTRA entries (numEntriesScannedInFilter - numDocs).  tests/test_filter_stats_cpu.py holds it against
// the oracle and the iterator walk; the device code is the same arithmetic.
inline int64_t leap2_extra_entries(const uint64_t* a_words, const uint64_t* b_words, int32_t num_docs) {
  const int64_t num_tiles = ((int64_t)num_docs + 2047) / 2048;
  const size_t nw = ((size_t)num_docs + 63) / 64;
  auto lane_word = [&](const uint64_t* w, int64_t tile, int lane) -> uint32_t {
    const int64_t first = tile * 2048 + (int64_t)lane * 32;
    if (first >= num_docs) return 0u;
    const size_t i = (size_t)(first >> 6);
    uint32_t v = i < nw ? (uint32_t)(w[i] >> (first & 63)) : 0u;
    const int64_t rem = (int64_t)num_docs - first;
    return rem >= 32 ? v : (v & ((1u << (int)rem) - 1u));
  };
  uint64_t total = 0;
  uint32_t g_all = 0, p_all = 1;
  int d_all = 0;                                  // (kept for symmetry with the device's summaries; the chain starts in state 0)
  for (int64_t tile = 0; tile < num_tiles; ++tile) {
    uint32_t A[64], B[64];
    uint64_t sum0[64];
    uint64_t Gm = 0, Pm = 0;
    for (int lane = 0; lane < 64; ++lane) {
      A[lane] = lane_word(a_words, tile, lane); B[lane] = lane_word(b_words, tile, lane);
      const uint32_t E = A[lane] | B[lane], X = A[lane] & ~B[lane], prop = ~E;
      sum0[lane] = (uint64_t)(X | prop) + (uint64_t)X;
      if ((uint32_t)(sum0[lane] >> 32)) Gm |= 1ull << lane;
      if (E == 0) Pm |= 1ull << lane;
    }
    unsigned long long wsum;
    const bool tile_carry = __builtin_uaddll_overflow(Gm | Pm, Gm, &wsum);
    const uint64_t cins = wsum ^ Pm;
    uint32_t cost0 = 0;
    for (int lane = 0; lane < 64; ++lane) {
      const uint32_t E = A[lane] | B[lane], X = A[lane] & ~B[lane], Y = B[lane] & ~A[lane], T = A[lane] & B[lane];
      const uint32_t S = (uint32_t)(sum0[lane] + ((cins >> lane) & 1ull)) ^ ~E;
      cost0 += (uint32_t)(__builtin_popcount(T) + __builtin_popcount(X & ~S) + __builtin_popcount(Y & S));
    }
    int delta = 0;
    const uint64_t with_events = ~Pm;
    if (with_events) {
      const int first = __builtin_ctzll(with_events);
      const uint32_t Ef = A[first] | B[first];
      const int bit = __builtin_ctz(Ef);
      delta = (int)(((B[first] & ~A[first]) >> bit) & 1u) - (int)(((A[first] & ~B[first]) >> bit) & 1u);
    }
    const uint32_t g = tile_carry ? 1u : 0u, pp = with_events == 0 ? 1u : 0u;
    total += cost0 + (g_all ? (int64_t)delta : 0);
    d_all += p_all ? delta : 0;
    g_all = g | (pp & g_all);
    p_all &= pp;
  }
  return (int64_t)total;
}

// The walk itself: DocIdSetOperator pulls next() until EOF (core/operator/DocIdSetOperator.java:66-90).
inline int64_t replay_generic(const DocIdSet& root, int32_t num_docs) {
  int64_t entries = 0;
  IteratorPtr it = make_iterator(root, num_docs, &entries);
  while (it->next() != kEof) {}
  return entries;
}

// chunk_docs / num_threads <= 0: the defaults (1 Mi docs per chunk, the host's cores up to 32; small segments stay on one thread).
// use_fast_paths = false forces the iterator objects (tests compare the two).
inline int64_t replay(const pg_query* q, int32_t num_docs, const std::vector<Words>& leaf_words, bool use_fast_paths = true, int32_t chunk_docs = 0, int num_threads = 0) {
  TreeBuilder builder(q, &leaf_words);
  const DocIdSet root = builder.trues(q->num_filter_nodes - 1);
  if (use_fast_paths && root.kind == DocIdSet::Kind::kAnd && root.children.size() >= 2) {
    std::vector<const uint64_t*> scans;
    for (const DocIdSet& c : root.children) if (c.kind == DocIdSet::Kind::kScan && c.words) scans.push_back(c.words->data());
    if (scans.size() == root.children.size()) {
      if (num_threads <= 0) num_threads = num_docs >= (1 << 22) ? (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u) : 1;
      return replay_and_of_scans_parallel(scans, num_docs, chunk_docs > 0 ? chunk_docs : (1 << 20), num_threads);
    }
  }
  return replay_generic(root, num_docs);
}

}  // namespace fstats
}  // namespace pg
