// numEntriesScannedInFilter of a leap-frogging root AND as a finite-state transducer over the docs -- what lets the DEVICE count it at any
// segment size (pg_filter_stats.h replays the reference's iterator objects on the host, one advance() at a time, up to 64 Mi docs).
//
// Shapes: a root AND whose children are scan leaves, index-based leaves (sorted docId ranges, inverted-index / null bitmaps) or ORs of
// such leaves -- `a AND b AND c`, `a AND (b OR c)`, the reference's own golden filter (sorted range AND (scan OR posting) AND scan AND scan).
// What the reference does with them (docidsets/AndDocIdSet.java:73-172, dociditerators/AndDocIdIterator.java:41-80,
// OrDocIdIterator.java:52-140, SVScanDocIdIterator.java:76-145), as a walk over the docs x = 0, 1, 2, ...:
//   * AndDocIdSet first merges its index-based children into one bitmap and and-s its scan children into it in list order
//     (ScanBasedDocIdIterator.applyAnd: one entry per docId still standing) -- per doc, an entry count that depends on the leaves' match
//     bits only; the merged bitmap then leads the remaining (OR) children as one index-based child.  Without an index-based child the
//     children leap-frog as they are.
//   * AndDocIdIterator: exactly one child LEADS at any doc.  A leading scan leaf looks at every doc (one entry each) until it matches;
//     at the leader's match the other children are asked about that doc in child order (a scan leaf: one entry) until one does not
//     contain it -- that one leads from there -- and when all contain it the doc is a result and child 0 leads from the next doc,
//     which it is asked about afresh.
//   * An OR child is asked through OrDocIdIterator.advance(t): every member whose look-ahead is behind t scans on from t to ITS next
//     match.  So a scan member of an OR looks at doc x iff x is a doc the OR is asked about, or it looked at x - 1 and did not match
//     there ("open").  One bit of state per such member.
// State = (leading child, "the next doc is asked afresh", the open bits); input = the leaves' match bits at the doc; output = entries.
// The reachable states are enumerated here (at most kFsmMaxStates, else the shape stays with the host replay) into a table
// delta[state << L | input] = next state | entries << 4.  Chunks of docs are functions {entry state} -> {exit state, entries}: lanes of
// 32 docs, tiles of 64 lanes, tiles chained in order -- fsm_count_tiled below is that structure on the host (the CPU tests hold it
// against the iterator replay and the oracle), the kernels in pg_fsm_kernels.h are the same arithmetic.
// Host-side C++ only.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <map>
#include <tuple>
#include <vector>

#include "pg_filter_stats.h"

namespace pg {
namespace fstats {

constexpr int kFsmMaxStates = 16, kFsmMaxInputs = 8;

struct Fsm {
  int num_inputs = 0;                  // L: leaf bitmaps the walk reads
  std::vector<int> input_predicate;    // [L] predicate index behind every input bit
  int num_states = 0;                  // S; state 0 is where doc 0 is entered
  std::vector<uint8_t> delta;          // [S << L] next state | entries << 4
  // A NOT child over a scan leaf (see "NOT children" below): the batches its leaf scans are charged per EPISODE, not per doc.
  std::vector<uint8_t> marks;          // [S << L] kMarkOpen / kMarkClose / 0; empty when the machine has no episodes
  uint32_t pending_states = 0;         // bit s: in state s an episode is open (the leaf's next() is scanning ahead) -- what the end of the docs closes
  // Further NOT children over scan leaves (round 6b): every such child has an episode stream of its own -- the same machine (delta), its
  // own marks and pending states; the episode pass runs once per stream.  `marks` / `pending_states` above are the first child's.
  struct EpisodeStream { std::vector<uint8_t> marks; uint32_t pending_states = 0; };
  std::vector<EpisodeStream> more_streams;
  bool has_episodes() const { return !marks.empty(); }
  int num_episode_streams() const { return marks.empty() ? 0 : 1 + (int)more_streams.size(); }
  const std::vector<uint8_t>& stream_marks(int k) const { return k == 0 ? marks : more_streams[(size_t)k - 1].marks; }
  uint32_t stream_pending(int k) const { return k == 0 ? pending_states : more_streams[(size_t)k - 1].pending_states; }
};
constexpr int kFsmMaxEpisodeStreams = 3;
constexpr uint8_t kMarkOpen = 1, kMarkClose = 2;

// What an episode costs: the leaf's next() scans whole batches of kScanBatch = 256 docs (pg_filter_stats.h) from `origin` until the batch that holds the first match it
// does not hand on (`close`; num_docs: there is none) -- SVScanDocIdIterator.java:76-98, the last batch stops at numDocs.
inline int64_t episode_entries(int64_t origin, int64_t close, int64_t num_docs) {
  if (origin >= num_docs) return 0;
  if (close >= num_docs) return num_docs - origin;
  return std::min<int64_t>(((close - origin) / kScanBatch + 1) * kScanBatch, num_docs - origin);
}

namespace fsm_detail {

struct Leaf { int input; bool scan; };                       // a leaf as the walk sees it: which input bit, and whether looking at a doc costs an entry
struct Child {
  enum Kind { kSet, kScan, kOr, kNot } kind = kSet;
  std::vector<Leaf> members;                                  // kSet: the leaves and-ed into it (one, or the merged ones); kScan: one leaf; kOr: its members; kNot: the leaf under it, or the members of the OR under it
  std::vector<int> open_bit;                                  // kOr: per member, its bit in the state's open mask (-1: an index-based member)
};
struct Model {
  int num_inputs = 0;
  std::vector<Child> children;                                // of the AndDocIdIterator
  std::vector<std::vector<int>> standing;                     // applyAnd: per and-ed scan leaf, the inputs that must all be set for its entry at a doc
  int num_open = 0;
};

inline bool contains(const Child& c, unsigned input) {
  if (c.kind == Child::kNot) { for (const Leaf& l : c.members) if ((input >> l.input) & 1u) return false; return true; }      // (one leaf, or the members of the OR under it)
  if (c.kind == Child::kOr) { for (const Leaf& l : c.members) if ((input >> l.input) & 1u) return true; return false; }
  for (const Leaf& l : c.members) if (!((input >> l.input) & 1u)) return false;
  return true;
}

// NOT children (NotDocIdIterator.java:36-76 over SVScanDocIdIterator.java:76-112).  The iterator knows ONE doc of its leaf ahead
// (_nextNonMatchingDocId).  Asked about a doc behind that one it answers without touching the leaf; asked about that very doc (or walking
// over it while it leads: the while loop of next()) it HANDS IT ON and pulls the leaf's next(), which scans whole 256-doc batches from
// where the leaf stands; asked about a doc beyond it, it advance()s the leaf -- doc by doc to the leaf's next match, which becomes the
// known doc, and the batches start afresh behind it.  Per doc that is three states of the NOT child:
//   kNotPending   the leaf's next() has been pulled and everything up to the leaf's next match is (being) scanned in batches -- an
//                 EPISODE is open; a match of the leaf that is handed on keeps it open, a match nobody asks about closes it (mark
//                 kMarkClose at that doc) and the child is stale
//   kNotStale     the known doc lies behind; the next time the child is asked it advances the leaf: one entry at that doc and
//   kNotAdvancing one entry per doc up to the leaf's next match.  If that match is asked about (handed on) an episode opens behind it
//                 (mark kMarkOpen: origin = that doc + 1) -> pending; if nobody asks, the child is stale again.
// The constructor pulls next() once: doc 0 is entered pending, origin 0.  An episode's batches are a function of its origin and of the
// doc that closes it (episode_entries above) -- the batch phase never becomes part of the state; the walk only has to say where episodes
// open and close, and they alternate.  Every such child keeps its own three states and its own episode stream (at most
// kFsmMaxEpisodeStreams per machine): `not_state` holds one base-3 digit per NOT child over a scan leaf, in child order, and step()'s
// `mark` two bits per stream.
constexpr int kNotPending = 0, kNotStale = 1, kNotAdvancing = 2;

struct State { int leader; int fresh; unsigned open; int not_state; };
inline bool operator<(const State& a, const State& b) { return std::tie(a.leader, a.fresh, a.open, a.not_state) < std::tie(b.leader, b.fresh, b.open, b.not_state); }

inline State step(const Model& m, State s, unsigned input, int* entries, int* mark = nullptr) {
  int inc = 0;
  for (const auto& need : m.standing) { bool all = true; for (int i : need) all = all && ((input >> i) & 1u); inc += all ? 1 : 0; }
  const int k = (int)m.children.size();
  std::vector<char> asked((size_t)k, 0);
  if (s.fresh) asked[0] = 1;                                   // child 0 is advanced to this doc afresh (AndDocIdIterator.next() after a result / at doc 0)
  State out = s;
  out.fresh = 0;
  if (m.children[(size_t)s.leader].kind == Child::kScan) inc += 1;           // the leading scan leaf looks at every doc
  if (contains(m.children[(size_t)s.leader], input)) {
    int next_leader = 0, result = 1;
    for (int i = 0; i < k; ++i) {
      if (i == s.leader) continue;
      asked[(size_t)i] = 1;
      if (m.children[(size_t)i].kind == Child::kScan) inc += 1;
      if (!contains(m.children[(size_t)i], input)) { next_leader = i; result = 0; break; }
    }
    out.leader = next_leader;
    out.fresh = result;
  }
  for (int c = 0; c < k; ++c) {
    const Child& ch = m.children[(size_t)c];
    if (ch.kind != Child::kOr) continue;
    for (size_t j = 0; j < ch.members.size(); ++j) {
      const int bit = ch.open_bit[j];
      if (bit < 0) continue;
      const bool cover = asked[(size_t)c] || ((s.open >> bit) & 1u);
      inc += cover ? 1 : 0;
      const bool still_open = cover && !((input >> ch.members[j].input) & 1u);
      out.open = (out.open & ~(1u << bit)) | ((still_open ? 1u : 0u) << bit);
    }
  }
  int mk = 0, stream = 0, digit_weight = 1;
  for (int c = 0; c < k; ++c) {
    const Child& ch = m.children[(size_t)c];
    if (ch.kind != Child::kNot) continue;
    const bool is_asked = asked[(size_t)c] || s.leader == c;          // (a leading child is asked about every doc it leads over)
    for (const Leaf& member : ch.members) {
    if (!member.scan) continue;                                       // (an index-based leaf: a bitmap iterator underneath, nothing is counted)
    const bool match = (input >> member.input) & 1u;
    const int mine = (s.not_state / digit_weight) % 3;
    int next_mine = mine, my_mark = 0;
    switch (mine) {
      case kNotPending:
        if (match && !is_asked) { my_mark = kMarkClose; next_mine = kNotStale; }
        break;                                                        // (a match that is asked about is handed on: next() again, the episode goes on)
      case kNotStale:
        if (is_asked) { inc += 1; if (match) { my_mark = kMarkOpen; next_mine = kNotPending; } else next_mine = kNotAdvancing; }
        break;
      default:                                                        // kNotAdvancing
        inc += 1;
        if (match) { if (is_asked) { my_mark = kMarkOpen; next_mine = kNotPending; } else next_mine = kNotStale; }
        break;
    }
    out.not_state += (next_mine - mine) * digit_weight;
    mk |= my_mark << (2 * stream);
    digit_weight *= 3;
    ++stream;
    }
  }
  *entries = inc;
  if (mark) *mark = mk;
  return out;
}

}  // namespace fsm_detail

// The root AND of `q` as a transducer; false when the shape is not one of the above or needs more than kFsmMaxStates states.
inline bool compile_fsm(const pg_query* q, Fsm* out) {
  using namespace fsm_detail;
  if (q->num_filter_nodes < 3 || malformed(q)) return false;
  const std::vector<Words> no_words((size_t)std::max(q->num_predicates, 1));
  TreeBuilder tb(q, &no_words);
  const int root = q->num_filter_nodes - 1;
  if (q->filter[root].op != PG_FILTER_AND) return false;
  Model m;
  std::vector<int> input_predicate;
  auto leaf_of = [&](int node, Leaf* l) {
    const pg_filter_node& n = q->filter[node];
    if (n.op != PG_FILTER_LEAF) return false;
    const LeafClass c = classify(q->predicates[n.predicate]);
    if (c == LeafClass::kMatchAll || c == LeafClass::kEmpty) return false;      // (FilterOperatorUtils folds constants away before a tree exists)
    auto it = std::find(input_predicate.begin(), input_predicate.end(), n.predicate);
    if (it == input_predicate.end()) { input_predicate.push_back(n.predicate); it = input_predicate.end() - 1; }
    l->input = (int)(it - input_predicate.begin());
    l->scan = c == LeafClass::kScan;
    return true;
  };
  struct Raw { Child child; bool index; };
  std::vector<Raw> raw;
  int num_not_scan = 0;
  for (int kid : tb.children_of(root)) {
    Raw r;
    r.index = false;
    Leaf l;
    if (leaf_of(kid, &l)) {
      r.child.kind = l.scan ? Child::kScan : Child::kSet;
      r.child.members.push_back(l);
      r.index = !l.scan;
    } else if (q->filter[kid].op == PG_FILTER_OR) {
      r.child.kind = Child::kOr;
      int num_sorted = 0, num_members = 0, num_scans = 0;
      for (int g : tb.children_of(kid)) {
        if (!leaf_of(g, &l)) return false;
        r.child.members.push_back(l);
        num_members++;
        num_scans += l.scan ? 1 : 0;
        num_sorted += classify(q->predicates[q->filter[g].predicate]) == LeafClass::kSorted ? 1 : 0;
      }
      if (num_members < 2) return false;
      // (OrDocIdSet.java:62-126 merges two or more SORTED members into one bitmap iterator; when no scan member remains the OR IS that
      //  iterator, an index-based child of the AND: a shape left to the replay.
      //  A KNOWING DEVIATION from the cited lines: this fork's OrDocIdSet.iterator() never fills `bitmapBasedDocIdIterators` (:80-82 count
      //  the member's entries and add it to no list), so with two or more sorted members the reference DROPS the bitmap members of the OR --
      //  docs that only a posting matches leave the result.  The oracle (ds_iterator, DS_OR), the host replay and this machine or the bitmap
      //  members into the merged iterator, i.e. they compute the OR the query asks for; upstream Pinot fills the list.  With one sorted
      //  member or none nothing is merged and the fork, the oracle and the machine agree line by line.  With a scan member present the
      //  machine walks the unmerged members: the same docs and the same entries -- an index-based member costs no entries and a doc of the
      //  OR is a doc of the OR -- held against the oracle by tests/test_filter_stats_cpu.py's random trees, which force this shape.  Round 4 bailed out only when EVERY member was sorted: `idx AND scan AND (bitmap OR sorted OR sorted)` was walked as a
      //  leap-frogging OR and counted 27 559 entries where the iterators count 15 468 -- found by the kernel-coverage table of round 5.)
      if (num_sorted > 1 && num_scans == 0) return false;
    } else if (q->filter[kid].op == PG_FILTER_NOT) {
      // NotFilterOperator.getTrues = the child's getFalses (NotFilterOperator.java:52-63); a leaf's getFalses is a NotDocIdSet over its
      // docId set (BaseFilterOperator.java:96-113): a leap-frogging child.  NOT over anything but a leaf or an OR of leaves stays with the replay.
      const std::vector<int> under = tb.children_of(kid);
      if (under.size() != 1) return false;
      r.child.kind = Child::kNot;
      if (leaf_of(under[0], &l)) {
        r.child.members.push_back(l);
        if (l.scan && ++num_not_scan > kFsmMaxEpisodeStreams) return false;      // an episode stream per such child
      } else if (q->filter[under[0]].op == PG_FILTER_OR) {
        // NOT over an OR of leaves (round 6c): OrFilterOperator.getFalses is a NotDocIdSet over the OrDocIdSet of the members' trues
        // (OrFilterOperator.java:61-88).  The NotDocIdIterator knows one doc of the OR ahead -- the smallest of the members' look-aheads
        // (OrDocIdIterator.java:52-108).  Asked about a doc beyond it, OrDocIdIterator.advance() advances exactly the members whose
        // look-ahead lies behind the doc (doc by doc to their next match); handing the known doc on, OrDocIdIterator.next() pulls next()
        // from exactly the members that stand AT it (whole 256-doc batches, from where each of them stands).  A member ahead of the doc is
        // not touched either way.  So every scan member is the three-state machine of a NOT child over a scan leaf, driven by the NOT
        // child's `asked` and by its OWN match bit: a stale member is advanced whenever the child is asked (the known doc lies behind with
        // it), a member standing at the doc is handed on whenever the child is asked (nothing smaller is left once the stale ones have
        // advanced), a match nobody asks about closes its episode.  One episode stream per scan member; the child contains a doc when no
        // member matches it.  Index-based members cost no entries.
        int num_sorted = 0, num_scans = 0;
        for (int g : tb.children_of(under[0])) {
          if (!leaf_of(g, &l)) return false;
          r.child.members.push_back(l);
          num_scans += l.scan ? 1 : 0;
          num_sorted += classify(q->predicates[q->filter[g].predicate]) == LeafClass::kSorted ? 1 : 0;
          if (l.scan && ++num_not_scan > kFsmMaxEpisodeStreams) return false;
        }
        if (r.child.members.size() < 2) return false;
        if (num_sorted > 1 && num_scans == 0) return false;     // (see the OR child above: the fork's OrDocIdSet and two sorted members)
      } else {
        return false;                                           // NOT over an AND / a NOT: the host replay's
      }
    } else {
      return false;                                             // nested AND under the root AND: the host replay's
    }
    raw.push_back(std::move(r));
  }
  if ((int)input_predicate.size() > kFsmMaxInputs) return false;
  int num_index = 0, num_scan = 0;
  for (const Raw& r : raw) { num_index += r.index ? 1 : 0; num_scan += r.child.kind == Child::kScan ? 1 : 0; }
  if ((num_index > 0 && num_scan > 0) || num_index > 1) {
    // AndDocIdSet.java:127-165: one bitmap of the index-based children, the scan children and-ed into it in list order
    Child merged;
    merged.kind = Child::kSet;
    for (const Raw& r : raw) if (r.index) merged.members.push_back(r.child.members[0]);
    for (const Raw& r : raw) {
      if (r.child.kind != Child::kScan) continue;
      std::vector<int> need;
      for (const Leaf& l : merged.members) need.push_back(l.input);
      m.standing.push_back(need);
      merged.members.push_back(Leaf{r.child.members[0].input, false});
    }
    m.children.push_back(merged);
    for (const Raw& r : raw) if (r.child.kind == Child::kOr || r.child.kind == Child::kNot) m.children.push_back(r.child);
  } else {
    for (const Raw& r : raw) m.children.push_back(r.child);
  }
  for (Child& c : m.children) {
    if (c.kind != Child::kOr) continue;
    for (const Leaf& l : c.members) c.open_bit.push_back(l.scan ? m.num_open++ : -1);
  }
  m.num_inputs = (int)input_predicate.size();
  const int L = m.num_inputs;
  // reachable states, breadth first from (child 0 leads, asked afresh, nothing open)
  std::map<State, int> id;
  std::vector<State> states;
  id[State{0, 1, 0u, kNotPending}] = 0;                         // (NotDocIdIterator's constructor has pulled next(): an episode from doc 0)
  states.push_back(State{0, 1, 0u, kNotPending});
  // (ids of the REACHABLE states are 16 bits wide: a machine with two NOT children reaches 19-34 states that minimise to 15 -- the
  //  kFsmMaxStates bound is applied to what the minimisation leaves)
  constexpr int kFsmMaxReachable = 64;
  std::vector<std::vector<uint16_t>> rows;                      // next state's id | entries << 12
  std::vector<std::vector<uint8_t>> mark_rows;
  for (size_t at = 0; at < states.size(); ++at) {
    std::vector<uint16_t> row((size_t)1 << L);
    std::vector<uint8_t> mark_row((size_t)1 << L);
    for (unsigned input = 0; input < (1u << L); ++input) {
      int inc = 0, mk = 0;
      const State nxt = step(m, states[at], input, &inc, &mk);
      mark_row[input] = (uint8_t)mk;
      auto it = id.find(nxt);
      if (it == id.end()) {
        if ((int)states.size() >= kFsmMaxReachable) return false;
        it = id.emplace(nxt, (int)states.size()).first;
        states.push_back(nxt);
      }
      if (inc > 15) return false;
      row[input] = (uint16_t)(it->second | (inc << 12));
    }
    rows.push_back(std::move(row));
    mark_rows.push_back(std::move(mark_row));
  }
  // Mealy minimisation: two states are one when, for every input, they count the same entries and go on to states that are one ("asked
  // afresh" only matters in front of an OR: `a AND b AND c` has three states, one per leading child).  Classes are numbered by first
  // appearance, so the entry state stays 0.
  const int S0 = (int)states.size();
  std::vector<int> cls((size_t)S0, 0);
  // (with a NOT child: the end of the docs closes an open episode -- an output of the STATE; pending and other states start in different classes)
  int num_start = 1;
  auto pending_mask = [&](int st) {                            // bit k: stream k's episode is open in this state
    unsigned mask = 0;
    int v = states[(size_t)st].not_state;
    for (int k = 0; k < num_not_scan; ++k, v /= 3) mask |= (v % 3 == kNotPending ? 1u : 0u) << k;
    return mask;
  };
  if (num_not_scan > 0) {
    std::map<unsigned, int> start_id;                          // numbered by first appearance: state 0 (everything pending) is class 0
    for (int st = 0; st < S0; ++st) cls[(size_t)st] = start_id.emplace(pending_mask(st), (int)start_id.size()).first->second;
    num_start = (int)start_id.size();
  }
  for (int num_classes = num_start;;) {
    std::map<std::vector<int>, int> sig_id;
    std::vector<int> next_cls((size_t)S0, 0);
    for (int st = 0; st < S0; ++st) {
      std::vector<int> sig{cls[(size_t)st]};
      for (unsigned input = 0; input < (1u << L); ++input) { const uint16_t d = rows[(size_t)st][input]; sig.push_back((d >> 12) | (mark_rows[(size_t)st][input] << 8)); sig.push_back(cls[(size_t)(d & 0xFFF)]); }
      next_cls[(size_t)st] = sig_id.emplace(std::move(sig), (int)sig_id.size()).first->second;
    }
    cls = next_cls;
    if ((int)sig_id.size() == num_classes) break;
    num_classes = (int)sig_id.size();
  }
  const int S = 1 + *std::max_element(cls.begin(), cls.end());
  if (S > kFsmMaxStates) return false;
  out->num_inputs = L;
  out->input_predicate = input_predicate;
  out->num_states = S;
  out->delta.assign((size_t)S << L, 0);
  for (int st = 0; st < S0; ++st)
    for (unsigned input = 0; input < (1u << L); ++input) {
      const uint16_t d = rows[(size_t)st][input];
      out->delta[((size_t)cls[(size_t)st] << L) | input] = (uint8_t)(cls[(size_t)(d & 0xFFF)] | ((d >> 12) << 4));
    }
  out->marks.clear();
  out->pending_states = 0;
  out->more_streams.clear();
  for (int k = 0; k < num_not_scan; ++k) {
    std::vector<uint8_t> marks((size_t)S << L, 0);
    uint32_t pending = 0;
    for (int st = 0; st < S0; ++st) {
      for (unsigned input = 0; input < (1u << L); ++input) marks[((size_t)cls[(size_t)st] << L) | input] = (uint8_t)((mark_rows[(size_t)st][input] >> (2 * k)) & 3);
      if ((pending_mask(st) >> k) & 1u) pending |= 1u << cls[(size_t)st];
    }
    if (k == 0) { out->marks = std::move(marks); out->pending_states = pending; }
    else { Fsm::EpisodeStream es; es.marks = std::move(marks); es.pending_states = pending; out->more_streams.push_back(std::move(es)); }
  }
  return true;
}

// The walk itself, doc by doc (reference for the tiled form below).
inline int64_t fsm_count_sequential(const Fsm& f, const std::vector<const uint64_t*>& leaf_words, int32_t num_docs) {
  const int streams = f.num_episode_streams();
  int64_t entries = 0, origin[kFsmMaxEpisodeStreams] = {0, 0, 0};      // (origin: of the episode that is open, per NOT child over a scan leaf)
  int state = 0;
  for (int32_t x = 0; x < num_docs; ++x) {
    unsigned input = 0;
    for (int i = 0; i < f.num_inputs; ++i) input |= (unsigned)((leaf_words[(size_t)i][(size_t)x >> 6] >> (x & 63)) & 1ull) << i;
    const size_t at = ((size_t)state << f.num_inputs) | input;
    const uint8_t d = f.delta[at];
    entries += d >> 4;
    state = d & 15;
    for (int k = 0; k < streams; ++k) {
      const uint8_t mk = f.stream_marks(k)[at];
      if (mk == kMarkClose) entries += episode_entries(origin[k], x, num_docs);
      else if (mk == kMarkOpen) origin[k] = (int64_t)x + 1;
    }
  }
  for (int k = 0; k < streams; ++k) if ((f.stream_pending(k) >> state) & 1u) entries += episode_entries(origin[k], num_docs, num_docs);
  return entries;
}

// One chunk's table: entry state -> exit state, entries.
struct FsmTable { uint8_t next[kFsmMaxStates]; uint32_t entries[kFsmMaxStates]; };

// The device's structure (pg_fsm_kernels.h) on the host: lanes of 32 docs walked from every entry state, 64 lane tables composed into
// the tile's table, tile tables chained in order from state 0.  Docs past numDocs do not exist: a lane stops at the last doc.
inline int64_t fsm_count_tiled(const Fsm& f, const std::vector<const uint64_t*>& leaf_words, int32_t num_docs) {
  const int S = f.num_states, L = f.num_inputs;
  const int64_t num_tiles = ((int64_t)num_docs + 2047) / 2048;
  int64_t entries = 0;
  int state = 0;
  for (int64_t tile = 0; tile < num_tiles; ++tile) {
    FsmTable tile_table;
    for (int s = 0; s < S; ++s) { tile_table.next[s] = (uint8_t)s; tile_table.entries[s] = 0; }
    for (int lane = 0; lane < 64; ++lane) {
      const int64_t first = tile * 2048 + (int64_t)lane * 32;
      const int docs = (int)std::max<int64_t>(0, std::min<int64_t>(32, (int64_t)num_docs - first));
      uint32_t w[kFsmMaxInputs];
      for (int i = 0; i < L; ++i) w[i] = docs > 0 ? (uint32_t)(leaf_words[(size_t)i][(size_t)first >> 6] >> (first & 63)) : 0u;
      FsmTable lane_table;
      for (int s = 0; s < S; ++s) {
        int cur = s;
        uint32_t e = 0;
        for (int d = 0; d < docs; ++d) {
          unsigned input = 0;
          for (int i = 0; i < L; ++i) input |= ((w[i] >> d) & 1u) << i;
          const uint8_t t = f.delta[((size_t)cur << L) | input];
          e += t >> 4;
          cur = t & 15;
        }
        lane_table.next[s] = (uint8_t)cur;
        lane_table.entries[s] = e;
      }
      for (int s = 0; s < S; ++s) {          // tile_table := lane_table after tile_table
        const int mid = tile_table.next[s];
        tile_table.entries[s] += lane_table.entries[mid];
        tile_table.next[s] = lane_table.next[mid];
      }
    }
    entries += tile_table.entries[state];
    state = tile_table.next[state];
  }
  return entries;
}

// The episodes of a machine with a NOT child (Fsm::marks), in the device's structure -- pg_fsm_kernels.h: fsm_chunk_states_kernel /
// fsm_tile_states_kernel hand every tile the state it is entered in (the tiles' tables walked from state 0), fsm_episode_tiles_kernel walks
// every lane's 32 docs from ITS entry state (the 64 lane tables walked from the tile's), leaves an open word and a close word per lane,
// pairs every close with the last open in front of it inside the tile and keeps, per tile, the close that has none (there is at most one:
// opens and closes alternate) and its last open; fsm_episode_finish_kernel pairs those across tiles and closes what the end of the docs
// leaves open.  Returns the episodes' entries only (fsm_count_tiled counts the per-doc ones).
inline int64_t fsm_episode_entries_tiled(const Fsm& f, const std::vector<const uint64_t*>& leaf_words, int32_t num_docs, int stream = -1) {
  if (!f.has_episodes() || num_docs <= 0) return 0;
  if (stream < 0) {                                             // every NOT child's stream: the pass once per stream, like the device
    int64_t all = 0;
    for (int k = 0; k < f.num_episode_streams(); ++k) all += fsm_episode_entries_tiled(f, leaf_words, num_docs, k);
    return all;
  }
  const std::vector<uint8_t>& marks = f.stream_marks(stream);
  const uint32_t pending_states = f.stream_pending(stream);
  const int S = f.num_states, L = f.num_inputs;
  const int64_t num_tiles = ((int64_t)num_docs + 2047) / 2048;
  auto lane_words = [&](int64_t first, int docs, uint32_t* w) { for (int i = 0; i < L; ++i) w[i] = docs > 0 ? (uint32_t)(leaf_words[(size_t)i][(size_t)first >> 6] >> (first & 63)) : 0u; };
  auto input_of = [&](const uint32_t* w, int d) { unsigned in = 0; for (int i = 0; i < L; ++i) in |= ((w[i] >> d) & 1u) << i; return in; };
  // the tiles' entry states
  std::vector<uint8_t> tile_state((size_t)num_tiles);
  std::vector<int32_t> tile_first_close((size_t)num_tiles, -1), tile_last_open((size_t)num_tiles, -1);
  int64_t sum = 0;
  int state = 0, final_pending = 0;
  for (int64_t tile = 0; tile < num_tiles; ++tile) {
    tile_state[(size_t)tile] = (uint8_t)state;
    uint8_t lane_next[64][kFsmMaxStates];
    for (int lane = 0; lane < 64; ++lane) {
      const int64_t first = tile * 2048 + (int64_t)lane * 32;
      const int docs = (int)std::max<int64_t>(0, std::min<int64_t>(32, (int64_t)num_docs - first));
      uint32_t w[kFsmMaxInputs];
      lane_words(first, docs, w);
      for (int s = 0; s < S; ++s) {
        int cur = s;
        for (int d = 0; d < docs; ++d) cur = f.delta[((size_t)cur << L) | input_of(w, d)] & 15;
        lane_next[lane][s] = (uint8_t)cur;
      }
    }
    // the lanes' entry states; every lane again from its own: the open / close words
    int cur = state;
    int32_t prev_open = -1;                                    // the last open of the lanes in front (the device: a prefix maximum over the wavefront)
    for (int lane = 0; lane < 64; ++lane) {
      const int64_t first = tile * 2048 + (int64_t)lane * 32;
      const int docs = (int)std::max<int64_t>(0, std::min<int64_t>(32, (int64_t)num_docs - first));
      uint32_t w[kFsmMaxInputs];
      lane_words(first, docs, w);
      uint32_t open_word = 0, close_word = 0;
      int st = cur;
      for (int d = 0; d < docs; ++d) {
        const size_t at = ((size_t)st << L) | input_of(w, d);
        open_word |= (marks[at] == kMarkOpen ? 1u : 0u) << d;
        close_word |= (marks[at] == kMarkClose ? 1u : 0u) << d;
        st = f.delta[at] & 15;
      }
      if (docs > 0 && first + docs == (int64_t)num_docs) final_pending = (int)((pending_states >> st) & 1u);      // the lane that holds the last doc
      for (uint32_t cw = close_word; cw != 0; cw &= cw - 1) {
        const int d = __builtin_ctz(cw);
        const uint32_t below = open_word & ((1u << d) - 1u);
        const int32_t open_at = below ? (int32_t)(first + 31 - __builtin_clz(below)) : prev_open;
        const int64_t x = first + d;
        if (open_at >= 0) sum += episode_entries((int64_t)open_at + 1, x, num_docs);
        else tile_first_close[(size_t)tile] = (int32_t)x;       // (its open lies in an earlier tile, or it is the episode of doc 0)
      }
      if (open_word) prev_open = (int32_t)(first + 31 - __builtin_clz(open_word));
      cur = lane_next[lane][cur];
    }
    tile_last_open[(size_t)tile] = prev_open;
    state = cur;
  }
  // across tiles, the way fsm_episode_finish_kernel does it: sixteen wavefronts, each a contiguous range of tiles read 64 at a time (lane l: tile
  // base + l); the last open in front of a tile = the maximum of the wavefronts in front, of the 64-tile groups in front (carry) and of the
  // lanes in front (a prefix maximum over the wavefront)
  const int64_t per_wave = (num_tiles + 15) / 16;
  int32_t wave_last[16], wave_carry[16];
  for (int wave = 0; wave < 16; ++wave) {
    const int64_t lo = std::min<int64_t>(wave * per_wave, num_tiles), hi = std::min<int64_t>(lo + per_wave, num_tiles);
    int32_t m = -1;
    for (int64_t i = lo; i < hi; ++i) m = std::max(m, tile_last_open[(size_t)i]);
    wave_last[wave] = m;
  }
  int32_t all_last = -1;
  for (int wave = 0; wave < 16; ++wave) { wave_carry[wave] = all_last; all_last = std::max(all_last, wave_last[wave]); }
  for (int wave = 0; wave < 16; ++wave) {
    const int64_t lo = std::min<int64_t>(wave * per_wave, num_tiles), hi = std::min<int64_t>(lo + per_wave, num_tiles);
    int32_t carry = wave_carry[wave];
    for (int64_t base = lo; base < hi; base += 64) {
      int32_t incl[64];
      for (int lane = 0; lane < 64; ++lane) { const int64_t i = base + lane; incl[lane] = i < hi ? tile_last_open[(size_t)i] : -1; }
      for (int off = 1; off < 64; off <<= 1)                        // (the device: __shfl_up rounds)
        for (int lane = 63; lane >= off; --lane) incl[lane] = std::max(incl[lane], incl[lane - off]);
      for (int lane = 0; lane < 64; ++lane) {
        const int64_t i = base + lane;
        const int32_t x = i < hi ? tile_first_close[(size_t)i] : -1;
        const int32_t prev = std::max(carry, lane ? incl[lane - 1] : -1);
        if (x >= 0) sum += episode_entries((int64_t)prev + 1, x, num_docs);
      }
      carry = std::max(carry, incl[63]);
    }
  }
  if (final_pending) sum += episode_entries((int64_t)all_last + 1, num_docs, num_docs);
  return sum;
}

// fsm_tiles_perm_kernel's arithmetic on the host (machines of at most four states and four inputs): a function {entry state} -> {exit
// state} is four bytes of one word, composition is a byte permute (V_PERM_B32: selector byte k in 0..3 picks byte k of the second source,
// 4..7 of the first, 0x0c gives zero), the entries ride along as bytes (lane) / 16-bit fields (tile) gathered by the same selectors, two
// docs per step.  -1 when the machine is larger.  The CPU tests hold it against the doc-by-doc walk; the kernel is the same arithmetic.
inline uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) {
  const uint64_t both = ((uint64_t)s0 << 32) | s1;
  uint32_t out = 0;
  for (int k = 0; k < 4; ++k) {
    const uint32_t c = (sel >> (8 * k)) & 0xFFu;
    const uint32_t byte = c < 8 ? (uint32_t)((both >> (8 * c)) & 0xFFu) : (c == 0x0c ? 0u : 0xFFu);      // (the sign-extending selectors 8..11 and 13+ are not used)
    out |= byte << (8 * k);
  }
  return out;
}
inline int64_t fsm_count_perm(const Fsm& f, const std::vector<const uint64_t*>& leaf_words, int32_t num_docs) {
  const int S = f.num_states, L = f.num_inputs;
  if (S > 4 || L > 4) return -1;
  for (uint8_t d : f.delta) if ((d >> 4) > 7) return -1;      // (a lane's entries per entry state are one byte: 32 docs x at most 7)
  auto delta = [&](uint32_t st, uint32_t in) -> uint32_t { return (st < (uint32_t)S && in < (1u << L)) ? f.delta[((size_t)st << L) | in] : 0u; };
  // pair_fn[idx]: leaf l's bits for docs d, d + 1 at bits 2l, 2l + 1 of idx
  std::vector<uint32_t> pair_next((size_t)1 << (2 * L)), pair_inc((size_t)1 << (2 * L));
  for (uint32_t idx = 0; idx < (1u << (2 * L)); ++idx) {
    uint32_t in0 = 0, in1 = 0, next = 0, inc = 0;
    for (int l = 0; l < L; ++l) { in0 |= ((idx >> (2 * l)) & 1u) << l; in1 |= ((idx >> (2 * l + 1)) & 1u) << l; }
    for (uint32_t st = 0; st < 4; ++st) {
      const uint32_t t0 = delta(st, in0), t1 = delta(t0 & 15u, in1);
      next |= ((t1 & 15u) & 3u) << (8 * st);
      inc |= ((t0 >> 4) + (t1 >> 4)) << (8 * st);
    }
    pair_next[idx] = next; pair_inc[idx] = inc;
  }
  const int64_t num_tiles = ((int64_t)num_docs + 2047) / 2048;
  int64_t entries = 0;
  uint32_t state = 0;
  for (int64_t tile = 0; tile < num_tiles; ++tile) {
    uint32_t F[64], EA[64], EB[64];
    for (int lane = 0; lane < 64; ++lane) {
      const int64_t first = tile * 2048 + (int64_t)lane * 32;
      const int docs = (int)std::max<int64_t>(0, std::min<int64_t>(32, (int64_t)num_docs - first));
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      for (int i = 0; i < L; ++i) w[i] = docs > 0 ? (uint32_t)(leaf_words[(size_t)i][(size_t)first >> 6] >> (first & 63)) : 0u;
      uint32_t fn = 0x03020100u, e = 0u;
      if (docs == 32) {
        for (int d = 0; d < 32; d += 2) {
          uint32_t idx = 0;
          for (int i = 0; i < L; ++i) idx |= ((w[i] >> d) & 3u) << (2 * i);
          e += perm_b32(pair_inc[idx], pair_inc[idx], fn);
          fn = perm_b32(pair_next[idx], pair_next[idx], fn);
        }
      } else {
        uint32_t st[4] = {0u, 1u, 2u, 3u}, ent[4] = {0u, 0u, 0u, 0u};
        for (int d = 0; d < docs; ++d) {
          uint32_t in = 0;
          for (int i = 0; i < L; ++i) in |= ((w[i] >> d) & 1u) << i;
          for (int c = 0; c < 4; ++c) { const uint32_t t = delta(st[c], in); ent[c] += t >> 4; st[c] = t & 3u; }
        }
        fn = st[0] | (st[1] << 8) | (st[2] << 16) | (st[3] << 24);
        e = ent[0] | (ent[1] << 8) | (ent[2] << 16) | (ent[3] << 24);
      }
      F[lane] = fn;
      EA[lane] = perm_b32(0u, e, 0x0c010c00u);
      EB[lane] = perm_b32(0u, e, 0x0c030c02u);
    }
    for (int j = 0; j < 6; ++j) {                    // the tree: every lane computes, like the wavefront does
      uint32_t nF[64], nA[64], nB[64];
      for (int lane = 0; lane < 64; ++lane) {
        const int from = (lane + (1 << j)) & 63;
        const uint32_t G = F[from], HA = EA[from], HB = EB[from];
        const uint32_t selA = (perm_b32(F[lane], F[lane], 0x01010000u) << 1) + 0x01000100u;
        const uint32_t selB = (perm_b32(F[lane], F[lane], 0x03030202u) << 1) + 0x01000100u;
        nA[lane] = EA[lane] + perm_b32(HB, HA, selA);
        nB[lane] = EB[lane] + perm_b32(HB, HA, selB);
        nF[lane] = perm_b32(G, G, F[lane]);
      }
      for (int lane = 0; lane < 64; ++lane) { F[lane] = nF[lane]; EA[lane] = nA[lane]; EB[lane] = nB[lane]; }
    }
    const uint32_t e_of_state = ((state < 2 ? EA[0] : EB[0]) >> (16 * (state & 1u))) & 0xFFFFu;
    entries += e_of_state;
    state = (F[0] >> (8 * state)) & 3u;
  }
  return entries;
}

// fsm_tiles_perm8_kernel's arithmetic on the host: five to eight states -- a function is eight bytes (two words: states 0..3, 4..7), one
// step is four byte permutes over the pair's {next, entries} words, the tree gathers 16-bit entries from four words (two candidates per
// field, picked by bit 2 of the selector).  -1 when the machine does not fit (more than eight states, more than four inputs, a doc of
// more than 7 entries).
inline int64_t fsm_count_perm8(const Fsm& f, const std::vector<const uint64_t*>& leaf_words, int32_t num_docs) {
  const int S = f.num_states, L = f.num_inputs;
  if (S > 8 || L > 4) return -1;
  for (uint8_t d : f.delta) if ((d >> 4) > 7) return -1;
  auto delta = [&](uint32_t st, uint32_t in) -> uint32_t { return (st < (uint32_t)S && in < (1u << L)) ? f.delta[((size_t)st << L) | in] : 0u; };
  struct Pair { uint32_t nlo, nhi, ilo, ihi; };
  std::vector<Pair> pair_fn((size_t)1 << (2 * L));
  for (uint32_t idx = 0; idx < (1u << (2 * L)); ++idx) {
    uint32_t in0 = 0, in1 = 0;
    for (int l = 0; l < L; ++l) { in0 |= ((idx >> (2 * l)) & 1u) << l; in1 |= ((idx >> (2 * l + 1)) & 1u) << l; }
    Pair p{0u, 0u, 0u, 0u};
    for (uint32_t st = 0; st < 8; ++st) {
      const uint32_t t0 = delta(st, in0), t1 = delta(t0 & 15u, in1);
      const uint32_t next = (t1 & 15u) & 7u, inc = (t0 >> 4) + (t1 >> 4);
      if (st < 4) { p.nlo |= next << (8 * st); p.ilo |= inc << (8 * st); } else { p.nhi |= next << (8 * (st - 4)); p.ihi |= inc << (8 * (st - 4)); }
    }
    pair_fn[idx] = p;
  }
  const int64_t num_tiles = ((int64_t)num_docs + 2047) / 2048;
  int64_t entries = 0;
  uint32_t state = 0;
  for (int64_t tile = 0; tile < num_tiles; ++tile) {
    uint32_t Flo[64], Fhi[64], E[4][64];
    for (int lane = 0; lane < 64; ++lane) {
      const int64_t first = tile * 2048 + (int64_t)lane * 32;
      const int docs = (int)std::max<int64_t>(0, std::min<int64_t>(32, (int64_t)num_docs - first));
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      for (int i = 0; i < L; ++i) w[i] = docs > 0 ? (uint32_t)(leaf_words[(size_t)i][(size_t)first >> 6] >> (first & 63)) : 0u;
      uint32_t flo = 0x03020100u, fhi = 0x07060504u, elo = 0u, ehi = 0u;
      if (docs == 32) {
        for (int d = 0; d < 32; d += 2) {
          uint32_t idx = 0;
          for (int i = 0; i < L; ++i) idx |= ((w[i] >> d) & 3u) << (2 * i);
          const Pair& t = pair_fn[idx];
          elo += perm_b32(t.ihi, t.ilo, flo);
          ehi += perm_b32(t.ihi, t.ilo, fhi);
          const uint32_t nlo = perm_b32(t.nhi, t.nlo, flo), nhi = perm_b32(t.nhi, t.nlo, fhi);
          flo = nlo; fhi = nhi;
        }
      } else {
        uint32_t st[8], ent[8];
        for (int c = 0; c < 8; ++c) { st[c] = (uint32_t)c; ent[c] = 0u; }
        for (int d = 0; d < docs; ++d) {
          uint32_t in = 0;
          for (int i = 0; i < L; ++i) in |= ((w[i] >> d) & 1u) << i;
          for (int c = 0; c < 8; ++c) { const uint32_t t = delta(st[c], in); ent[c] += t >> 4; st[c] = t & 7u; }
        }
        flo = st[0] | (st[1] << 8) | (st[2] << 16) | (st[3] << 24);
        fhi = st[4] | (st[5] << 8) | (st[6] << 16) | (st[7] << 24);
        elo = ent[0] | (ent[1] << 8) | (ent[2] << 16) | (ent[3] << 24);
        ehi = ent[4] | (ent[5] << 8) | (ent[6] << 16) | (ent[7] << 24);
      }
      Flo[lane] = flo; Fhi[lane] = fhi;
      E[0][lane] = perm_b32(0u, elo, 0x0c010c00u); E[1][lane] = perm_b32(0u, elo, 0x0c030c02u);
      E[2][lane] = perm_b32(0u, ehi, 0x0c010c00u); E[3][lane] = perm_b32(0u, ehi, 0x0c030c02u);
    }
    for (int j = 0; j < 6; ++j) {
      uint32_t nFlo[64], nFhi[64], nE[4][64];
      for (int lane = 0; lane < 64; ++lane) {
        const int from = (lane + (1 << j)) & 63;
        const uint32_t Glo = Flo[from], Ghi = Fhi[from];
        const uint32_t H01 = E[0][from], H23 = E[1][from], H45 = E[2][from], H67 = E[3][from];
        for (int r = 0; r < 4; ++r) {
          const uint32_t src = r < 2 ? Flo[lane] : Fhi[lane];
          const uint32_t dup = perm_b32(src, src, (r & 1) ? 0x03030202u : 0x01010000u);
          const uint32_t sel = ((dup & 0x03030303u) << 1) + 0x01000100u;
          const uint32_t lo = perm_b32(H23, H01, sel), hi = perm_b32(H67, H45, sel);
          const uint32_t m = (dup >> 2) & 0x01010101u;
          const uint32_t mask = (m << 8) - m;
          nE[r][lane] = E[r][lane] + ((hi & mask) | (lo & ~mask));
        }
        nFlo[lane] = perm_b32(Ghi, Glo, Flo[lane]);
        nFhi[lane] = perm_b32(Ghi, Glo, Fhi[lane]);
      }
      for (int lane = 0; lane < 64; ++lane) { Flo[lane] = nFlo[lane]; Fhi[lane] = nFhi[lane]; for (int r = 0; r < 4; ++r) E[r][lane] = nE[r][lane]; }
    }
    entries += (E[state >> 1][0] >> (16 * (state & 1u))) & 0xFFFFu;
    state = ((state < 4 ? Flo[0] : Fhi[0]) >> (8 * (state & 3u))) & 7u;
  }
  return entries;
}

}  // namespace fstats
}  // namespace pg
