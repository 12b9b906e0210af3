"""One-process-per-GPU plumbing for segment-sharded execution.

Segments are independent units, so there is no collective on the data path: every rank runs its own segment(s) on its
own GPU and only the few-byte partials travel.  They are brought to every rank with one all_gather (RCCL when the
tensors live on GPUs, gloo on CPU) and merged on the host with the reference's merge rules
(AggregationFunction.merge: SUM '+' on doubles, COUNT '+' on longs, MIN/MAX min/max, AVG pairwise;
core/operator/combine/merger/AggregationResultsBlockMerger.java:34-44).
"""
import torch
import torch.distributed as dist


def gather_partials(values_i64, device):
    """all_gather a small list of int64 partials; returns a list (one entry per rank) of python int lists."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    t = torch.tensor([int(v) for v in values_i64], dtype=torch.int64, device=device)
    if world == 1:
        return [[int(x) for x in t.tolist()]]
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [[int(x) for x in p.tolist()] for p in parts]


def merge_sum_count(per_rank):
    """per_rank: [[exact_sum_i64, count], ...] -> (double sum merged like SumAggregationFunction.merge, long count)."""
    merged_sum = 0.0
    merged_count = 0
    for s, c in per_rank:
        merged_sum = merged_sum + float(s)   # each segment's intermediate result is a Double
        merged_count += c
    return merged_sum, merged_count


def max_over_ranks(seconds, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- group-by across ranks ----
# DictIds are segment-local (every segment has its own dictionaries), so group-by partials can only be merged on the VALUES of the
# keys, like GroupByCombineOperator does when it upserts `Key(Object[] values)` records into its indexed table
# (core/operator/combine/GroupByCombineOperator.java:132-147, core/data/table/Key.java).  Each rank turns its raw group ids into
# value tuples with its own dictionaries first; only those rows travel (all_gather_object: tens of bytes per group).

def group_rows(result, segment_data, group_by_columns):
    """[(key values tuple, [(function, AggValue)...])] of one segment's group-by result: raw group id -> per-column dictIds
    (DictionaryBasedGroupKeyGenerator.java:306-324: id = sum dictId_j * prod_{k<j} cardinality_k) -> dictionary values."""
    cards = [segment_data.columns[c].cardinality for c in group_by_columns]
    rows = []
    for gid, values in result.groups.items():
        key, rest = [], gid
        for c, card in zip(group_by_columns, cards):
            key.append(segment_data.columns[c].value_of(rest % card))
            rest //= card
        rows.append((tuple(key), [(f, v.count, v.sum, v.sum_i64, v.min, v.max) for f, v in zip(result.functions, values)]))
    return rows


def merge_group_rows(per_rank_rows):
    """Value-keyed merge with the reference's merge rules per function (AggregationFunction.merge: SUM '+' on doubles, COUNT '+',
    MIN / MAX Math.min / max, AVG pairwise sum and count).  Returns {key tuple: [(function, count, sum, sum_i64, min, max)...]}."""
    table = {}
    for rows in per_rank_rows:
        for key, values in rows:
            if key not in table:
                table[key] = [tuple(v) for v in values]
                continue
            merged = []
            for (f, c0, s0, i0, mn0, mx0), (_, c1, s1, i1, mn1, mx1) in zip(table[key], values):
                merged.append((f, c0 + c1, s0 + s1, i0 + i1, min(mn0, mn1), max(mx0, mx1)))
            table[key] = merged
    return table


def gather_group_rows(rows):
    """all_gather of the value-keyed rows (python objects over the process group's CPU backend)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [rows]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, rows)
    return out
