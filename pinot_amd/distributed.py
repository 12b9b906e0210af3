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
