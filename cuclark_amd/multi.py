"""Multi-GPU plumbing for the two ways the path shards.  One process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

read-sharded : table replicated, reads dealt to ranks; no collective on the data path.
table-sharded: rank r holds buckets [r*ceil(H/N), ...) (the reference's m_partPointer ranges, CuClarkDB.cu:566-574),
               every rank probes all reads; per-read sparse rows are exchanged so that rank r owns read range r:
                   all_to_all_single(rows by read range) -> N-1 merges (sum by target) -> best/second
               replacing the reference's cudaMemcpyPeer + mergeKernel tree into device 0 (CuClarkDB.cu:954-974).
"""
import torch
import torch.distributed as dist


def shard_range(htsize, world, rank):
    per = (htsize + world - 1) // world
    return rank * per, min(htsize, (rank + 1) * per)


def read_range(n_reads, world, rank):
    per = (n_reads + world - 1) // world
    return rank * per, min(n_reads, (rank + 1) * per), per


def padded_rows(n_reads, world, row_words, device, dtype=torch.int32):
    """Row buffer whose length is a multiple of `world` so it splits evenly; pad rows have n = 0."""
    per = (n_reads + world - 1) // world
    return torch.zeros((per * world, row_words), dtype=dtype, device=device)


def exchange_rows(rows, world, out=None):
    """rows: [world*per, row_words] of this rank's shard -> [world, per, row_words]: slice r of every rank's rows
    lands on rank r (index 0 of the result = rows computed by rank 0's shard, ...)."""
    per = rows.shape[0] // world
    if out is None:
        out = torch.empty((world, per, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    dist.all_to_all_single(out.view(-1), rows.view(-1))
    return out


def merge_exchanged(recv, merge_fn):
    """Fold the `world` row sets of this rank's read range with merge_fn(a, b) -> a (+) b."""
    cur = recv[0]
    for r in range(1, recv.shape[0]):
        cur = merge_fn(cur, recv[r])
    return cur


ROW_INVALID = -1   # MIC_ROW_INVALID (0xFFFFFFFF) as int32


def overflowed_reads(merged_rows, world, rank, n_reads):
    """Reads of this rank's range whose merged sparse row does not fit (row[0] == MIC_ROW_INVALID), as GLOBAL read
    ids gathered from every rank: (all_ids [total], offsets [world + 1]) - rank r's reads are all_ids[offsets[r]:offsets[r+1]].
    The reference truncates such rows (CuClarkDB.cu:1200-1211); here they are completed exactly (complete_overflowed)."""
    lo, hi, per = read_range(n_reads, world, rank)
    bad = (merged_rows[: max(hi - lo, 0), 0] == ROW_INVALID).nonzero().flatten().to(torch.int64) + lo
    n_mine = torch.tensor([bad.numel()], dtype=torch.int64, device=merged_rows.device)
    counts = [torch.zeros_like(n_mine) for _ in range(world)]
    dist.all_gather(counts, n_mine)
    counts = [int(c.item()) for c in counts]
    top = max(counts) if counts else 0
    offsets = [0]
    for c in counts:
        offsets.append(offsets[-1] + c)
    if top == 0:
        return torch.zeros(0, dtype=torch.int64, device=merged_rows.device), offsets
    pad = torch.full((top,), -1, dtype=torch.int64, device=merged_rows.device)
    pad[: bad.numel()] = bad
    gathered = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad)
    return torch.cat([g[:c] for g, c in zip(gathered, counts)]), offsets


def complete_overflowed(merged_rows, world, rank, n_reads, count_dense_fn):
    """Exact completion of the reads whose merged row overflowed: every rank counts those reads densely against ITS
    bucket range (count_dense_fn(global_ids) -> int32 [n, T]), the counts are summed over the ranks (all_reduce), and
    each rank gets back (local row indices, dense counts) of the reads it owns.  One small exchange; flagged reads only."""
    ids, offsets = overflowed_reads(merged_rows, world, rank, n_reads)
    if ids.numel() == 0:
        return None, None
    counts = count_dense_fn(ids).contiguous()
    dist.all_reduce(counts)                      # per-target counts are additive across shards (CuClarkDB.cu:1385-1388)
    lo = read_range(n_reads, world, rank)[0]
    mine = slice(offsets[rank], offsets[rank + 1])
    return (ids[mine] - lo), counts[mine]


def gather_results(res_part, world):
    """[per, 8] per rank -> [world*per, 8] on every rank, in read order."""
    out = torch.empty((world * res_part.shape[0], res_part.shape[1]), dtype=res_part.dtype, device=res_part.device)
    dist.all_gather_into_tensor(out, res_part.contiguous())
    return out
