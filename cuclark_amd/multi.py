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


def gather_results(res_part, world):
    """[per, 8] per rank -> [world*per, 8] on every rank, in read order."""
    out = torch.empty((world * res_part.shape[0], res_part.shape[1]), dtype=res_part.dtype, device=res_part.device)
    dist.all_gather_into_tensor(out, res_part.contiguous())
    return out
