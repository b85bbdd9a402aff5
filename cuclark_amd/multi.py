"""Multi-GPU plumbing for the two ways the path shards.  One process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

read-sharded : table replicated, reads dealt to ranks; no collective on the data path.
table-sharded: rank r holds part r of the table (mic_db_set_part: super-k-mer layouts a slot range of the resident table,
               other layouts the reference's bucket ranges, CuClarkDB.cu:566-574), every rank of a group probes the group's
               reads; per-read sparse rows are exchanged so that group rank j owns read sub-range j:
                   all_to_all_single(rows by read range) -> P-1 merges (sum by target) -> best/second
               replacing the reference's cudaMemcpyPeer + mergeKernel tree into device 0 (CuClarkDB.cu:954-974).
               The exchange is issued per CHUNK of reads and asynchronously, so it overlaps the query kernel of the next chunk.
2-D          : N = R x P ranks: P parts of the table (as many as the table needs to fit), R groups of P ranks; group g takes
               the g-th 1/R of the reads, the exchange stays inside a group.  P = N is the reference's mode, P = 1 read-sharding.
"""
import torch
import torch.distributed as dist


def shard_range(htsize, world, rank):
    per = (htsize + world - 1) // world
    return rank * per, min(htsize, (rank + 1) * per)


def read_range(n_reads, world, rank):
    per = (n_reads + world - 1) // world
    return rank * per, min(n_reads, (rank + 1) * per), per


def grid(world, rank, parts):
    """2-D layout: `parts` ranks per group hold the parts of the table, world // parts groups split the reads.
    Returns (part index, group index, number of groups, global ranks of this rank's group)."""
    assert parts >= 1 and world % parts == 0, f"--parts {parts} does not divide {world} ranks"
    g = rank // parts
    return rank % parts, g, world // parts, list(range(g * parts, (g + 1) * parts))


def padded_rows(n_reads, world, row_words, device, dtype=torch.int32):
    """Row buffer whose length is a multiple of `world` so it splits evenly; pad rows have n = 0."""
    per = (n_reads + world - 1) // world
    return torch.zeros((per * world, row_words), dtype=dtype, device=device)


def exchange_rows(rows, world, out=None, group=None, async_op=False):
    """rows: [world*per, row_words] of this rank's shard -> [world, per, row_words]: slice r of every rank's rows
    lands on rank r (index 0 of the result = rows computed by rank 0's shard, ...).  `world` = size of `group`.
    async_op: returns (out, work); work.wait() before `out` is read (the exchange then overlaps whatever is queued next)."""
    per = rows.shape[0] // world
    if out is None:
        out = torch.empty((world, per, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    work = dist.all_to_all_single(out.view(-1), rows.view(-1), group=group, async_op=async_op)
    return (out, work) if async_op else out


def merge_exchanged(recv, merge_fn):
    """Fold the `world` row sets of this rank's read range with merge_fn(a, b) -> a (+) b."""
    cur = recv[0]
    for r in range(1, recv.shape[0]):
        cur = merge_fn(cur, recv[r])
    return cur


ROW_INVALID = -1   # MIC_ROW_INVALID (0xFFFFFFFF) as int32


def overflowed_reads(merged_rows, world, rank, n_reads, group=None):
    """Reads of this rank's range whose merged sparse row does not fit (row[0] == MIC_ROW_INVALID), as GLOBAL read
    ids gathered from every rank: (all_ids [total], offsets [world + 1]) - rank r's reads are all_ids[offsets[r]:offsets[r+1]].
    The reference truncates such rows (CuClarkDB.cu:1200-1211); here they are completed exactly (complete_overflowed)."""
    lo, hi, per = read_range(n_reads, world, rank)
    bad = (merged_rows[: max(hi - lo, 0), 0] == ROW_INVALID).nonzero().flatten().to(torch.int64) + lo
    n_mine = torch.tensor([bad.numel()], dtype=torch.int64, device=merged_rows.device)
    counts = [torch.zeros_like(n_mine) for _ in range(world)]
    dist.all_gather(counts, n_mine, group=group)
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
    dist.all_gather(gathered, pad, group=group)
    return torch.cat([g[:c] for g, c in zip(gathered, counts)]), offsets


def complete_overflowed(merged_rows, world, rank, n_reads, count_dense_fn, group=None):
    """Exact completion of the reads whose merged row overflowed: every rank counts those reads densely against ITS
    bucket range (count_dense_fn(global_ids) -> int32 [n, T]), the counts are summed over the ranks (all_reduce), and
    each rank gets back (local row indices, dense counts) of the reads it owns.  One small exchange; flagged reads only."""
    ids, offsets = overflowed_reads(merged_rows, world, rank, n_reads, group)
    if ids.numel() == 0:
        return None, None
    counts = count_dense_fn(ids).contiguous()
    dist.all_reduce(counts, group=group)         # per-target counts are additive across shards (CuClarkDB.cu:1385-1388)
    lo = read_range(n_reads, world, rank)[0]
    mine = slice(offsets[rank], offsets[rank + 1])
    return (ids[mine] - lo), counts[mine]


def gather_results(res_part, world, group=None):
    """[per, 8] per rank -> [world*per, 8] on every rank, in read order."""
    out = torch.empty((world * res_part.shape[0], res_part.shape[1]), dtype=res_part.dtype, device=res_part.device)
    dist.all_gather_into_tensor(out, res_part.contiguous(), group=group)
    return out


class ShardedPass:
    """One table-sharded pass over `n_reads` reads starting at read `first`, on the P ranks of `group` (each holds one part
    of the table).  The reads are cut into `chunks`; per chunk: query (all reads of the chunk against this rank's part) ->
    all_to_all of the sparse rows by read sub-range (asynchronous: it overlaps the next chunk's query) -> P-1 merges ->
    best / second -> completion of overflowed rows.  Group rank j ends up with the results of sub-range j of every chunk.

    ops (device work, supplied by the caller - bench.py binds them to an engine, the CPU tests to a reference implementation):
      query(first, count, rows)            rows[:count] <- sparse rows of reads [first, first + count) against this rank's part
      merge(a, b, out, n)                  out[:n] <- a (+) b   (mergeKernel, CuClarkDB.cu:1321-1415)
      result(rows, res, n)                 res[:n] <- best / second of rows   (resultKernel, CuClarkDB.cu:1421-1471)
      count_dense(ids) -> [len(ids), T]    dense counts of the reads `ids` (global ids) against this rank's part
      result_from_dense(counts, idx, res)  res[idx] <- best / second of dense counts
      staged: rows travel through host memory (gloo)"""

    def __init__(self, ops, first, n_reads, row_words, device, group, group_size, group_rank, chunks=4, staged=False):
        self.ops, self.first, self.n, self.rw, self.dev = ops, first, n_reads, row_words, device
        self.group, self.P, self.j, self.staged = group, group_size, group_rank, staged
        chunks = max(1, min(chunks, max(1, n_reads // max(group_size, 1))))
        per_chunk = (n_reads + chunks - 1) // chunks
        self.cuts = [min(n_reads, c * per_chunk) for c in range(chunks + 1)]
        self.per = [read_range(self.cuts[c + 1] - self.cuts[c], self.P, 0)[2] for c in range(chunks)]
        self.rows = [torch.zeros((self.per[c] * self.P, row_words), dtype=torch.int32, device=device) for c in range(chunks)]
        self.recv = [torch.zeros((self.P, self.per[c], row_words), dtype=torch.int32, device=device) for c in range(chunks)]
        self.acc = [torch.zeros((2, self.per[c], row_words), dtype=torch.int32, device=device) for c in range(chunks)]
        self.res = [torch.zeros((self.per[c], 8), dtype=torch.int32, device=device) for c in range(chunks)]
        self.completed = 0

    def _finish_chunk(self, c, work):
        if work is not None:
            work.wait()
        recv, per = self.recv[c], self.per[c]
        cur = recv[0]
        for r in range(1, self.P):
            out = self.acc[c][r & 1]
            self.ops["merge"](cur, recv[r], out, per)
            cur = out
        self.ops["result"](cur, self.res[c], per)
        n_c = self.cuts[c + 1] - self.cuts[c]
        base = self.first + self.cuts[c]
        rows_view = cur.cpu() if self.staged else cur
        idx, counts = complete_overflowed(rows_view, self.P, self.j, n_c, lambda ids: self.ops["count_dense"](ids + base), self.group)
        if idx is not None and idx.numel():
            self.ops["result_from_dense"](counts, idx, self.res[c])
            self.completed += int(idx.numel())

    def step(self):
        self.completed = 0
        pending = None
        for c in range(len(self.per)):
            n_c = self.cuts[c + 1] - self.cuts[c]
            self.ops["query"](self.first + self.cuts[c], n_c, self.rows[c])
            if self.staged:
                if torch.device(self.dev).type == "cuda":
                    torch.cuda.synchronize()
                self.recv[c].copy_(exchange_rows(self.rows[c].cpu(), self.P, group=self.group))
                work = None
            else:
                _, work = exchange_rows(self.rows[c], self.P, out=self.recv[c], group=self.group, async_op=True)
            if pending is not None:
                self._finish_chunk(*pending)         # chunk c-1: its exchange ran under chunk c's query
            pending = (c, work)
        self._finish_chunk(*pending)

    def gather(self):
        """results of all reads of the pass, in read order, on every rank of the group: [n_reads, 8]"""
        out = []
        for c in range(len(self.per)):
            part = self.res[c].cpu() if self.staged else self.res[c]
            g = gather_results(part, self.P, self.group)
            out.append(g[: self.cuts[c + 1] - self.cuts[c]].to(self.dev))
        return torch.cat(out)
