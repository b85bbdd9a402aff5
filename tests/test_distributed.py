"""world_size-2 tests on gloo (CPU) of the multi-GPU plumbing (cuclark_amd/multi.py).  The per-shard sparse rows that
the HIP kernel would produce are computed here by the oracle with the same shard filter (CuClarkDB.cu:1272-1274); the
exchange, the ownership of read ranges, the fold order and the final gather are the product code under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows_u32(o, counts, row_words):
    rows = np.zeros((counts.shape[0], row_words), np.uint32)
    for r in range(counts.shape[0]):
        n, row = o.sparse_row(counts[r], row_words - 1)
        assert n <= row_words - 1
        rows[r, 0] = n
        rows[r, 1:1 + n] = (row[2:2 + 2 * n:2].astype(np.uint32) << 16) | row[1:1 + 2 * n:2]
    return rows


def _merge_u32(a, b):
    """reference merge (sum by target) on the product's u32 row format, numpy."""
    out = np.zeros_like(a)
    for r in range(a.shape[0]):
        d = {}
        for row in (a[r], b[r]):
            for v in row[1:1 + row[0]]:
                d[int(v) & 0xFFFF] = d.get(int(v) & 0xFFFF, 0) + (int(v) >> 16)
        out[r, 0] = len(d)
        for i, t in enumerate(sorted(d)):
            out[r, 1 + i] = (d[t] << 16) | t
    return out


INVALID = 0xFFFFFFFF


def _rows_u32_or_invalid(o, counts, row_words):
    """the query kernel's rows: a read with more targets than the row holds carries MIC_ROW_INVALID"""
    rows = np.zeros((counts.shape[0], row_words), np.uint32)
    for r in range(counts.shape[0]):
        nz = np.nonzero(counts[r])[0]
        if nz.size > row_words - 1:
            rows[r, 0] = INVALID
            continue
        rows[r, 0] = nz.size
        rows[r, 1:1 + nz.size] = (counts[r][nz].astype(np.uint32) << 16) | nz.astype(np.uint32)
    return rows


def _merge_u32_or_invalid(a, b):
    """merge_rows_kernel's rule: invalid in, or too many targets out -> invalid"""
    out = np.zeros_like(a)
    for r in range(a.shape[0]):
        if a[r, 0] == INVALID or b[r, 0] == INVALID:
            out[r, 0] = INVALID
            continue
        d = {}
        for row in (a[r], b[r]):
            for v in row[1:1 + row[0]]:
                d[int(v) & 0xFFFF] = d.get(int(v) & 0xFFFF, 0) + (int(v) >> 16)
        if len(d) > a.shape[1] - 1:
            out[r, 0] = INVALID
            continue
        out[r, 0] = len(d)
        for i, t in enumerate(sorted(d)):
            out[r, 1 + i] = (d[t] << 16) | t
    return out


def _overflow_worker(rank, world, port, ret):
    """Table-sharded ranks, reads that hit 3 .. 40 targets with 15-entry rows: the rows of the crowded reads overflow (in a
    shard or only after the merge) and are completed exactly from dense counts summed over the ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuclark_amd import multi
        o = gu.oracle()
        rng = np.random.default_rng(21)                 # same data on every rank
        k, T, RW, htsize = 27, 40, 16, 4099
        sizes, keys, labels, canon = gu.random_db(rng, htsize, 3000, k, 8, T)
        odb = o.db_from_arrays(sizes, keys, labels, 1)
        by_label = {t: [c for c, l in zip(canon, labels) if l == t] for t in range(T)}
        recs = []
        for i, n_t in enumerate([3, 40, 14, 15, 16, 17, 30, 2, 25, 16, 9, 40]):
            parts = [gu.kmer_to_ascii(by_label[t][(i + j) % len(by_label[t])], k) for j, t in enumerate(rng.permutation(T)[:n_t])]
            recs.append(f">r{i}\n" + "N".join(parts) + "\n")
        data = "".join(recs).encode()
        ix = o.index_reads(data)
        rp, ct = o.pack_batch(data, ix["seq_s"], ix["seq_e"], ix["length"], k)
        n = rp.size - 1
        whole, _ = odb.query_batch(k, rp, ct, T)
        expect = o.result_from_counts(whole)
        assert (np.count_nonzero(whole, axis=1) > RW - 1).sum() >= 6
        s0, s1 = multi.shard_range(htsize, world, rank)
        counts, _ = odb.query_batch(k, rp, ct, T, part=(s0, s1))
        rows = multi.padded_rows(n, world, RW, "cpu")
        rows[:n] = torch.from_numpy(_rows_u32_or_invalid(o, counts, RW).view(np.int32))
        recv = multi.exchange_rows(rows, world)
        merged = multi.merge_exchanged(recv, lambda a, b: torch.from_numpy(
            _merge_u32_or_invalid(a.numpy().view(np.uint32), b.numpy().view(np.uint32)).view(np.int32)))
        lo, hi, per = multi.read_range(n, world, rank)
        m = merged.numpy().view(np.uint32)
        res = np.zeros((per, 8), np.uint32)
        for i in range(hi - lo):
            if m[i, 0] == INVALID:
                continue
            row16 = np.zeros(2 * RW, np.uint16)
            row16[0] = m[i, 0]
            row16[1:1 + 2 * m[i, 0]:2] = m[i, 1:1 + m[i, 0]] & 0xFFFF
            row16[2:2 + 2 * m[i, 0]:2] = m[i, 1:1 + m[i, 0]] >> 16
            res[i, :5] = o.result_from_row(row16)
        idx, dense = multi.complete_overflowed(merged, world, rank, n, lambda ids: torch.from_numpy(
            counts[ids.numpy()].astype(np.int32)))
        n_fixed = 0
        if idx is not None:
            res[idx.numpy(), :5] = o.result_from_counts(dense.numpy().view(np.uint32))
            n_fixed = int(idx.numel())
        allres = multi.gather_results(torch.from_numpy(res.view(np.int32)), world).numpy().view(np.uint32)
        t = torch.tensor([n_fixed])
        dist.all_reduce(t)
        ret[rank] = bool((allres[:n, :5] == expect).all()) and int(t.item()) == int((np.count_nonzero(whole, axis=1) > RW - 1).sum())
    finally:
        dist.destroy_process_group()


def _sharded_pass_worker(rank, world, port, ret):
    """cuclark_amd/multi.py: ShardedPass on two gloo ranks: reads cut into chunks, rows of every chunk exchanged, merged, finished,
    overflowed rows completed from dense counts - with the per-rank rows computed by the oracle under the product's partition
    (slot range of the resident table: oracle/part_rule.c) instead of the HIP kernel."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuclark_amd import multi
        o = gu.oracle()
        rng = np.random.default_rng(22)
        k, T, RW, htsize = 27, 40, 16, 4099
        sizes, keys, labels, canon = gu.random_db(rng, htsize, 3000, k, 8, T)
        odb = o.db_from_arrays(sizes, keys, labels, 1)
        by_label = {t: [c for c, l in zip(canon, labels) if l == t] for t in range(T)}
        recs = []
        for i, n_t in enumerate([3, 40, 14, 15, 16, 17, 30, 2, 25, 16, 9, 40, 1, 0, 5, 33, 2]):
            parts = [gu.kmer_to_ascii(by_label[t][(i + j) % len(by_label[t])], k) for j, t in enumerate(rng.permutation(T)[:n_t])]
            recs.append(f">r{i}\n" + ("N".join(parts) if parts else "ACGT" * 10) + "\n")
        data = "".join(recs).encode()
        ix = o.index_reads(data)
        rp, ct = o.pack_batch(data, ix["seq_s"], ix["seq_e"], ix["length"], k)
        n = rp.size - 1
        whole, _ = odb.query_batch(k, rp, ct, T)
        expect = o.result_from_counts(whole)
        mine, _ = odb.query_batch_slot_part(k, 20, True, 1000, rank, world, rp, ct, T)          # this rank's part of the table
        other, _ = odb.query_batch_slot_part(k, 20, True, 1000, 1 - rank, world, rp, ct, T)
        assert (mine + other == whole).all() and mine.sum() > 0 and other.sum() > 0

        def query(first, count, rows):
            rows[:count] = torch.from_numpy(_rows_u32_or_invalid(o, mine[first:first + count], RW).view(np.int32))

        def merge(a, b, out, nn):
            out[:nn] = torch.from_numpy(_merge_u32_or_invalid(a[:nn].numpy().view(np.uint32), b[:nn].numpy().view(np.uint32)).view(np.int32))

        def result(rows, res, nn):
            m = rows.numpy().view(np.uint32)
            for i in range(nn):
                if m[i, 0] == INVALID:
                    continue
                row16 = np.zeros(2 * RW, np.uint16)
                row16[0] = m[i, 0]
                row16[1:1 + 2 * m[i, 0]:2] = m[i, 1:1 + m[i, 0]] & 0xFFFF
                row16[2:2 + 2 * m[i, 0]:2] = m[i, 1:1 + m[i, 0]] >> 16
                res[i, :5] = torch.from_numpy(o.result_from_row(row16).astype(np.int64)).to(torch.int32)

        def result_from_dense(counts, idx, res):
            res[idx, :5] = torch.from_numpy(o.result_from_counts(counts.numpy().view(np.uint32)).astype(np.int64)).to(torch.int32)
        ops = dict(query=query, merge=merge, result=result, result_from_dense=result_from_dense,
                   count_dense=lambda ids: torch.from_numpy(mine[ids.numpy()].astype(np.int32)))
        ok = True
        for chunks in (1, 3):
            sp = multi.ShardedPass(ops, 0, n, RW, "cpu", None, world, rank, chunks=chunks, staged=False)
            sp.step()
            got = sp.gather().numpy().view(np.uint32)
            t = torch.tensor([sp.completed])
            dist.all_reduce(t)
            ok = ok and bool((got[:, :5] == expect).all()) and int(t.item()) == int((np.count_nonzero(whole, axis=1) > RW - 1).sum())
            ok = ok and len(sp.per) == chunks
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharded_pass_in_chunks():
    world = 2
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_pass_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_grid_of_parts_and_read_groups():
    from cuclark_amd import multi
    for world, parts in ((8, 8), (8, 2), (8, 1), (4, 2), (2, 2), (1, 1)):
        seen = set()
        for rank in range(world):
            p, g, ng, ranks = multi.grid(world, rank, parts)
            assert ng == world // parts and rank in ranks and len(ranks) == parts and ranks.index(rank) == p
            seen.add((p, g))
        assert len(seen) == world


def test_two_ranks_complete_overflowed_rows():
    world = 2
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _worker(rank, world, port, mode, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuclark_amd import multi
        o = gu.oracle()
        odb, meta = gu.oracle_db_from_golden("light_k31_u64")
        data = open(os.path.join(gu.GOLDEN, "reads_k31.fa"), "rb").read()
        ix = o.index_reads(data)
        rp, ct = o.pack_batch(data, ix["seq_s"], ix["seq_e"], ix["length"], 31)
        n, T, RW = rp.size - 1, 6, 16
        whole, _ = odb.query_batch(31, rp, ct, T)
        expect = o.result_from_counts(whole)
        if mode == "db":
            s0, s1 = multi.shard_range(meta["htsize"], world, rank)
            counts, _ = odb.query_batch(31, rp, ct, T, part=(s0, s1))
            rows = multi.padded_rows(n, world, RW, "cpu")
            rows[:n] = torch.from_numpy(_rows_u32(o, counts, RW).astype(np.int64)).to(torch.int32)
            recv = multi.exchange_rows(rows, world)
            merged = multi.merge_exchanged(recv, lambda a, b: torch.from_numpy(
                _merge_u32(a.numpy().view(np.uint32), b.numpy().view(np.uint32)).view(np.int32)))
            lo, hi, per = multi.read_range(n, world, rank)
            m = merged.numpy().view(np.uint32)
            res = np.zeros((per, 8), np.uint32)
            for i in range(hi - lo):
                row16 = np.zeros(2 * RW, np.uint16)
                row16[0] = m[i, 0]
                row16[1:1 + 2 * m[i, 0]:2] = m[i, 1:1 + m[i, 0]] & 0xFFFF
                row16[2:2 + 2 * m[i, 0]:2] = m[i, 1:1 + m[i, 0]] >> 16
                res[i, :5] = o.result_from_row(row16)
            allres = multi.gather_results(torch.from_numpy(res.view(np.int32)), world).numpy().view(np.uint32)
            ok = bool((allres[:n, :5] == expect).all())
        else:  # read-sharded: each rank classifies its own read range against the whole table; no collective
            lo, hi, per = multi.read_range(n, world, rank)
            mine = o.result_from_counts(whole[lo:hi])
            ok = bool((mine == expect[lo:hi]).all()) and (hi - lo) > 0
            t = torch.tensor([hi - lo])
            dist.all_reduce(t)
            ok = ok and int(t.item()) == n
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["db", "read"])
def test_two_ranks_gloo(mode):
    world = 2
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_ranges_cover_everything():
    from cuclark_amd import multi
    for H, W in ((1610612741, 8), (57777779, 4), (1009, 2), (7, 8)):
        edges = [multi.shard_range(H, W, r) for r in range(W)]
        assert edges[0][0] == 0 and max(e[1] for e in edges) == H
        for a, b in zip(edges[:-1], edges[1:]):
            assert a[1] == b[0] or b[0] >= H
    for n, W in ((10_000_000, 8), (131, 2), (5, 8)):
        got = sum(max(0, multi.read_range(n, W, r)[1] - multi.read_range(n, W, r)[0]) for r in range(W))
        assert got == n
