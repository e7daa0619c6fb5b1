"""Multi-process (gloo, world_size 2, CPU) tests of the sharded-table plumbing of
tlsan_amd/dist.py: mod-G partition, id routing, row fetch, gradient push.  The arithmetic on the
rows is HIP-only and covered by the gpu-marked tests; here the checker is plain numpy."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tlsan_amd.dist import ModPartition, RowExchange
        n, width = 37, 6
        rng = np.random.RandomState(0)
        table = rng.randn(n, width).astype(np.float32)           # the global table (same on all ranks)
        part = ModPartition(n, world)
        shard = torch.as_tensor(table[rank::world].copy())
        assert shard.shape[0] == part.local_count(rank)
        x = RowExchange(part)
        rs = np.random.RandomState(100 + rank)
        ids = np.unique(rs.randint(0, n, 25)).astype(np.int64)    # sorted unique, rank specific
        plan = x.plan(torch.as_tensor(ids))
        got = x.fetch(plan, shard).numpy()
        assert np.array_equal(got, table[ids])                     # rows arrive in uniq order
        # push: every rank sends value rows for its ids; owners get (local_row, value) pairs
        vals = (np.arange(len(ids))[:, None] + 1000.0 * rank + np.zeros((1, 3))).astype(np.float32)
        rows, recv = x.push(plan, torch.as_tensor(vals))
        # reference: gather everybody's (id, value) on every rank and keep what this rank owns
        all_ids = [None] * world
        all_vals = [None] * world
        dist.all_gather_object(all_ids, ids)
        dist.all_gather_object(all_vals, vals)
        exp = np.zeros((part.local_count(rank), 3), np.float64)
        for src in range(world):
            for i, v in zip(all_ids[src], all_vals[src]):
                if i % world == rank:
                    exp[i // world] += v
        acc = np.zeros_like(exp)
        np.add.at(acc, rows.numpy(), recv.numpy().astype(np.float64))
        assert np.allclose(acc, exp)
        # contributions are concatenated in source-rank order (deterministic reduction order)
        src_of = (recv.numpy()[:, 0] // 1000).astype(int)
        assert np.all(np.diff(src_of) >= 0)
        # empty request from one rank
        plan0 = x.plan(torch.as_tensor(ids if rank == 0 else ids[:0]))
        got0 = x.fetch(plan0, shard)
        assert got0.shape[0] == (len(ids) if rank == 0 else 0)
        ret[rank] = "ok"
    except Exception as e:  # pragma: no cover
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_row_exchange_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_mod_partition():
    from tlsan_amd.dist import ModPartition
    p = ModPartition(10, 4)
    assert [p.local_count(r) for r in range(4)] == [3, 3, 2, 2]
    ids = torch.arange(10)
    assert torch.equal(p.owner(ids), ids % 4) and torch.equal(p.local_row(ids), ids // 4)
    assert p.global_ids(1).tolist() == [1, 5, 9]


def _worker_router(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tlsan_amd.dist import KeyRouter, torch_scan
        I, U, W = 23, 17, 5
        rng = np.random.RandomState(0)
        items = rng.randn(I, W).astype(np.float32)
        users = rng.randn(U, W).astype(np.float32)
        r = KeyRouter(I, U, world, rank)
        shard = np.zeros((r.R, W), np.float32)
        gi, gu = np.arange(rank, I, world), np.arange(rank, U, world)
        shard[:len(gi)] = items[gi]
        shard[r.cI:r.cI + len(gu)] = users[gu]
        shard_t = torch.as_tensor(shard)
        rs = np.random.RandomState(10 + rank)
        it = torch.as_tensor(rs.randint(0, I, 40))
        us = torch.as_tensor(rs.randint(0, U, 9))
        keys = torch.cat([r.item_keys(it), r.user_keys(us)])
        plan = r.plan(keys, torch_scan)
        table = r.fetch(plan, shard_t).numpy()
        comp = plan["prefix"][keys].numpy()
        # every id of the batch finds its own row in the compact table
        assert np.array_equal(table[comp[:40]], items[it.numpy()])
        assert np.array_equal(table[comp[40:]], users[us.numpy()])
        assert plan["n"] == len(np.unique(keys.numpy()))
        # push: per-row values come back to their owners exactly once per requesting rank
        vals = torch.as_tensor(np.arange(plan["n"], dtype=np.float32)[:, None] + 100.0 * rank + np.zeros((1, 2), np.float32))
        rows, recv = r.push(plan, vals)
        all_keys = [None] * world
        dist.all_gather_object(all_keys, plan["uniq"].numpy())
        exp_rows = np.concatenate([k[k // r.R == rank] % r.R for k in all_keys])
        assert np.array_equal(rows.numpy().astype(np.int64), exp_rows.astype(np.int64))   # source-rank order
        assert recv.shape[0] == len(exp_rows)
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_key_router_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_router, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)
