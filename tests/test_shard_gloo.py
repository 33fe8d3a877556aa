"""CPU suite: the N > 1 path (scan / pair sharding + descriptor all-gather) on world_size 2, gloo."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_bounds_cover_everything():
    from mr_slam_amd import shard
    for n in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            b = [shard.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == shard.shard_sizes(n, world)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mr_slam_amd import shard
    from oracle import corr_oracle as K
    rng = np.random.default_rng(0)                      # same data on both ranks
    n_db = 5
    db = rng.normal(size=(n_db, 1, 120, 120)).astype(np.float32)
    db = np.stack([K.ring_normalize(x).numpy() for x in db])
    q = np.stack([np.roll(db[3], 9, axis=1), db[1]])
    lo, hi = shard.shard_bounds(n_db, rank, world)      # ragged: 3 + 2
    local = torch.from_numpy(db[lo:hi])
    full = shard.allgather_ragged(local)
    assert torch.equal(full, torch.from_numpy(db))
    # the exchange format of the GPU path: fp16 replicas travel, the owner keeps fp32
    rep = shard.allgather_ragged(local.to(torch.float16))
    assert rep.dtype == torch.float16 and torch.equal(rep, torch.from_numpy(db).to(torch.float16))

    def sweep(qq, dd):                                  # CPU stand-in for ring.corr_sweep (the checker)
        D = torch.zeros((qq.shape[0], dd.shape[0])); A = torch.zeros((qq.shape[0], dd.shape[0]), dtype=torch.int32)
        for i in range(qq.shape[0]):
            a = torch.fft.fft2(qq[i], dim=-2, norm="ortho")
            for j in range(dd.shape[0]):
                b = torch.fft.fft2(dd[j], dim=-2, norm="ortho")
                d, g, _ = K.fast_corr(a, b)
                D[i, j] = float(d); A[i, j] = g
        return D, A

    d_sh, a_sh = shard.sharded_sweep(torch.from_numpy(q), local, sweep)
    d_full, a_full = sweep(torch.from_numpy(q), torch.from_numpy(db))
    assert torch.equal(d_sh, d_full) and torch.equal(a_sh, a_full)
    assert int(a_full[0, 3]) == -9 and int(torch.argmin(d_full[1])) == 1
    # top-k only exchange over the sharded database == single-rank sweep + stable sort
    dk, ak, rk = shard.sharded_topk_sweep(torch.from_numpy(q), local, sweep, 3)
    order = torch.argsort(d_full, dim=1, stable=True)[:, :3]
    assert torch.equal(dk, torch.gather(d_full, 1, order)) and torch.equal(ak, torch.gather(a_full, 1, order))
    assert torch.equal(rk, order)
    # static-shape form (no exchange of the shard sizes, one packed collective): the same numbers
    sizes_now = [int(v) for v in shard.allgather_ragged(torch.tensor([local.shape[0]]))]
    dkp, akp, rkp = shard.sharded_topk_sweep(torch.from_numpy(q), local, sweep, 3, shard_rows=sizes_now, packed=True)
    assert torch.equal(dkp, dk) and torch.equal(akp, ak) and torch.equal(rkp, rk)
    dk9, _, rk9 = shard.sharded_topk_sweep(torch.from_numpy(q), local, sweep, 9)     # k larger than the database
    assert torch.isinf(dk9[:, n_db:]).all() and (rk9[:, n_db:] == -1).all() and torch.equal(rk9[:, :n_db], torch.argsort(d_full, dim=1, stable=True))
    # owner re-scoring: replicas are a lossy copy; results within the margin of the threshold are recomputed by the
    # row's owner on its exact entry.  Stand-in scores: exact = |q - entry|, replica = exact + a known error.
    exact_db = torch.arange(10, dtype=torch.float32)                 # rows 0..4 owned by rank 0, 5..9 by rank 1
    own = exact_db[5 * rank: 5 * rank + 5]
    qv = torch.tensor([2.3005, 7.5, 6.3, 0.1], dtype=torch.float32) + rank   # different queries per rank
    cand_row = torch.tensor([2, 8, 6, 9])
    exact = (qv - exact_db[cand_row]).abs()
    err = torch.tensor([1.5e-3, -1.0e-3, 0.0, 1e-3])
    replica = exact + err
    thr = float(exact[0]) + 1e-3                                      # entry 0: replica above, exact below the threshold
    rs = shard.OwnerRescorer(thr, margin=2e-3, slots=3)
    got_d, got_a = rs.rescore(replica, torch.zeros(4, dtype=torch.int32), cand_row, qv[:, None],
                              lambda rows: rows // 5, lambda rows: rows % 5,
                              lambda qq, lr: ((qq[:, 0] - own[lr]).abs(), torch.full((qq.shape[0],), 7, dtype=torch.int32)))
    amb = (replica - thr).abs() < 2e-3
    assert amb[0] and torch.equal(got_d[amb], exact[amb]) and torch.equal(got_d[~amb], replica[~amb])
    assert (got_a[amb] == 7).all() and rs.stats["flipped"] >= 1 and rs.stats["rounds"] == 1
    # more ambiguous candidates than slots: further rounds until no rank has any left (nothing is decided on a replica score)
    assert int(amb.sum()) == 2
    rs1 = shard.OwnerRescorer(thr, margin=2e-3, slots=1)
    got_d1, got_a1 = rs1.rescore(replica, torch.zeros(4, dtype=torch.int32), cand_row, qv[:, None],
                                 lambda rows: rows // 5, lambda rows: rows % 5,
                                 lambda qq, lr: ((qq[:, 0] - own[lr]).abs(), torch.full((qq.shape[0],), 7, dtype=torch.int32)))
    assert torch.equal(got_d1, got_d) and torch.equal(got_a1, got_a) and rs1.stats["rounds"] == 2 and rs1.stats["requested"] == 2
    # empty shard on one rank
    e = shard.allgather_ragged(torch.zeros((0 if rank else 2, 3)))
    assert e.shape == (2, 3)
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_allgather_and_sharded_sweep_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _fetch_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mr_slam_amd import shard
    rng = np.random.default_rng(7)                                       # the same database on every rank, sharded raggedly
    sizes = [5, 0, 3][:world] if world == 3 else [4, 6]
    n_db = sum(sizes)
    db = torch.from_numpy(rng.normal(size=(n_db, 3, 2)).astype(np.float32))
    cdb = torch.view_as_complex(torch.from_numpy(rng.normal(size=(n_db, 4, 2)).astype(np.float32)))   # complex64 like the spectra
    lo = sum(sizes[:rank])
    local, clocal = db[lo:lo + sizes[rank]].clone(), cdb[lo:lo + sizes[rank]].clone()
    rr = np.random.default_rng(100 + rank)                               # different requests per rank: repeats, own rows, remote rows
    want = torch.from_numpy(rr.integers(0, n_db, size=11 + 3 * rank))
    got = shard.fetch_rows(local, want, sizes)
    assert got.shape == (want.numel(), 3, 2) and torch.equal(got, db[want])
    cgot = shard.fetch_rows(clocal, want, sizes)
    assert cgot.dtype == torch.complex64 and torch.equal(torch.view_as_real(cgot), torch.view_as_real(cdb[want]))
    # the same with the request phase done ahead of time: one all-to-all per fetch, also asynchronously, plan reused
    plan = shard.RowFetchPlan(want, sizes)
    assert torch.equal(plan.fetch(local), db[want])
    work, finish = plan.fetch(clocal, async_op=True)
    if work is not None:
        work.wait()
    assert torch.equal(torch.view_as_real(finish()), torch.view_as_real(cdb[want]))
    assert torch.equal(plan.fetch(local * 2), db[want] * 2)
    own = int(((want >= lo) & (want < lo + sizes[rank])).sum())
    assert plan.bytes_in(24) == 24 * (want.numel() - own)
    # a rank that asks for nothing still takes part in the collectives
    none = shard.fetch_rows(local, torch.zeros(0, dtype=torch.int64) if rank == 0 else want[:2], sizes)
    assert none.shape[0] == (0 if rank == 0 else 2) and (rank == 0 or torch.equal(none, db[want[:2]]))
    # only rows of one owner
    owner0 = torch.arange(sizes[0]).flip(0)
    assert torch.equal(shard.fetch_rows(local, owner0, sizes), db[owner0])
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_fetch_rows_returns_the_owners_rows_in_request_order(tmp_path, world):
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_fetch_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_fetch_rows_single_process():
    from mr_slam_amd import shard
    db = torch.arange(12, dtype=torch.float32).view(6, 2)
    rows = torch.tensor([5, 0, 0, 3])
    assert torch.equal(shard.fetch_rows(db, rows, [6]), db[rows])
    with pytest.raises(AssertionError):
        shard.fetch_rows(db, torch.tensor([6]), [6])
