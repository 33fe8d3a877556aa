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
