"""world_size-2 gloo test (CPU) of the multi-GPU layer: view sharding and the single packed all-reduce."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deodr_amd.distributed import PackedGradients, allreduce_shared_gradients, shard_views


def test_shard_views_partition():
    for n, w in [(8, 1), (8, 2), (8, 8), (7, 4), (3, 8), (64, 8)]:
        got = [i for r in range(w) for i in shard_views(n, r, w)]
        assert got == list(range(n))
        sizes = [len(shard_views(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V, C, n_views = 11, 3, 5
    rs = np.random.RandomState(0)
    per_view_ij = rs.randn(n_views, V, 2)
    per_view_col = rs.randn(n_views, V, C)
    mine = list(shard_views(n_views, rank, world))
    ij_b = torch.as_tensor(per_view_ij[mine].sum(0)) if mine else torch.zeros(V, 2, dtype=torch.float64)
    col_b = torch.as_tensor(per_view_col[mine].sum(0)) if mine else torch.zeros(V, C, dtype=torch.float64)
    packed = PackedGradients([(V, 2), (V, C)])
    tot_ij, tot_col = allreduce_shared_gradients(packed, [ij_b, col_b])
    ok = np.allclose(tot_ij.numpy(), per_view_ij.sum(0)) and np.allclose(tot_col.numpy(), per_view_col.sum(0))
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write(str(int(ok)))
    dist.destroy_process_group()


def test_packed_allreduce_gloo_world2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(2)] == ["1", "1"]


def test_pack_unpack_roundtrip():
    a, b = torch.arange(6.0, dtype=torch.float64).reshape(3, 2), torch.arange(4.0, dtype=torch.float64)
    p = PackedGradients([a.shape, b.shape])
    x, y = allreduce_shared_gradients(p, [a, b])  # no process group: plain pack / unpack
    assert torch.equal(x, a) and torch.equal(y, b)
