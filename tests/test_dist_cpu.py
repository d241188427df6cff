"""The sharded (multi-GPU) LU driver on CPU: gloo, world_size 2, 3 and 4 (look-ahead included: the next
owner factors and broadcasts its panel before it finishes the update; more blocks than ranks, fewer blocks
than ranks, ragged last block).

Checks the distribution logic of linalg_solver_amd/dist.py -- block-cyclic ownership,
local offsets, one panel broadcast per step, interchanges on the non-owned columns --
against the single-process partial-pivot twin of the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from linalg_solver_amd import gen
from linalg_solver_amd.dist import ShardedLU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, nb, kind, chunks=1, dist_block=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_ops import CpuOps
        from oracle import capi

        slu = ShardedLU(CpuOps(), n, nb, rank, world, device=torch.device("cpu"), chunks=chunks, dist_block=dist_block)
        # every column block is owned exactly once
        owned = torch.zeros(slu.nblocks, dtype=torch.int32)
        for b in slu.my_blocks:
            owned[b] = 1
        dist.all_reduce(owned)
        assert torch.all(owned == 1)
        A = slu.fill(kind, 21)
        full0 = slu.gather_to_full(A).numpy().copy()
        assert np.array_equal(full0, gen.fill(kind, 21, n, n)), "distributed fill differs from the generator"
        ipiv, info = slu.factor_(A)
        LU = slu.gather_to_full(A).numpy()
        oLU, oipiv, oinfo = capi.getrf(full0)
        assert int(info[0]) == oinfo == 0
        assert np.array_equal(ipiv.numpy(), oipiv), "pivot sequence differs from the single-process twin"
        assert np.max(np.abs(LU - oLU)) <= 1e-11 * max(1.0, np.max(np.abs(oLU)))
        # every rank holds the same interchange list
        chk = ipiv.clone()
        dist.broadcast(chk, src=0)
        assert torch.equal(chk, ipiv)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nb,kind", [(2, 96, 32, gen.U11), (2, 200, 32, gen.U11), (2, 130, 64, gen.INT5),
                                             (3, 150, 16, gen.U11), (2, 40, 64, gen.U11),
                                             (4, 260, 32, gen.U11), (4, 97, 16, gen.INT5), (3, 64, 64, gen.U11)])
def test_sharded_lu_matches_single_process_twin(world, n, nb, kind):
    mp.spawn(_worker, args=(world, _free_port(), n, nb, kind), nprocs=world, join=True)


@pytest.mark.parametrize("world,n,nb,kind,chunks,db", [(2, 200, 32, gen.U11, 1, 2), (4, 260, 32, gen.U11, 3, 2),
                                                       (3, 150, 16, gen.INT5, 4, 3), (2, 40, 64, gen.U11, 4, 2),
                                                       (4, 97, 16, gen.INT5, 1, 2), (2, 130, 64, gen.U11, 2, 4)])
def test_sharded_lu_with_wider_distribution_blocks(world, n, nb, kind, chunks, db):
    """dist_block consecutive column blocks per owner (every dist_block-th step only has a broadcast in front of the
    next panel): same factors and pivots as the single-process twin, ragged orders and fewer groups than ranks
    included."""
    mp.spawn(_worker, args=(world, _free_port(), n, nb, kind, chunks, db), nprocs=world, join=True)


@pytest.mark.parametrize("world,n,nb,kind,chunks", [(2, 200, 32, gen.U11, 4), (4, 260, 32, gen.U11, 3),
                                                    (3, 150, 16, gen.INT5, 4), (2, 40, 64, gen.U11, 4)])
def test_sharded_lu_with_the_panel_sent_in_row_chunks(world, n, nb, kind, chunks):
    """The chunked exchange (header + top rows first, the update consuming each row range as it lands): same
    factors and pivots as the single-process twin, at world 2, 3 and 4."""
    mp.spawn(_worker, args=(world, _free_port(), n, nb, kind, chunks), nprocs=world, join=True)


def test_block_cyclic_layout_arithmetic():
    slu = ShardedLU(object(), 1000, 128, 1, 4, device=torch.device("cpu"))
    assert slu.nblocks == 8 and slu.my_blocks == [1, 5]
    assert slu.widths == {1: 128, 5: 128} and slu.offset == {1: 0, 5: 128} and slu.local_cols == 256
    last = ShardedLU(object(), 1000, 128, 3, 4, device=torch.device("cpu"))
    assert last.my_blocks == [3, 7] and last.widths[7] == 1000 - 7 * 128 and last.local_cols == 128 + 104
    assert [ShardedLU(object(), 1000, 128, r, 4, device=torch.device("cpu")).local_cols for r in range(4)] == \
        [256, 256, 256, 232]
