"""Rehearsal of the sharded LU with the REAL HIP building blocks: 2 ranks sharing the one GPU
of the test box, gloo backend, the panel broadcast staged through the host (RCCL refuses two
ranks on one device; on the 8-GPU node bench.py uses backend "nccl" = RCCL over xGMI)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, nb, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from linalg_solver_amd import gen
        from linalg_solver_amd.device import DeviceSolver
        from linalg_solver_amd.dist import ShardedLU

        torch.cuda.set_device(0)
        dev = DeviceSolver(0)

        def bcast_via_host(t, src):
            h = t.cpu()
            dist.broadcast(h, src=src)
            t.copy_(h)

        slu = ShardedLU(dev, n, nb, rank, world, bcast=bcast_via_host)
        A = slu.fill(gen.U11, 33)
        ipiv, info = slu.factor_(A)
        torch.cuda.synchronize()
        full = torch.zeros((n, n), dtype=torch.float64)
        for b in slu.my_blocks:
            o, w = slu.offset[b], slu.widths[b]
            full[:, b * nb:b * nb + w] = A[:, o:o + w].cpu()
        dist.all_reduce(full)
        if rank == 0:
            np.savez(os.path.join(outdir, "sharded.npz"), LU=full.numpy(), ipiv=ipiv.cpu().numpy(),
                     info=info.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,nb", [(1024, 128), (1000, 128), (640, 64)])
def test_two_ranks_one_gpu_match_single_gpu_bit_for_bit(tmp_path, n, nb):
    mp.spawn(_worker, args=(2, _free_port(), n, nb, str(tmp_path)), nprocs=2, join=True)
    z = np.load(tmp_path / "sharded.npz")
    import linalg_solver_amd as la
    from linalg_solver_amd import dense, gen

    h = la.default_handle()
    h.set_option("nb", nb)
    try:
        LU, ipiv, info = dense.lu_factor(gen.fill(gen.U11, 33, n, n))
    finally:
        h.set_option("nb", 128)
    assert info == 0 and int(z["info"][0]) == 0
    assert np.array_equal(z["ipiv"], ipiv)
    # same kernels, same k-order in every dot product: the shards reproduce the single-GPU bits
    assert np.array_equal(z["LU"], LU)
