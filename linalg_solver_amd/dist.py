"""Multi-GPU LU: the trailing update sharded across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  The
matrix is distributed 1-D block-cyclic by COLUMNS: column
block b (width nb) lives on rank b % P, each rank holding all n rows of its
blocks as one row-major local matrix.  Per block step:

    owner      factors the panel (pivot search needs whole columns: they are local)
    owner -->  ONE broadcast: the factored panel [(n-k) x jb] + its jb pivots
    everyone   applies the interchanges to its other columns, solves its slice of
               U12 with L11, and updates its slice of A22 -= L21 * U12 (MFMA)

No reduction is needed: the trailing update is embarrassingly parallel over
column blocks; the only exchange is the panel broadcast, which the reference has
no counterpart for (it is single-threaded).

The local kernels are reached through an ``ops`` object.  The product passes
``linalg_solver_amd.device.DeviceSolver`` (HIP kernels via the C ABI); the CPU
tests (gloo, world_size 2) pass a numpy stand-in built on the oracle so that
the distribution logic is exercised without a GPU.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class ShardedLU:
    def __init__(self, ops, n: int, nb: int, rank: int, world: int, dtype=torch.float64, device=None,
                 group=None, bcast=None):
        if nb < 1 or n < 1:
            raise ValueError("n and nb must be positive")
        if dtype != torch.float64:
            raise TypeError("the sharded driver is fp64 (BASELINE config #4)")
        self.ops, self.n, self.nb, self.rank, self.world = ops, n, nb, rank, world
        self.dtype = dtype
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                                         if torch.cuda.is_available() else torch.device("cpu"))
        self.group = group
        # the exchange primitive: RCCL broadcast by default; tests may stage through the host
        self._bcast = bcast if bcast is not None else (lambda t, src: dist.broadcast(t, src=src, group=self.group))
        self.nblocks = (n + nb - 1) // nb
        # global block ids owned by this rank, their widths and local column offsets
        self.my_blocks: List[int] = [b for b in range(self.nblocks) if b % world == rank]
        self.widths = {b: min(nb, n - b * nb) for b in self.my_blocks}
        self.offset = {}
        off = 0
        for b in self.my_blocks:
            self.offset[b] = off
            off += self.widths[b]
        self.local_cols = off
        # broadcast buffer: [gather list: 512 int32 = 256 T | has-list flag | info | pivots (as T) | panel rows]
        self._buf = torch.zeros((258 + nb + n * nb,), dtype=dtype, device=self.device)
        self._moves = self._buf[:256].view(torch.int32)  # raw int32 view of the first 2 KB
        self._fast_swaps = hasattr(ops, "panel_moves_") and hasattr(ops, "laswp_moves_")

    # -- distribution helpers -------------------------------------------------
    def owner(self, b: int) -> int:
        return b % self.world

    def empty_local(self) -> torch.Tensor:
        return torch.empty((self.n, max(self.local_cols, 1)), dtype=self.dtype, device=self.device)

    def fill(self, kind: int, seed: int) -> torch.Tensor:
        """This rank's columns of the synthetic matrix (same generator as the single-GPU path)."""
        A = self.empty_local()
        for b in self.my_blocks:
            o, w = self.offset[b], self.widths[b]
            self.ops.fill_(A[:, o:o + w], kind, seed, 0, b * self.nb)
        return A

    def scatter_from(self, full: torch.Tensor) -> torch.Tensor:
        """Take this rank's column blocks out of a replicated full matrix (tests)."""
        A = self.empty_local()
        for b in self.my_blocks:
            o, w = self.offset[b], self.widths[b]
            A[:, o:o + w] = full[:, b * self.nb:b * self.nb + w]
        return A

    def gather_to_full(self, A: torch.Tensor) -> torch.Tensor:
        """All-gather the distributed matrix into a replicated full one (tests / small n)."""
        full = torch.zeros((self.n, self.n), dtype=self.dtype, device=self.device)
        for b in self.my_blocks:
            o, w = self.offset[b], self.widths[b]
            full[:, b * self.nb:b * self.nb + w] = A[:, o:o + w]
        dist.all_reduce(full, group=self.group)  # disjoint supports: the sum is the union
        return full

    def _has_list_host(self, buf, own) -> bool:
        """Whether the broadcast carried a gather list.  The flag is known on the host without a
        device read: the owner's panel kernel either always or never emits one for a given shape,
        and every rank runs the same library -- so ask the local ops object once per shape."""
        key = "list"
        if not hasattr(self, "_list_cache"):
            self._list_cache = {}
        if key not in self._list_cache:
            self._list_cache[key] = bool(buf[256].item() > 0.5)  # one sync, first step only
        return self._list_cache[key]

    def _first_local_block_after(self, b: int) -> Optional[int]:
        for mb in self.my_blocks:
            if mb > b:
                return mb
        return None

    # -- the factorisation ------------------------------------------------------
    def factor_(self, A: torch.Tensor):
        """In-place P A = L U of the distributed matrix.  Returns (ipiv, info): ipiv is the
        full LAPACK-style interchange list (replicated on every rank, device tensor int32)."""
        n, nb = self.n, self.nb
        ops = self.ops
        ipiv = torch.zeros(n, dtype=torch.int32, device=self.device)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._list_cache = {}
        for b in range(self.nblocks):
            k = b * nb
            jb = min(nb, n - k)
            m = n - k
            own = self.owner(b)
            buf = self._buf[: 258 + jb + m * jb]
            panel = buf[258 + jb:].view(m, jb)
            if own == self.rank:
                o = self.offset[b]
                P = A[k:, o:o + jb]
                ops.panel_(P, k, ipiv[k:k + jb], info)
                panel.copy_(P)
                buf[258:258 + jb].copy_(ipiv[k:k + jb])   # int32 -> T in one copy kernel
                buf[257:258].copy_(info)
                has_list = self._fast_swaps and ops.panel_moves_(self._moves)
                buf[256:257].fill_(1.0 if has_list else 0.0)
            # the one exchange of the step: factored panel + pivots (+ gather list), owner -> everyone
            self._bcast(buf, own)
            if own != self.rank:
                ipiv[k:k + jb].copy_(buf[258:258 + jb])   # exact: row indices < 2^53
                info.copy_(buf[257:258])
            # interchanges on this rank's other columns (the owner's panel is already swapped)
            piv = ipiv[k:k + jb]
            use_list = self._fast_swaps and self._has_list_host(buf, own)

            def swap_rows(view):
                if use_list:
                    ops.laswp_moves_(view, k, self._moves)
                else:
                    ops.laswp_(view, k, jb, piv)
            nxt = self._first_local_block_after(b)
            left_cols = self.offset[b] if own == self.rank else (self.offset[nxt] if nxt is not None
                                                                  else self.local_cols)
            if left_cols > 0:
                swap_rows(A[:, :left_cols])
            if nxt is not None:
                ro = self.offset[nxt]
                right = A[:, ro:self.local_cols]
                swap_rows(right)
                # U12 slice and trailing update of this rank's columns right of the panel
                U12 = A[k:k + jb, ro:self.local_cols]
                ops.trsm_lu_(panel[:jb, :], U12)
                if m > jb:
                    ops.gemm_sub_(A[k + jb:, ro:self.local_cols], panel[jb:, :], U12)
        return ipiv, info
