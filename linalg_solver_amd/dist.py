"""Multi-GPU LU: the trailing update sharded across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  The
matrix is distributed 1-D block-cyclic by COLUMNS: column
block b (width nb) lives on rank b % P, each rank holding all n rows of its
blocks as one row-major local matrix.  Per block step:

    owner      factors the panel (pivot search needs whole columns: they are local)
    owner -->  ONE broadcast: the factored panel [(n-k) x jb] + its jb pivots
    everyone   applies the interchanges to its other columns, solves its slice of
               U12 with L11, and updates its slice of A22 -= L21 * U12 (MFMA)

Look-ahead (depth 1): the owner of panel b+1 updates that one column block first, factors it and starts
its broadcast, THEN finishes its share of update b; the other ranks post the receive for panel b+1 before
they start update b.  The panel factorisation (latency-bound, ~2 us per column) and the broadcast thus run
under the trailing update instead of in front of it.  Two broadcast buffers alternate.

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
                 group=None, bcast=None, chunks: int = 1, dist_block: int = 1):
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
        # dist_block consecutive column blocks share an owner (ScaLAPACK's distribution block > panel width): inside
        # such a group the chain panel -> next block's update -> next panel stays on one GPU, so only every
        # dist_block-th step has a broadcast in front of the next panel; the others send theirs beside it
        self.dist_block = max(1, int(dist_block))
        self.my_blocks: List[int] = [b for b in range(self.nblocks) if self.owner(b) == rank]
        self.widths = {b: min(nb, n - b * nb) for b in self.my_blocks}
        self.offset = {}
        off = 0
        for b in self.my_blocks:
            self.offset[b] = off
            off += self.widths[b]
        self.local_cols = off
        # broadcast buffers (two, alternating with the block step):
        # [gather list: 512 int32 = 256 T | has-list flag | info | pivots (as T) | panel rows]
        self._bufs = [torch.zeros((258 + nb + n * nb,), dtype=dtype, device=self.device) for _ in range(2)]
        self._fast_swaps = hasattr(ops, "panel_moves_") and hasattr(ops, "laswp_moves_")
        self._custom_bcast = bcast is not None
        # chunks > 1: the panel travels in that many row chunks (header + top rows first); a receiver starts its
        # interchanges and U12 behind the first and updates each row range as its chunk lands, so the transfer of
        # a 16 MB panel overlaps the update instead of preceding it (the C driver lsx_getrf_mg_f64 does the same)
        self.chunks = max(1, int(chunks))

    # -- distribution helpers -------------------------------------------------
    def owner(self, b: int) -> int:
        return (b // self.dist_block) % self.world

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

    def _has_list_host(self, buf) -> bool:
        """Whether the broadcasts carry a gather list.  The owner's panel kernel either always or never
        emits one for a given build, and every rank runs the same library -- so read the flag once per
        factorisation (one sync, first step only)."""
        if "list" not in self._list_cache:
            self._list_cache["list"] = bool(buf[256].item() > 0.5)
        return self._list_cache["list"]

    def _first_local_block_after(self, b: int) -> Optional[int]:
        for mb in self.my_blocks:
            if mb > b:
                return mb
        return None

    # -- the exchange ---------------------------------------------------------
    def _chunk_rows(self, m: int, jb: int):
        """Row ranges [r0, r1) of the chunks of an m-row panel: multiples of 64 rows, the first holds at least the
        top jb rows (L11)."""
        per = max(((m + self.chunks - 1) // self.chunks + 63) // 64 * 64, jb)
        out = []
        for c in range(self.chunks):
            r0, r1 = min(m, c * per), (m if c == self.chunks - 1 else min(m, (c + 1) * per))
            out.append((r0, r1))
        return out

    def _bcast_start(self, t: torch.Tensor, src: int, m: int = 0, jb: int = 0):
        """Start the broadcast of one step: a list with one work handle per chunk (None: done already).  RCCL:
        asynchronous (the transfers run on the collective's own stream beside the update kernels).  A
        caller-supplied exchange (tests, host staging) is blocking."""
        pieces = [t]
        if self.chunks > 1 and m > 0:
            hdr = 258 + jb
            pieces = [t[(0 if c == 0 else hdr + r0 * jb):hdr + r1 * jb] for c, (r0, r1) in enumerate(self._chunk_rows(m, jb))]
        works = []
        for piece in pieces:
            if piece.numel() == 0:
                works.append(None)
            elif self._custom_bcast:
                self._bcast(piece, src)
                works.append(None)
            else:
                works.append(dist.broadcast(piece, src=src, group=self.group, async_op=True))
        return works

    @staticmethod
    def _bcast_wait(work):
        for w in (work if isinstance(work, (list, tuple)) else [work]):
            if w is not None:
                w.wait()

    def _pack_panel(self, A, b, ipiv, info):
        """Owner: factor panel b in place and fill its broadcast buffer."""
        n, nb, ops = self.n, self.nb, self.ops
        k = b * nb
        jb = min(nb, n - k)
        m = n - k
        buf = self._bufs[b & 1][: 258 + jb + m * jb]
        o = self.offset[b]
        P = A[k:, o:o + jb]
        ops.panel_(P, k, ipiv[k:k + jb], info)
        buf[258 + jb:].view(m, jb).copy_(P)
        buf[258:258 + jb].copy_(ipiv[k:k + jb])   # int32 -> T in one copy kernel
        buf[257:258].copy_(info)
        has_list = self._fast_swaps and ops.panel_moves_(buf[:256].view(torch.int32))
        buf[256:257].fill_(1.0 if has_list else 0.0)
        return buf

    # -- the factorisation ------------------------------------------------------
    def factor_(self, A: torch.Tensor):
        """In-place P A = L U of the distributed matrix.  Returns (ipiv, info): ipiv is the
        full LAPACK-style interchange list (replicated on every rank, device tensor int32)."""
        n, nb = self.n, self.nb
        ops = self.ops
        ipiv = torch.zeros(n, dtype=torch.int32, device=self.device)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._list_cache = {}

        def shape(b):
            k = b * nb
            jb = min(nb, n - k)
            return k, jb, n - k

        def buf_of(b):
            k, jb, m = shape(b)
            return self._bufs[b & 1][: 258 + jb + m * jb]

        def unpack(b):
            """Non-owner: pivots and info of panel b out of its buffer."""
            k, jb, m = shape(b)
            buf = buf_of(b)
            ipiv[k:k + jb].copy_(buf[258:258 + jb])   # exact: row indices < 2^53
            info.copy_(buf[257:258])

        works = {}   # panel -> its chunk work handles still to be waited for (receivers)
        sends = {}   # buffer slot -> work handles of this rank's own broadcast from it, waited for only when the
                     # slot is used again: the owner goes on with the next block (and, inside a distribution
                     # group, the next panel) while its panel is still on the wire

        def start(bb):
            """Begin the broadcast of panel bb (send or receive) once its buffer slot is free."""
            self._bcast_wait(sends.pop(bb & 1, None) or [])
            kk, jj, mm = shape(bb)
            return self._bcast_start(buf_of(bb), self.owner(bb), mm, jj)

        def apply_panel(b, col0, col1):
            """Interchanges, U12 and trailing update of local columns [col0, col1) with panel b."""
            pend = works.get(b)
            if col1 <= col0:
                self._bcast_wait(pend or [])
                works.pop(b, None)
                return
            k, jb, m = shape(b)
            buf = buf_of(b)
            panel = buf[258 + jb:].view(m, jb)
            view = A[:, col0:col1]
            if self._fast_swaps and self._has_list_host(buf):
                ops.laswp_moves_(view, k, buf[:256].view(torch.int32))
            else:
                ops.laswp_(view, k, jb, ipiv[k:k + jb])
            U12 = A[k:k + jb, col0:col1]
            ops.trsm_lu_(panel[:jb, :], U12)
            if m > jb and pend and len(pend) > 1:
                for c, (r0, r1) in enumerate(self._chunk_rows(m, jb)):   # each row range as its chunk lands
                    self._bcast_wait(pend[c])
                    r0 = max(r0, jb)
                    if r1 > r0:
                        ops.gemm_sub_(A[k + r0:k + r1, col0:col1], panel[r0:r1, :], U12)
                works.pop(b, None)
            elif m > jb:
                self._bcast_wait(pend or [])
                works.pop(b, None)
                ops.gemm_sub_(A[k + jb:, col0:col1], panel[jb:, :], U12)

        def swap_left(b, ncols):
            if ncols <= 0:
                return
            k, jb, m = shape(b)
            buf = buf_of(b)
            if self._fast_swaps and self._has_list_host(buf):
                ops.laswp_moves_(A[:, :ncols], k, buf[:256].view(torch.int32))
            else:
                ops.laswp_(A[:, :ncols], k, jb, ipiv[k:k + jb])

        # panel 0: factored and sent before the loop
        own0 = self.owner(0)
        if own0 == self.rank:
            self._pack_panel(A, 0, ipiv, info)
        self._bcast_wait(start(0))
        if own0 != self.rank:
            unpack(0)

        for b in range(self.nblocks):
            own = self.owner(b)
            has_next = b + 1 < self.nblocks
            own_next = self.owner(b + 1) if has_next else -1
            nxt = self._first_local_block_after(b)       # first local block right of panel b
            right0 = self.offset[nxt] if nxt is not None else self.local_cols
            work = None
            if has_next and own_next == self.rank:
                # look-ahead: bring block b+1 up to date, factor it, start its broadcast ...
                assert nxt == b + 1
                w = self.widths[nxt]
                apply_panel(b, right0, right0 + w)
                self._bcast_wait(sends.pop((b + 1) & 1, None) or [])   # the slot's previous broadcast has left it
                self._pack_panel(A, b + 1, ipiv, info)
                work = start(b + 1)
                right0 += w                                # ... then the rest of update b
            elif has_next:
                work = start(b + 1)                        # receives posted before the update
            apply_panel(b, right0, self.local_cols)
            # interchanges on the columns left of the panel (the owner's panel is already swapped)
            swap_left(b, self.offset[b] if own == self.rank else (self.offset[nxt] if nxt is not None
                                                                  else self.local_cols))
            if has_next:
                if own_next == self.rank:
                    sends[(b + 1) & 1] = work              # its own sends: see `sends`
                elif self.chunks == 1:
                    self._bcast_wait(work)                 # unchunked: the whole panel
                else:
                    self._bcast_wait(work[0])              # header + top rows; the other chunks are waited for
                    works[b + 1] = work                    # where the update consumes them
                if own_next != self.rank:
                    unpack(b + 1)
        for w_ in list(sends.values()):
            self._bcast_wait(w_ or [])
        return ipiv, info


class MultiDeviceLU:
    """One process driving P devices through ONE C call (lsx_getrf_mg_f64, csrc/mg.hip): the same 1-D block-cyclic
    column distribution as ShardedLU, the panel written to every peer directly over xGMI in row chunks -- no
    torch.distributed.  `devices` lists the device index of every shard; the same index may appear several times
    (several handles on one GPU: the rehearsal the one-GPU test box allows)."""

    def __init__(self, n: int, devices, nb: int = 128):
        import ctypes as C

        from . import _native as N

        self.n, self.nb, self.P = n, nb, len(devices)
        self.devices = list(devices)
        self.handles = [N.Handle(d) for d in self.devices]
        for h in self.handles:
            h.set_option("nb", nb)
        self.nblocks = (n + nb - 1) // nb
        self.blocks = [[b for b in range(self.nblocks) if b % self.P == d] for d in range(self.P)]
        self.local_cols = [sum(min(nb, n - b * nb) for b in bl) for bl in self.blocks]
        self._C, self._N = C, N

    def scatter(self, full: torch.Tensor):
        """Replicated full matrix (on any device) -> the P local matrices."""
        out = []
        for d in range(self.P):
            cols = [full[:, b * self.nb:b * self.nb + min(self.nb, self.n - b * self.nb)] for b in self.blocks[d]]
            loc = torch.cat(cols, dim=1) if cols else torch.empty(self.n, 0, dtype=full.dtype)
            out.append(loc.to(torch.device("cuda", self.devices[d])).contiguous())
        return out

    def gather(self, locs) -> torch.Tensor:
        full = torch.empty(self.n, self.n, dtype=locs[0].dtype, device=locs[0].device)
        for d in range(self.P):
            off = 0
            for b in self.blocks[d]:
                w = min(self.nb, self.n - b * self.nb)
                full[:, b * self.nb:b * self.nb + w] = locs[d][:, off:off + w].to(full.device)
                off += w
        return full

    def factor_(self, locs):
        """In-place P A = L U of the distributed matrix.  Returns (ipiv per device, info per device)."""
        C, N = self._C, self._N
        P = self.P
        ipiv = [torch.zeros(self.n, dtype=torch.int32, device=t.device) for t in locs]
        info = [torch.zeros(1, dtype=torch.int32, device=t.device) for t in locs]
        for t in locs:
            if t.dtype != torch.float64 or t.stride(1) != 1:
                raise TypeError("local matrices must be row-major fp64")
        hs = (C.c_void_p * P)(*[h.ptr for h in self.handles])
        dA = (C.c_void_p * P)(*[t.data_ptr() for t in locs])
        lda = (C.c_int * P)(*[max(t.stride(0), 1) for t in locs])
        dp = (C.c_void_p * P)(*[t.data_ptr() for t in ipiv])
        di = (C.c_void_p * P)(*[t.data_ptr() for t in info])
        for d in set(self.devices):
            torch.cuda.synchronize(d)
        N.check(self.handles[0].lib.lsx_getrf_mg_f64(hs, P, self.n, dA, lda, dp, di), "lsx_getrf_mg_f64")
        return ipiv, info
