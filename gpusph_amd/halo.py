"""Transports of the slab decomposition's exchange (gpusph_amd/multigpu.py): who moves the edge layers and reduces dt.

TorchTransport   torch.distributed point-to-point and collectives (backend "nccl" = RCCL over xGMI; "gloo" in the CPU test
                 rigs and for several ranks on one GPU).
CapiTransport    the library's own entry points (include/sphx.h sphx_halo_*, gpusph_amd/csrc/halo.hip): the calls a GPUSPH host
                 makes in place of GPUWorker::transferBursts -- RCCL send / recv for one process per device, peer copies between
                 the worker threads of one process.  The Python driver uses it to exercise that C ABI (tests) and, with
                 SPHX_HALO=capi, in bench.py.

All take dim-0 row ranges of torch tensors; payloads travel as bytes."""
import ctypes as C

import numpy as np
import torch

from . import capi


class TorchTransport:
    def __init__(self, dist, is_cuda):
        self.dist = dist
        self.stage = is_cuda and dist.get_backend() == "gloo"     # test rigs only: gloo moves host memory

    def exchange(self, tensors, left, right, send_l, recv_l, send_r, recv_r):
        """send my edge layers / receive the halo layers of every tensor (dim-0 ranges), as one grouped batch of
        point-to-point operations (RCCL has no 16-bit integer type for the ushort4 particleinfo: bytes)"""
        dist = self.dist
        ops, copies = [], []
        moved = 0

        def send(t, rng, peer):
            if rng[1] > rng[0]:
                v = t[rng[0]:rng[1]].view(torch.uint8)
                ops.append(dist.P2POp(dist.isend, v.cpu() if self.stage else v, peer))

        def recv(t, rng, peer):
            if rng[1] > rng[0]:
                v = t[rng[0]:rng[1]].view(torch.uint8)
                if self.stage:
                    h = torch.empty(v.shape, dtype=torch.uint8)
                    copies.append((v, h))
                    v = h
                ops.append(dist.P2POp(dist.irecv, v, peer))

        # two slabs on a ring (a periodic split axis) are each other's neighbour on both sides: messages between one pair of
        # ranks are matched in the order they are posted, and my LEFT layer is my peer's RIGHT halo, so the receives are
        # posted right halo first
        same_peer = left is not None and left == right
        for t in tensors:
            row = t[0:1].view(torch.uint8).numel() if t.shape[0] else 0
            if same_peer:
                send(t, send_l, left); send(t, send_r, right); recv(t, recv_r, right); recv(t, recv_l, left)
                moved += row * (send_l[1] - send_l[0] + recv_l[1] - recv_l[0] + send_r[1] - send_r[0] + recv_r[1] - recv_r[0])
                continue
            if left is not None:
                send(t, send_l, left); recv(t, recv_l, left)
                moved += row * (send_l[1] - send_l[0] + recv_l[1] - recv_l[0])
            if right is not None:
                send(t, send_r, right); recv(t, recv_r, right)
                moved += row * (send_r[1] - send_r[0] + recv_r[1] - recv_r[0])
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        for v, h in copies:
            v.copy_(h)
        return moved

    def allgather_pair(self, a, b, device):
        mine = torch.tensor([a, b], dtype=torch.int64, device=device)
        allc = [torch.zeros_like(mine) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(allc, mine)
        return [(int(c[0].item()), int(c[1].item())) for c in allc]

    def allreduce_min(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)

    def allreduce_sum(self, t):
        self.dist.all_reduce(t)


class CapiTransport:
    """sphx_halo_* behind the interface above.  `kernels` is the rank's HipKernels (its context owns the device)."""

    def __init__(self, kernels, rank, world, group=None, unique_id=None):
        self.k, self.rank, self.world = kernels, rank, world
        self.lib = kernels.lib
        h = C.c_void_p()
        if group is not None:
            capi.check(self.lib.sphx_halo_create_threads(group, kernels.ctx.handle, rank, C.byref(h)))
        else:
            assert unique_id is not None and len(unique_id) == 128
            buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
            capi.check(self.lib.sphx_halo_create_rccl(kernels.ctx.handle, buf, rank, world, C.byref(h)))
        self.handle = h

    @staticmethod
    def new_group(lib, world):
        g = C.c_void_p()
        capi.check(lib.sphx_halo_group_create(world, C.byref(g)))
        return g

    @staticmethod
    def new_unique_id(lib):
        buf = (C.c_char * 128)()
        capi.check(lib.sphx_halo_unique_id(buf))
        return bytes(buf)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sphx_halo_destroy(self.handle)
            self.handle = None

    def exchange(self, tensors, left, right, send_l, recv_l, send_r, recv_r):
        tensors = [t for t in tensors if t.shape[0]]
        n = len(tensors)
        bufs = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in tensors])
        rows = (C.c_uint32 * max(n, 1))(*[t[0:1].view(torch.uint8).numel() for t in tensors])
        sl, rl, sr, rr = (send_l, recv_l, send_r, recv_r)
        capi.check(self.lib.sphx_halo_exchange(self.handle, n, bufs, rows,
                                               -1 if left is None else left, sl[0], sl[1] - sl[0], rl[0], rl[1] - rl[0],
                                               -1 if right is None else right, sr[0], sr[1] - sr[0], rr[0], rr[1] - rr[0], self.k._s()))
        moved = 0
        for r in rows[:n]:
            if left is not None:
                moved += r * (sl[1] - sl[0] + rl[1] - rl[0])
            if right is not None:
                moved += r * (sr[1] - sr[0] + rr[1] - rr[0])
        return moved

    def allgather_pair(self, a, b, device):
        mine = (C.c_uint64 * 2)(int(a), int(b))
        out = (C.c_uint64 * (2 * self.world))()
        capi.check(self.lib.sphx_halo_allgather_u64x2(self.handle, mine, out, self.k._s()))
        return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(self.world)]

    def allreduce_min(self, t):
        assert t.dtype == torch.float32 and t.numel() == 1
        capi.check(self.lib.sphx_halo_allreduce_min_f32(self.handle, capi.ptr(t), self.k._s()))

    def allreduce_sum(self, t):
        """float32 or float64 (the body totals), reduced in the tensor's own precision like TorchTransport does"""
        if t.dtype == torch.float64:
            n = t.numel()
            for k in range(0, n, 8):         # worker threads reduce at most 8 values per call
                v = t.view(-1)[k:k + 8]
                capi.check(self.lib.sphx_halo_allreduce_sum_f64(self.handle, capi.ptr(v), v.numel(), self.k._s()))
            return
        assert t.dtype == torch.float32
        n = t.numel()
        for k in range(0, n, 8):
            v = t.view(-1)[k:k + 8]
            capi.check(self.lib.sphx_halo_allreduce_sum_f32(self.handle, capi.ptr(v), v.numel(), self.k._s()))
