"""Multi-GPU plumbing of the overlap path: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Contract (option 1 "central commit"): the order-dependent part of `wtzmo -t 1` (closed pairs, contained-read
masking, coverage saturation, record order) is one sequential stream by definition, so rank 0 plans and commits; the PURE device
stages - seed lookup per query, pair seeding / windows / alignment / CIGAR rendering per pair - are dealt round-robin over the
ranks, each with the reads and both indexes replicated in its own HBM.  Requests (read ids, pair lists: a few bytes per pair) go
out from rank 0, results (48-byte pair summaries, window boxes, 72-byte alignment results, the CIGAR text) come back with
send / recv over xGMI; the CIGAR text - the bulk, ~6.4 KB per record - is sent from the device buffer it was rendered into (a tensor that
aliases the library's memory: `_send_dev`), so on the nccl backend it goes GPU -> xGMI -> rank 0's GPU without touching the sender's host.  One .ovl, written by rank 0, identical to `wtzmo -t 1` for ANY number of ranks; total work is fixed.

The C host driver (smartdenovo_amd/csrc/host/wtzmo_main.c, loaded as libwtzmo_host.so) drives the exchange through three hooks
set with wtzmo_set_dist(rank, world, bcast, send, recv); `RankExchange` implements them on torch.distributed.  The same class
runs on the gloo backend with CPU tensors (tests/test_multi_rank_gloo.py)."""
from __future__ import annotations

import ctypes as C

import torch

BCAST = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64)
SEND = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_int)
RECV = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_int)
SEND_DEV = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_int)
_CHUNK = 1 << 30      # bytes per message (int32 element counts inside the collectives)


class _DevBytes:
    """n bytes of device memory owned by libwtzmo_hip.so, exposed through the CUDA array interface so that torch can alias them (no copy)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_view(ptr, n):
    """uint8 tensor aliasing [ptr, ptr + n) on the current device"""
    return torch.as_tensor(_DevBytes(ptr, n), device="cuda")


def _host_view(ptr, n):
    return torch.frombuffer((C.c_char * n).from_address(ptr), dtype=torch.uint8)


class RankExchange:
    """The three exchange hooks of the host driver on a torch.distributed process group."""

    def __init__(self, dist, device: str):
        self.dist, self.device = dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.bytes_sent = self.bytes_received = self.messages = 0
        self._cb = (BCAST(self._bcast), SEND(self._send), RECV(self._recv))      # keep the callbacks alive
        self._cb_dev = SEND_DEV(self._send_dev)
        self.bytes_sent_from_device = 0
        self._stage = None
        self.dev_is_host = False        # the emulated device layer of the CPU tests: "device" pointers are host pointers

    def _stage_buf(self, m):
        """nccl moves device tensors only: ONE grow-only device staging buffer for the host-resident messages (round 5 allocated a tensor per message);
        on gloo the host view itself is the message - no copy on either side"""
        if self._stage is None or self._stage.numel() < m:
            self._stage = torch.empty(max(m, 1 << 20), dtype=torch.uint8, device=self.device)
        return self._stage[:m]

    def _bcast(self, ptr, n):
        for o in range(0, n, _CHUNK):
            m = min(_CHUNK, n - o)
            h = _host_view(ptr + o, m)
            if self.device == "cpu":
                self.dist.broadcast(h, 0)
                continue
            t = self._stage_buf(m)
            if self.rank == 0:
                t.copy_(h)
            self.dist.broadcast(t, 0)
            if self.rank != 0:
                h.copy_(t)
        self.messages += 1

    def _send(self, ptr, n, dst):
        for o in range(0, n, _CHUNK):
            m = min(_CHUNK, n - o)
            h = _host_view(ptr + o, m)
            if self.device == "cpu":
                self.dist.send(h, dst)
            else:
                t = self._stage_buf(m)
                t.copy_(h)
                self.dist.send(t, dst)
        self.bytes_sent += n
        self.messages += 1

    def _send_dev(self, ptr, n, dst):
        """the bytes are in this rank's device memory: on the nccl backend the tensor handed to send() aliases them (GPU -> xGMI -> rank 0's GPU);
        on gloo (test aid: several ranks on one GPU) they are staged through the host here, like _send does for host buffers"""
        for o in range(0, n, _CHUNK):
            m = min(_CHUNK, n - o)
            if self.dev_is_host:
                t = _host_view(ptr + o, m)
            else:
                t = device_view(ptr + o, m)
                if self.device == "cpu":
                    t = t.cpu()
            self.dist.send(t, dst)
        self.bytes_sent += n
        self.bytes_sent_from_device += n
        self.messages += 1

    def _recv(self, ptr, n, src):
        for o in range(0, n, _CHUNK):
            m = min(_CHUNK, n - o)
            h = _host_view(ptr + o, m)
            if self.device == "cpu":
                self.dist.recv(h, src)
            else:
                t = self._stage_buf(m)
                self.dist.recv(t, src)
                h.copy_(t)
        self.bytes_received += n
        self.messages += 1

    def install(self, host_lib):
        """wtzmo_set_dist on the loaded host driver (libwtzmo_host.so or the emulated one of the tests)"""
        host_lib.wtzmo_set_dist.argtypes = [C.c_int, C.c_int, BCAST, SEND, RECV]
        host_lib.wtzmo_set_dist.restype = None
        host_lib.wtzmo_set_dist(self.rank, self.world, *self._cb)
        if hasattr(host_lib, "wtzmo_set_dist_dev"):
            host_lib.wtzmo_set_dist_dev.argtypes = [SEND_DEV]
            host_lib.wtzmo_set_dist_dev.restype = None
            host_lib.wtzmo_set_dist_dev(self._cb_dev)
