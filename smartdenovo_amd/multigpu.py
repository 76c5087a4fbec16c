"""Multi-GPU plumbing of the overlap path: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by QUERY (contract 2 "job striping"): rank g of N runs the reference's own
`wtzmo -P N -p g` striping (wtzmo.c:1291,1314) on reads + indexes replicated in its own HBM, so there is no data-path
collective; the only exchange is the gather of the finished overlap records, done here with all_gather over xGMI.
The same functions run with the gloo backend on CPU tensors (tests/test_multi_rank_gloo.py)."""
from __future__ import annotations

import torch


def stripe_argv(world: int, rank: int):
    """argv fragment that makes one wtzmo process take its stripe of the queries."""
    return ["-P", str(world), "-p", str(rank)] if world > 1 else []


def gather_records(dist, data: bytes, device: str):
    """All-gather variable-length record blobs; returns the list of per-rank blobs (on every rank)."""
    world = dist.get_world_size()
    payload = torch.frombuffer(bytearray(data or b"\n"), dtype=torch.uint8).to(device)
    n = torch.tensor([len(data)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    mx = max(1, max(int(s.item()) for s in sizes))
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    pad[:payload.numel()] = payload
    bufs = [torch.empty(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [bytes(b[:int(s.item())].cpu().numpy().tobytes()) for b, s in zip(bufs, sizes)]
