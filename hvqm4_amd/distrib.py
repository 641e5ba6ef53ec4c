"""One-process-per-GPU plumbing for the independent-clip sharding: clips are dealt
round-robin to ranks, there is NO collective on the data path; torch.distributed is used only for the
start/stop barrier and to reduce timings / counters / checksums at the end.  The backend is gloo (host
TCP) by default: a barrier and three scalars per run need no RCCL, `north_star` asks for none, and eight
ranks then cannot fail in a communicator they do not use (HVQM4_DIST_BACKEND=nccl selects RCCL)."""
from __future__ import annotations

import os
import zlib
from typing import List, Sequence, Tuple


def shard(n_items: int, rank: int, world: int) -> List[int]:
    """clip i -> rank i mod world"""
    return [i for i in range(n_items) if i % world == rank]


def checksum_of_checksums(crcs: Sequence[int]) -> int:
    """order-independent digest of per-picture CRC32s: sum of a 64-bit mix of each, mod 2^64.
    Digests of disjoint shards combine by addition mod 2^64 (Group.sum64)."""
    s = 0
    for c in crcs:
        z = (c + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        s = (s + (z ^ (z >> 31))) & 0xFFFFFFFFFFFFFFFF
    return s


def picture_crc(buf) -> int:
    return zlib.crc32(memoryview(buf)) & 0xFFFFFFFF


class Group:
    """Thin wrapper so that bench.py and the CPU tests share one code path."""

    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.torch = None
        self.device = None
        self._out_fd = None
        if self.world > 1 or os.environ.get("HVQM4_DIST_FORCE"):      # FORCE: rehearse the RCCL path with one rank
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            backend = backend or os.environ.get("HVQM4_DIST_BACKEND") or "gloo"
            # RCCL prints a version banner on stdout when the first communicator is created: keep stdout for the
            # one JSON result line (emit) and send everything else to stderr
            import sys
            sys.stdout.flush()
            self._out_fd = os.dup(1)
            os.dup2(2, 1)
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group(backend="nccl", device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend="gloo")
            self.dist, self.torch = dist, torch

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            if self.device.type == "cuda":
                self.torch.cuda.synchronize()

    def _reduce(self, value: float, op) -> float:
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, value: float) -> float:
        return self._reduce(value, self.dist.ReduceOp.MAX if self.dist else None)

    def sum(self, value: float) -> float:
        return self._reduce(value, self.dist.ReduceOp.SUM if self.dist else None)

    def sum64(self, value: int) -> int:
        """sum of 64-bit digests over ranks, mod 2^64"""
        if self.dist is None:
            return value
        t = self.torch.tensor([value & 0x7FFFFFFFFFFFFFFF, value >> 63], dtype=self.torch.int64, device=self.device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        r = 0
        for o in out:
            r = (r + (int(o[0].item()) | (int(o[1].item()) << 63))) & 0xFFFFFFFFFFFFFFFF
        return r

    def emit(self, text: str):
        """write the result line to the process's real stdout"""
        if self._out_fd is None:
            print(text, flush=True)
        else:
            os.write(self._out_fd, (text + "\n").encode())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
