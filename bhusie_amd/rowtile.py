"""Row tiling of one frame across the GPUs of a node (one process per GPU, torch.distributed).

The reference is single-device (no collectives).  Pixels are independent
(/root/reference/src/renderer/shaders/ray.wgsl:167-243 reads only uniforms, read-only scene data
and the previous ladder level), so the frame shards by rows with ONE exchange step: a gather of the
per-rank HDR rows to the root rank (RCCL over xGMI when the backend is "nccl").

Partition (same rule as bhray_config.row_*): frame row r belongs to rank (r // stripe_rows) % world.
Interleaved stripes rather than contiguous bands, because the rows that cross the hole and the disk
cost far more integrator steps than sky rows.  Each rank recomputes the coarse ladder
rows its stripes depend on (a halo of one coarse row per level), so nothing is exchanged before the
gather.

Everything here works on CPU tensors with the gloo backend too (tests/test_rowtile_gloo.py).
"""
from __future__ import annotations

import numpy as np


def partition_rows(frame_h: int, world: int, stripe_rows: int = 27):
    """rows[rank] = increasing frame-row indices owned by `rank` (must equal bhray_local_row_index)."""
    r = np.arange(frame_h)
    owner = (r // stripe_rows) % world
    return [r[owner == k] for k in range(world)]


def max_local_rows(frame_h: int, world: int, stripe_rows: int = 27) -> int:
    return max(len(x) for x in partition_rows(frame_h, world, stripe_rows))


class FrameGather:
    """Gathers packed per-rank row blocks to `dst` and de-interleaves them into the frame.

    Each rank contributes a (max_rows, W, 4) f32 block (ranks with fewer rows pad at the end), so the
    collective is a plain equal-size gather: torch.distributed.gather (ncclSend/ncclRecv group under
    the nccl backend).  `assemble` runs on the root only: frame[rows_of_rank] = block[:len(rows)].
    """

    def __init__(self, frame_w: int, frame_h: int, rank: int, world: int, stripe_rows: int = 27, dst: int = 0,
                 device="cpu", group=None):
        import torch
        self.torch = torch
        self.w, self.h, self.rank, self.world, self.dst, self.group = frame_w, frame_h, rank, world, dst, group
        self.rows = partition_rows(frame_h, world, stripe_rows)
        self.max_rows = max(len(x) for x in self.rows)
        self.device = device
        self.local = torch.zeros((self.max_rows, frame_w, 4), dtype=torch.float32, device=device)
        if rank == dst:
            self.blocks = [torch.zeros_like(self.local) for _ in range(world)]
            self.frame = torch.zeros((frame_h, frame_w, 4), dtype=torch.float32, device=device)
            self.row_index = [torch.as_tensor(x, dtype=torch.long, device=device) for x in self.rows]
        else:
            self.blocks, self.frame, self.row_index = None, None, None

    def gather(self, async_op: bool = False):
        """Collective over all ranks.  Returns the work handle (or None)."""
        import torch.distributed as dist
        if self.world == 1 and not dist.is_initialized():
            self.blocks[0] = self.local
            return None
        return dist.gather(self.local, self.blocks if self.rank == self.dst else None, dst=self.dst,
                           group=self.group, async_op=async_op)

    def assemble(self):
        """Root only: de-interleave the gathered blocks into the (H, W, 4) frame."""
        if self.rank != self.dst:
            return None
        for k in range(self.world):
            n = len(self.rows[k])
            if n:
                self.frame.index_copy_(0, self.row_index[k], self.blocks[k][:n])
        return self.frame
