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


class _Done:
    """Work handle of a collective that has already completed."""

    def wait(self):
        return True


class FrameGather:
    """Gathers packed per-rank row blocks to `dst` and de-interleaves them into the frame(s).

    Each rank contributes a (batch, max_rows, W, 4) f32 block — the frames of one launch batch (bhray_config.frames_per_batch;
    batch = 1: a single frame), ranks with fewer rows pad at the end — so the collective is ONE plain equal-size gather per
    batch: torch.distributed.gather (ncclSend/ncclRecv group under the nccl backend).  `assemble` runs on the root only:
    frames[:, rows_of_rank] = block[:, :len(rows)].  `local[i]` is the buffer frame i of the batch is rendered into.
    """

    def __init__(self, frame_w: int, frame_h: int, rank: int, world: int, stripe_rows: int = 27, dst: int = 0,
                 device="cpu", group=None, batch: int = 1):
        import torch
        self.torch = torch
        self.w, self.h, self.rank, self.world, self.dst, self.group = frame_w, frame_h, rank, world, dst, group
        self.batch = batch
        self.rows = partition_rows(frame_h, world, stripe_rows)
        self.max_rows = max(len(x) for x in self.rows)
        self.device = device
        self._local = torch.zeros((batch, self.max_rows, frame_w, 4), dtype=torch.float32, device=device)
        # every rank owns the same number of whole stripes: the de-interleave is then ONE strided copy (a permutation of
        # [rank][frame][cycle][stripe row] into [frame][cycle][rank][stripe row]) instead of one index_copy per rank
        self.regular = frame_h % (stripe_rows * world) == 0
        self.cycles = frame_h // (stripe_rows * world) if self.regular else 0
        self.stripe_rows = stripe_rows
        if rank == dst:
            self.stacked = torch.zeros((world,) + tuple(self._local.shape), dtype=torch.float32, device=device)
            self.blocks = list(self.stacked.unbind(0))            # contiguous views: what the gather fills
            self.frames = torch.zeros((batch, frame_h, frame_w, 4), dtype=torch.float32, device=device)
            self.row_index = [torch.as_tensor(x, dtype=torch.long, device=device) for x in self.rows]
        else:
            self.stacked, self.blocks, self.frames, self.row_index = None, None, None, None

    @property
    def local(self):
        """The (max_rows, W, 4) buffer of a single-frame gather (batch == 1); `local_frame(i)` in general."""
        return self._local[0]

    def local_frame(self, i: int):
        return self._local[i]

    @property
    def frame(self):
        return None if self.frames is None else self.frames[0]

    def gather(self, async_op: bool = False):
        """Collective over all ranks.  Returns the work handle (or None)."""
        import torch.distributed as dist
        if self.world == 1 and not dist.is_initialized():
            self.blocks[0].copy_(self._local)
            return None
        if self._local.is_cuda and dist.get_backend(self.group) == "gloo":
            # functional-test path (several ranks on ONE GPU, bench.py --backend gloo): gloo cannot move device tensors, so
            # the block is staged through host memory, synchronously.  Same buffers, same de-interleave.
            self.torch.cuda.current_stream().synchronize()
            host = self._local.cpu()
            parts = [self.torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
            dist.gather(host, parts, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                for k in range(self.world):
                    self.blocks[k].copy_(parts[k])
            return _Done() if async_op else None
        return dist.gather(self._local, self.blocks if self.rank == self.dst else None, dst=self.dst,
                           group=self.group, async_op=async_op)

    def assemble(self):
        """Root only: de-interleave the gathered blocks into the (batch, H, W, 4) frames; returns frames[0] when batch == 1."""
        if self.rank != self.dst:
            return None
        if self.regular:
            B, C, R, sr, W = self.batch, self.cycles, self.world, self.stripe_rows, self.w
            self.frames.view(B, C, R, sr, W, 4).copy_(self.stacked.view(R, B, C, sr, W, 4).permute(1, 2, 0, 3, 4, 5))
        else:
            for k in range(self.world):
                n = len(self.rows[k])
                if n:
                    self.frames.index_copy_(1, self.row_index[k], self.blocks[k][:, :n])
        return self.frames[0] if self.batch == 1 else self.frames
