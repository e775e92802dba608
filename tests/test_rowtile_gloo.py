"""CPU, world_size 2 over gloo: the N>1 path of bench.py — row partition, equal-size gather of the packed
row blocks to rank 0, de-interleave into the frame.  The per-rank rows come from the CPU oracle here
(no GPU in this container); on the GPU box the same FrameGather carries device tensors over RCCL."""
import os
import socket

import numpy as np
import pytest

from bhusie_amd.rowtile import FrameGather, max_local_rows, partition_rows


def test_partition_rule():
    rows = partition_rows(1080, 8, 27)
    assert sum(len(r) for r in rows) == 1080
    assert sorted(np.concatenate(rows).tolist()) == list(range(1080))
    assert all(len(r) == 135 for r in rows)                      # 40 stripes / 8 ranks
    for k, r in enumerate(rows):
        assert np.all((r // 27) % 8 == k)
    rows = partition_rows(110, 3, 9)
    assert [len(r) for r in rows] == [38, 36, 36] and max_local_rows(110, 3, 9) == 38
    assert len(partition_rows(5, 8, 27)[0]) == 5 and len(partition_rows(5, 8, 27)[1]) == 0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, w, h, stripe, frame_path, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(frame_path)
    g = FrameGather(w, h, rank, world, stripe, 0, device="cpu")
    mine = g.rows[rank]
    g.local.zero_()
    g.local[:len(mine)] = torch.from_numpy(full[mine])           # what bhray_read_hdr / bind_output would hold
    for _ in range(2):                                            # twice: buffers are reusable
        work = g.gather(async_op=True)
        if work is not None:
            work.wait()
        frame = g.assemble()
    # a batch of 3 frames (bhray_config.frames_per_batch): one collective, frames de-interleaved together
    gb = FrameGather(w, h, rank, world, stripe, 0, device="cpu", batch=3)
    for i in range(3):
        gb.local_frame(i).zero_()
        gb.local_frame(i)[:len(mine)] = torch.from_numpy(full[mine]) * float(i + 1)
    gb.gather()
    frames = gb.assemble()
    # another root (bench.py rotates the root over the batch slots)
    g1 = FrameGather(w, h, rank, world, stripe, world - 1, device="cpu")
    g1.local.zero_(); g1.local[:len(mine)] = torch.from_numpy(full[mine])
    g1.gather()
    f1 = g1.assemble()
    assert (f1 is not None) == (rank == world - 1)
    if rank == world - 1:
        assert np.array_equal(f1.numpy(), full, equal_nan=True)
    if rank == 0:
        assert frames.shape == (3, h, w, 4)
        for i in range(3):
            assert np.array_equal(frames[i].numpy(), full * np.float32(i + 1), equal_nan=True)
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,stripe", [(2, 9), (2, 27), (2, 10)])      # 40 rows: (2, 10) is the regular case (one strided copy)
def test_gather_reassembles_the_frame(tmp_path, world, stripe):
    import torch.multiprocessing as mp
    from bhusie_amd import assets
    from oracle import host_oracle as H
    from oracle import oracle as O
    w, h = 70, 40
    sc = O.OracleScene(H.camera_uniform(), H.black_hole_uniform(), H.ray_details(integration_method=1),
                       assets.temp_lut(16), assets.disk_texture(32), assets.sky_texture(64, 32))
    full = O.render_ladder(sc, [(24, 14), (70, 40)])[-1]
    fp, op = str(tmp_path / "frame.npy"), str(tmp_path / "out.npy")
    np.save(fp, full)
    mp.spawn(_worker, args=(world, _free_port(), w, h, stripe, fp, op), nprocs=world, join=True)
    got = np.load(op)
    assert np.array_equal(got, full)
