"""Multi-process tests of the row-band partition + gather (the N>1 path of bench.py) on CPU:
torch.distributed with the gloo backend, world sizes 2 and 3, 127.0.0.1 rendezvous.
The render kernel itself is replaced by a row-pattern writer here -- what is under test
is that every rank's band layout matches the C ABI's rtc_partition_rows and that the gathered
image is exactly the un-partitioned one, for ragged heights too."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ray_tracer_challenge_amd as P
from ray_tracer_challenge_amd import _lib as L
from ray_tracer_challenge_amd.dist import BandGather, band_layout


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _global_rows(height, n_parts, part, band_rows=64):
    rows = []
    for b in range(part, (height + band_rows - 1) // band_rows, n_parts):
        rows.extend(range(b * band_rows, min((b + 1) * band_rows, height)))
    return rows


def _worker(rank, world_size, port, height, width, ok, extra_parts=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        g = BandGather(height, width, 3, torch.float32, torch.device("cpu"), rank, world_size, extra_parts=extra_parts)
        n_parts = world_size + extra_parts
        assert g.n_parts == n_parts and g.local_rows == len(_global_rows(height, n_parts, rank))
        assert g.parts() == ([rank] + list(range(world_size, n_parts)) if rank == 0 else [rank])

        def fill(slot, frame):
            # stand-in for the kernel: pixel value encodes (frame, global row, column, channel); one "launch" per owned part
            for part in g.parts():
                local = g.local_view(slot, part)
                rows = _global_rows(height, n_parts, part)
                assert local.shape[0] == len(rows)
                for i, y in enumerate(rows):
                    local[i] = (frame * 1e6 + y * 1000.0 + torch.arange(width, dtype=torch.float32)[:, None]
                                + torch.tensor([0.0, 0.25, 0.5]))

        def expected(frame):
            return (frame * 1e6 + torch.arange(height, dtype=torch.float32)[:, None, None] * 1000.0
                    + torch.arange(width, dtype=torch.float32)[None, :, None] + torch.tensor([0.0, 0.25, 0.5]))

        def check(image, frame):
            if rank == 0:
                assert image.shape == (height, width, 3) and torch.equal(image, expected(frame)), frame
            else:
                assert image is None

        # unpipelined
        fill(0, 0)
        check(g.gather(0), 0)
        # bench.py's software pipeline: start(i), then finish(i-1), over 5 frames and 2 slots
        n_frames = 5
        for i in range(n_frames):
            fill(i % 2, i + 1)
            g.start(i % 2)
            if i > 0:
                check(g.finish((i - 1) % 2), i)
        check(g.finish((n_frames - 1) % 2), n_frames)
        dist.barrier()
        ok[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,height", [(2, 256), (2, 200), (3, 1000), (2, 63)])
def test_band_gather_gloo(world_size, height):
    ok = mp.get_context("spawn").Array("i", [0] * world_size)
    mp.spawn(_worker, args=(world_size, _free_port(), height, 17, ok), nprocs=world_size, join=True)
    assert list(ok) == [1] * world_size


@pytest.mark.parametrize("world_size,height,extra", [(2, 256, 2), (2, 1000, 3), (3, 700, 1), (2, 63, 2)])
def test_band_gather_with_extra_parts_on_the_root(world_size, height, extra):
    """The bandwidth-aware split: N + E parts, rank 0 renders E of them locally; the assembled image is the same."""
    ok = mp.get_context("spawn").Array("i", [0] * world_size)
    mp.spawn(_worker, args=(world_size, _free_port(), height, 17, ok, extra), nprocs=world_size, join=True)
    assert list(ok) == [1] * world_size


def test_band_layout_matches_c_abi_partition():
    for height in (1, 63, 64, 65, 400, 1000, 4096, 8192):
        for n in (1, 2, 3, 4, 8):
            rows, max_rows, perm = band_layout(height, n)
            for p in range(n):
                q = L.rtc_partition(64, n, p)
                assert rows[p] == P.lib().rtc_partition_rows(height, C.byref(q)) == len(_global_rows(height, n, p))
            assert sum(rows) == height and max_rows == max(rows)
            # perm maps every global row to a distinct (part, local row) slot
            assert len(set(perm.tolist())) == height
            for p in range(n):
                for i, y in enumerate(_global_rows(height, n, p)[:3]):
                    assert perm[y] == p * max_rows + i


def test_single_rank_gather_is_identity():
    g = BandGather(100, 8, 3, torch.float32, torch.device("cpu"), 0, 1)
    for slot in (0, 1):
        g.local_view(slot).copy_(torch.arange(100 * 8 * 3, dtype=torch.float32).view(100, 8, 3) + slot)
        g.start(slot)
    for slot in (0, 1):
        assert torch.equal(g.finish(slot), torch.arange(100 * 8 * 3, dtype=torch.float32).view(100, 8, 3) + slot)
