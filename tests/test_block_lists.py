"""The host side of the frame-to-frame feedback (rtc_device.hip refine_block_list, simulate_dispatch) -- no device needed.

Whatever the wave times say, the list made from them must render every pixel of the partition exactly once: a block list is a
tiling of the frame by blocks of 16 x 16, 16 x 8, 8 x 8, 8 x 4 or 4 x 4 pixels (1, 2, 4, 8, 16 lanes per pixel), one size per 16 x 16 tile.
"""
import ctypes as C

import numpy as np
import pytest

import ray_tracer_challenge_amd as P

U32P = C.POINTER(C.c_uint32)


def _block(s):
    return 16 >> (s >> 1), 16 >> ((s + 1) >> 1)  # width, height in pixels (render_body: tw, th of four waves)


def _list(width, rows, s_of_tile):
    out = []
    for y0 in range(0, rows, 16):
        for x0 in range(0, width, 16):
            s = s_of_tile(x0 // 16, y0 // 16)
            bw, bh = _block(s)
            for dy in range(0, 16, bh):
                for dx in range(0, 16, bw):
                    if y0 + dy < rows and x0 + dx < width:
                        out.append((s & 3) << 30 | ((x0 + dx) // 4) << 16 | (s >> 2) << 15 | ((y0 + dy) // 4))
    return np.array(out, dtype=np.uint32)


def _refine(lst, ticks, width, rows, slots, up, down):
    cap = 8 * len(lst) + 16
    out = np.zeros(cap, dtype=np.uint32)
    n = P.lib().rtc_diag_refine_block_list(lst.ctypes.data_as(U32P), ticks.ctypes.data_as(U32P), len(lst), width, rows, slots, up, down,
                                           out.ctypes.data_as(U32P), cap)
    assert n <= cap
    return out[:n]


def _coverage(lst, width, rows):
    cover = np.zeros((rows, width), dtype=np.int32)
    lanes = {}
    for t in lst.tolist():
        s, x0, y0 = (t >> 30) | ((t >> 13) & 4), ((t >> 16) & 0x3fff) << 2, (t & 0x7fff) << 2
        bw, bh = _block(s)
        assert x0 % bw == 0 and y0 % bh == 0 and x0 < width and y0 < rows, (s, x0, y0)
        cover[y0:y0 + bh, x0:x0 + bw] += 1
        lanes.setdefault((x0 // 16, y0 // 16), set()).add(s)
    return cover, lanes


@pytest.mark.parametrize("width,rows,seed", [(64, 48, 1), (100, 37, 2), (333, 130, 3), (16, 16, 4), (1000, 400, 5), (8, 4, 6)])
def test_a_refined_list_tiles_the_partition(width, rows, seed):
    rng = np.random.default_rng(seed)
    tiles_x = (width + 15) // 16
    start = rng.integers(0, 5, size=((rows + 15) // 16, tiles_x))
    lst = _list(width, rows, lambda tx, ty: int(start[ty, tx]))
    for trial in range(6):
        kind = trial % 3
        if kind == 0:
            ticks = rng.integers(0, 100000, size=4 * len(lst)).astype(np.uint32)        # anything
        elif kind == 1:
            ticks = (rng.pareto(1.2, size=4 * len(lst)) * 500).clip(0, 4e9).astype(np.uint32)  # a few very long waves
        else:
            ticks = np.zeros(4 * len(lst), dtype=np.uint32)                                # a launch that never ran
        up, down = [(0.85, 0.4), (0.1, 0.0), (float("inf"), 0.0), (0.5, 5.0)][trial % 4]
        new = _refine(lst, ticks, width, rows, 6144.0 * 0.85, up, down)
        cover, lanes = _coverage(new, width, rows)
        assert (cover == 1).all(), "pixels rendered %s times" % sorted(set(cover.ravel().tolist()))
        assert all(len(v) == 1 for v in lanes.values()), "a tile with blocks of two sizes"
        if up == float("inf") and down == 0.0:  # ordering only: every tile keeps its lanes
            before = _coverage(lst, width, rows)[1]
            assert lanes == before


def test_tiles_start_in_the_order_of_their_longest_wave():
    width, rows = 128, 64
    lst = _list(width, rows, lambda tx, ty: 0)
    rng = np.random.default_rng(7)
    ticks = rng.integers(1, 10000, size=4 * len(lst)).astype(np.uint32)
    new = _refine(lst, ticks, width, rows, 1e9, float("inf"), 0.0)  # (so many wave slots that no tile is re-cut)
    longest = {int(t): int(ticks[4 * i:4 * i + 4].max()) for i, t in enumerate(lst.tolist())}
    seq = [longest[int(t)] for t in new.tolist()]
    assert seq == sorted(seq, reverse=True) and len(new) == len(lst)


@pytest.mark.parametrize("seed,top", [(1, 4), (2, 300), (3, 2 ** 31), (4, 2 ** 32 - 1)])
def test_equal_tiles_keep_their_order_in_the_list(seed, top):
    """The order is a stable one (a radix sort of the predictions' bit patterns): tiles of equal longest wave start in the order
    the list had them -- few distinct values, and values across the whole range of the tick counter."""
    width, rows = 512, 256
    lst = _list(width, rows, lambda tx, ty: 0)
    rng = np.random.default_rng(seed)
    ticks = rng.integers(0, top, size=4 * len(lst), endpoint=True).astype(np.uint32)
    new = _refine(lst, ticks, width, rows, 1e12, float("inf"), 0.0)
    longest = [int(ticks[4 * i:4 * i + 4].max()) for i in range(len(lst))]
    expected = [int(lst[i]) for i in sorted(range(len(lst)), key=lambda i: -longest[i])]  # (sorted() is stable)
    assert new.tolist() == expected


def test_blocks_of_a_padded_grid_outside_the_image_are_left_out():
    width, rows = 40, 24  # a grid padded to 4 x 4 blocks of 16 x 16 holds blocks that start outside the image
    lst = np.array([(bx * 4) << 16 | (by * 4) for by in range(4) for bx in range(4)], dtype=np.uint32)
    ticks = np.arange(4 * len(lst), dtype=np.uint32)
    new = _refine(lst, ticks, width, rows, 100.0, float("inf"), 0.0)
    cover, _ = _coverage(new, width, rows)
    assert (cover == 1).all() and len(new) == 3 * 2


def test_the_dispatcher_model():
    sim = lambda cost, slots: P.lib().rtc_diag_simulate_dispatch(np.asarray(cost, dtype=np.uint32).ctypes.data_as(U32P), len(cost), slots)
    assert sim([5, 5, 5, 5], 4) == 5.0 and sim([5, 5, 5, 5], 2) == 10.0 and sim([5, 5, 5, 5], 1) == 20.0
    assert sim([1, 1, 1, 9], 2) == 10.0 and sim([9, 1, 1, 1], 2) == 9.0  # the long one last / first
    assert sim([], 8) == 0.0
