"""Row-band partition of one image over the ranks of a torch.distributed job
(one process per GPU) and the gather of the rendered bands to rank 0.

Pixels are independent given the (replicated, < 10 KB) scene, so the render
itself needs no collective.  The only exchange is the final gather of Canvas
rows: every peer sends its compact band buffer straight to rank 0 -- one xGMI
hop each on an MI355X node -- via torch.distributed (backend "nccl" is RCCL).
RCCL has no gather primitive of its own; dist.gather is grouped send/recv,
which is exactly the star we want (an all-gather would move N x the bytes).

Bands are dealt round-robin (band b -> rank b mod N) because cost per row
varies strongly with image content.  Jitter keys use global pixel indices, so
the assembled image is bit-identical to a single-GPU render.

The gather is double-buffered: `start(slot)` enqueues the gather of frame i on
the communication stream (it waits for the render that filled the slot, not for
the host), and the caller can launch frame i+1's render into the other slot
before `finish(slot)` of frame i -- so xGMI traffic overlaps the next frame's
kernel.
"""
import torch
import torch.distributed as dist

BAND_ROWS = 64


def band_layout(height, n_parts, band_rows=BAND_ROWS):
    """-> (rows per part, max rows, perm) with perm[y] = part * max_rows + local_row for global row y."""
    n_bands = (height + band_rows - 1) // band_rows
    rows = [0] * n_parts
    where = []
    for b in range(n_bands):
        part = b % n_parts
        y0, y1 = b * band_rows, min((b + 1) * band_rows, height)
        for y in range(y0, y1):
            where.append((part, rows[part] + (y - y0)))
        rows[part] += y1 - y0
    max_rows = max(rows) if rows else 0
    perm = torch.tensor([p * max_rows + r for p, r in where], dtype=torch.long)
    return rows, max_rows, perm


class BandGather:
    """Pre-allocated, double-buffered gather of (rows_r, W, C) band buffers to rank `dst`.

    `extra_parts` = E > 0 deals the bands over N + E parts instead of N: rank r renders part r as before, and `dst`
    additionally renders parts N .. N+E-1 straight into its own memory.  xGMI is point-to-point -- each peer's rows
    reach `dst` over that peer's one link -- so when moving a peer's share takes longer than rendering it, giving
    `dst` (whose rows never travel) a larger share shortens the step; the messages stay equal-sized, so the
    collective is the same dist.gather.  bench.py picks E from a measured gather / render ratio."""

    def __init__(self, height, width, channels, dtype, device, rank, world_size, dst=0, band_rows=BAND_ROWS,
                 slots=2, extra_parts=0, force_collective=False):
        if world_size == 1:
            extra_parts = 0
        # force_collective: a job of ONE rank still goes through dist.gather (bench.py --force-dist: the whole RCCL path --
        # communicator, communication stream, stream hand-off in finish() -- on a box with a single GPU)
        self.collective = world_size > 1 or force_collective
        self.rank, self.world_size, self.dst, self.slots = rank, world_size, dst, slots
        self.height, self.width, self.channels = height, width, channels
        self.extra_parts, self.n_parts = extra_parts, world_size + extra_parts
        self.rows, self.max_rows, perm = band_layout(height, self.n_parts, band_rows)
        self.local_rows = self.rows[rank]
        self.work = [None] * slots
        if rank == dst and self.collective:
            # one buffer per slot holding every part: [0, N) filled by the gather, [N, N+E) rendered here
            self.all = [torch.zeros((self.n_parts, self.max_rows, width, channels), dtype=dtype, device=device)
                        for _ in range(slots)]
            self.send = [torch.zeros((self.max_rows, width, channels), dtype=dtype, device=device) for _ in range(slots)]
            self.perm = perm.to(device)
            self.image = [torch.zeros((height, width, channels), dtype=dtype, device=device) for _ in range(slots)]
        else:
            # send buffers padded to the largest part so that every message has one size
            self.send = [torch.zeros((self.max_rows, width, channels), dtype=dtype, device=device) for _ in range(slots)]
            self.all = self.perm = self.image = None

    def parts(self):
        """The parts this rank renders: its own, plus the extra ones on dst."""
        own = [self.rank]
        if self.rank == self.dst and self.world_size > 1:
            own += list(range(self.world_size, self.n_parts))
        return own

    def local_view(self, slot=0, part=None):
        """Where the renderer writes part `part` (default: this rank's gather part) for `slot` -- compact, band after band."""
        if part is None or part == self.rank:
            return self.send[slot][: self.local_rows]
        assert self.rank == self.dst and self.world_size <= part < self.n_parts
        return self.all[slot][part][: self.rows[part]]

    def start(self, slot=0):
        """Enqueue the gather of `slot` (after whatever filled it on the current stream); returns immediately."""
        if not self.collective:
            return
        if self.rank == self.dst:
            self.work[slot] = dist.gather(self.send[slot], [self.all[slot][r] for r in range(self.world_size)],
                                          dst=self.dst, async_op=True)
        else:
            self.work[slot] = dist.gather(self.send[slot], None, dst=self.dst, async_op=True)

    def finish(self, slot=0):
        """Complete the gather of `slot`; returns the assembled image on dst, None elsewhere."""
        if not self.collective:
            return self.send[slot][: self.height]
        if self.work[slot] is not None:
            self.work[slot].wait()  # NCCL: the current stream waits for the comm stream, the host does not block
            self.work[slot] = None
        if self.rank != self.dst:
            return None
        flat = self.all[slot].view(self.n_parts * self.max_rows, self.width, self.channels)
        torch.index_select(flat, 0, self.perm, out=self.image[slot])
        return self.image[slot]

    def gather(self, slot=0):
        """start + finish (no overlap)."""
        self.start(slot)
        return self.finish(slot)
