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
    """Pre-allocated, double-buffered gather of (rows_r, W, C) band buffers to rank `dst`."""

    def __init__(self, height, width, channels, dtype, device, rank, world_size, dst=0, band_rows=BAND_ROWS,
                 slots=2):
        self.rank, self.world_size, self.dst, self.slots = rank, world_size, dst, slots
        self.height, self.width, self.channels = height, width, channels
        self.rows, self.max_rows, perm = band_layout(height, world_size, band_rows)
        self.local_rows = self.rows[rank]
        # send buffers padded to the largest part so that every message has one size
        self.send = [torch.zeros((self.max_rows, width, channels), dtype=dtype, device=device) for _ in range(slots)]
        self.work = [None] * slots
        if rank == dst and world_size > 1:
            self.recv = [torch.zeros((world_size, self.max_rows, width, channels), dtype=dtype, device=device)
                         for _ in range(slots)]
            self.perm = perm.to(device)
            self.image = [torch.zeros((height, width, channels), dtype=dtype, device=device) for _ in range(slots)]
        else:
            self.recv = self.perm = self.image = None

    def local_view(self, slot=0):
        """Where the renderer writes this rank's rows for `slot` (compact, band after band)."""
        return self.send[slot][: self.local_rows]

    def start(self, slot=0):
        """Enqueue the gather of `slot` (after whatever filled it on the current stream); returns immediately."""
        if self.world_size == 1:
            return
        if self.rank == self.dst:
            self.work[slot] = dist.gather(self.send[slot], [self.recv[slot][r] for r in range(self.world_size)],
                                          dst=self.dst, async_op=True)
        else:
            self.work[slot] = dist.gather(self.send[slot], None, dst=self.dst, async_op=True)

    def finish(self, slot=0):
        """Complete the gather of `slot`; returns the assembled image on dst, None elsewhere."""
        if self.world_size == 1:
            return self.send[slot][: self.height]
        if self.work[slot] is not None:
            self.work[slot].wait()  # NCCL: the current stream waits for the comm stream, the host does not block
            self.work[slot] = None
        if self.rank != self.dst:
            return None
        flat = self.recv[slot].view(self.world_size * self.max_rows, self.width, self.channels)
        torch.index_select(flat, 0, self.perm, out=self.image[slot])
        return self.image[slot]

    def gather(self, slot=0):
        """start + finish (no overlap)."""
        self.start(slot)
        return self.finish(slot)
