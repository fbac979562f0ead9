"""Multi-GPU decomposition of one frame: one process per GPU, horizontal strips of H/N full-res
rows, every input plane replicated, SSAO/blur halos recomputed locally (crychic_ssao_compute), and exactly one
collective per frame: the all-gather of the composed RGBA8 strips (RCCL over xGMI through torch.distributed).
"""
import ctypes as C

import torch
import torch.distributed as dist

from ._lib import check, lib


def strip_rows(H, nranks, rank):
    """Full-res rows (row0, rows) owned by `rank` (crychic_strip_rows: even-row aligned, last rank takes the rest)."""
    r0, rn = C.c_uint32(), C.c_uint32()
    check(lib.crychic_strip_rows(int(H), int(nranks), int(rank), C.byref(r0), C.byref(rn)))
    return r0.value, rn.value


class FrameGather:
    """All-gather of the per-rank RGBA8 strips into a full frame on every rank.

    Two frame slots alternate so the gather of frame i (issued asynchronously; RCCL runs it on its own stream
    behind an event on the render stream) overlaps the rendering of frame i+1.  Strips are equal-sized
    (H/2 divisible by nranks) so the collective is a plain all_gather_into_tensor; a ragged last strip falls
    back to all_gather with a padded send buffer.
    """

    SLOTS = 2

    def __init__(self, W, H, nranks, rank, device, group=None):
        self.W, self.H, self.nranks, self.rank, self.group = W, H, nranks, rank, group
        self.row0, self.rows = strip_rows(H, nranks, rank)
        self.uniform = (H // 2) % nranks == 0
        self.max_rows = max(strip_rows(H, nranks, r)[1] for r in range(nranks))
        # each slot: a full back buffer the strip is rendered into (crychic addresses rows of the whole frame) ...
        self.render = [torch.zeros((H, W, 4), dtype=torch.uint8, device=device) for _ in range(self.SLOTS)]
        # ... and the gathered frame
        self.frames = [torch.zeros((nranks * self.max_rows, W, 4), dtype=torch.uint8, device=device) for _ in range(self.SLOTS)]
        self.pad = None if self.uniform else [torch.zeros((self.max_rows, W, 4), dtype=torch.uint8, device=device)
                                              for _ in range(self.SLOTS)]
        self.pending = [None] * self.SLOTS

    def strip_buffer(self, i):
        """Back buffer for frame i; waits for the gather that last read this slot."""
        s = i % self.SLOTS
        if self.pending[s] is not None:
            self.pending[s].wait()
            self.pending[s] = None
        return self.render[s]

    def launch(self, i):
        s = i % self.SLOTS
        strip = self.render[s][self.row0:self.row0 + self.rows]
        if self.uniform:
            send = strip
        else:
            self.pad[s][:self.rows].copy_(strip)
            send = self.pad[s]
        self.pending[s] = dist.all_gather_into_tensor(self.frames[s].view(-1), send.reshape(-1), group=self.group, async_op=True)

    def wait_all(self):
        for s in range(self.SLOTS):
            if self.pending[s] is not None:
                self.pending[s].wait()
                self.pending[s] = None

    def frame(self, i):
        """The gathered H x W x 4 frame of step i (valid after wait_all / the slot's wait)."""
        s = i % self.SLOTS
        f = self.frames[s]
        if self.uniform:
            return f[:self.H]
        parts = []
        for r in range(self.nranks):
            rows = strip_rows(self.H, self.nranks, r)[1]
            parts.append(f[r * self.max_rows:r * self.max_rows + rows])
        return torch.cat(parts, dim=0)
