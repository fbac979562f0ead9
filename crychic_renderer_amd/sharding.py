"""Multi-GPU decomposition of one frame: one process per GPU, horizontal strips of H/N full-res
rows, every input plane replicated, SSAO/blur halos recomputed locally (crychic_ssao_compute), and exactly one
collective per frame: the all-gather of the composed RGBA8 strips (RCCL over xGMI through torch.distributed).
"""
import ctypes as C

import torch
import torch.distributed as dist

from ._lib import check, lib


class _Batch:
    """The requests of one batch_isend_irecv call behind the wait() of a single Work."""

    def __init__(self, reqs):
        self.reqs = reqs

    def wait(self):
        for r in self.reqs:
            r.wait()


def strip_rows(H, nranks, rank):
    """Full-res rows (row0, rows) owned by `rank` (crychic_strip_rows: even-row aligned, last rank takes the rest)."""
    r0, rn = C.c_uint32(), C.c_uint32()
    check(lib.crychic_strip_rows(int(H), int(nranks), int(rank), C.byref(r0), C.byref(rn)))
    return r0.value, rn.value


class StripBalancer:
    """Measured load balancing of the row strips.

    Keeps a cost estimate per half-res row pair, seeded from the depth plane's coverage (a covered row costs about
    `covered_row_weight` sky rows: the lighting pass only works on covered pixels, SSAO taps reach further on near
    geometry).  After every rank has timed its strip, update() rescales the estimates inside each strip so that they sum to
    the measured time (the shape inside a strip is kept, so the estimate sharpens as the boundaries move) and plan() cuts
    the cumulative cost into equal parts, moving each boundary `damping` of the way (the per-strip fixed cost -- launches,
    blur halos -- does not move with the rows, so an undamped step overshoots).  Everything is a pure function of the
    depth plane and the all-gathered times, so every rank derives the same plan without further communication."""

    def __init__(self, depth, nranks, covered_row_weight=3.0, damping=0.7):
        H = int(depth.shape[0])
        self.h2, self.n, self.damping = H // 2, int(nranks), float(damping)
        covered = ((depth.reshape(H, -1).to(torch.int64) & 0xFFFFFF) != 0xFFFFFF).sum(dim=1)     # exact integer counts
        cov = (covered.reshape(self.h2, 2).sum(dim=1).to(torch.float64) / (2.0 * depth.shape[1])).cpu()
        self.cost = (1.0 + (covered_row_weight - 1.0) * cov).tolist()
        self.edges = self._cut([0] * (self.n + 1), 1.0)

    def _cut(self, old_edges, damping):
        total, edges, acc, row = sum(self.cost), [0], 0.0, 0
        for k in range(1, self.n):
            target = total * k / self.n
            while row < self.h2 and acc + self.cost[row] < target:
                acc += self.cost[row]
                row += 1
            pos = row + ((target - acc) / self.cost[row] if row < self.h2 else 0.0)
            e = int(round(old_edges[k] + damping * (pos - old_edges[k])))
            edges.append(max(edges[-1] + 1, min(e, self.h2 - (self.n - k))))      # every strip keeps at least one row pair
        edges.append(self.h2)
        return edges

    def bounds(self):
        """[(row0, rows)] per rank in full-res rows (even-aligned)."""
        return [(2 * self.edges[r], 2 * (self.edges[r + 1] - self.edges[r])) for r in range(self.n)]

    def update(self, times):
        """times[r] = what rank r measured for its current strip; returns the new bounds()."""
        for r in range(self.n):
            a, b = self.edges[r], self.edges[r + 1]
            est = sum(self.cost[a:b])
            f = max(float(times[r]), 1e-9) / est
            for i in range(a, b):
                self.cost[i] *= f
        self.edges = self._cut(self.edges, self.damping)
        return self.bounds()


def balanced_strips(depth, nranks, covered_row_weight=3.0):
    """The coverage-based first guess of StripBalancer (no measurements)."""
    return StripBalancer(depth, nranks, covered_row_weight).bounds()


class FrameGather:
    """All-gather of the per-rank RGBA8 strips into a full frame on every rank.

    SLOTS frame slots rotate so the gather of frame i (issued asynchronously; RCCL runs it on its own stream
    behind an event on the render stream) overlaps the rendering of frame i+1.  Strips are equal-sized
    (H/2 divisible by nranks) so the collective is a plain all_gather_into_tensor; a ragged last strip falls
    back to all_gather with a padded send buffer.
    """

    SLOTS = 4

    def __init__(self, W, H, nranks, rank, device, group=None, bounds=None):
        """bounds = [(row0, rows)] per rank selects the ragged mode for strips of any height (StripBalancer): every rank
        renders into a full-frame buffer and receives its peers' rows in place, so the gathered frame of a slot is
        render[slot] itself.  On the nccl (= RCCL) backend that is ONE call per frame -- all_gather with a list of
        different-sized row views, which the backend runs as a coalesced group of broadcasts (one per strip; on the fully
        connected xGMI mesh every link carries each strip once) -- because per-call host cost matters once a strip renders
        in 70 us: a batch of 2(N-1) isend/irecv ops costs roughly 15 us of Python/dispatcher time per op.  Backends
        without uneven all_gather (gloo, used by the CPU tests) fall back to that batch of sends / receives.
        Without bounds: equal strips and one all_gather_into_tensor."""
        self.W, self.H, self.nranks, self.rank, self.group = W, H, nranks, rank, group
        self.bounds = [tuple(int(v) for v in b) for b in bounds] if bounds is not None else None
        if self.bounds is not None:
            assert len(self.bounds) == nranks and sum(b[1] for b in self.bounds) == H and self.bounds[0][0] == 0
            self.row0, self.rows = self.bounds[rank]
            self.render = [torch.zeros((H, W, 4), dtype=torch.uint8, device=device) for _ in range(self.SLOTS)]
            self.pending = [None] * self.SLOTS
            self.p2p = dist.get_backend(group) != "nccl"
            self.views = [[buf[r0:r0 + rn] for (r0, rn) in self.bounds] for buf in self.render]
            return
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
        if self.bounds is not None and not self.p2p:
            self.pending[s] = dist.all_gather(self.views[s], self.views[s][self.rank], group=self.group, async_op=True)
            return
        if self.bounds is not None:
            buf = self.render[s]
            ops = []
            for p in range(self.nranks):
                if p == self.rank:
                    continue
                r0, rn = self.bounds[p]
                peer = p if self.group is None else dist.get_global_rank(self.group, p)
                ops.append(dist.P2POp(dist.isend, buf[self.row0:self.row0 + self.rows], peer, self.group))
                ops.append(dist.P2POp(dist.irecv, buf[r0:r0 + rn], peer, self.group))
            self.pending[s] = _Batch(dist.batch_isend_irecv(ops)) if ops else None
            return
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
        if self.bounds is not None:
            return self.render[s]
        f = self.frames[s]
        if self.uniform:
            return f[:self.H]
        parts = []
        for r in range(self.nranks):
            rows = strip_rows(self.H, self.nranks, r)[1]
            parts.append(f[r * self.max_rows:r * self.max_rows + rows])
        return torch.cat(parts, dim=0)


class StripExchange:
    """The frame all-gather behind the C ABI (crychic_allgather_frame, csrc/comm.cpp): RCCL over xGMI, enqueued on the
    stream that rendered the strip, no host wait and no Python collective per frame.

    Every rank renders its strip IN PLACE into one of `slots` full-frame buffers and the exchange fills in the peers'
    rows; a slot is reused only by work on the same stream, so stream order alone keeps frame i's gather ahead of
    frame i + slots' lighting pass.  bounds = None: the crychic_strip_rows plan (equal strips, one in-place
    ncclAllGather); bounds = [(row0, rows)] per rank: any heights (one group of in-place ncclBroadcasts)."""

    def __init__(self, ctx, W, H, nranks, rank, unique_id, bounds=None, slots=4):
        self.ctx, self.W, self.H, self.nranks, self.rank = ctx, int(W), int(H), int(nranks), int(rank)
        assert len(unique_id) == _lib_comm_id_bytes()
        self.bounds = [tuple(int(v) for v in b) for b in bounds] if bounds is not None else None
        if self.bounds is not None:
            assert len(self.bounds) == nranks and sum(b[1] for b in self.bounds) == H and self.bounds[0][0] == 0
            flat = [v for b in self.bounds for v in b]
            self._bounds_arr = (C.c_uint32 * len(flat))(*flat)
            self.row0, self.rows = self.bounds[rank]
        else:
            self._bounds_arr = None
            self.row0, self.rows = strip_rows(H, nranks, rank)
        self.handle = C.c_void_p()
        idbuf = (C.c_uint8 * len(unique_id)).from_buffer_copy(bytes(unique_id))
        check(lib.crychic_comm_create(ctx.handle, self.nranks, self.rank, idbuf, C.byref(self.handle)))
        self.render = [torch.zeros((H, W, 4), dtype=torch.uint8, device=ctx.device) for _ in range(int(slots))]

    def set_bounds(self, bounds):
        """Switch the strip plan (None = equal strips) on the same communicator; the caller makes sure every rank does so
        between the same two frames."""
        self.bounds = [tuple(int(v) for v in b) for b in bounds] if bounds is not None else None
        if self.bounds is not None:
            assert len(self.bounds) == self.nranks and sum(b[1] for b in self.bounds) == self.H and self.bounds[0][0] == 0
            flat = [v for b in self.bounds for v in b]
            self._bounds_arr = (C.c_uint32 * len(flat))(*flat)
            self.row0, self.rows = self.bounds[self.rank]
        else:
            self._bounds_arr = None
            self.row0, self.rows = strip_rows(self.H, self.nranks, self.rank)

    @staticmethod
    def new_unique_id():
        """Rank 0 calls this and hands the bytes to its peers out of band (ncclGetUniqueId)."""
        buf = (C.c_uint8 * _lib_comm_id_bytes())()
        check(lib.crychic_comm_unique_id(buf))
        return bytes(buf)

    def strip_buffer(self, i):
        return self.render[i % len(self.render)]

    def launch(self, i, stream=None):
        """Enqueue the exchange of frame i's buffer on `stream` (default: torch's current stream)."""
        s = torch.cuda.current_stream(self.ctx.device) if stream is None else stream
        check(lib.crychic_allgather_frame(self.handle, C.c_void_p(self.render[i % len(self.render)].data_ptr()), self.W, self.H,
                                          self._bounds_arr, C.c_void_p(s.cuda_stream)))

    def draw(self, app, i, parts):
        """Frame i on torch's current stream through crychic_draw_hot_path_shared: app's strip into this slot's buffer and the
        exchange, overlapped in `parts` row ranges (1 = the strip, then one exchange -- what Draw + launch do)."""
        app.mBackBuffer = self.render[i % len(self.render)]
        app.Draw(self.row0, self.rows, shared=(self.handle, self._bounds_arr, parts))

    def barrier(self, stream=None):
        """Stream-ordered rendezvous of all ranks (one-word all-reduce); the caller synchronises the stream."""
        s = torch.cuda.current_stream(self.ctx.device) if stream is None else stream
        check(lib.crychic_comm_barrier(self.handle, C.c_void_p(s.cuda_stream)))

    def wait_all(self):
        torch.cuda.synchronize(self.ctx.device)
        check(lib.crychic_comm_async_error(self.handle))

    def frame(self, i):
        return self.render[i % len(self.render)]

    def abort(self):
        if self.handle:
            lib.crychic_comm_abort(self.handle)

    def close(self):
        if self.handle:
            lib.crychic_comm_destroy(self.handle)
            self.handle = C.c_void_p()


def _lib_comm_id_bytes():
    from ._lib import COMM_ID_BYTES
    return COMM_ID_BYTES
