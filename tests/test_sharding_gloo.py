"""Multi-GPU path on CPU: world_size 2 and 3 over gloo.  Each rank renders only its own row strip (the oracle stands
in for the GPU renderer -- this file tests the strip plan and the frame all-gather of crychic_renderer_amd.sharding,
not the kernels) and every rank must end up with the complete frame, for uniform and ragged strip sizes, with the
two-slot pipelining used by bench.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_lib, scene_util
from crychic_renderer_amd import sharding
from crychic_renderer_amd._lib import lib
W, H = int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "equal"
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
orc = oracle_lib.load()
pl = scene_util.cpu_scene(W, H, 256, 32); p = scene_util.np_planes(pl); c = pl["consts"]
scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
pcb = oracle_lib.as_oracle_cb(c.pass_cb, oracle_lib.OrPassConstants)
ao = orc.compute_ssao(scb, p["normal"], p["depth"], p["randvec"], 2)
if mode == "balanced":          # cost-aware strips of different heights + point-to-point exchange
    bounds = sharding.balanced_strips(torch.from_numpy(p["depth"].view(np.int32)), world)
    assert sum(b[1] for b in bounds) == H and all(b[1] >= 2 and b[1] % 2 == 0 for b in bounds)
    row0, rows = bounds[rank]
    g = sharding.FrameGather(W, H, world, rank, torch.device("cpu"), bounds=bounds)
else:
    row0, rows = sharding.strip_rows(H, world, rank)
    g = sharding.FrameGather(W, H, world, rank, torch.device("cpu"))
assert (g.row0, g.rows) == (row0, rows)
full = [orc.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], nl, 0.0) for nl in (1, 2, 3)]
for i, nl in enumerate((1, 2, 3)):           # three frames through the two slots
    buf = g.strip_buffer(i)
    buf.fill_(7)
    strip = orc.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], nl, 0.0, row0=row0, rows=rows)
    buf[row0:row0 + rows] = torch.from_numpy(strip[row0:row0 + rows])
    g.launch(i)
g.wait_all()
for i in (1, 2):                              # slots still hold frames 1 and 2
    got = g.frame(i).numpy()
    assert got.shape == (H, W, 4), got.shape
    assert np.array_equal(got, full[i]), "rank %d frame %d differs" % (rank, i)
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,W,H,mode", [(2, 64, 48, "equal"), (3, 64, 40, "equal"), (2, 34, 34, "equal"),
                                            (2, 64, 48, "balanced"), (3, 64, 40, "balanced")])
def test_strip_gather_gloo(built_lib, oracle, tmp_path, world, W, H, mode):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(W), str(H), mode], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-3000:])
        assert "rank %d ok" % r in out


def test_strip_balancer_converges(built_lib):
    """sharding.StripBalancer against a synthetic cost model (fixed cost per strip + cheap sky rows, expensive ground rows):
    plans stay valid (even rows, >= 2 each, contiguous, sum = H) and the slowest strip gets faster."""
    import torch
    from crychic_renderer_amd import sharding
    H, Wd = 2160, 16
    depth = torch.full((H, Wd), 0xFFFFFF, dtype=torch.int32)
    depth[1000:] = 12345                                               # ground below full-res row 1000
    cost_row = np.where(np.arange(H // 2) < 500, 0.05, 0.22)          # microseconds per half-res row

    def times(bounds):
        return [60.0 + float(cost_row[b[0] // 2:(b[0] + b[1]) // 2].sum()) for b in bounds]

    for n in (2, 3, 4, 8):
        equal = [sharding.strip_rows(H, n, r) for r in range(n)]
        bal = sharding.StripBalancer(depth, n)
        bounds = bal.bounds()
        assert bounds[0][1] > bounds[-1][1]                            # the sky strip starts out taller
        for _ in range(5):
            bounds = bal.update(times(bounds))
            assert sum(b[1] for b in bounds) == H and bounds[0][0] == 0
            assert all(b[1] >= 2 and b[1] % 2 == 0 and b[0] % 2 == 0 for b in bounds)
            assert all(bounds[i][0] + bounds[i][1] == bounds[i + 1][0] for i in range(n - 1))
        t0, t1 = times(equal), times(bounds)
        assert max(t1) < max(t0) * 0.93, (n, max(t0), max(t1))
        assert max(t1) - min(t1) < 0.25 * (max(t0) - min(t0)), (n, t0, t1)
    # same inputs -> same plan (every rank computes it independently)
    a, b = sharding.StripBalancer(depth, 4), sharding.StripBalancer(depth, 4)
    assert a.update([1.0, 2.0, 3.0, 4.0]) == b.update([1.0, 2.0, 3.0, 4.0])
