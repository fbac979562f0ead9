"""Runs on the GPU box: the benchmark frame on the CRY_PROBE_TIMING build; prints the average stage times of a lit wavefront
of light_kernel (s_memtime stamps, 100 MHz constant clock)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRYCHIC_LIB"] = os.path.join(ROOT, "tools", "_probe", "lib_timing.so")
sys.path.insert(0, ROOT)
import torch
from crychic_renderer_amd import Context, Crychic, scene
from crychic_renderer_amd._lib import lib
W, H = 3840, 2160
cam = sys.argv[1] if len(sys.argv) > 1 else "reference"
ctx = Context(0)
consts = scene.Constants(W, H, 4096, cam=scene.covered_camera(W, H) if cam == "covered" else scene.default_camera(W, H))
planes = scene.make_scene(W, H, shadow_dim=4096, cube_dim=256, device="cuda:0", consts=consts)
app = Crychic(ctx, W, H, planes["randvec"], planes["cube"], shadow_dim=4096)
app.load_scene(planes)
app.blurCount, app.numDirLights, app.pcfSearchRadius = 4, 3, 0.0
app.mBackBuffer = planes["out"]
for _ in range(20): app.Draw(0, H)
torch.cuda.synchronize()
import numpy as np
NW = (W // 64) * H
buf = np.zeros((NW, 8), dtype=np.uint64)
lib.crychic_probe_read.argtypes = [C.c_void_p, C.c_int]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(20): app.Draw(0, H)
torch.cuda.synchronize()
app.set_profiling(True)
app.Draw(0, H)
lt = app.last_pass_times()["light_ms"]
torch.cuda.synchronize()
lib.crychic_probe_read(buf.ctypes.data, NW)
t = buf.astype(np.int64)
lit = t[:, 2] != 0
tpu = 2270.0      # s_memtime ticks per us: the shader clock of the loaded GPU (SQ_BUSY_CYCLES / 32 over the kernel's duration in the PMC passes: 2.27 GHz);
                  # the counter is per XCD, so only differences inside one wavefront mean anything
print("camera", cam, "light pass %.1f us by events (probe build)" % (lt * 1e3))
print("lit wavefronts", int(lit.sum()), "of", NW)
names = ["entry -> depth arrived", "depth -> G-buffer arrived", "G -> decode, ambient + cube gathers arrived", "-> cascade lookups done", "-> lights done", "-> tone map (table loads) + reflection done"]
for k, name in enumerate(names):
    d = (t[lit, k + 1] - t[lit, k])
    print("%-46s mean %.3f us   median %.3f us" % (name, d.mean() / tpu, np.median(d) / tpu))
d = t[lit, 6] - t[lit, 0]
print("%-46s mean %.3f us" % ("lit wavefront, entry -> last stamp", d.mean() / tpu))
sk = ~lit
print("%-46s mean %.3f us (%d waves)" % ("sky wavefront, entry -> depth arrived", ((t[sk, 1] - t[sk, 0]).mean() / tpu) if sk.any() else 0.0, int(sk.sum())))
