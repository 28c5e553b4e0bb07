"""Round 5, VERDICT item 4: two half-batches of the bench workload on two streams -- does the latency-bound set-up / finalize of one
half hide under the issue-bound forward raster of the other?

    python tools/two_stream_probe.py [--views 8] [--steps 200] [--lib path]

Modes (all move the same 8 views per step; a "step" of a two-stream mode = one fit step of each half):
  one        8 views, one call, one stream (the bench's step)
  free2      4 + 4 views, two DeviceScenes / workspaces / streams, free-running (no ordering between the streams, none between steps)
  offset2    as free2, the second stream started half a step late once (a device-side sleep), then free-running
  lock2      as free2 with the streams in lock step through the step-done flags: half B's step k is queued behind a wait for half A's
             step k - 0 ... (A(k) -> B(k) -> A(k+1): no overlap at all; the lower bound of what ordering costs)
  stag2      staggered by flags: A's step k + 1 waits for B's step k - 1, B's step k waits for A's step k - 0's FORWARD to be under way is
             not expressible with end-of-step flags; so: A(k+1) waits for done(B, k-1), B(k) waits for done(A, k-1): at most one step of
             skew, the overlap is left to the hardware queues
Prints ms per 8 views and the Mpixel/s of each."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import hip_renderer as hr
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

arg = lambda name, default: type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
if "--lib" in sys.argv:
    hr.LIB_PATH = os.path.abspath(arg("--lib", ""))
B, S, steps = arg("--views", 8), arg("--size", 1024), arg("--steps", 200)
dev = torch.device("cuda:0")
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
s0 = views[0]


class Half:
    def __init__(self, vs):
        stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in vs])
        self.ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                              stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
                              vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)  # fmt: skip
        self.r = HipRasterizer.for_scene(self.ds)
        n, C = len(vs), self.ds.nb_colors
        self.obs = torch.rand((n, S, S, C), dtype=torch.float32, device=dev)
        self.image = torch.empty((n, S, S, C), dtype=torch.float32, device=dev)
        self.z = torch.empty((n, S, S), dtype=torch.float32, device=dev)
        self.grads = self.ds.zero_grads()
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.k = 0  # flagged steps so far (= the value the flag holds once the last of them has finished)
        self.r.render(self.ds, 1.0, out=(self.image, self.z), check_overflow=True)

    def fit(self, flag=False):
        self.k += 1 if flag else 0
        self.r.render_fit(self.ds, self.obs, 1.0, grads=self.grads, out=(self.image, self.z), check_overflow=False, clear_grads=True,
                          done_flag=(self.flag, self.k) if flag else None)  # fmt: skip

    def wait_for(self, other, k):
        if k > 0:
            hr.wait_flag(other.flag, k, status=self.status, timeout=2.0)


def timed(step, n, sync):
    for _ in range(30):
        step()
    sync()
    best = 1e9
    for _rep in range(3):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


def report(name, t):
    print(f"{name:9s} {t * 1e3:.4f} ms per {B} views = {B * S * S / t / 1e6:.0f} Mpixel/s   (8d fraction of 8 TB/s: {43.22e6 * B / t / 8e12:.3f})", flush=True)


only = arg("--only", "")
if only == "free2":  # (for a rocprofv3 --kernel-trace run: tools/trace_timeline.py reads the csv)
    a, b = Half(views[: B // 2]), Half(views[B // 2 :])
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(steps):
        with torch.cuda.stream(sa):
            a.fit()
        with torch.cuda.stream(sb):
            b.fit()
    torch.cuda.synchronize()
    sys.exit(0)
whole = Half(views)
report("one", timed(whole.fit, steps, torch.cuda.synchronize))
del whole
a, b = Half(views[: B // 2]), Half(views[B // 2 :])
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def free2():
    with torch.cuda.stream(sa):
        a.fit()
    with torch.cuda.stream(sb):
        b.fit()


report("free2", timed(free2, steps, torch.cuda.synchronize))
torch.cuda.synchronize()
with torch.cuda.stream(sb):
    torch.cuda._sleep(int(60e-6 * 2.4e9))  # ~ half a step at 2.4 GHz
report("offset2", timed(free2, steps, torch.cuda.synchronize))


def stag2():
    with torch.cuda.stream(sa):
        a.wait_for(b, b.k - 1)
        a.fit(True)
    with torch.cuda.stream(sb):
        b.wait_for(a, a.k - 1)
        b.fit(True)


report("stag2", timed(stag2, steps, torch.cuda.synchronize))
assert int(a.status.item()) == 0 and int(b.status.item()) == 0, "a flag wait timed out"


def lock2():
    with torch.cuda.stream(sa):
        a.wait_for(b, b.k)
        a.fit(True)
    with torch.cuda.stream(sb):
        b.wait_for(a, a.k)
        b.fit(True)


report("lock2", timed(lock2, steps, torch.cuda.synchronize))
assert int(a.status.item()) == 0 and int(b.status.item()) == 0, "a flag wait timed out"
# four quarter-batches, free-running
qs = [Half(views[i * B // 4 : (i + 1) * B // 4]) for i in range(4)]
ss = [torch.cuda.Stream() for _ in range(4)]


def free4():
    for h, st in zip(qs, ss):
        with torch.cuda.stream(st):
            h.fit()


report("free4", timed(free4, steps, torch.cuda.synchronize))
