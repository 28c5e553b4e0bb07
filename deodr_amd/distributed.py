"""Multi-GPU layer of the hot path: independent views shard across ranks, ONE all-reduce of the shared gradients.

The reference renders the frames of a multi-view fit in a Python loop on one core and sums the gradients of the shared
parameters as it goes (deodr/mesh_fitter.py:518-527, 536-546).  Here every rank (one process per GPU, torch.distributed
with the "nccl" backend = RCCL over xGMI; "gloo" in the CPU tests) renders its own views and the only communication is the
sum of the packed shared-parameter gradient.  Nothing on the data path of a view ever crosses GPUs.
"""

import torch
import torch.distributed as dist


def shard_views(n_views, rank, world_size):
    """Contiguous block of view indices owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(int(n_views), int(world_size))
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


class PackedGradients:
    """Flat buffer holding several gradient tensors back to back, so that ONE collective moves all of them (a 6 KB message
    for the hand mesh is latency-bound, ~13 MB with a 1024^2 texture is one xGMI-link-bound ring)."""

    def __init__(self, shapes, dtype=torch.float64, device="cpu"):
        self.shapes = [tuple(s) for s in shapes]
        self.sizes = [int(torch.Size(s).numel()) for s in self.shapes]
        self.flat = torch.zeros(sum(self.sizes), dtype=dtype, device=device)

    def pack(self, tensors):
        off = 0
        for t, n in zip(tensors, self.sizes):
            self.flat[off : off + n].copy_(t.reshape(-1))
            off += n
        return self.flat

    def unpack(self):
        out, off = [], 0
        for shape, n in zip(self.shapes, self.sizes):
            out.append(self.flat[off : off + n].view(shape))
            off += n
        return out


def allreduce_shared_gradients(packed, tensors, group=None):
    """Sum `tensors` (already reduced over the local views) over all ranks; returns views into the packed buffer."""
    packed.pack(tensors)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed.flat, op=dist.ReduceOp.SUM, group=group)
    return packed.unpack()


class OverlappedViewsReduction:
    """The reduction of a sharded multi-view fit -- the gradient of what the views share (mesh vertices through every view's camera, vertex
    colours), summed over the local views and over the ranks -- on a COMMUNICATION STREAM, while the render stream already works on the next
    step.  Device-resident (ROCm tensors, "nccl" = RCCL, or gloo in tests); `sets` alternating sets of gradient buffers.

        red = OverlappedViewsReduction(ds, camera, posed)
        for every step:
            slot = red.begin()                          # (the host waits until the reduction that last read this set has read it)
            rasterizer.render_fit(ds, obs, sigma, grads=slot.grads, clear_grads=True, done_flag=slot.done_flag)
            red.reduce(slot)                            # queued on the communication stream
            ... slot.vertices_b [V,3], slot.colors_b [V,C] are the all-reduced sums once red.wait(slot) returns
        red.finish()

    The hand-over between the two streams is a word of device memory that the fit step stores when its gradients are complete
    (``DeodrHipFitOptions::done_flag``) and a one-lane kernel on the communication stream that waits for it (``deodr_hip_wait_flag``): an
    event recorded on the render stream and waited for by a second hardware queue costs the render stream ~8 us per step on MI355X, the flag
    1.6.  ``reduce(slot, event=...)`` takes an event instead, for steps that cannot store the flag (the two-call path).
    The collective is issued under the communication stream; with one rank it is skipped unless ``always_collective``."""

    class Slot:
        __slots__ = ("index", "grads", "shared", "vertices_b", "colors_b", "done_flag", "read", "step")

    def __init__(self, ds, camera, posed, group=None, sets=2, always_collective=False, wait_timeout=2.0):
        self.device = ds.device
        self.camera, self.posed, self.group = camera, posed, group
        V, C = int(posed.shape[1]), int(ds.nb_colors)
        self.comm = torch.cuda.Stream(device=self.device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.wait_status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.wait_timeout = float(wait_timeout)
        self.collective = always_collective or (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1)
        self.steps = 0
        self.slots = []
        for i in range(int(sets)):
            s = self.Slot()
            s.index, s.grads, s.read, s.step = i, ds.zero_grads(), None, 0
            s.shared = torch.zeros(V * (3 + C), dtype=torch.float64, device=self.device)  # the packed buffer of the collective
            s.vertices_b, s.colors_b = s.shared[: 3 * V].view(V, 3), s.shared[3 * V :].view(V, C)
            s.done_flag = None
            self.slots.append(s)

    def begin(self):
        """-> the set of buffers of the next step (its ``grads`` for the fit step, its ``done_flag`` = (tensor, step number))"""
        s = self.slots[self.steps % len(self.slots)]
        self.steps += 1
        if s.read is not None:
            s.read.synchronize()  # (the host, not the render stream: one packet fewer between two steps)
        s.step = self.steps
        s.done_flag = (self.flag, s.step)
        return s

    def reduce(self, slot, event=None):
        from . import fronthalf, hip_renderer

        with torch.cuda.stream(self.comm):
            if event is not None:
                self.comm.wait_event(event)
            else:
                hip_renderer.wait_flag(self.flag, slot.step, status=self.wait_status, timeout=self.wait_timeout)
            # one launch: every view's projection adjoint applied to ij_b and summed over the views, the colour gradients summed over the views,
            # both straight into the packed buffer (which the collective of `sets` steps ago, earlier on this stream, has left)
            fronthalf.views_gradient_sum(self.posed, self.camera, slot.grads["ij_b"], slot.vertices_b, colors_b=slot.grads["colors_b"], colors_sum=slot.colors_b)
            slot.read = torch.cuda.Event()
            slot.read.record()
            if self.collective:
                dist.all_reduce(slot.shared, op=dist.ReduceOp.SUM, group=self.group)

    def wait(self, slot=None):
        """the current stream waits for the reductions queued so far (of `slot`, or of all)"""
        torch.cuda.current_stream(self.device).wait_stream(self.comm)

    def finish(self):
        """host-side: every queued reduction has finished; raises if a flag wait timed out (its reduction read incomplete gradients)"""
        self.comm.synchronize()
        if int(self.wait_status.item()) != 0:
            raise RuntimeError("deodr_hip: a wait for the step-done flag timed out (deodr_hip_wait_flag): the shared gradient of that step is incomplete")
