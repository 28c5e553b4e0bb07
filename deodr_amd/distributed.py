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


class SharedGradientBuffer:
    """What the views of a multi-view fit share, as ONE flat buffer for ONE collective (the packed buffer
    ``[vertices_b | colors_b (| uv_b | texture_b)]``, 13 MB with a 1024^2 RGB texture in float32):

        [ texture_b  Ht x Wt x C | vertices_b  V x 3 | colors_b  V x C | uv_b  Vuv x 2 ]

    ``texture_b`` and ``uv_b`` are fields of the reference's ``struct Scene`` that the views of one scene share
    (DifferentiableRenderer.h:56-90; ``bilinear_sample_B``, H.h:563-631, adds every view's taps into the same array); ``vertices_b`` is the
    sum over the views of the camera adjoint applied to ``ij_b`` (deodr/mesh_fitter.py:518-527), ``colors_b`` the sum over the views.  The
    texture part leads so that it stays 16-byte aligned whatever V.  ``dtype``: the buffer's (and the collective's) element type."""

    def __init__(self, V, C, n_uv=0, texture_shape=None, dtype=torch.float64, device="cpu"):
        self.V, self.C, self.n_uv = int(V), int(C), int(n_uv)
        self.texture_shape = None if texture_shape is None else tuple(int(x) for x in texture_shape)
        n_tex = 0 if self.texture_shape is None else int(torch.Size(self.texture_shape).numel())
        n_small = self.V * (3 + self.C) + 2 * self.n_uv
        self.flat = torch.zeros(n_tex + n_small, dtype=dtype, device=device)
        self.texture_b = self.flat[:n_tex].view(self.texture_shape) if n_tex else None
        self.small = self.flat[n_tex:]
        self.vertices_b, self.colors_b, self.uv_b = self.split_small(self.small)

    def split_small(self, small):
        """(vertices_b [V,3], colors_b [V,C], uv_b [Vuv,2] | None) as views of a flat tensor laid out like ``self.small``"""
        V, C, n_uv = self.V, self.C, self.n_uv
        return small[: 3 * V].view(V, 3), small[3 * V : V * (3 + C)].view(V, C), (small[V * (3 + C) :].view(n_uv, 2) if n_uv else None)

    def all_reduce(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        return self


class OverlappedViewsReduction:
    """The reduction of a sharded multi-view fit -- the gradient of what the views share (mesh vertices through every view's camera, vertex
    colours, and for a textured scene uv_b and texture_b), summed over the local views and over the ranks -- on a COMMUNICATION STREAM, while
    the render stream already works on the next step.  Device-resident (ROCm tensors, "nccl" = RCCL, or gloo in tests); `sets` alternating
    sets of gradient buffers.

        red = OverlappedViewsReduction(ds, camera, posed)
        for every step:
            slot = red.begin()                          # (the host waits until the reduction that last read this set has read it)
            rasterizer.render_fit(ds, obs, sigma, grads=slot.grads, clear_grads=True, done_flag=slot.done_flag)
            red.reduce(slot)                            # queued on the communication stream
            ... slot.vertices_b [V,3], slot.colors_b [V,C] (slot.uv_b [Vuv,2], slot.texture_b [Ht,Wt,C]) are the all-reduced sums once
            red.wait(slot) returns
        red.finish()

    The hand-over between the two streams is a word of device memory that the fit step stores when its gradients are complete
    (``DeodrHipFitOptions::done_flag``) and a one-lane kernel on the communication stream that waits for it (``deodr_hip_wait_flag``): an
    event recorded on the render stream and waited for by a second hardware queue costs the render stream ~8 us per step on MI355X, the flag
    1.6.  ``reduce(slot, event=...)`` takes an event instead, for steps that cannot store the flag (the two-call path).
    The collective is issued under the communication stream; with one rank it is skipped unless ``always_collective``.  ``status_every``: steps
    between two looks at the time-out word of the flag waits (``begin()`` raises at the next look; 64 by default, 1 = every step: see ``__init__``).

    Textured scenes (``texture`` = None: when the scene has one): the library adds every local view's texture taps and uv adjoints into ONE
    array each (they are per-scene fields of the reference's struct), so ``slot.grads["texture_b"]`` IS the texture part of the packed buffer
    (no copy of the 12.6 MB) and ``uv_b`` lies next to the vertex sums.  The buffer has the scene's PIXEL dtype -- float32 buffers: one 13 MB
    float32 collective for a 1024^2 RGB texture, the float64 vertex / colour / uv sums (0.3 % of it) converted on the way in; float64
    buffers: no conversion.  ``shade_b`` stays per view: its way to the shared parameters is the shading adjoint of the caller
    (``deodr_hip_vertex_shade_b``)."""

    class Slot:
        __slots__ = ("index", "grads", "buffer", "shared", "stage", "stage_parts", "vertices_b", "colors_b", "uv_b", "texture_b", "done_flag", "read", "step",
                     "status_host")

    def __init__(self, ds, camera, posed, group=None, sets=2, always_collective=False, wait_timeout=2.0, texture=None, status_every=64):
        from . import fronthalf

        self.device = ds.device
        self.camera, self.posed, self.group = camera, posed, group
        V, C = int(posed.shape[1]), int(ds.nb_colors)
        if ds.vertex_dtype != torch.float64:
            raise ValueError("OverlappedViewsReduction: the scene's vertex arrays must be float64 (deodr_hip_views_gradient_sum reads float64)")
        if not (posed.is_cuda and posed.dtype == torch.float64 and posed.is_contiguous() and tuple(posed.shape) == (ds.n_views, V, 3)):
            raise ValueError(f"OverlappedViewsReduction: posed must be a contiguous float64 ROCm tensor [{ds.n_views}, {V}, 3]")
        self.textured = (ds.texture is not None) if texture is None else bool(texture)
        if self.textured and ds.texture is None:
            raise ValueError("OverlappedViewsReduction(texture=True): the scene has no texture")
        self.comm = torch.cuda.Stream(device=self.device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.wait_status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.wait_timeout = float(wait_timeout)
        # Steps between two looks at the time-out word of the flag waits.  A wait that timed out lets every later wait return at once, so between the
        # time-out and the next look begin() keeps handing out sets whose reductions read INCOMPLETE gradients -- for up to `status_every` steps.
        # 64: the copy of the word to pinned memory costs the communication stream ~10 us of host time per look, more than the kernels it follows;
        # 1: every step is checked (a fit loop that cannot afford a bad step; finish() always checks).  The word only changes when a flag wait gives
        # up after `wait_timeout` seconds, i.e. when the render stream has hung or died.
        self.status_every = max(1, int(status_every))
        self.collective = always_collective or (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1)
        self.steps = 0
        self.slots = []
        for i in range(int(sets)):
            s = self.Slot()
            s.index, s.grads, s.read, s.step = i, ds.zero_grads(), None, 0
            if self.textured:
                s.buffer = SharedGradientBuffer(V, C, int(ds.uv.shape[0]), tuple(ds.texture.shape), ds.pixel_dtype, self.device)
                # the float64 sums of the launch below (and the library's uv_b) in the buffer itself, or in a float64 stage next to it
                s.stage = s.buffer.small if ds.pixel_dtype == torch.float64 else torch.zeros_like(s.buffer.small, dtype=torch.float64)
                s.grads["texture_b"] = s.buffer.texture_b
                s.grads["uv_b"] = s.buffer.split_small(s.stage)[2]
            else:
                s.buffer = SharedGradientBuffer(V, C, 0, None, torch.float64, self.device)
                s.stage = s.buffer.small
            s.stage_parts = s.buffer.split_small(s.stage)  # where the launch of reduce() writes (views formed once: no tensor ops per step)
            fronthalf._validate_views_gradient_sum(posed, camera, s.grads["ij_b"], s.stage_parts[0], None, s.grads["colors_b"], s.stage_parts[1])
            s.shared = s.buffer.flat  # the packed buffer of the collective
            s.vertices_b, s.colors_b, s.uv_b, s.texture_b = s.buffer.vertices_b, s.buffer.colors_b, s.buffer.uv_b, s.buffer.texture_b
            s.done_flag = None
            s.status_host = torch.zeros(1, dtype=torch.int32).pin_memory()  # the time-out word as the communication stream last saw it
            self.slots.append(s)

    def begin(self):
        """-> the set of buffers of the next step (its ``grads`` for the fit step, its ``done_flag`` = (tensor, step number)).  Raises when
        a flag wait of an earlier step timed out (every later wait returns at once then: the reductions since are incomplete)."""
        s = self.slots[self.steps % len(self.slots)]
        self.steps += 1
        if s.read is not None:
            s.read.synchronize()  # (the host, not the render stream: one packet fewer between two steps)
            # the status word was copied to pinned memory on the communication stream ahead of that event: reading it here synchronises
            # nothing (an .item() on the device word would wait for the RENDER stream at every step)
            if int(s.status_host[0]) != 0:
                self._raise_timed_out()
        s.step = self.steps
        s.done_flag = (self.flag, s.step)
        return s

    def _raise_timed_out(self):
        raise RuntimeError("deodr_hip: a wait for the step-done flag timed out (deodr_hip_wait_flag): the shared gradient of that step, and of "
                           "every step since, is incomplete")  # fmt: skip

    def reduce(self, slot, event=None):
        from . import fronthalf, hip_renderer

        with torch.cuda.stream(self.comm):
            if event is not None:
                self.comm.wait_event(event)
            else:
                hip_renderer.wait_flag(self.flag, slot.step, status=self.wait_status, timeout=self.wait_timeout)
            # one launch: every view's projection adjoint applied to ij_b and summed over the views, the colour gradients summed over the views,
            # both straight into the packed buffer (which the collective of `sets` steps ago, earlier on this stream, has left)
            fronthalf.views_gradient_sum(self.posed, self.camera, slot.grads["ij_b"], slot.stage_parts[0], colors_b=slot.grads["colors_b"],
                                         colors_sum=slot.stage_parts[1], validate=False)  # (validated once, in __init__)
            if slot.stage is not slot.buffer.small:
                slot.buffer.small.copy_(slot.stage)  # float64 sums -> the float32 buffer (one cast kernel over V (3 + C) + 2 Vuv values)
            if slot.step % self.status_every < len(self.slots):
                # (every step would be a device-to-host copy per step on this stream: ~10 us of host time, more than the kernels it follows)
                slot.status_host.copy_(self.wait_status, non_blocking=True)
            if not self.textured:
                slot.read = torch.cuda.Event()
                slot.read.record()
            if self.collective:
                dist.all_reduce(slot.shared, op=dist.ReduceOp.SUM, group=self.group)
            if self.textured:
                # the library accumulates the next step's texture_b / uv_b straight into this buffer: the set is free again only once the
                # collective has read it (with two sets that is a step ago by the time begin() asks)
                slot.read = torch.cuda.Event()
                slot.read.record()

    def wait(self, slot=None):
        """the current stream waits for the reductions queued so far (of `slot`, or of all)"""
        torch.cuda.current_stream(self.device).wait_stream(self.comm)

    def finish(self):
        """host-side: every queued reduction has finished; raises if a flag wait timed out (its reduction read incomplete gradients)"""
        self.comm.synchronize()
        if int(self.wait_status.item()) != 0:
            self._raise_timed_out()
