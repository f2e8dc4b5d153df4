"""hipGraph capture of one whole training step (forward + loss + backward).

At IEMOCAP sizes a step is a few hundred short kernels; launched eagerly the host (Python + autograd
dispatch) is the bottleneck, not the GPU.  For a fixed batch signature (same dialogue lengths) the step is
static, so it is captured once into a hipGraph and replayed: one launch per step.  Gradients are written
into persistent tensors (views of the data-parallel flat bucket when one is given), so an eager RCCL
all-reduce / optimizer step can follow each replay.
"""
import torch

from . import ops
from .layout import recording


class CapturedStep:
    def __init__(self, model, step_fn, warmup=3, bucket=None, reduce_in_graph=False):
        """step_fn() must run forward + backward on STATIC input tensors and return the loss tensor.
        The step is captured with every ``.grad`` set to None, so autograd simply hands its gradient
        buffers over (no zero-fill, no accumulate kernels); with a ``bucket`` the captured graph ends with
        the single multi-tensor pack into the flat all-reduce buffer and, with ``reduce_in_graph``, the RCCL
        all-reduce of that buffer as a graph node (no host launch between backward and the collective)."""
        self.model = model
        self.bucket = bucket
        # the graph bakes the addresses of every tensor step_fn closes over (static inputs, index tensors, labels):
        # holding the closure keeps them allocated for the lifetime of the graph
        self._step_fn = step_fn

        def tail():
            ops.join_weight_grads()          # flush weight gradients still queued (normally the engine callback did)
            if bucket is not None:
                bucket.flatten()
                if reduce_in_graph:
                    bucket.reduce_flat()
        # the warm-up passes and the capture draw dropout flags: put BOTH generators back afterwards -- torch's CUDA generator
        # and the package's device-resident Philox state (ops.draw_flags), which replays advance on the device -- so that
        # building a captured step consumes no random numbers.  Replays then move torch's generator by what the graph
        # consumed (recorded here), so an eager draw that follows (the warm-up of the next cache miss) continues the stream
        # instead of re-seeding it from a stale offset (overlapping dropout masks during the first epoch).
        dev_index = self._dev_index = torch.cuda.current_device()
        rng_state = torch.cuda.get_rng_state(dev_index)
        flag_snap = ops.flag_state_snapshot(dev_index)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                model.zero_grad(set_to_none=True)
                step_fn()
                tail()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        model.zero_grad(set_to_none=True)
        consumed0 = ops.flags_consumed(dev_index)
        from . import dialogue_model, layout
        # the graph bakes raw pointers of the cached index tensors (dialogue layout, pad-strip index): keep every
        # cache entry used during the capture alive for as long as this object lives, whatever the caches evict
        # weight piece planes the step reads (ops_linear.weight_planes): fresh after the warm-up passes, so the graph contains no
        # cut; replay() re-cuts the ones an optimizer step has made stale, in front of the graph launch
        with recording(layout._LAYOUT_CACHE, dialogue_model._FLAT_CACHE) as used, ops.planes_recording() as planes:
            # thread_local: a communication-library watchdog thread polling its own events must not invalidate the capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss = step_fn()
                tail()
        self._pinned = list(used)
        self._planes = list(planes)
        self._rng_per_replay = ops.flags_consumed(dev_index) - consumed0      # Philox counters one replay consumes
        torch.cuda.set_rng_state(rng_state, dev_index)
        ops.flag_state_restore(dev_index, flag_snap)
        self.grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        # the parameter storages are baked in as well: a later re-pointing of p.data (FlatAdam._materialise, .to(),
        # load_state_dict(assign=True)) would make the graph update stale memory -- checked on every replay
        self._param_ptrs = [(p, p.data_ptr()) for p in model.parameters()]

    def close(self):
        """Release the captured graph NOW (hipGraphExec + its private memory pool) instead of whenever the last reference to this
        object dies.  A graph that holds a collective node keeps the RCCL communicator's work objects alive: it has to be gone
        before ``torch.distributed.destroy_process_group()`` (bench.py, the RCCL tests).  The object is unusable afterwards."""
        graph, self.graph = self.graph, None
        self.loss = None
        self.grads = {}
        self._pinned = []
        self._planes = []
        self._step_fn = None
        if graph is not None:
            torch.cuda.synchronize()
            graph.reset()

    def replay(self):
        if self.graph is None:
            raise RuntimeError("CapturedStep: replay() after close()")
        for p, ptr in self._param_ptrs:
            if p.data_ptr() != ptr:
                raise RuntimeError("CapturedStep: a parameter's storage moved after the capture (optimizer "
                                   "materialisation / .to() / load_state_dict(assign=True)); capture again")
        if self._rng_per_replay:
            ops.flag_state_sync(self._dev_index)           # torch.manual_seed / a restored RNG state since the last draw
        if self._planes:
            ops.refresh_planes(self._planes)               # (host-side version check; a launch only after a weight update)
        self.graph.replay()
        if self._rng_per_replay:
            ops.flags_advance_host(self._dev_index, self._rng_per_replay)
        return self.loss
