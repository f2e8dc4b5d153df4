"""Data parallelism over dialogues: one process per GPU, ONE flat-bucket gradient all-reduce per step.

The reference is single-process; dialogues never interact (GRUs are per sequence,
the adjacency is block-diagonal per dialogue, the loss is a mean over utterances), so the batch
shards with no data-path collective.  The only exchange is the gradient sum: all live-parameter
gradients are views into ONE contiguous fp32 buffer (~4-5 MB) which is all-reduced in a single
RCCL call over xGMI (backend "nccl" on ROCm) -- one large message instead of ~50 small ones,
which is what a point-to-point xGMI ring wants.  Parameters the MM-DFN configuration never
reaches (38 of 86 tensors get no gradient) are excluded, as Adam skips them in the reference.

Exactness vs one big batch: FocalLoss is a mean over the GLOBAL utterance count, so each rank
scales its local mean loss by n_local * world / n_global before backward and the bucket is
averaged (sum / world).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend=backend)


def all_reduce_scalar(value, device=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item())


def shard_dialogues(lengths, world, rank):
    """Balance dialogues over ranks by sum(L_i^2) (adjacency cost), greedy longest-first; returns indices."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    load = [0] * world
    bins = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        bins[r].append(i)
        load[r] += lengths[i] * lengths[i]
    return sorted(bins[rank])


def bucket_order(model, live):
    """Flat-buffer order of the live parameters: model order, except that every ``weight_ih_l*_reverse`` / ``bias_ih_l*_reverse`` directly
    follows its forward twin -- the fused GRU path reads the two as one stacked (600, K) operand (gru._stacked_view), and FlatAdam
    lays the parameters out in this order."""
    names = {id(p): n for n, p in model.named_parameters()}
    by_name = {names[id(p)]: p for p in live}
    out, placed = [], set()
    for p in live:
        if id(p) in placed:
            continue
        out.append(p)
        placed.add(id(p))
        n = names[id(p)]
        if (".weight_ih_l" in "." + n or ".bias_ih_l" in "." + n) and not n.endswith("_reverse"):
            twin = by_name.get(n + "_reverse")
            if twin is not None and id(twin) not in placed:
                out.append(twin)
                placed.add(id(twin))
    return out


# ---- flat-buffer slots.  A parameter occupies ``slot_size(p)`` floats of the flat gradient / parameter / moment buffers:
# its element count rounded up to a multiple of 4 (every slot starts 16-byte aligned: the kernels fetch weight rows in
# 16-byte units), or rows x padded width for a dense weight whose contraction width is not a multiple of 4 (ops.py,
# "row padding": the parameter and its gradient are (N, K) views of (N, Kp) blocks whose pad columns are zero).
def _row_padded(p):
    return p.dim() == 2 and p.shape[1] % 4 != 0


def slot_size(p):
    if _row_padded(p):
        return p.shape[0] * ((p.shape[1] + 3) & ~3)
    return (p.numel() + 3) & ~3


def slot_view(flat, off, p):
    """The tensor of p's shape that lives in flat[off : off + slot_size(p)]."""
    if _row_padded(p):
        Kp = (p.shape[1] + 3) & ~3
        return flat[off:off + p.shape[0] * Kp].view(p.shape[0], Kp)[:, :p.shape[1]]
    return flat[off:off + p.numel()].view_as(p)


_ZERO_PAD = {}     # (device, n) -> n zeros: the tail of a slot whose parameter has numel % 4 != 0 (entries are never replaced)


def slot_pieces(t, p):
    """1-D tensors that, concatenated, hold ``t`` (a value of p's shape) in slot layout: no copy when ``t`` already has
    that layout (a row-padded view written by the weight-gradient batch; an aligned contiguous tensor), a cached run of
    zeros behind a value whose element count is not a multiple of 4 (so packing stays ONE torch.cat per step)."""
    n = slot_size(p)
    if _row_padded(p):
        Kp = (p.shape[1] + 3) & ~3
        full = None
        if t.is_cuda:
            from . import ops
            full = ops.row_padded_view(t)
        if full is None:
            full = t.new_zeros(p.shape[0], Kp)
            full[:, :p.shape[1]].copy_(t)
        return [full.reshape(-1)]
    flat = t.reshape(-1)
    if n == flat.numel():
        return [flat]
    key = (t.device, n - flat.numel())
    z = _ZERO_PAD.get(key)
    if z is None:
        z = _ZERO_PAD[key] = torch.zeros(n - flat.numel(), dtype=t.dtype, device=t.device)
    return [flat, z]


def register_slots(flat, params):
    """Tell ops.py which blocks of ``flat`` are row-padded weights (their pad columns are zero and nobody writes them)."""
    if not flat.is_cuda:
        return
    from . import ops
    off = 0
    for p in params:
        if _row_padded(p):
            ops.register_row_padded(flat, (flat.storage_offset() + off, p.shape[0], (p.shape[1] + 3) & ~3), K=p.shape[1])
        off += slot_size(p)


# parameters whose gradients are complete when the graph part of the backward pass ends (graph stack, fusion modules, head):
# the first part of a two-part bucket
EARLY_PREFIXES = ("graph_model.", "graph_net_", "smax_fc.", "mfn.", "gatedatt.")
# ... except the speaker / modality embeddings of MM_GCN (use_speaker / use_modal): they are added to the features BEFORE the
# adjacency is built, so their gradients come out of the adjacency builder's backward, i.e. after the point where the first
# part is packed (with them in the early part the two-part path never engaged, or packed incomplete gradients)
LATE_EXCEPTIONS = ("speaker_embeddings", "modal_embeddings", "_spk_embs")


def is_early(name):
    return name.startswith(EARLY_PREFIXES) and not any(tag in name for tag in LATE_EXCEPTIONS)


class GradientBucket:
    """Flat fp32 gradient bucket over the parameters that receive gradients.

    ``flatten()`` packs all live gradients into ONE contiguous buffer with a single multi-tensor copy
    (torch.cat -> one or two launches instead of one add/copy per parameter) and re-points every ``.grad``
    at its slice, so the all-reduce result is what the optimizer reads.  Backward passes therefore always
    run with ``.grad = None`` (autograd hands over its buffers, no accumulation kernels).

    ``parts=2`` (off by default; bench.py --two-part-bucket): the flat buffer is laid out [graph stack + head | encoders]
    and, from the second step on, reduced as TWO collectives -- the first is packed and started where the graph part of
    the backward pass ends (ops.set_graph_backward_done_hook; the weight gradients queued so far leave first) and runs
    while the encoders' backward (the GRU recurrences, ~40 % of a step) is still computing; the second follows the
    backward pass.  Element-wise sums: the reduced buffer equals the one-collective result (bit for bit at two ranks,
    tests/test_distributed_gloo.py; up to the ring's summation order beyond)."""

    def __init__(self, model, average=True, parts=1):
        self.model = model
        self.average = average
        self.parts = int(parts)
        self.flat = None
        self.params = None
        self._all_params = None
        self.split = 0                 # floats of the first part
        self._early = None             # state of the step in progress: None | "packed" (first part packed and reduced / in flight)
        self._work = None
        self._comm_stream = None

    # ---- layout
    def _layout(self, live):
        self.params = bucket_order(self.model, live)
        if self.parts == 2:
            names = {id(p): n for n, p in self.model.named_parameters()}
            early = [p for p in self.params if is_early(names[id(p)])]
            late = [p for p in self.params if not is_early(names[id(p)])]
            self.params = early + late
            self._n_early = len(early)
            self.split = sum(slot_size(p) for p in early)
        self._ids = [id(p) for p in live]

    def _check_live(self, live):
        if [id(p) for p in live] != self._ids:
            # the bucket layout is frozen at the first step (flat optimizer state and all-reduce offsets depend on
            # it); a parameter that starts / stops receiving a gradient later would silently never be reduced or
            # updated -- or crash in the pack.  The reference's Adam skips grad-less parameters step by step
            # (run_train_erc.py:512), so a changing live set needs a new bucket (and optimizer state), not a patch.
            names = {id(p): n for n, p in self.model.named_parameters()}
            now, was = set(id(p) for p in live), set(self._ids)
            raise RuntimeError("GradientBucket: the set of parameters receiving gradients changed since the first step "
                               "(new: %s; missing: %s)" % (sorted(names[i] for i in now - was),
                                                           sorted(names[i] for i in was - now)))

    def flatten(self, attach=True):
        """``attach=False``: pack only (the fused optimizer reads the flat buffer; ``.grad`` keeps pointing at the tensors
        the backward pass produced -- ~100 fewer view operations on the host per step)."""
        # (the module tree is walked once: 86 parameters() resumptions per step were 0.19 ms of the pass loop's ~0.8 ms of host
        # time per step, tools/streamed_gap.py; a module that gains parameters later needs a new bucket anyway)
        allp = self._all_params
        if allp is None:
            allp = self._all_params = list(self.model.parameters())
        live = [p for p in allp if p.requires_grad and p.grad is not None]
        if self.params is None:
            self._layout(live)
        else:
            self._check_live(live)
        if self._early == "packed":
            # the first part left during the backward pass: only the encoders' gradients are packed now
            late = self.params[self._n_early:]
            torch.cat([piece for p in late for piece in slot_pieces(p.grad, p)], out=self.flat[self.split:])
        else:
            grads = [piece for p in self.params for piece in slot_pieces(p.grad, p)]
            if self.flat is None:
                self.flat = torch.cat(grads)
                register_slots(self.flat, self.params)
            else:
                torch.cat(grads, out=self.flat)
        if attach:
            self.attach_views()
        return self.flat

    def attach_views(self):
        off = 0
        for p in self.params:
            p.grad = slot_view(self.flat, off, p)
            off += slot_size(p)

    # ---- two-part protocol
    def arm(self):
        """Call before a backward pass (train.backward does not know about buckets): registers the hook that packs and
        reduces the first part where the graph part of that backward pass ends.  A no-op for one-part buckets and
        before the layout exists (the first step)."""
        self._early, self._work = None, None
        if self.parts == 2 and self.flat is not None:
            from . import ops
            ops.set_graph_backward_done_hook(self.early_part)

    def early_part(self):
        """The graph stack's and the head's gradients are complete: pack them and start their all-reduce."""
        from . import ops
        ops.set_graph_backward_done_hook(None)
        if self.parts != 2 or self.flat is None or self._early is not None:
            return
        early = self.params[:self._n_early]
        ops.flush_queued_wgrads_now()                  # weight gradients queued so far (all of them belong to this part)
        if any(p.grad is None for p in early):
            # not complete (an unusual graph): this step reduces in one piece -- said once, a silent fallback hides that the
            # overlap never engages
            if not getattr(self, "_warned_incomplete", False):
                import warnings
                names = {id(p): n for n, p in self.model.named_parameters()}
                missing = [names.get(id(p), "?") for p in early if p.grad is None]
                warnings.warn("GradientBucket(parts=2): %d early-part gradients are not ready where the graph part of the "
                              "backward pass ends (%s ...); reducing in one piece" % (len(missing), ", ".join(missing[:3])))
                self._warned_incomplete = True
            return
        torch.cat([piece for p in early for piece in slot_pieces(p.grad, p)], out=self.flat[:self.split])
        self._early = "packed"
        part = self.flat[:self.split]
        if dist.is_initialized():
            if part.is_cuda:
                self._work = dist.all_reduce(part, async_op=True)      # RCCL's stream: overlaps the encoders' backward
            else:
                dist.all_reduce(part)

    def reduce_flat(self):
        if self._early == "packed":
            dist.all_reduce(self.flat[self.split:])
            if self._work is not None:
                self._work.wait()
            self._early, self._work = None, None
        else:
            dist.all_reduce(self.flat)
        if self.average:
            self.flat.div_(dist.get_world_size())
        return self.flat

    def all_reduce(self):
        """Eager step: pack, all-reduce, leave .grad pointing into the reduced bucket."""
        self.flatten()
        return self.reduce_flat()
