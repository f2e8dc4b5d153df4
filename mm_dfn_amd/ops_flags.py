"""Dropout keep flags (one generator launch per step) and the mask-scale node.

Part of the operator layer over the C-ABI kernels (libmmdfn_hip.so); `mm_dfn_amd.ops` re-exports every name.
Every function launches hand-written gfx950 kernels on the current HIP stream; there is no CPU / eager fallback.
"""
import torch

from . import _hip


# ---- dropout keep flags: one generator launch per step -----------------------------------------------------------------
# Every dropout site of the fused path (GRU inter-layer dropout, the GCN stack, the head) consumes 0 / 1 keep flags that
# its kernel scales by 1/(1-p).  Inside a ``flag_pool()`` scope (the models open one per forward) the sites share one
# buffer per (device, p) drawn by ONE bernoulli_ launch: the first request of a step draws as many flags as the previous
# step with the same scope key used, later requests take slices (a request that does not fit draws its own buffer).
# A fresh tensor per draw: slices saved for backward are never overwritten.  Outside a scope every request draws its own.
_FLAG_SCOPE = None
_FLAG_HINT = {}


class flag_pool:
    def __init__(self, key=None):
        self.key = key
        self.bufs = {}       # (device, p) -> [buffer, offset, used]

    def __enter__(self):
        global _FLAG_SCOPE
        self.outer = _FLAG_SCOPE
        if self.outer is None:
            _FLAG_SCOPE = self
        return self

    def __exit__(self, *exc):
        global _FLAG_SCOPE
        if self.outer is None:
            _FLAG_SCOPE = None
            for k, (_, _, used) in self.bufs.items():
                if len(_FLAG_HINT) > 256:
                    _FLAG_HINT.clear()
                _FLAG_HINT[(self.key,) + k] = used
        return False


def keep_scale(p):
    """The factor dropout multiplies the kept elements by, 1 / (1 - p); p = 1 drops everything (all flags are 0), and the
    factor is 0 rather than inf so that 0 * inf never appears."""
    return 0.0 if p >= 1.0 else 1.0 / (1.0 - p)


_FLAG_STATE = {}       # device index -> [device state (seed, offset, workgroup counter), host mirror (seed, offset)]
_FLAG_CONSUMED = {}    # device index -> Philox counters consumed so far (eager and captured launches alike; host-side tally)


def flags_consumed(idx):
    return _FLAG_CONSUMED.get(idx, 0)


def flag_state_snapshot(idx):
    """(device state clone, host mirror) of the keep-flag generator, or None before the first draw on the device."""
    ent = _FLAG_STATE.get(idx)
    return None if ent is None else (ent[0].clone(), ent[1])


def flag_state_restore(idx, snap):
    """Put the keep-flag generator back where ``flag_state_snapshot`` found it (graphs.CapturedStep: building a captured
    step consumes no random numbers).  With no earlier state the device generator is re-seeded from torch's CUDA generator."""
    ent = _FLAG_STATE.get(idx)
    if ent is None:
        return
    if snap is None:
        ent[1] = None
        flag_state_sync(idx)
    else:
        ent[0].copy_(snap[0])
        ent[1] = snap[1]


def flag_state_sync(idx):
    """Re-seed the device generator from torch's CUDA generator if the two disagree (torch.manual_seed, a restored RNG
    state, torch's own random ops since the last draw).  Eager draws do this themselves; a captured step calls it before a
    replay, whose launches read the device state as it is."""
    ent = _FLAG_STATE.get(idx)
    if ent is None:
        return
    gen = torch.cuda.default_generators[idx]
    now = (int(gen.initial_seed()), int(gen.get_offset()))
    if ent[1] != now:
        seed = now[0] - (1 << 64) if now[0] >= (1 << 63) else now[0]
        ent[0].copy_(torch.tensor([seed, now[1], 0, 0], dtype=torch.int64), non_blocking=False)
        ent[1] = now


def flags_advance_host(idx, counters):
    """After a replay that consumed ``counters`` Philox counters on the device: move torch's generator (and the host mirror)
    by the same amount, so that the next eager draw continues the stream instead of re-seeding it backwards."""
    ent = _FLAG_STATE.get(idx)
    if ent is None or counters <= 0:
        return
    gen = torch.cuda.default_generators[idx]
    gen.set_offset(int(gen.get_offset()) + int(counters))
    ent[1] = (int(gen.initial_seed()), int(gen.get_offset()))


# The step's draw as a RIDER of the first GRU layer's forward recurrence launch (MMDFN_FLAG_RIDER=0 turns it off): the first
# consumer of a step's flags is the dropout behind that layer, and the recurrence leaves CUs idle (cfg2: 96 of 256 for 69 us), so the
# generator launch (7.7 us at cfg2, a link of the step's dependent chain) runs as extra workgroups of the recurrence launch
# instead (csrc/gru.hip gru_seq_fwd_io_flags_kernel, include/mmdfn_hip.h mmdfn_keep_flags_stage).
FLAG_RIDER = __import__("os").environ.get("MMDFN_FLAG_RIDER", "1") == "1"
_FLAG_STAGED = [False]


def stage_flag_draw(p, device, rows):
    """In front of the plain forward launch of the first GRU layer (groups of ``rows`` sequences): if the step's flag pool has no
    buffer for (device, p) yet and the previous step with the same scope key says how many flags the step uses, draw them NOW
    as riders of that launch.  ``finish_flag_draw()`` must follow the launch."""
    scope = _FLAG_SCOPE
    if not FLAG_RIDER or scope is None or _FLAG_STAGED[0]:
        return
    device = torch.device(device)
    k = (device, float(p))
    if k in scope.bufs:
        return
    want = _FLAG_HINT.get((scope.key,) + k, 0)
    if want <= 0 or not _hip.lib().mmdfn_gru_seq_fwd_takes_flags(len(rows), _hip.int_array(rows)):
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if _FLAG_STATE.get(idx) is None:
        return                        # (the first draw on a device sets the generator state up: the ordinary way)
    scope.bufs[k] = [draw_flags(want, p, device, stage=True), 0, 0]
    _FLAG_STAGED[0] = True


def finish_flag_draw():
    """Behind that launch: a staged draw the launch did not take (another kernel form) is launched now."""
    if _FLAG_STAGED[0]:
        _FLAG_STAGED[0] = False
        _hip.check(_hip.lib().mmdfn_keep_flags_flush(_hip.stream()), "mmdfn_keep_flags_flush")


def draw_flags(n, p, device, stage=False):
    """n (a multiple of 4) fresh fp32 keep flags from the package's Philox kernel (csrc/encoder_glue.hip).  The generator state
    lives on the device and every launch advances it, so replays of a captured graph draw new flags.  In eager mode the state
    follows torch's CUDA generator: it is re-seeded from (initial_seed, offset) whenever those differ from what this function
    left behind (torch.manual_seed, a restored RNG state, other random ops in between), and the generator's offset is advanced
    by the counters consumed -- `torch.manual_seed(s)` reproduces a run exactly as it does for torch's own dropout."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ent = _FLAG_STATE.get(idx)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None:
        if capturing:
            raise RuntimeError("the first dropout draw on a device cannot happen inside a stream capture (run one eager step first)")
        ent = _FLAG_STATE[idx] = [torch.zeros(4, dtype=torch.int64, device=device), None]
    n8 = ((n + 7) // 8 + 63) // 64 * 64    # Philox counters one launch consumes (8 flags each, whole waves)
    if not capturing:
        gen = torch.cuda.default_generators[idx]
        now = (int(gen.initial_seed()), int(gen.get_offset()))
        if ent[1] != now:
            seed = now[0] - (1 << 64) if now[0] >= (1 << 63) else now[0]
            ent[0].copy_(torch.tensor([seed, now[1], 0, 0], dtype=torch.int64), non_blocking=False)
        # leave torch's generator behind the counters this launch consumes (its offset moves in multiples of 4)
        gen.set_offset(now[1] + 4 * ((n8 + 3) // 4))
        ent[1] = (now[0], int(gen.get_offset()))
    _FLAG_CONSUMED[idx] = _FLAG_CONSUMED.get(idx, 0) + 4 * ((n8 + 3) // 4)
    out = torch.empty(n, dtype=torch.float32, device=device)
    fn = _hip.lib().mmdfn_keep_flags_stage if stage else _hip.lib().mmdfn_keep_flags
    _hip.check(fn(_hip.ptr(out), n, float(1.0 - p), _hip.ptr(ent[0]), _hip.stream()), "mmdfn_keep_flags")
    return out


def keep_flags(n, p, device):
    """n fp32 keep flags (1 with probability 1 - p), 16-byte aligned."""
    n = int(n)
    scope = _FLAG_SCOPE
    if scope is None:
        return draw_flags((n + 3) & ~3, p, device)[:n]
    k = (device, float(p))
    ent = scope.bufs.get(k)
    if ent is None:
        ent = scope.bufs[k] = [None, 0, 0]
    n4 = (n + 3) & ~3
    if ent[0] is None or ent[1] + n4 > ent[0].numel():
        want = max(n4, _FLAG_HINT.get((scope.key,) + k, 0) - ent[2])
        ent[0] = draw_flags(want, p, device)
        ent[1] = 0
    out = ent[0][ent[1]:ent[1] + n]
    ent[1] += n4
    ent[2] += n4
    return out


class _MaskScale(torch.autograd.Function):
    """outs[g] = xs[g] * masks[g] * scale for up to 4 tensors in ONE launch (csrc/encoder_glue.hip); the backward pass is
    the same launch on the incoming gradients.  masks: flat 0 / 1 keep flags (ops.keep_flags)."""

    @staticmethod
    def forward(ctx, scale, masks, *xs):
        _hip.require_cuda(*xs)
        _hip.require_f32(*xs, *masks)
        xs = [x.contiguous() for x in xs]
        outs = [torch.empty_like(x) for x in xs]
        ctx.masks, ctx.scale = list(masks), float(scale)
        _MaskScale._launch(xs, ctx.masks, outs, ctx.scale)
        return tuple(outs)

    @staticmethod
    def _launch(xs, masks, outs, scale):
        for x, m in zip(xs, masks):
            if m.numel() != x.numel() or x.numel() % 4 or x.data_ptr() % 16 or m.data_ptr() % 16:
                raise _hip.HipLibraryError("mask_scale: flags must match the tensor (multiple of 4 elements, 16-byte aligned)")
        rc = _hip.lib().mmdfn_mask_scale(len(xs), _hip.ptr_array(xs), _hip.ptr_array(masks), _hip.ptr_array(outs),
                                         _hip.long_array([x.numel() for x in xs]), scale, _hip.stream())
        _hip.check(rc, "mmdfn_mask_scale")

    @staticmethod
    def backward(ctx, *douts):
        live = [i for i, d in enumerate(douts) if d is not None]
        grads = [None] * len(douts)
        if live:
            ds = [douts[i].contiguous() for i in live]
            outs = [torch.empty_like(d) for d in ds]
            _MaskScale._launch(ds, [ctx.masks[i] for i in live], outs, ctx.scale)
            for i, o in zip(live, outs):
                grads[i] = o
        return (None, None) + tuple(grads)


def mask_scale(xs, masks, scale):
    """Dropout as a multiply by precomputed keep flags for a list of (<= 4) tensors, one launch each way."""
    if len(xs) > 4:
        return tuple(o for k in range(0, len(xs), 4) for o in mask_scale(xs[k:k + 4], masks[k:k + 4], scale))
    return _MaskScale.apply(scale, list(masks), *xs)
