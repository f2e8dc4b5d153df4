"""Dialogue layout and the block-tile adjacency of the multimodal dialogue graph.

The reference stores the graph as one dense (M*N x M*N) fp32 matrix
(model_mm.py:125).  Its structure is block-sparse: per dialogue and modality a
dense L_i x L_i tile plus M(M-1) cross-modal diagonals.  ``BlockTileAdjacency``
keeps exactly that content (layout documented in include/mmdfn_hip.h).
"""
from collections import OrderedDict

import numpy as np
import torch


class PinnedLRU:
    """Small LRU of host-built device index tensors keyed by batch signature.

    Entries are evicted one at a time (least recently used first), never wholesale.  A hipGraph bakes the raw device
    pointers of the entries used while it was captured but holds no Python reference to them, so ``recording()``
    collects every entry handed out during a capture and the graph's owner (graphs.CapturedStep) keeps that list
    alive: an evicted entry that a captured step still replays against stays allocated."""

    def __init__(self, maxsize=64):
        self.maxsize = maxsize
        self.data = OrderedDict()
        self._recorders = []

    def get(self, key, make):
        val = self.data.get(key)
        if val is None:
            val = make()
            self.data[key] = val
            while len(self.data) > self.maxsize:
                self.data.popitem(last=False)
        else:
            self.data.move_to_end(key)
        for rec in self._recorders:
            rec.append(val)
        return val

    def clear(self):
        self.data.clear()

    def __len__(self):
        return len(self.data)


class recording:
    """``with recording(cache_a, cache_b) as used:`` -- every entry the caches hand out inside the block is appended to
    ``used`` (what a captured step must keep alive)."""

    def __init__(self, *caches):
        self.caches = caches
        self.used = []

    def __enter__(self):
        for c in self.caches:
            c._recorders.append(self.used)
        return self.used

    def __exit__(self, *exc):
        for c in self.caches:
            c._recorders.remove(self.used)
        return False


_LAYOUT_CACHE = PinnedLRU(64)


class IndexScope:
    """Owner of the host-built device index tensors of ONE captured step that is replayed for batches with different
    dialogue lengths (train.StepGraphCache, bucketed entries).  Inside ``with scope:`` the layout / index caches hand out
    PRIVATE objects (never shared through the LRUs, so re-pointing their contents cannot corrupt another user) and remember
    how each was built; ``scope.retarget(lengths)`` rewrites all of them, in place, for another batch of the same bucket
    (same number of dialogues, same total and maximum length): the hipGraph's baked pointers stay valid, only the index
    data changes.  The kernels read lengths, row offsets and tile offsets from these arrays; their grids depend on
    (B, M, N, max_len) only."""

    _active = []

    def __init__(self, slots=4):
        self.layouts = {}           # (M, device) -> DialogueLayout
        self.tensors = {}           # key -> (device tensor, maker(lengths) -> numpy array)
        self.lengths = None
        # pinned staging for retarget(): the host runs ahead of the stream, so a staging buffer is rewritten only after the
        # copies that read it have executed (a ring of ``slots`` generations, one event each)
        self._slots = [dict(event=None, bufs={}) for _ in range(max(2, int(slots)))]
        self._turn = 0

    def __enter__(self):
        IndexScope._active.append(self)
        return self

    def __exit__(self, *exc):
        IndexScope._active.pop()
        return False

    @staticmethod
    def current():
        return IndexScope._active[-1] if IndexScope._active else None

    def layout(self, lengths, M, device):
        key = (int(M), str(device))
        lay = self.layouts.get(key)
        if lay is None:
            lay = self.layouts[key] = DialogueLayout(lengths, M, device, capacity=True)
        elif lay.lengths != [int(x) for x in lengths]:
            raise RuntimeError("IndexScope: one step, two different dialogue-length lists")
        return lay

    def tensor(self, key, lengths, maker, device):
        ent = self.tensors.get(key)
        if ent is None:
            ent = self.tensors[key] = (torch.from_numpy(maker(lengths)).to(device), maker)
        return ent[0]

    def _stage(self, slot, name, arr, dev):
        host = slot["bufs"].get(name)
        if host is None or tuple(host.shape) != arr.shape or host.dtype != dev.dtype:
            # (the caching pinned allocator; `.pin_memory()` of an existing tensor registers its pages: milliseconds per call)
            host = torch.empty(arr.shape, dtype=dev.dtype, pin_memory=bool(dev.is_cuda))
            slot["bufs"][name] = host
        host.numpy()[...] = arr
        dev.copy_(host, non_blocking=True)

    def retarget(self, lengths):
        lengths = [int(x) for x in lengths]
        # validate EVERYTHING first: a batch that does not fit must leave host state and device arrays as they were
        for lay in self.layouts.values():
            lay.check_retarget(lengths)
        made = {}
        for key, (dev, maker) in self.tensors.items():
            arr = made[key] = maker(lengths)
            if arr.shape != tuple(dev.shape):
                raise RuntimeError("IndexScope.retarget: the batch does not fit this step's bucket")
        slot = self._slots[self._turn % len(self._slots)]
        self._turn += 1
        if slot["event"] is not None:
            slot["event"].synchronize()
        cuda = False
        for key, lay in self.layouts.items():
            i32, tile_base = lay.retarget_host(lengths)
            self._stage(slot, ("i32",) + key, i32, lay._i32)
            self._stage(slot, ("tb",) + key, tile_base, lay.tile_base)
            cuda = cuda or lay._i32.is_cuda
        for key, (dev, maker) in self.tensors.items():
            self._stage(slot, ("t",) + tuple(key), made[key], dev)
            cuda = cuda or dev.is_cuda
        if cuda:
            slot["event"] = torch.cuda.Event()
            slot["event"].record()
        self.lengths = lengths


def pair_list(M):
    return [(m, n) for m in range(M) for n in range(m + 1, M)]


class DialogueLayout:
    """Index arrays describing a batch of B dialogues x M modalities."""

    def __init__(self, lengths, M, device, capacity=False):
        """``capacity``: a layout owned by an IndexScope -- ``tile_elems`` (what callers allocate tile arrays with) is the
        upper bound M N round4(max_len) over every batch of the same (B, N, max_len), so retarget() never outgrows a buffer
        a captured step has baked in."""
        self.M = int(M)
        self.device = torch.device(device)
        self._capacity = bool(capacity)
        i32, tile_base = self._host_arrays(lengths)
        self.tile_elems = int(tile_base[-1])
        if capacity:
            self.tile_elems = self.M * self.N * ((self.max_len + 3) & ~3)
        self._i32 = torch.from_numpy(i32).to(self.device)
        self.dia_len = self._i32[:self.B]
        self.row_start = self._i32[self.B:]
        self.tile_base = torch.from_numpy(tile_base).to(self.device)

    def _host_arrays(self, lengths):
        lens = np.asarray([int(x) for x in lengths], dtype=np.int64)
        if lens.ndim != 1 or lens.size == 0 or (lens <= 0).any():
            raise ValueError("dialogue lengths must be a non-empty list of positive ints")
        self.lengths = [int(x) for x in lens]
        self.B = int(lens.size)
        row_start = np.zeros(self.B + 1, dtype=np.int64)
        row_start[1:] = np.cumsum(lens)
        ld = (lens + 3) & ~3
        tile_base = np.zeros(self.B + 1, dtype=np.int64)
        tile_base[1:] = np.cumsum(self.M * lens * ld)
        self.N = int(row_start[-1])
        self.max_len = int(lens.max())
        self.npairs = self.M * (self.M - 1) // 2
        self.nnz = int((self.M * lens * lens + self.M * (self.M - 1) * lens).sum())
        self.row_start_host = row_start
        self.tile_base_host = tile_base
        self.ld_host = ld
        return np.concatenate([lens.astype(np.int32), row_start.astype(np.int32)]), tile_base

    def retarget_host(self, lengths):
        """Re-point this (scope-owned) layout's HOST side at another batch with the same number of dialogues, the same total
        and the same maximum length; returns the new (dia_len | row_start, tile_base) arrays -- IndexScope.retarget writes them
        over the device arrays in place (a captured step keeps reading those)."""
        if not self._capacity:
            raise RuntimeError("only layouts owned by an IndexScope can be retargeted (shared ones live in an LRU)")
        self.check_retarget(lengths)
        return self._host_arrays(lengths)

    def check_retarget(self, lengths):
        """Raise unless ``lengths`` fits this layout's (B, N, max_len) and tile capacity -- WITHOUT touching the host state, so a
        rejected batch leaves the layout describing the batch its device arrays still hold."""
        lens = np.asarray([int(x) for x in lengths], dtype=np.int64)
        if lens.ndim != 1 or lens.size == 0 or (lens <= 0).any():
            raise ValueError("dialogue lengths must be a non-empty list of positive ints")
        new = (int(lens.size), int(lens.sum()), int(lens.max()))
        tiles = int((self.M * lens * ((lens + 3) & ~3)).sum())
        if new != (self.B, self.N, self.max_len) or tiles > self.tile_elems:
            raise RuntimeError("DialogueLayout.retarget: (B, N, max_len) = %s does not match this layout's %s"
                               % (new, (self.B, self.N, self.max_len)))

    @staticmethod
    def get(lengths, M, device):
        scope = IndexScope.current()
        if scope is not None:
            return scope.layout(lengths, M, device)
        key = (tuple(int(x) for x in lengths), int(M), str(device))
        return _LAYOUT_CACHE.get(key, lambda: DialogueLayout(lengths, M, device))

    # algorithmic bytes / flops of one propagate call
    def propagate_bytes(self, d):
        return 4 * self.nnz + 8 * self.M * self.N * d

    def propagate_flops(self, d):
        return 2 * d * self.nnz


class BlockTileAdjacency:
    """Normalised adjacency in block-tile storage (never dense on the hot path)."""

    def __init__(self, layout, tiles, cross, symmetric=False, stacked_feats=None):
        self.layout = layout
        self.tiles = tiles
        self.cross = cross
        self.symmetric = bool(symmetric)
        self.stacked_feats = stacked_feats  # (M, N, D) view of cat([a, v, l], 0), if available

    @property
    def requires_grad(self):
        return self.tiles.requires_grad or self.cross.requires_grad

    @property
    def shape(self):
        n = self.layout.M * self.layout.N
        return (n, n)

    def to_dense(self):
        """Dense (MN x MN) matrix (tests / interoperability only)."""
        lay = self.layout
        M, N = lay.M, lay.N
        dense = self.tiles.new_zeros(M * N, M * N)
        for i, L in enumerate(lay.lengths):
            ld = int(lay.ld_host[i])
            rs = int(lay.row_start_host[i])
            base = int(lay.tile_base_host[i])
            for m in range(M):
                t = self.tiles[base + m * L * ld: base + (m + 1) * L * ld].view(L, ld)[:, :L]
                dense[m * N + rs:m * N + rs + L, m * N + rs:m * N + rs + L] = t
        ar = torch.arange(N, device=self.tiles.device)
        for k, (m, n) in enumerate(pair_list(M)):
            dense[m * N + ar, n * N + ar] = self.cross[k]
            dense[n * N + ar, m * N + ar] = self.cross[k]
        return dense

    @staticmethod
    def from_parts(layout, tile_list, cross, device=None, symmetric=False):
        """Pack per-(dialogue, modality) L x L tensors (dialogue-major, then modality) + cross (npairs, N)."""
        device = device or layout.device
        flat = torch.zeros(layout.tile_elems, dtype=torch.float32, device=device)
        it = iter(tile_list)
        for i, L in enumerate(layout.lengths):
            ld = int(layout.ld_host[i])
            base = int(layout.tile_base_host[i])
            for m in range(layout.M):
                t = next(it).to(device=device, dtype=torch.float32)
                flat[base + m * L * ld: base + (m + 1) * L * ld].view(L, ld)[:, :L] = t
        cross = cross.to(device=device, dtype=torch.float32).contiguous()
        return BlockTileAdjacency(layout, flat, cross, symmetric)
