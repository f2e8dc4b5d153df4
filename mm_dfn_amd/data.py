"""Data ingest for the hot path: the reference's pickle -> Dataset -> collate contract (dataloader.py), its
loader factories (run_train_erc.py:29-88), and what the device adds: length-bucketed batches (less padding in
the (L, B, D) tensors the encoders walk) and a pinned, double-buffered host->HBM prefetcher on a side stream.

Pickle layout (dataloader.py:12-14, 40-42): a 9-tuple (IEMOCAP) or 10-tuple (MELD) of dicts keyed by dialogue id
  videoIDs, videoSpeakers, videoLabels, videoText, videoAudio, videoVisual, videoSentence, trainVid, testVid[, _]
IEMOCAP speakers are 'M' / 'F' strings -> one-hot [1,0] / [0,1]; MELD speakers are already one-hot rows.
A dataset item is (text, visual, audio, qmask, umask, label, vid) (dataloader.py:18-27); a batch is
[textf (L,B,Dt), visuf (L,B,Dv), acouf (L,B,Da), qmask (L,B,P), umask (B,L), label (B,L), vids] (:34).
The .pkl feature files themselves are not distributed with the reference (.MISSING_LARGE_BLOBS);
``write_synthetic_pickle`` produces files of the same layout for tests and smoke runs.
"""
import pickle

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import DataLoader, Dataset, Sampler
from torch.utils.data.sampler import SubsetRandomSampler


def _odd_feature(t, slot):
    """``slot``: the tensor's position in the reference's batch tuple (textf, visuf, acouf, qmask, umask, label)."""
    from . import ops
    return ops.is_odd_feature_tensor(t, slot in ops.FEATURE_SLOTS)


def _collate(data):
    """dataloader.py:31-34 without the pandas round trip: columns 0-3 time-major, 4-5 batch-major, 6 a list."""
    cols = list(zip(*data))
    out = []
    for i, col in enumerate(cols):
        if i < 4:
            out.append(pad_sequence(list(col)))
        elif i < 6:
            out.append(pad_sequence(list(col), True))
        else:
            out.append(list(col))
    return out


class _DialogueDataset(Dataset):
    def __len__(self):
        return self.len

    def collate_fn(self, data):
        return _collate(data)

    def lengths(self):
        """Utterances per dialogue, in dataset order (what the bucketing sampler sorts by)."""
        return [len(self.videoLabels[k]) for k in self.keys]

    def return_labels(self):
        """dataloader.py:61-65."""
        out = []
        for key in self.keys:
            out += list(self.videoLabels[key])
        return out


class IEMOCAPDataset(_DialogueDataset):
    """dataloader.py:9-34."""

    def __init__(self, path=None, train=True):
        with open(path, 'rb') as f:
            (self.videoIDs, self.videoSpeakers, self.videoLabels, self.videoText, self.videoAudio, self.videoVisual,
             self.videoSentence, self.trainVid, self.testVid) = pickle.load(f, encoding='latin1')
        self.keys = [x for x in (self.trainVid if train else self.testVid)]
        self.len = len(self.keys)

    def __getitem__(self, index):
        vid = self.keys[index]
        return (torch.FloatTensor(np.asarray(self.videoText[vid], dtype=np.float32)),
                torch.FloatTensor(np.asarray(self.videoVisual[vid], dtype=np.float32)),
                torch.FloatTensor(np.asarray(self.videoAudio[vid], dtype=np.float32)),
                torch.FloatTensor([[1, 0] if x == 'M' else [0, 1] for x in self.videoSpeakers[vid]]),
                torch.FloatTensor([1] * len(self.videoLabels[vid])),
                torch.LongTensor(self.videoLabels[vid]),
                vid)


class MELDDataset(_DialogueDataset):
    """dataloader.py:37-69."""

    def __init__(self, path=None, train=True):
        with open(path, 'rb') as f:
            (self.videoIDs, self.videoSpeakers, self.videoLabels, self.videoText, self.videoAudio, self.videoVisual,
             self.videoSentence, self.trainVid, self.testVid, self.aaa) = pickle.load(f, encoding='latin1')
        self.keys = [x for x in (self.trainVid if train else self.testVid)]
        self.len = len(self.keys)

    def __getitem__(self, index):
        vid = self.keys[index]
        return (torch.FloatTensor(np.asarray(self.videoText[vid], dtype=np.float32)),
                torch.FloatTensor(np.asarray(self.videoVisual[vid], dtype=np.float32)),
                torch.FloatTensor(np.asarray(self.videoAudio[vid], dtype=np.float32)),
                torch.FloatTensor(np.asarray(self.videoSpeakers[vid], dtype=np.float32)),
                torch.FloatTensor([1] * len(self.videoLabels[vid])),
                torch.LongTensor(self.videoLabels[vid]),
                vid)


def get_train_valid_sampler(trainset, valid=0.1):
    """run_train_erc.py:29-33: the first ``valid`` fraction of the train split is the validation set."""
    size = len(trainset)
    idx = list(range(size))
    split = int(valid * size)
    return SubsetRandomSampler(idx[split:]), SubsetRandomSampler(idx[:split])


class LengthBucketedBatchSampler(Sampler):
    """Batches of dialogues of similar length: the index set is shuffled, cut into windows of ``bucket`` batches,
    each window sorted by dialogue length and sliced into batches, and the batch order shuffled again.  Padding
    (rows the projections / GRUs walk for nothing, reference model.py:1065-1154 runs them at full padded length)
    drops from ~70 % extra rows to ~25 % for lengths spread uniformly over 5..110 at batch 16 (tests/test_data_and_fit.py);
    every dialogue is still visited once per epoch."""

    def __init__(self, indices, lengths, batch_size, bucket=8, shuffle=True, seed=0):
        self.indices = list(indices)
        self.lengths = lengths
        self.batch_size = batch_size
        self.bucket = max(1, bucket)
        self.shuffle = shuffle
        self.seed = seed
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        rs = np.random.RandomState(self.seed + self.epoch)
        idx = list(self.indices)
        if self.shuffle:
            rs.shuffle(idx)
        win = self.batch_size * self.bucket
        batches = []
        for s in range(0, len(idx), win):
            chunk = sorted(idx[s:s + win], key=lambda i: self.lengths[i])
            batches += [chunk[b:b + self.batch_size] for b in range(0, len(chunk), self.batch_size)]
        if self.shuffle:
            rs.shuffle(batches)
        return iter(batches)


def _loaders(trainset, testset, batch_size, valid_rate, num_workers, pin_memory, bucketed):
    if bucketed:
        size = len(trainset)
        split = int(valid_rate * size)
        lens = trainset.lengths()
        mk = lambda ds, ids, shuffle: DataLoader(
            ds, batch_sampler=LengthBucketedBatchSampler(ids, ds.lengths(), batch_size, shuffle=shuffle),
            collate_fn=ds.collate_fn, num_workers=num_workers, pin_memory=pin_memory)
        del lens
        return (mk(trainset, range(split, size), True), mk(trainset, range(split), False),
                mk(testset, range(len(testset)), False))
    train_sampler, valid_sampler = get_train_valid_sampler(trainset, valid_rate)
    mk = lambda ds, sampler: DataLoader(ds, batch_size=batch_size, sampler=sampler, collate_fn=ds.collate_fn,
                                        num_workers=num_workers, pin_memory=pin_memory)
    return mk(trainset, train_sampler), mk(trainset, valid_sampler), mk(testset, None)


def get_IEMOCAP_loaders(data_path=None, batch_size=32, valid_rate=0.1, num_workers=0, pin_memory=False, bucketed=False):
    """run_train_erc.py:63-88 (``bucketed`` is the device-side extension, off = the reference's sampling)."""
    return _loaders(IEMOCAPDataset(path=data_path), IEMOCAPDataset(path=data_path, train=False), batch_size, valid_rate,
                    num_workers, pin_memory, bucketed)


def get_MELD_loaders(data_path=None, batch_size=32, valid_rate=0.1, num_workers=0, pin_memory=False, bucketed=False):
    """run_train_erc.py:36-60."""
    return _loaders(MELDDataset(data_path), MELDDataset(data_path, train=False), batch_size, valid_rate, num_workers,
                    pin_memory, bucketed)


class DeviceBatch(list):
    """The reference's batch list with the host-side dialogue lengths attached (``.lengths``)."""


class DevicePrefetcher:
    """Wraps a loader: batch k+1 is staged in pinned host memory and copied to HBM on a side stream while batch k
    computes; the consumer stream waits on the copy's event only.  Yields the reference's batch list with the six
    tensors resident on ``device`` (the trainer is then called with cuda_flag=False: nothing left to move)."""

    def __init__(self, loader, device="cuda", depth=2, recycle=True, graph_cache=None, train_flag=False):
        """``graph_cache`` (a train.StepGraphCache) + ``train_flag``: a batch whose signature has a captured step is copied
        from pinned host memory STRAIGHT INTO that step's static input buffers (no device-side staging copy; the pass loop
        receives those very tensors and copies nothing)."""
        self.loader = loader
        self.recycle = recycle
        self.graph_cache = graph_cache
        self.train_flag = bool(train_flag)
        self.device = torch.device(device)
        self.depth = max(1, depth)
        # device staging buffers are recycled (depth + 2 sets, each tensor slot grown to the largest batch seen): ragged
        # batches otherwise ask the caching allocator for a new size every step and every other step ends in a hipMalloc
        self._ring = [dict() for _ in range(self.depth + 2)]
        self._ring_free = [None] * (self.depth + 2)      # event on the consumer's stream: the set's last batch has been used
        self._turn = 0

    def _device_buffer(self, slot, shape, dtype):
        """A (shape) view of this turn's staging buffer for tensor ``slot`` (grown when a larger batch arrives)."""
        numel = 1
        for d in shape:
            numel *= int(d)
        ring = self._ring[self._turn % len(self._ring)]
        buf = ring.get((slot, dtype))
        if buf is None or buf.numel() < numel:
            buf = ring[(slot, dtype)] = torch.empty(max(numel, 1), dtype=dtype, device=self.device)
        return buf[:numel].view(*shape)

    def __len__(self):
        return len(self.loader)

    def bind_graph_cache(self, graph_cache, train_flag):
        """Called by train.train_or_eval_graph_model at the start of a pass that replays captured steps."""
        self.graph_cache, self.train_flag = graph_cache, bool(train_flag)

    def _stage(self, batch, stream):
        tensors, rest = batch[:6], batch[6:]
        # dialogue lengths from the HOST copy of umask (run_train_erc.py:194 reads them back from the device with B
        # syncs): they key the trainer's captured-step cache without a device round trip
        # (numpy, not torch: a torch CPU reduction forks the whole intra-op thread pool -- 6-27 ms per batch measured on
        # the 256-thread host of the GPU box, tools/prof_stream.py)
        um = tensors[4].numpy()
        lengths = [int(v) for v in ((um == 1) * np.arange(1, um.shape[1] + 1)[None, :]).max(1)]
        ent = None
        if self.graph_cache is not None:
            ent = self.graph_cache.claim_static([t.shape for t in tensors], lengths, self.train_flag)
        if ent is not None:
            if ent.get("done") is not None:
                stream.wait_event(ent["done"])       # the entry's last replay has read its inputs
            with torch.cuda.stream(stream):
                dev = []
                for t, dst in zip(tensors, ent["static"]):
                    h = t if t.is_pinned() else t.pin_memory()
                    base = getattr(dst, "_mmdfn_padbase", None)
                    if base is not None:             # row-padded static buffer: pad on the host, one contiguous copy
                        K = t.shape[-1]
                        hp = torch.empty(tuple(base.shape), dtype=torch.float32, pin_memory=True)
                        hn = hp.numpy()
                        hn[..., :K] = t.numpy()
                        hn[..., K:] = 0.0
                        base.copy_(hp, non_blocking=True)
                    else:
                        dst.copy_(h, non_blocking=True)
                    dev.append(dst)
                ev = torch.cuda.Event()
                ev.record(stream)
            return dev, list(rest), ev, lengths, None
        self._turn += 1
        free = self._ring_free[self._turn % len(self._ring)]
        if free is not None:
            stream.wait_event(free)          # the copy engine must not overwrite a set the consumer's stream still reads
        with torch.cuda.stream(stream):
            dev = []
            for slot, t in enumerate(tensors):
                if _odd_feature(t, slot):
                    # feature width not a multiple of 4 (1582-d audio, 342-d visual features): the pinned staging buffer
                    # is row-padded to the next multiple of 4 (pad columns zero), so the device copy is the operand the
                    # MFMA kernels fetch in 16-byte units (ops.py "row padding") -- no pad launch on the device
                    from . import ops
                    K, Kp = t.shape[-1], ops.pad4(t.shape[-1])
                    h = torch.empty(t.shape[0], t.shape[1], Kp, dtype=torch.float32, pin_memory=True)
                    hn = h.numpy()
                    hn[..., :K] = t.numpy()
                    hn[..., K:] = 0.0
                    base = ops.register_row_padded(h.to(self.device, non_blocking=True), K=K)
                    view = base[..., :K]
                    view._mmdfn_padbase = base
                    dev.append(view)
                    continue
                h = t if t.is_pinned() else t.pin_memory()
                if not self.recycle:
                    dev.append(h.to(self.device, non_blocking=True))
                    continue
                d = self._device_buffer(slot, tuple(h.shape), h.dtype)
                d.copy_(h, non_blocking=True)
                dev.append(d)
            ev = torch.cuda.Event()
            ev.record(stream)
        return dev, list(rest), ev, lengths, self._turn % len(self._ring)

    def __iter__(self):
        stream = torch.cuda.Stream(device=self.device)
        queue = []
        it = iter(self.loader)
        if self.graph_cache is not None:
            self.graph_cache.forget_queued()       # (announcements of a pass that was abandoned mid-way)
        try:
            while len(queue) < self.depth:
                queue.append(self._stage(next(it), stream))
        except StopIteration:
            pass
        while queue:
            dev, rest, ev, lengths, ring_slot = queue.pop(0)
            torch.cuda.current_stream(self.device).wait_event(ev)
            cur = torch.cuda.current_stream(self.device)
            for t in dev:
                base = getattr(t, "_mmdfn_padbase", None)
                if ring_slot is not None and base is not None:
                    # row-padded features are fresh allocations made on the COPY stream (not ring buffers): the consumer's
                    # stream must be recorded, or the allocator may hand the block to a later copy while kernels queued on
                    # the consumer's stream still read it
                    base.record_stream(cur)
                elif not self.recycle:
                    t.record_stream(cur)
            try:
                queue.append(self._stage(next(it), stream))
            except StopIteration:
                pass
            out = DeviceBatch(dev + rest)
            out.lengths = lengths
            yield out
            # the consumer has enqueued everything that reads this batch: its staging set may be refilled after that
            if ring_slot is not None:
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))
                self._ring_free[ring_slot] = done


def write_synthetic_pickle(path, dataset="IEMOCAP", n_train=24, n_test=8, max_len=40, min_len=3, D_t=100, D_a=100,
                           D_v=512, n_classes=6, n_speakers=2, seed=0):
    """A feature file with the reference's tuple layout (see module docstring), RandomState-seeded."""
    rs = np.random.RandomState(seed)
    ids = ["dlg%03d" % i for i in range(n_train + n_test)]
    vIDs, vSpk, vLab, vText, vAud, vVis, vSent = {}, {}, {}, {}, {}, {}, {}
    for vid in ids:
        n = int(rs.randint(min_len, max_len + 1))
        vIDs[vid] = list(range(n))
        spk = rs.randint(0, n_speakers, size=n)
        if dataset == "IEMOCAP":
            vSpk[vid] = ['M' if s == 0 else 'F' for s in spk]
        else:
            vSpk[vid] = np.eye(n_speakers, dtype=np.float32)[spk].tolist()
        vLab[vid] = [int(x) for x in rs.randint(0, n_classes, size=n)]
        vText[vid] = rs.randn(n, D_t).astype(np.float32)
        vAud[vid] = rs.randn(n, D_a).astype(np.float32)
        vVis[vid] = rs.randn(n, D_v).astype(np.float32)
        vSent[vid] = ["utt %d" % j for j in range(n)]
    train, test = ids[:n_train], ids[n_train:]
    blob = (vIDs, vSpk, vLab, vText, vAud, vVis, vSent, train, test)
    if dataset != "IEMOCAP":
        blob = blob + (None,)
    with open(path, "wb") as f:
        pickle.dump(blob, f)
    return path
