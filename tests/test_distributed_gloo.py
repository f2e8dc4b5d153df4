"""CPU, world_size 2 over gloo: the data-parallel path (flat gradient bucket + global-N loss scaling)
reproduces the single-process gradient of one big batch.  The model here is a small torch stand-in
(the HIP modules need a GPU); what is under test is mm_dfn_amd.distributed itself."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from mm_dfn_amd import distributed
from mm_dfn_amd.loss import FocalLoss


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.dead = torch.nn.Linear(3, 3)     # never reached: must stay out of the bucket
        self.b = torch.nn.Linear(16, 6)

    def forward(self, x):
        return torch.log_softmax(self.b(torch.relu(self.a(x))), 1)


def make_data():
    rs = np.random.RandomState(0)
    lengths = [9, 4, 7, 2, 5, 11]
    xs = [torch.from_numpy(rs.randn(n, 8).astype(np.float32)) for n in lengths]
    ys = [torch.from_numpy(rs.randint(0, 6, size=n)) for n in lengths]
    return lengths, xs, ys


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    distributed.init(backend="gloo")
    torch.manual_seed(0)
    model = Tiny()
    lengths, xs, ys = make_data()
    mine = distributed.shard_dialogues(lengths, world, rank)
    x = torch.cat([xs[i] for i in mine])
    y = torch.cat([ys[i] for i in mine])
    n_local = x.shape[0]
    n_global = distributed.all_reduce_scalar(n_local, device="cpu")
    bucket = distributed.GradientBucket(model)
    loss_f = FocalLoss(gamma=0.5)
    for _ in range(2):   # second step exercises re-packing into the existing bucket
        model.zero_grad(set_to_none=True)
        loss = loss_f(model(x), y) * (n_local * world / n_global)
        loss.backward()
        flat = bucket.all_reduce()
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    if rank == 0:
        torch.save({"grads": grads, "n_global": n_global, "bucket": flat.numel(), "shard": mine}, out)
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_equals_single_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0.pt")
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = Tiny()
    lengths, xs, ys = make_data()
    FocalLoss(gamma=0.5)(model(torch.cat(xs)), torch.cat(ys)).backward()
    assert got["n_global"] == sum(lengths)
    assert got["bucket"] == sum(distributed.slot_size(p) for n, p in model.named_parameters() if not n.startswith("dead"))
    assert "dead.weight" not in got["grads"]
    for k, p in model.named_parameters():
        if k.startswith("dead"):
            continue
        assert (got["grads"][k] - p.grad).abs().max() < 1e-6, k


def test_shard_dialogues_balanced_and_complete():
    lengths = [110, 27, 64, 90, 33, 110, 45, 71]
    parts = [distributed.shard_dialogues(lengths, 4, r) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(8))
    loads = [sum(lengths[i] ** 2 for i in p) for p in parts]
    assert max(loads) / min(loads) < 1.6


def test_bucket_rejects_a_changing_live_parameter_set():
    """A parameter that starts receiving a gradient after the bucket layout was frozen must not be silently skipped."""
    torch.manual_seed(0)
    model = Tiny()
    x = torch.randn(5, 8)
    bucket = distributed.GradientBucket(model)
    model(x).sum().backward()
    bucket.flatten()
    n0 = bucket.flat.numel()
    model.zero_grad(set_to_none=True)
    (model(x).sum() + model.dead(torch.randn(2, 3)).sum()).backward()     # 'dead' comes alive
    with pytest.raises(RuntimeError, match="dead"):
        bucket.flatten()
    assert bucket.flat.numel() == n0
    model.zero_grad(set_to_none=True)
    model.a(x).sum().backward()                                            # 'b' gets no gradient this time
    with pytest.raises(RuntimeError, match="missing"):
        bucket.flatten()


class TinySeq(torch.nn.Module):
    """GRU-bearing stand-in with the parameter names of the real encoders (nn.GRU keys weight_ih_l*[ _reverse ]), so
    that distributed.bucket_order's twin-adjacency rule and the live-set check run on the real ordering code."""

    def __init__(self):
        super().__init__()
        self.proj = torch.nn.Linear(8, 8)
        self.enc = torch.nn.GRU(8, 4, num_layers=2, bidirectional=True)
        self.dead = torch.nn.Linear(3, 3)
        self.head = torch.nn.Linear(8, 6)

    def forward(self, xs):
        outs = []
        for x in xs:                                     # one dialogue at a time: (L_i, 8) -> (L_i, 6)
            y, _ = self.enc(self.proj(x).unsqueeze(1))
            outs.append(self.head(y.squeeze(1)))
        return torch.log_softmax(torch.cat(outs), 1)


def seq_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    distributed.init(backend="gloo")
    torch.manual_seed(0)
    model = TinySeq()
    lengths, xs, ys = make_data()
    mine = distributed.shard_dialogues(lengths, world, rank)
    y = torch.cat([ys[i] for i in mine])
    n_local = y.shape[0]
    n_global = distributed.all_reduce_scalar(n_local, device="cpu")
    bucket = distributed.GradientBucket(model)
    loss_f = FocalLoss(gamma=0.5)
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        (loss_f(model([xs[i] for i in mine]), y) * (n_local * world / n_global)).backward()
        bucket.all_reduce()
    names = {id(p): n for n, p in model.named_parameters()}
    order = [names[id(p)] for p in bucket.params]
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    # every gradient is a view of the one flat buffer, in bucket order
    off, views_ok = 0, True
    for p in bucket.params:
        views_ok &= p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * off
        off += distributed.slot_size(p)        # (every slot starts 16-byte aligned)
    # a parameter coming alive after the layout was frozen must raise on every rank (before any collective is issued)
    model.zero_grad(set_to_none=True)
    (loss_f(model([xs[i] for i in mine]), y) + model.dead(torch.ones(1, 3)).sum()).backward()
    try:
        bucket.all_reduce()
        raised = ""
    except RuntimeError as exc:
        raised = str(exc)
    torch.save({"grads": grads, "order": order, "views_ok": views_ok, "raised": raised, "bucket": bucket.flat.numel()},
               out + ".%d" % rank)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_bucket_order_keeps_gru_direction_pairs_adjacent(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "seq.pt")
    mp.spawn(seq_worker, args=(2, port, out), nprocs=2, join=True)
    got = [torch.load(out + ".%d" % r) for r in range(2)]
    torch.manual_seed(0)
    model = TinySeq()
    lengths, xs, ys = make_data()
    FocalLoss(gamma=0.5)(model(xs), torch.cat(ys)).backward()
    for g in got:
        order = g["order"]
        assert g["views_ok"] and "dead" in g["raised"] and not any(n.startswith("dead") for n in order)
        assert g["bucket"] == sum(distributed.slot_size(p) for n, p in model.named_parameters() if not n.startswith("dead"))
        for layer in range(2):
            for kind in ("weight_ih", "bias_ih"):
                n = "enc.%s_l%d" % (kind, layer)
                assert order.index(n + "_reverse") == order.index(n) + 1, order       # the fused GRU's stacked operand
        # the recurrent weights keep model order (no pairing rule for them)
        assert order.index("enc.weight_hh_l0") < order.index("enc.weight_hh_l0_reverse")
        assert order == got[0]["order"]                                              # identical layout on every rank
        for k, p in model.named_parameters():
            if not k.startswith("dead"):
                assert (g["grads"][k] - p.grad).abs().max() < 1e-6, k


# ------------------------------------------------------------------------------------------------------------------
# two-part bucket: the graph / head half is packed and reduced where the graph part of the backward pass ends
# ------------------------------------------------------------------------------------------------------------------
class _GraphDone(torch.autograd.Function):
    """Identity between the 'encoder' and the 'graph' half of the stand-in below; its backward is where the product's
    adjacency builder calls ops.flush_queued_wgrads_early() (-> the hook a two-part bucket registers)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        from mm_dfn_amd import ops
        ops.flush_queued_wgrads_early()
        return g


class TinyTwoHalves(torch.nn.Module):
    """Parameter names of the real model's two halves: an encoder (projection + bidirectional GRU: late part) feeding
    'graph_model.' / 'smax_fc.' layers (early part)."""

    def __init__(self):
        super().__init__()
        self.linear_l = torch.nn.Linear(8, 8)
        self.lstm_l = torch.nn.GRU(8, 4, num_layers=1, bidirectional=True)
        self.graph_model = torch.nn.Sequential(torch.nn.Linear(8, 10), torch.nn.ReLU(), torch.nn.Linear(10, 7))
        self.smax_fc = torch.nn.Linear(7, 6)
        self.fired = 0

    def forward(self, xs):
        outs = []
        for x in xs:
            y, _ = self.lstm_l(self.linear_l(x).unsqueeze(1))
            outs.append(y.squeeze(1))
        h = _GraphDone.apply(torch.cat(outs))
        return torch.log_softmax(self.smax_fc(self.graph_model(h)), 1)


def two_part_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    distributed.init(backend="gloo")
    lengths, xs, ys = make_data()
    mine = distributed.shard_dialogues(lengths, world, rank)
    y = torch.cat([ys[i] for i in mine])
    loss_f = FocalLoss(gamma=0.5)
    res = {}
    for parts in (1, 2):
        torch.manual_seed(0)
        model = TinyTwoHalves()
        bucket = distributed.GradientBucket(model, average=True, parts=parts)
        started_early = []
        for step in range(3):
            model.zero_grad(set_to_none=True)
            bucket.arm()
            loss_f(model([xs[i] for i in mine]), y).backward()
            started_early.append(bucket._early == "packed")
            bucket.all_reduce()
        names = {id(p): n for n, p in model.named_parameters()}
        res[parts] = dict(flat=bucket.flat.clone(), order=[names[id(p)] for p in bucket.params], split=bucket.split,
                          early=started_early, grads={n: p.grad.clone() for n, p in model.named_parameters()})
    if rank == 0:
        torch.save(res, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_part_bucket_reduces_to_the_same_bits_as_one_collective(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0.pt")
    mp.spawn(two_part_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    one, two = res[1], res[2]
    # the first step lays the bucket out (one collective); from the second step on the graph / head half leaves early
    assert two["early"] == [False, True, True] and one["early"] == [False, False, False]
    n_early = sum(1 for n in two["order"] if n.startswith(distributed.EARLY_PREFIXES))
    assert n_early == 6 and all(n.startswith(distributed.EARLY_PREFIXES) for n in two["order"][:n_early])
    assert not any(n.startswith(distributed.EARLY_PREFIXES) for n in two["order"][n_early:])
    assert 0 < two["split"] < two["flat"].numel()
    # same reduced gradients, bit for bit (two ranks: a + b in either collective)
    for n, g in one["grads"].items():
        assert torch.equal(g, two["grads"][n]), n
    # and the flat buffers hold the same values in their own orders
    assert torch.equal(one["flat"].sort().values, two["flat"].sort().values)
