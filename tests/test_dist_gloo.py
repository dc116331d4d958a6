"""The N>1 path on CPU: two gloo ranks shard candidates / calibration batches and combine with one
all-reduce; results must equal the single-process ones.  The per-element arithmetic is supplied
by the CPU oracle here (the product's only implementation is the HIP kernel)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_row_reduce(weights, grads, out, accumulate):
    from oracle import importance as oi
    pos = 0
    for w, g in zip(weights, grads):
        r = oi.taylor_rows(w, g)
        if accumulate:
            out[pos: pos + len(r)] += r
        else:
            out[pos: pos + len(r)] = r
        pos += len(r)


def _candidate_inputs(i, n_px=4096, c=7):
    rng = np.random.default_rng(100 + i)
    t = rng.integers(0, c, n_px)
    p = np.where(rng.random(n_px) < 0.2 + 0.05 * i, rng.integers(0, c, n_px), t)
    t[rng.random(n_px) < 0.05] = 255
    return p, t


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_golden
    from oracle import confusion as oc
    from spice_amd import dist as sdist
    from spice_amd.pruning.sensitivity_utils import compute_taylor_scores
    from toy_models import ToyNet

    # (a) candidate tables
    n, c = 11, 7
    table = torch.zeros(n, c, c, dtype=torch.int64)
    visited = []

    def evaluate(i):
        visited.append(i)
        p, t = _candidate_inputs(i)
        m = np.zeros((c, c), dtype=np.int64)
        oc.update_confusion(m, p, t, c)
        table[i] += torch.from_numpy(m)

    costs = [1.0 + (i % 3) for i in range(n)]
    sdist.sharded_tables(n, list(range(n)), costs, evaluate, [table])

    # (b) Taylor scores with calibration batches dealt round-robin to the ranks
    ty = load_golden("taylor.npz")
    net = ToyNet()
    net.load_state_dict({k[len("state/"):]: torch.from_numpy(ty[k]) for k in ty.files if k.startswith("state/")})
    batches = [(torch.from_numpy(ty[f"batch{b}/images"]), torch.from_numpy(ty[f"batch{b}/targets"])) for b in range(3)]
    scores = compute_taylor_scores(net, batches, "cpu", max_batches=100, shard=True, _row_reduce=_oracle_row_reduce)
    by_name = dict(net.named_modules())
    np.savez(os.path.join(out_dir, f"rank{rank}_of{world}.npz"), table=table.numpy(), visited=np.array(visited),
             **{f"taylor/{n_}": scores[by_name[n_]].numpy() for n_ in [str(x) for x in ty["conv_names"]]})
    if world > 1:
        dist.destroy_process_group()


def _run(world, out_dir):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(out_dir)), nprocs=world, join=True)
    return [np.load(os.path.join(out_dir, f"rank{r}_of{world}.npz")) for r in range(world)]


def test_two_ranks_equal_one_rank(tmp_path):
    from conftest import load_golden
    one = _run(1, tmp_path)[0]
    two = _run(2, tmp_path)
    # every candidate evaluated exactly once over the two ranks, none on both
    seen = sorted(list(two[0]["visited"]) + list(two[1]["visited"]))
    assert seen == list(range(11)) and len(two[0]["visited"]) > 0 and len(two[1]["visited"]) > 0
    for r in two:
        np.testing.assert_array_equal(r["table"], one["table"])  # integer tables: bit-exact on every rank
    ty = load_golden("taylor.npz")
    for n in [str(x) for x in ty["conv_names"]]:
        for r in two + [one]:
            np.testing.assert_allclose(r[f"taylor/{n}"], ty[f"tuple_all/{n}"], rtol=1e-5, atol=1e-9)
        np.testing.assert_array_equal(two[0][f"taylor/{n}"], two[1][f"taylor/{n}"])


def _grad_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from spice_amd import dist as sdist
    from spice_amd.pruning.dataset import PruningDataHandler
    from toy_models import ToyNet

    torch.manual_seed(5)
    net = ToyNet()
    net.pw.bias.requires_grad_(False)             # a frozen parameter stays outside the buckets
    data = torch.Generator().manual_seed(77)
    xs = torch.randn(4, 3, 12, 12, generator=data)
    ys = torch.randint(0, 4, (4, 12, 12), generator=data)
    lo, hi = (rank * 4 // world, (rank + 1) * 4 // world)
    buckets = sdist.FlatGradBuckets(net.parameters(), bucket_bytes=256)   # tiny: several buckets
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.1)
    n_buckets = len(buckets.buckets)
    for it in range(2):                           # second pass checks zeroing and hook re-arming
        net.train(it == 1)                        # pass 0 in eval mode: samples are independent
        buckets.zero_grad()
        loss = torch.nn.functional.cross_entropy(net(xs[lo:hi]), ys[lo:hi])
        loss.backward()
        buckets.finish()
        if it == 0:
            grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.requires_grad}
        opt.step()
    # gradient accumulation: two backward passes per step, exchange only after the second
    net.eval()
    buckets.zero_grad()
    mid = (lo + hi) // 2 if hi - lo > 1 else hi
    parts = [(lo, mid), (mid, hi)] if mid < hi else [(lo, hi)]
    for k, (a, b) in enumerate(parts):
        buckets.hold = k < len(parts) - 1
        (torch.nn.functional.cross_entropy(net(xs[a:b]), ys[a:b], reduction="sum") / (12 * 12 * 4 / world)).backward()
    buckets.finish()
    acc_grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.requires_grad}
    buckets.zero_grad()
    (torch.nn.functional.cross_entropy(net(xs[lo:hi]), ys[lo:hi], reduction="sum") / (12 * 12 * 4 / world)).backward()
    buckets.finish()
    one_grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.requires_grad}
    acc_err = max(float((acc_grads[n] - one_grads[n]).abs().max() / (one_grads[n].abs().max() + 1e-12)) for n in one_grads)
    net.train()
    sdist.average_buffers_(net)
    # loaders: the training split is dealt to the ranks without overlap, validation by stride
    h = PruningDataHandler(data_root=os.path.join(out_dir, "nodata"), batch_size=2, seed=3, num_classes=4,
                           input_size=(32, 32), synthetic_length=10)
    train, val = h.get_train_val_loaders()
    train_ids = [int(i) for i in train.sampler] if world > 1 else list(range(10))
    val_ids = list(val.dataset.indices) if world > 1 else list(range(10))
    cal_first = next(iter(h.get_calibration_dataloader()))[1][0, 0, :4].tolist()
    np.savez(os.path.join(out_dir, f"grad_rank{rank}_of{world}.npz"), n_buckets=n_buckets, acc_err=acc_err,
             train_ids=np.array(train_ids), val_ids=np.array(val_ids), cal_first=np.array(cal_first),
             **{f"g/{k}": v.numpy() for k, v in grads.items()},
             **{f"p/{k}": v.detach().numpy() for k, v in net.named_parameters()},
             **{f"b/{k}": v.numpy() for k, v in net.named_buffers() if v.is_floating_point()})
    buckets.close()
    if world > 1:
        dist.destroy_process_group()


def test_gradient_buckets_average_over_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_grad_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    one = np.load(os.path.join(tmp_path, "grad_rank0_of1.npz"))
    port = _free_port()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    two = [np.load(os.path.join(tmp_path, f"grad_rank{r}_of2.npz")) for r in range(2)]
    assert int(two[0]["n_buckets"]) > 1
    assert float(two[0]["acc_err"]) < 1e-5 and float(two[1]["acc_err"]) < 1e-5 and float(one["acc_err"]) < 1e-5
    keys = [k for k in one.files if k.startswith("g/")]
    assert keys and "g/pw.bias" not in keys
    for k in keys:
        # mean over the two half-batch gradients == full-batch gradient (eval-mode pass)
        np.testing.assert_array_equal(two[0][k], two[1][k])
        np.testing.assert_allclose(two[0][k], one[k], rtol=1e-4, atol=1e-7)
    for k in [k for k in one.files if k.startswith("p/") or k.startswith("b/")]:
        np.testing.assert_array_equal(two[0][k], two[1][k])   # replicas stay in lock step
    # data sharding: disjoint, complete
    tr = sorted(list(two[0]["train_ids"]) + list(two[1]["train_ids"]))
    assert tr == list(range(10))
    assert sorted(list(two[0]["val_ids"]) + list(two[1]["val_ids"])) == list(range(10))
    assert not set(two[0]["val_ids"]) & set(two[1]["val_ids"])
    np.testing.assert_array_equal(two[0]["cal_first"], two[1]["cal_first"])  # same calibration order on all ranks
