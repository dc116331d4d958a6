"""One rank, backend nccl (= RCCL), on cuda:0: the collectives of the sharded path on device buffers of its
shapes and dtypes, the asynchronous bucketed gradient all-reduce, and a sharded Taylor calibration.
Run by tests/test_gpu_api.py::test_rccl_single_rank_collectives in a process of its own."""
import os
import sys

os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                  MASTER_PORT=os.environ.get("MASTER_PORT", "29541"), SPICE_DIST_FORCE_INIT="1")
os.environ.pop("SPICE_DIST_BACKEND", None)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from spice_amd import dist as sdist  # noqa: E402
from spice_amd.pruning.sensitivity_utils import compute_taylor_scores  # noqa: E402
from toy_models import ToyNet  # noqa: E402


def main():
    rank, world, local = sdist.init_from_env("nccl")
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", local)
    # the sweep's tables: [n_candidates, C, C] int64 and [n, 3] fp32
    table = torch.arange(414 * 21 * 21, dtype=torch.int64, device=dev).reshape(414, 21, 21)
    kd = torch.rand(414, 3, device=dev)
    want_t, want_k = table.clone(), kd.clone()
    sdist.all_reduce_sum_(table)
    sdist.all_reduce_sum_(kd)
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    sdist.all_reduce_max_(t)
    sdist.barrier()
    torch.cuda.synchronize()
    assert torch.equal(table, want_t) and torch.equal(kd, want_k) and float(t.item()) == 1.25
    # gradient buckets: asynchronous all-reduce launched from the backward hooks, then finish()
    torch.manual_seed(0)
    net = ToyNet().to(dev)
    buckets = sdist.FlatGradBuckets(net.parameters(), bucket_bytes=4096)
    buckets.zero_grad()
    x = torch.randn(2, 3, 16, 16, device=dev)
    net(x).square().mean().backward()
    buckets.finish()
    ref = ToyNet().to(dev)
    ref.load_state_dict(net.state_dict())
    ref(x).square().mean().backward()
    for a, b in zip(net.parameters(), ref.parameters()):   # same kernels, but MIOpen may pick another algorithm per call
        assert torch.allclose(a.grad, b.grad, rtol=1e-4, atol=1e-7)
    buckets.close()
    # sharded Taylor calibration: one flat all-reduce carries scores and batch count
    loader = [(torch.randn(2, 3, 16, 16), torch.randint(0, 4, (2, 16, 16))) for _ in range(3)]
    one = compute_taylor_scores(net, loader, str(dev), max_batches=3, shard=False)
    two = compute_taylor_scores(net, loader, str(dev), max_batches=3, shard=True)
    for m in one:
        assert torch.allclose(one[m], two[m], rtol=1e-4, atol=1e-9)
    dist.destroy_process_group()
    print("rccl ok")


if __name__ == "__main__":
    main()
