"""Worker of test_gpu_api.py::test_two_rank_sharded_evaluation: baseline.eval.inference with the
evaluation batches dealt to the ranks and one all-reduce of the confusion matrix."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from spice_amd import dist as sdist  # noqa: E402
from spice_amd.baseline.eval import inference  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, world, local = sdist.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # small-integer weights and inputs: every logit is exact in fp16 and fp32 whatever convolution
    # algorithm a process happens to pick, so the matrices of different processes are comparable
    g = torch.Generator().manual_seed(12)
    model = torch.nn.Conv2d(3, 6, 1)
    with torch.no_grad():
        model.weight.copy_(torch.randint(-2, 3, model.weight.shape, generator=g).float())
        model.bias.copy_(torch.randint(-1, 2, model.bias.shape, generator=g).float())
    loader = []
    for _ in range(5):
        y = torch.randint(0, 6, (2, 64, 64), generator=g)
        y[torch.rand(2, 64, 64, generator=g) < 0.1] = 255
        loader.append((torch.randint(-3, 4, (2, 3, 64, 64), generator=g).float(), y))
    res = inference(model, loader, dev, 6)
    with open(os.path.join(out_dir, f"eval_rank{rank}_of{world}.json"), "w") as f:
        json.dump(res, f)
    sdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
