"""Warm MIOpen's find database for the fixed convolution shapes of a model.

The sweep runs MIOpen with MIOPEN_FIND_MODE=FAST: a convolution whose problem is in the find
database uses the recorded best solver, any other problem falls back to the heuristic choice at
once (no search).  That is the right trade for the one to three NEW shapes every candidate
brings, but on a machine whose database is empty it also leaves the ~70 UNCHANGED convolutions of
every candidate on heuristic picks (about 8.5 ms per DeepLabV3+-R50 forward at 8x3x512x512 fp16
against 5.2 ms with searched solvers, measured on MI355X).  This module runs one real search
(MIOPEN_FIND_MODE=NORMAL, about 35 ms per shape) for the un-pruned model in a CHILD process, which
writes the results to the user find database the parent then reads.  The find mode is read once
per process, hence the child; start it before the parent initialises the GPU.

    python -m spice_amd.miopen_tuning --arch DeepLabV3Plus --encoder resnet50 --classes 21 \
        --batch 8 --size 512 --autocast fp16 --device 0
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import time


def default_db_dir(local_rank: int = 0) -> str:
    """A per-user, per-rank database directory (ranks of one node do not share a lock)."""
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"spice_miopen_{os.getuid()}_r{local_rank}")


def ensure_tuned(arch: str, encoder: str, classes: int, batch: int, size: int, autocast: str = "fp16",
                 device: int = 0, timeout: float = 600.0) -> float:
    """Run the search in a child process (blocking); returns its wall time in seconds.  A stamp file
    in the database directory makes repeated calls for the same problem free."""
    db = os.environ.setdefault("MIOPEN_USER_DB_PATH", default_db_dir(device))
    os.makedirs(db, exist_ok=True)
    # the convolutions that reach MIOpen differ with SPICE_TORCH_FORWARD (the child inherits it): one stamp each
    route = "torch" if os.environ.get("SPICE_TORCH_FORWARD", "0") == "1" else "lib"
    stamp = os.path.join(db, f"tuned_{arch}_{encoder}_{classes}_{batch}_{size}_{autocast}_{route}.stamp")
    if os.path.exists(stamp):
        return 0.0
    env = dict(os.environ, MIOPEN_FIND_MODE="NORMAL", MIOPEN_USER_DB_PATH=db)
    cmd = [sys.executable, "-m", "spice_amd.miopen_tuning", "--arch", arch, "--encoder", encoder,
           "--classes", str(classes), "--batch", str(batch), "--size", str(size), "--autocast", autocast,
           "--device", str(device)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.perf_counter()
    done = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)
    if done.returncode != 0:
        raise RuntimeError(f"MIOpen tuning child failed ({done.returncode}):\n{done.stdout[-2000:]}{done.stderr[-2000:]}")
    with open(stamp, "w") as f:
        f.write(done.stdout[-500:])
    return time.perf_counter() - t0


def _child(args) -> None:
    os.environ["MIOPEN_FIND_MODE"] = "NORMAL"  # before the first convolution of this process
    import torch

    from . import nets
    dev = torch.device("cuda", args.device)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    model = nets.create_model(args.arch, args.encoder, args.classes).to(dev).eval()
    x = torch.randn(args.batch, 3, args.size, args.size, device=dev)
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": None}[args.autocast]
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(2):
            if dtype is None:
                model(x)
            else:
                with torch.autocast("cuda", dtype=dtype):
                    model(x)
    torch.cuda.synchronize()
    print(f"searched {sum(1 for m in model.modules() if isinstance(m, torch.nn.Conv2d))} convolutions "
          f"in {time.perf_counter() - t0:.1f} s -> {os.environ.get('MIOPEN_USER_DB_PATH', '~/.config/miopen')}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="DeepLabV3Plus")
    ap.add_argument("--encoder", default="resnet50")
    ap.add_argument("--classes", type=int, default=21)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--autocast", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--device", type=int, default=0)
    _child(ap.parse_args())
