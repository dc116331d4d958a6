"""Per-convolution forward time of a model (HIP events via forward hooks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from spice_amd import nets

def run(dtype, channels_last, benchmark):
    torch.backends.cudnn.benchmark = benchmark
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
    x = torch.randn(8, 3, 512, 512, device=dev)
    if channels_last:
        m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    recs = {}
    def pre(mod, inp):
        e = torch.cuda.Event(enable_timing=True); e.record(); recs.setdefault(mod, []).append([e, None])
    def post(mod, inp, out):
        e = torch.cuda.Event(enable_timing=True); e.record(); recs[mod][-1][1] = e
    names = {}
    for n, mod in m.named_modules():
        if isinstance(mod, nn.Conv2d):
            names[mod] = n
            mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)
    import time
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            if dtype is not None:
                with torch.autocast("cuda", dtype=dtype):
                    y = m(x)
            else:
                y = m(x)
        torch.cuda.synchronize(); print(f"  iter {it}: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
    rows = []
    for mod, lst in recs.items():
        a, b = lst[-1]
        rows.append((a.elapsed_time(b), names[mod], tuple(mod.weight.shape), mod.stride, mod.dilation, mod.groups))
    rows.sort(reverse=True)
    print(f"dtype={dtype} channels_last={channels_last} benchmark={benchmark}: conv total {sum(r[0] for r in rows):.1f} ms")
    for r in rows[:12]:
        print("   %.3f ms %s %s stride=%s dil=%s groups=%d" % r)

for cfg in [(torch.float16, False, False), (torch.float16, True, False), (None, False, False), (torch.float16, False, True)]:
    run(*cfg)
