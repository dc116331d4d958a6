"""Host enqueue time of the ways to run W[cout,cin] @ X[n,cin,hw] in fp16 through torch."""
import time
import torch
dev = torch.device("cuda:0")
n, cin, h, w, cout = 8, 256, 128, 128, 64
x = torch.randn(n, cin, h, w, device=dev).half()
w2 = (torch.randn(cout, cin, device=dev) / 16).half()
we = w2.expand(n, cout, cin)
out = torch.empty(n, cout, h * w, device=dev, dtype=torch.float16)
def bench(fn, reps=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * host / reps, 1e6 * (time.perf_counter() - t0) / reps
bmm = torch.ops.aten.bmm.default
bmm_out = torch.ops.aten.bmm.out
for name, fn in (("matmul", lambda: torch.matmul(w2, x.view(n, cin, h * w)).view(n, cout, h, w)),
                 ("bmm expand each call", lambda: torch.bmm(w2.expand(n, cout, cin), x.view(n, cin, h * w)).view(n, cout, h, w)),
                 ("bmm pre-expanded", lambda: torch.bmm(we, x.view(n, cin, h * w)).view(n, cout, h, w)),
                 ("aten.bmm pre-expanded", lambda: bmm(we, x.view(n, cin, h * w)).view(n, cout, h, w)),
                 ("bmm out=", lambda: torch.bmm(we, x.view(n, cin, h * w), out=out)),
                 ("empty only", lambda: torch.empty(n, cout, h, w, device=dev, dtype=torch.float16)),
                 ("view only", lambda: x.view(n, cin, h * w))):
    with torch.no_grad():
        a = bench(fn)
        with torch.autocast("cuda", dtype=torch.float16):
            b = bench(fn)
    print(f"{name:24s} host {a[0]:5.1f} us (wall {a[1]:5.1f}) | under autocast host {b[0]:5.1f} us", flush=True)
y1 = torch.matmul(w2, x.view(n, cin, h * w)); y2 = torch.bmm(we, x.view(n, cin, h * w))
print("bmm == matmul:", torch.equal(y1, y2))
