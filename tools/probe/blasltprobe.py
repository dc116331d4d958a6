"""Host time of torch.bmm / torch.matmul with rocBLAS and with hipBLASLt as torch's preferred BLAS."""
import time
import torch
dev = torch.device("cuda:0")
n, cin, h, w, cout = 8, 1024, 32, 32, 256
x = torch.randn(n, cin, h, w, device=dev).half()
w2 = (torch.randn(cout, cin, device=dev) / 32).half()
def bench(fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * host / reps, 1e6 * (time.perf_counter() - t0) / reps
ref = None
for lib in ("default", "cublas", "cublaslt"):
    if lib != "default":
        torch.backends.cuda.preferred_blas_library(lib)
    print("preferred:", torch.backends.cuda.preferred_blas_library())
    y = torch.bmm(w2.expand(n, -1, -1), x.view(n, cin, h * w))
    ref = y if ref is None else ref
    a = bench(lambda: torch.bmm(w2.expand(n, -1, -1), x.view(n, cin, h * w)))
    xt = x.view(n, cin, h * w)
    b = bench(lambda: torch.matmul(w2, xt))
    # one flat GEMM over the whole batch: X as [cin, n*hw] needs a transpose of the batch; time the NHWC-style linear instead
    xl = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    c = bench(lambda: torch.nn.functional.linear(xl, w2))
    print(f"  bmm host {a[0]:.1f} us (wall {a[1]:.1f}) | matmul host {b[0]:.1f} | linear [n*hw,cin] host {c[0]:.1f} (wall {c[1]:.1f}) | equal to first: {torch.equal(y, ref)}", flush=True)
