"""Host time and results of the 1x1-convolution product through rocBLAS called directly (ctypes)
against torch.bmm."""
import ctypes, time
import torch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
rb = ctypes.CDLL("librocblas.so.5")
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
rb.rocblas_create_handle.argtypes = [ctypes.POINTER(P)]
rb.rocblas_set_stream.argtypes = [P, P]
rb.rocblas_gemm_strided_batched_ex.argtypes = [P, I, I, I, I, I, P, P, I, I, L, P, I, I, L, P, P, I, I, L, P, I, I, L, I, I, I,
                                               ctypes.c_int32, ctypes.c_uint32]
h = P()
assert rb.rocblas_create_handle(ctypes.byref(h)) == 0
alpha, beta = ctypes.c_float(1.0), ctypes.c_float(0.0)
F16, F32, NONE = 150, 151, 111
raw = torch._C._cuda_getCurrentRawStream

def gemm(w2, x, out):
    n, cin, hh, ww = x.shape
    cout, hw = w2.shape[0], hh * ww
    rb.rocblas_set_stream(h, raw(0))
    return rb.rocblas_gemm_strided_batched_ex(h, NONE, NONE, hw, cout, cin, ctypes.byref(alpha), x.data_ptr(), F16, hw, cin * hw,
                                              w2.data_ptr(), F16, cin, 0, ctypes.byref(beta), out.data_ptr(), F16, hw, cout * hw,
                                              out.data_ptr(), F16, hw, cout * hw, n, F32, 0, 0, 0)

def bench(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * host / reps, 1e6 * (time.perf_counter() - t0) / reps

for n, cin, hh, ww, cout in [(8, 64, 128, 128, 256), (8, 256, 128, 128, 64), (8, 512, 64, 64, 128), (8, 1024, 32, 32, 256),
                             (8, 2048, 32, 32, 512), (8, 256, 32, 32, 1024), (8, 1280, 32, 32, 256), (8, 304, 128, 128, 256),
                             (8, 100, 128, 128, 256), (8, 230, 32, 32, 1000)]:
    x = torch.randn(n, cin, hh, ww, device=dev).half()
    w2 = (torch.randn(cout, cin, device=dev) / cin ** 0.5).half()
    out = torch.empty(n, cout, hh, ww, device=dev, dtype=torch.float16)
    rc = gemm(w2, x, out)
    ref = torch.bmm(w2.expand(n, -1, -1), x.view(n, cin, hh * ww)).view(n, cout, hh, ww)
    torch.cuda.synchronize()
    diff = int((out != ref).sum())
    a = bench(lambda: gemm(w2, x, torch.empty(n, cout, hh, ww, device=dev, dtype=torch.float16)))
    b = bench(lambda: torch.bmm(w2.expand(n, -1, -1), x.view(n, cin, hh * ww)).view(n, cout, hh, ww))
    print(f"[{n},{cin},{hh},{ww}]->{cout}: rc {rc} differing {diff} of {out.numel()} max {float((out.float()-ref.float()).abs().max()):.1e} | "
          f"rocBLAS host {a[0]:5.1f} us wall {a[1]:6.1f} | bmm host {b[0]:5.1f} us wall {b[1]:6.1f}", flush=True)
