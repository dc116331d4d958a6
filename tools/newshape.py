"""Host cost of the first convolution call on a new problem shape (MIOPEN_FIND_MODE=FAST), on the
calling thread and after another thread has already run the same problem."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spice_amd  # sets MIOPEN_FIND_MODE=FAST
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")

def call(cin, cout, k, hw, stream=None):
    x = torch.zeros(8, cin, hw, hw, device=dev, dtype=torch.float16)
    w = torch.zeros(cout, cin, k, k, device=dev, dtype=torch.float16)
    t0 = time.perf_counter()
    y = F.conv2d(x, w, None, 1, k // 2)
    return 1e3 * (time.perf_counter() - t0)

call(64, 64, 3, 64); torch.cuda.synchronize()
for k, hw, cout in ((3, 64, 256), (1, 128, 256), (3, 32, 512)):
    firsts, seconds = [], []
    for cin in range(100, 110):
        firsts.append(call(cin, cout, k, hw)); seconds.append(call(cin, cout, k, hw))
    torch.cuda.synchronize()
    print(f"k={k} hw={hw} cout={cout}: first call {sum(firsts)/10:.3f} ms, second {sum(seconds)/10:.3f} ms", flush=True)

# another thread runs the problems first
res = {}
def helper():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        res["helper_first"] = [call(cin, 256, 3, 64) for cin in range(120, 130)]
    s.synchronize()
th = threading.Thread(target=helper); th.start(); th.join()
main_after = [call(cin, 256, 3, 64) for cin in range(120, 130)]
main_again = [call(cin, 256, 3, 64) for cin in range(120, 130)]
torch.cuda.synchronize()
print(f"helper thread first {sum(res['helper_first'])/10:.3f} ms; main thread afterwards {sum(main_after)/10:.3f} ms; main again {sum(main_again)/10:.3f} ms")
# overlap: helper warms shapes while the main thread is busy launching known-shape convolutions
def helper2():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for cin in range(140, 160):
            call(cin, 256, 3, 64)
    s.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    call(64, 64, 3, 64)
torch.cuda.synchronize(); alone = time.perf_counter() - t0
th = threading.Thread(target=helper2); th.start()
t0 = time.perf_counter()
for _ in range(2000):
    call(64, 64, 3, 64)
torch.cuda.synchronize(); together = time.perf_counter() - t0
th.join()
print(f"2000 known-shape calls alone {1e3*alone:.1f} ms, with a helper warming 20 shapes {1e3*together:.1f} ms")
