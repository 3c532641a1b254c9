"""GPU, torch + the HIP runtime only (none of this repository's kernels): do MEMSET / MEMCPY nodes in a captured graph survive eager work between replays?
The pattern of the training engine before round 6: zero a buffer with hipMemsetAsync, accumulate into it with kernels, copy results with hipMemcpyAsync D2D -
captured once, replayed, with the same sequence run EAGERLY on other buffers between replays (what a save point's validation did), and captured a second time
after the first graph is gone (what the phase switch did).  Every replay must reproduce the eager result.
  MEMOPS=1 (default): runtime memory operations -> MEMSET / MEMCPY nodes;  MEMOPS=0: the same with torch kernels (zero_(), add into a copy)."""
import ctypes as C, os, sys
import torch
hip = C.CDLL("libamdhip64.so")
MEMOPS = os.environ.get("MEMOPS", "1") == "1"
SET_NODE, CPY_NODE = MEMOPS and os.environ.get("ONLY", "") != "memcpy", MEMOPS and os.environ.get("ONLY", "") != "memset"
EAGER_BETWEEN = int(os.environ.get("EAGER_BETWEEN", "3"))
N, CHAIN = 1 << 16, int(os.environ.get("CHAIN", "60"))

def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
def zero(t):
    if SET_NODE: assert hip.hipMemsetAsync(C.c_void_p(t.data_ptr()), 0, C.c_size_t(t.numel() * 4), stream()) == 0
    else: t.zero_()
def copy(d, s):
    if CPY_NODE: assert hip.hipMemcpyAsync(C.c_void_p(d.data_ptr()), C.c_void_p(s.data_ptr()), C.c_size_t(s.numel() * 4), 3, stream()) == 0      # 3 = device to device
    else: torch.add(s, 0.0, out=d)

class Work:
    def __init__(self, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.x = torch.rand(N, device="cuda", generator=g)
        self.acc, self.tmp, self.out = (torch.empty(N, device="cuda") for _ in range(3))
        self.small = torch.empty(192, device="cuda"); self.small_out = torch.empty(192, device="cuda")
    def body(self):
        zero(self.acc); zero(self.small)
        y = self.x
        for i in range(CHAIN):
            y = torch.sin(y) * 1.01 + 0.1
            self.acc.add_(y)
            if i % 5 == 4:
                copy(self.tmp, self.acc)
                y = self.tmp * 0.5 + y
                self.small.add_(y[:192])
        copy(self.out, self.acc); copy(self.small_out, self.small)

def capture(w):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        w.body()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        w.body()
    return g

bad = 0
other = Work(99)
for cap in range(3):                       # three captures in one process, each after the previous graph is gone
    w = Work(cap)
    w.body(); torch.cuda.synchronize()
    ref, ref_small = w.out.clone(), w.small_out.clone()
    g = capture(w)
    for k in range(12):
        w.out.fill_(-1.0); w.small_out.fill_(-1.0)
        g.replay(); torch.cuda.synchronize()
        if not (torch.equal(w.out, ref) and torch.equal(w.small_out, ref_small)):
            bad += 1
            idx = torch.nonzero(w.out != ref).reshape(-1)
            if bad <= 3:
                print(f"capture {cap} replay {k}: differs in {idx.numel()} of {N} / {int((w.small_out != ref_small).sum())} of 192 elements; index mod 4 of the wrong ones: "
                      f"{torch.unique(idx % 4).tolist()}, first {idx[:4].tolist()}, got {w.out[idx[:3]].tolist()} want {ref[idx[:3]].tolist()}")
        for _ in range(EAGER_BETWEEN):      # eager work between replays
            other.body()
        torch.cuda.synchronize()
    del g
print(f"MEMOPS={int(MEMOPS)} ONLY={os.environ.get('ONLY', '-')} EAGER_BETWEEN={EAGER_BETWEEN} PACKET_CAPTURE={os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE', 'default')}: {bad} of 36 replays differ from the eager result")
