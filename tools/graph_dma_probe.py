"""Does a kernel node of a replayed hipGraph see what a host-to-device DMA copy wrote into the buffer it reads?  (A hypothesis for the NaN parameters of the
single-staging-buffer variant of the captured training step, DESIGN 5.4c.  Answer on MI355X / ROCm 7.2: yes, always - the hypothesis does not hold.)
A graph of one kernel y = x + 0 (x: N floats, first touched by a fill kernel so that its lines sit in the L2); per iteration a new value reaches x either
 (a) directly: x.copy_(pinned, non_blocking=True)                  - the copy engine writes the buffer the graph reads,
 (b) through a kernel: tmp.copy_(pinned, non_blocking=True); x.copy_(tmp)
then the graph replays and y is compared with the value sent (argv[1]: number of unrelated kernel nodes in front of the reading one).  Also (c): (a) followed by an ordinary (non-graph) launch of the same kernel.
    python3 tools/graph_dma_probe.py"""
import torch

import sys
dev = torch.device("cuda")
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 0
print("kernel nodes in front of the reader:", DEPTH)
for N in (6, 3264, 1 << 20):
    x = torch.zeros(N, device=dev)
    tmp = torch.zeros(N, device=dev)
    y = torch.zeros(N, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            y.copy_(x + 0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    w = torch.ones(1 << 22, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(DEPTH):                 # DEPTH kernel nodes that do not touch x in front of the one that reads it
            w.mul_(1.0)
        y.copy_(x + 0)
    pins = [torch.zeros(N).pin_memory() for _ in range(8)]
    res = {}
    for mode in ("a: DMA -> x, graph", "b: DMA -> tmp, kernel -> x, graph", "c: DMA -> x, plain launch"):
        bad = 0
        for it in range(200):
            v = float(it + 1)
            p = pins[it % 8]
            p.fill_(v)
            if mode[0] == "b":
                tmp.copy_(p, non_blocking=True)
                x.copy_(tmp)
            else:
                x.copy_(p, non_blocking=True)
            if mode[0] == "c":
                y.copy_(x + 0)
            else:
                g.replay()
            if it % 8 == 7:
                torch.cuda.synchronize()        # (the pinned images are reused every 8 iterations)
            got = y.clone()
            torch.cuda.synchronize()
            bad += int((got != v).sum() > 0)
        res[mode] = bad
    print(f"N = {N:8d}:", {k: f"{v} of 200 replays read stale data" for k, v in res.items()}, flush=True)
