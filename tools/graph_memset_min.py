"""GPU: the smallest form of tools/graph_memop_repro.py.  A hipMemsetAsync inside a torch.cuda.graph capture is a MEMSET node; on ROCm 7.2 (AQL packet capture
of graph nodes on) replays intermittently leave part of the range un-zeroed.  Expected: acc == K * x after every replay.  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: clean."""
import ctypes, torch
hip, N, K = ctypes.CDLL("libamdhip64.so"), 1 << 16, 40
x, acc = torch.rand(N, device="cuda"), torch.empty(N, device="cuda")
def body():
    hip.hipMemsetAsync(ctypes.c_void_p(acc.data_ptr()), 0, ctypes.c_size_t(4 * N), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(K):
        acc.add_(x)
body(); torch.cuda.synchronize(); ref = acc.clone()
bad = 0
for cap in range(3):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for k in range(12):
        g.replay(); torch.cuda.synchronize()
        wrong = torch.nonzero(acc != ref).reshape(-1)
        if wrong.numel():
            bad += 1
            print(f"capture {cap} replay {k}: {wrong.numel()} of {N} elements wrong, index mod 4 in {torch.unique(wrong % 4).tolist()}, acc/ref = {float(acc[wrong[0]] / ref[wrong[0]]):.2f}")
print(f"{bad} of 36 replays wrong")
