"""HBM bandwidth by direction on the box at hand (torch elementwise kernels, 1 GiB tensors, HIP events): write only (fill), read only
(sum), copy (read + write).  Context for DESIGN 5.4b: a layer-wise training GEMM reads Z_in and writes Z_out - its floor is the sum of
a read stream and a write stream, and the write direction is the slow one."""
import torch
n = 1 << 28                      # 1 GiB of fp32
x = torch.empty(n, device="cuda")
y = torch.empty(n, device="cuda")
def t(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = n * 4 / 1e12
for _ in range(2):
    print("write only  (fill_)   %.2f TB/s" % (gb / t(lambda: x.fill_(1.5))))
    print("read only   (sum)     %.2f TB/s" % (gb / t(lambda: x.sum())))
    print("copy        (copy_)   %.2f TB/s of read + write" % (2 * gb / t(lambda: y.copy_(x))))
    print("read 2 + write 1 (add) %.2f TB/s" % (3 * gb / t(lambda: torch.add(x, y, out=y))))
