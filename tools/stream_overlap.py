"""Does overlapping consecutive render steps on two HIP streams pay?  The step (group network -> fused field kernel -> compositing) of step i + 1 is
queued on the other stream with its own scratch buffers: its small kernels and the head of its persistent field kernel fill the CUs the tail of
step i's field kernel leaves idle.   python tools/stream_overlap.py [steps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
import season_nerf_amd as sn

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda")
R, S, NC = bench.R, bench.S, bench.NC
L = sn._lib.lib()
net = sn.T_NeRF(bench.W, NC)
net.load_state_dict(sn.synthetic_state_dict(net, 0))
net = net.to(dev).eval()
model = net.device_model()
d = bench.synth(0, dev)
top, bot, sun, tim = d["Top"], d["Bot"], d["Sun_Angle"], d["Time_Encoded"]
tv = sn.sample_parameters(S, eval_mode=True).to(dev)
e = lambda *s: torch.empty(*s, device=dev)


class Slot:
    def __init__(self, stream):
        self.stream = stream
        self.st = C.c_void_p(stream.cuda_stream)
        self.cls, self.sky_raw, self.sky = e(R, NC), e(R, 3), e(R, 3)
        self.rho, self.sv, self.col, self.rgb = e(R * S), e(R * S), e(R * S, 3), e(R, 3)
        self.fo = sn._lib.FieldOut(d_rho=self.rho.data_ptr(), d_solar_vis=self.sv.data_ptr(), d_col=self.col.data_ptr())
        self.co = sn._lib.CompositeOut(d_rgb=self.rgb.data_ptr())

    def step(self):
        s = self
        sn._lib.check(L.snerf_group_forward(model, R, tim.data_ptr(), sun.data_ptr(), s.cls.data_ptr(), s.sky_raw.data_ptr(), s.sky.data_ptr(), s.st), "group")
        sn._lib.check(L.snerf_field_forward_rays(model, 0, R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), 1, sun.data_ptr(), s.cls.data_ptr(), C.byref(s.fo), s.st), "field")
        sn._lib.check(L.snerf_composite_rays(R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), s.rho.data_ptr(), s.col.data_ptr(), s.sv.data_ptr(), s.sky.data_ptr(), 0, None, 1.0,
                                             C.byref(s.co), s.st), "composite")


def run(n_streams, reps=5):
    slots = [Slot(torch.cuda.Stream()) for _ in range(n_streams)]
    for k in range(200):
        slots[k % n_streams].step()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            slots[k % n_streams].step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    ref = slots[0].rgb.clone()
    return sorted(out), ref


base, ref1 = run(1)
for n in (1, 2, 3, 4):
    t, ref = run(n)
    print(f"{n} stream(s): ms per step median {t[len(t)//2]:.4f}  min {t[0]:.4f}   = {R * S / (t[len(t)//2] * 1e-3):.4e} ray-samples/s   rgb identical to 1-stream: {bool(torch.equal(ref, ref1))}")
