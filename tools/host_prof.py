"""Where the HOST time of a training step goes (bench.py --workload train, configs[2]): every step is enqueued onto an IDLE GPU
(synchronize before it), so the wall time of the enqueue is the host's own cost - no back-pressure from a full queue - and
cProfile covers the step loop only.   python tools/host_prof.py [steps]"""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sys.argv = ["bench.py", "--workload", "train", "--no-cpu-baseline", "--steps", "3", "--warmup", "2"]
import torch
import bench
import season_nerf_amd as sn

orig = sn.Net_tool.train_step
times = []
pr = cProfile.Profile()
state = {"on": False}


def timed(self, d, k):
    if not state["on"]:
        return orig(self, d, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = orig(self, d, k)
    times.append((time.perf_counter() - t0) * 1e3)
    return r


sn.Net_tool.train_step = timed
bench.main()                      # warm: engines built, autograd registered, kernels loaded
state["on"] = True
sys.argv = ["bench.py", "--workload", "train", "--no-cpu-baseline", "--steps", str(n), "--warmup", "3"]
pr.enable()
bench.main()
pr.disable()
times.sort()
print(f"host enqueue per step onto an idle GPU (under cProfile): median {times[len(times)//2]:.2f} ms, min {times[0]:.2f}, max {times[-1]:.2f}  ({len(times)} steps)")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue()[:12000])
