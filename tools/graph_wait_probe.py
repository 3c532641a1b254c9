"""GPU: the event-wait trigger of the captured training step's failure family - history; the cause is hipGraph MEMSET nodes under the runtime's
packet capture (tools/graph_memop_repro.py), which the engine no longer issues.  With SNERF_TRAIN_MEMOPS=1 (the engine's old hipMemcpyAsync / hipMemsetAsync
calls back) this reproduces what round 6 first saw:

The driver test `tests/test_net_tool.py::test_driver_with_use_graph_switches_to_the_captured_step` captures the step twice in one process (DSM-prior phase, then
the free phase) and compares 20 steps with the eager driver at the level of the losses.  With an event wait between the current stream and torch's capture stream
in front of a capture - two lines, no kernel - the second graph's replays drift from the eager step by 1-4 % from its 4th replay on, in most runs (waitonly 4 of 6,
dummy 3 of 6, 1: 3 of 6).  With kernel nodes only every variant is 0 of 6.  Variants (trainer.GraphedTrainStep._capture, SNERF_GRAPH_PREPARE):
    0         nothing touches the capture stream
    waitonly  cs.wait_stream(cur); cur.wait_stream(cs)
    nowait    a small kernel on cs, no waits
    dummy     waits + a small kernel on cs
    1         waits + the loss scratch created on cs (the default since the cause was removed)
    cs_waits_cur / cur_waits_cs   one of the two waits only;   other   both waits, with a fresh stream that is NOT the capture stream
usage (GPU box): python3 tools/graph_wait_probe.py [runs per variant, default 6] [comma-separated variants]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ("0", "nowait", "waitonly", "dummy", "1")
for mode in modes:
    fails = 0
    for _ in range(n):
        r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_net_tool.py", "-x", "-q", "-m", "gpu", "-k", "driver_with_use_graph"], cwd=REPO,
                           env=dict(os.environ, SNERF_GRAPH_PREPARE=mode), capture_output=True, text=True)
        fails += r.returncode != 0
    print(f"SNERF_GRAPH_PREPARE={mode:9s} {fails} of {n} runs drift from the eager driver", flush=True)
