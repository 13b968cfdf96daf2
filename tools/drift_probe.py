"""GPU: does the time per CG iteration of ONE engine drift over the first tens of seconds of a process (clock / power
state ramp of a box that was idle)?  256^3 fp32 pressure loop, 0.25 s batches, one line per batch.
usage: python tools/drift_probe.py [seconds] [idle_seconds_before_second_pass]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
import bench
from mfs.pcg import PcgEngine
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda:0"); g = (256, 256, 256)
t_start = time.perf_counter()
wx, wy, wz, lphi, (b, x, d, r, q) = bench.build_problem(torch, dev, torch.float32, g, g, 0, None)
eng = PcgEngine(g, torch.float32, dev)
eng.setup(lphi, wx, wy, wz); eng.bind(b, x, d, r, q)
eng.begin(0.0); eng.iterate(5); torch.cuda.synchronize()
def run(tag, secs):
    out = []
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        t0 = time.perf_counter(); eng.iterate(2000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out.append((round(t0 - t_start, 2), round(dt / 2000 * 1e6, 2)))
    print(json.dumps({"pass": tag, "us_per_iteration_by_time_since_start": out}))
run("first", secs)
if idle > 0:
    time.sleep(idle)
    run(f"after {idle} s idle", secs / 2)
