"""GPU: which of the two modes of the dense plain stencil apply (about 75 us or about 83 us at 256^3 fp32) does this
process get, and where do its arrays lie?  One line: the apply time, the unfused loop's iteration time, and the device addresses
(mod 2 MiB, in KiB) of the CG vectors and the engine's coefficient arrays.  usage: python tools/mode_probe.py [pad_kib]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
import bench
dev = torch.device("cuda:0")
pad_kib = int(sys.argv[1]) if len(sys.argv) > 1 else -1
junk = torch.empty(max(pad_kib, 0) * 1024 + 16, dtype=torch.uint8, device=dev) if pad_kib >= 0 else None   # shifts later allocations
gres = (256, 256, 256)
wx, wy, wz, lphi, (b, x, d, r, q) = bench.build_problem(torch, dev, torch.float32, gres, gres, 0, (0, 256))
from mfs.pcg import PcgEngine
eng = PcgEngine(gres, torch.float32, dev)
eng.setup(lphi, wx, wy, wz)
eng.bind(b, x, d, r, q)
eng.set_compress(False); eng.set_fuse(False)
eng.begin(0.0); eng.iterate(5); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
for a, e in ev:
    a.record(); eng.native_apply(); e.record(); eng.native_finish()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(e) for a, e in ev)
t0 = time.perf_counter(); eng.iterate(100); torch.cuda.synchronize(); it_us = (time.perf_counter() - t0) * 1e4
a0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng.apply(d, q); a0.record()
for _ in range(40): eng.apply(d, q)
e0.record(); torch.cuda.synchronize()
M = 1 << 21
addr = {n: (t.data_ptr() % M) // 1024 for n, t in (("b", b), ("x", x), ("d", d), ("r", r), ("q", q), ("ws", eng.workspace))}
print(json.dumps({"apply_us_in_loop_median": round(ts[20] * 1e3, 1), "apply_us_back_to_back": round(a0.elapsed_time(e0) / 40 * 1e3, 1),
                  "iteration_us_unfused_dense": round(it_us, 1), "addr_mod_2MiB_KiB": addr, "pad_kib": pad_kib}))
