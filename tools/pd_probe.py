"""GPU: same-process A/B of the stencil kernel's grid size (workgroups per CU, mfs_pcg3d_tune) on the bench workload -- the timed
loop (fused, compressed, sparse lists), the dense plain apply inside the unfused loop, fp32 and fp64 state.
usage: python tools/pd_probe.py [f32|f64]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
import bench
from mfs.pcg import PcgEngine
dev = torch.device("cuda:0")
tdt = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else torch.float32
gres = (256, 256, 256)
wx, wy, wz, lphi, (b, x, d, r, q) = bench.build_problem(torch, dev, tdt, gres, gres, 0, (0, 256))
eng = PcgEngine(gres, tdt, dev)
eng.setup(lphi, wx, wy, wz)
eng.bind(b, x, d, r, q)


def loop_us(n=200):
    eng.begin(0.0); eng.iterate(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.iterate(n); torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e6, 2)


def apply_us():
    eng.begin(0.0); eng.iterate(5); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, e in ev:
        a.record(); eng.native_apply(); e.record(); eng.native_finish()
    torch.cuda.synchronize()
    return round(sorted(a.elapsed_time(e) for a, e in ev)[20] * 1e3, 1)


res = {"dtype": str(tdt)}
for rep in range(2):
    for bpc in (1, 2, 3, 4):
        eng.tune(2, 0, bpc, -1)
        eng.set_compress(True); eng.set_fuse(True); eng.set_sparse(True)
        res.setdefault(f"timed_loop_us_bpc{bpc}", []).append(loop_us())
        res.setdefault(f"fused_launch_us_bpc{bpc}", []).append(apply_us())
        eng.set_sparse(False)
        res.setdefault(f"loop_lists_off_us_bpc{bpc}", []).append(loop_us())
        eng.set_compress(False); eng.set_fuse(False)
        res.setdefault(f"dense_plain_apply_us_bpc{bpc}", []).append(apply_us())
print(json.dumps(res))
