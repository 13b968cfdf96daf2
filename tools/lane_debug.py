"""GPU: the lane-mask flag of the pressure engine through one solve of a ball-in-a-box problem (diagnostics)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO, os.path.join(REPO, "tests")]
import torch
from mfs import _lib
from mfs.pcg import PcgEngine
import test_pressure_gpu as TP
dt = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else torch.float32
gres = (160, 96, 144)
lphi, wx, wy, wz, b = TP._blob_problem(gres, (50.0, 40.0, 60.0), 22.0, 1)
eng = PcgEngine(gres, dt, "cuda:0")
eng.setup(lphi.to(dt), wx.to(dt), wy.to(dt), wz.to(dt))
x, d, r, q = (torch.zeros(gres, dtype=dt, device="cuda:0") for _ in range(4))
eng.bind(b.to(dt), x, d, r, q)
eng.begin(1e-6)
torch.cuda.synchronize()
print("after begin", eng.scalars.cpu().numpy()[[4, 5, 11, 12, 13, 14, 15]], eng.sparse_info(), eng.loop_info())
eng.iterate(16)
print("poll", eng.poll())
print("after 16", eng.scalars.cpu().numpy()[[4, 5, 11, 12, 13, 14, 15]])
eng.iterate(16)
print("poll", eng.poll())
print("after 32", eng.scalars.cpu().numpy()[[4, 5, 11, 12, 13, 14, 15]])
