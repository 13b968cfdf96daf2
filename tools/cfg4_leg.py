"""GPU: bench.py's config4_rank_share leg alone (one rank's slab of the 512^3 problem through the window loop on a 1-rank
window) -- for rocprofv3 (tools/prof_cmd.sh cfg4 50 tools/cfg4_leg.py)."""
import argparse, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "python-fluid-simulation_amd"), REPO]
import torch
import bench
args = argparse.Namespace(dtype="f32", steps=200, warmup=10)
out = bench.config4_rank_share_leg(args, torch, torch.device("cuda:0"), 0, 0.0)
print(json.dumps({k: v for k, v in out.items() if k != "parity_check"}))
