"""profiles/rNN_pmc_apply.json (what bench.py quotes as roofline.traffic) from the two summaries tools/pmc_bench.sh wrote:
   python tools/make_pmc_json.py gpurun_out/r03_pmc_summary.json gpurun_out/r03_pmc_dense_summary.json profiles/r03_pmc_apply.json "<collected>" "<commit>"
The default run's dominant kernel is the fused stencil launch (FUSE, XDEF, BOOK); the dense run's (--dense-coefficients
--unfused) the plain stencil apply."""
import json, sys
main, dense, out, collected, commit = sys.argv[1:6]
m, d = json.load(open(main)), json.load(open(dense))
pick = lambda dd, pred: max(((k, v) for k, v in dd.items() if pred(k) and (v["launches"] or 0) >= 20), key=lambda kv: kv[1]["launches"])  # noqa: E731
km, vm = pick(m, lambda k: "k_pcg_apply_march" in k)
kd, vd = pick(d, lambda k: "k_pcg_apply_march" in k)
kx, vx = pick(m, lambda k: "k_update_xr" in k)
N = 256
alg_all = (10 * N ** 3 + 3 * N ** 2) * 4
alg_dense = (6 * N ** 3 + 3 * N ** 2) * 4
res = {"workload": "256x256x256 f32", "kernel": km, "hbm_bytes_per_launch": int(vm["hbm_bytes_per_launch"]),
       "algorithmic_bytes_all_cells": alg_all, "ratio_all_cells": round(vm["hbm_bytes_per_launch"] / alg_all, 4),
       "dense_kernel": kd, "dense_hbm_bytes_per_launch": int(vd["hbm_bytes_per_launch"]), "dense_algorithmic_bytes": alg_dense,
       "dense_ratio": round(vd["hbm_bytes_per_launch"] / alg_dense, 4),
       "update_xr_kernel": kx, "update_xr_hbm_bytes_per_launch": int(vx["hbm_bytes_per_launch"]),
       "collected": collected, "commit": commit,
       "command": "tools/pmc_bench.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py "
                  "--no-cpu-baseline --timed-loop-only [--dense-coefficients --unfused]",
       "method": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads; "
                 "MI355X_MICROARCH.md HBM section); per-kernel averages over the timed loop only (one configuration per run)",
       "note": "main = the bench's dominant kernel with the solve's sparse work list (the launch visits the listed (tile, plane) pairs only; "
               "bench.py prices it on those cells) and compressed coefficient access.  dense = plain stencil apply, every coefficient array "
               "read in full, every pair visited: vs the 6N^3+3N^2 scalars of SURVEY.md 8(d)",
       "all_kernels_default_run": m, "all_kernels_dense_run": d}
json.dump(res, open(out, "w"), indent=1)
print({k: res[k] for k in ("kernel", "hbm_bytes_per_launch", "ratio_all_cells", "dense_hbm_bytes_per_launch", "dense_ratio", "update_xr_hbm_bytes_per_launch")})
