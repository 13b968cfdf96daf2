"""Copies the outputs of tools/runs/profiles_r03.sh (gpurun_out/) into profiles/r03_*, each with a header saying what ran.
usage: python tools/collect_profiles_r03.py "<commit the GPU run was made from>" """
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(REPO, "gpurun_out"), os.path.join(REPO, "profiles")
commit = sys.argv[1] if len(sys.argv) > 1 else "working tree"


def last_json(name):
    for line in reversed(open(os.path.join(G, name), errors="replace").read().strip().splitlines()):
        if line.strip().startswith("{"):
            return json.loads(line)
    raise ValueError(name)


def copy(src, dst, header):
    body = open(os.path.join(G, src), errors="replace").read()
    body = "\n".join(l for l in body.splitlines() if "amdgpu.ids" not in l and not l.startswith("no counters"))
    open(os.path.join(P, dst), "w").write("".join(f"# {h}\n" for h in header) + f"# tree: {commit}\n" + body + "\n")


copy("r03_prof_summary.txt", "r03_bench_kernel_stats.txt",
     ["round 3: rocprofv3 --kernel-trace --stats of the default bench command (tools/prof_bench.sh): ALL legs of the run, so the",
      "stencil kernel's average mixes the timed loop (sparse work list) with the lists-off / dense / all-mixed legs --",
      "r03_bench_timed_loop_kernel_stats.txt holds the timed loop alone"])
copy("r03_prof_loop_summary.txt", "r03_bench_timed_loop_kernel_stats.txt",
     ["round 3: rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --timed-loop-only` (tools/prof_bench.sh): the two",
      "kernels of the timed CG loop; k_pcg_apply_march<..., FUSE, XDEF, BOOK> is the dominant kernel of bench.py's `roofline`"])
copy("r03_visc_kernel_stats_256.txt", "r03_visc_kernel_stats_256.txt",
     ["round 3: rocprofv3 kernel stats of tools/bench_viscosity.py 256 f32 100 (tools/prof_visc.sh): a whole solve (mu = 50), then",
      "100 timed loop iterations + 50 stand-alone applies.  k_vcg_apply_march<..., COMP=true>: the loop's launches run on the work",
      "list (~55 us), the stand-alone ones visit every pair (~110 us) -- the average mixes both.  k_update_xr<.., true, 1> = r update,",
      "k_update_d<.., true, true> = d = r + beta d with x += alpha d, both on the live 32-unknown chunks"])
body = open(os.path.join(G, "r03_visc_pmc_256_dense.txt"), errors="replace").read()
copy("r03_visc_pmc_256.txt", "r03_visc_pmc_256.txt",
     ["round 3: PMC counters of the viscosity CG kernels, 256^3 fp32 (tools/pmc_visc.sh: separate rocprofv3 --pmc passes over",
      "tools/bench_viscosity.py 256 f32 20; FETCH_SIZE x2 per the gfx950 note, KiB units; averages per launch).",
      "FIRST BLOCK: default engine (compressed class access, sparse lists).  SECOND BLOCK: MFS_VISC_COMPRESS=0 MFS_VISC_SPARSE=0",
      "(every class array read in full, every pair visited): the march moves 844 MB + 201 MB = 1.18 x its 889 MB"])
open(os.path.join(P, "r03_visc_pmc_256.txt"), "a").write("#\n# ---- MFS_VISC_COMPRESS=0 MFS_VISC_SPARSE=0 ----\n" +
                                                          "\n".join(l for l in body.splitlines() if "amdgpu.ids" not in l and not l.startswith("no counters")) + "\n")
copy("r03_ts256_prof.txt", "r03_ts256_prof.txt",
     ["round 3: GPU-busy time by kernel of the notebook's whole time step at 256^3, 16.8 M particles, fp32 state",
      "(MFS_PRECISION=fp32 tools/prof_total.sh r03ts tools/bench_timestep.py 256 2: set-up + 2 steps under rocprofv3)"])
subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "make_pmc_json.py"), os.path.join(G, "r03_pmc_summary.json"),
                       os.path.join(G, "r03_pmc_dense_summary.json"), os.path.join(P, "r03_pmc_apply.json"),
                       "round 3, final tree (sparse work list, live 32-cell chunks)", commit])
# the notebook's own scene
nb = last_json("r3_nbscene_final.log")
open(os.path.join(P, "r03_notebook_scene.txt"), "w").write(
    "# round 3: the notebook's own scene (code cell 9: 48x80x48 grid, 84 889 particles, mu = 1), tools/run_notebook_scene.py 100\n"
    f"# tree: {commit}\n# round 2 (r02_notebook_scene.txt): 15.0 ms per step (density 3.0, viscosity 6.7, pressure 3.8)\n"
    "# round 3: the viscosity CG loop resident (24.5 -> 11.1 us per iteration), one look at the scalar block per solve\n"
    + json.dumps(nb, indent=1) + "\n")
# time steps: the final lines appended to the table of the round
for name, lab in (("r3_ts256_f64_final.log", "final tree"), ("r3_ts128_final.log", "final tree")):
    d = last_json(name)
    with open(os.path.join(P, "r03_time_steps.txt"), "a") as f:
        f.write(f"{name:24s} {d['workload'][19:40]:22s} {d['state_precision']}  s_per_step {d['s_per_step']:<7}  stages ms {json.dumps(d['stage_ms_per_step'])}  "
                f"iterations(density,viscosity,pressure) {d['cg_iterations(density,viscosity,pressure)'][0]}   [{lab}: {commit}]\n")
print("ok")
