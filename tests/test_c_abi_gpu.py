"""The C ABI from C: tests/c_abi/abi_{pressure,viscosity}_solve.c (plain C99, HIP runtime C API, no Python / torch in the process) run the
two hot paths -- the reference's `PressureCGSolver3D.solve` and `ViscosityCGSolver3D.solve` -- through libmfs_hip.so and check the CG
part against the oracle's C restatement linked into the same test program.  The CPU half of this file only builds it (the boundary really is `extern "C"`
with plain pointers: a C compiler accepts the header and the linker finds every symbol the program uses)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = {"pressure": os.path.join(REPO, "tests", "c_abi", "abi_pressure_solve.c"),
        "viscosity": os.path.join(REPO, "tests", "c_abi", "abi_viscosity_solve.c")}
LIBDIR = os.path.join(REPO, "python-fluid-simulation_amd", "mfs")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp_path, which="pressure"):
    sys.path.insert(0, REPO)
    import __graft_entry__ as G
    if not os.path.exists(os.path.join(LIBDIR, "libmfs_hip.so")):
        G.build()
    exe = str(tmp_path / f"abi_{which}_solve")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), "-I", os.path.join(ROCM, "include"), SRCS[which],
           os.path.join(REPO, "oracle", "mfs_oracle_c.c"), "-fopenmp", "-o", exe, "-L", LIBDIR, "-lmfs_hip", "-L", os.path.join(ROCM, "lib"),
           "-lamdhip64", "-lm", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{os.path.join(ROCM, 'lib')}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("which,symbols", [
    ("pressure", {"mfs_solid_frac3d", "mfs_pressure_rhs3d", "mfs_pcg3d_create", "mfs_pcg3d_setup", "mfs_pcg3d_bind", "mfs_pcg3d_solve",
                  "mfs_pcg3d_history", "mfs_pressure_update3d", "mfs_pcg3d_destroy", "mfs_last_error", "mfs_abi_version"}),
    ("viscosity", {"mfs_visc_extrapolate3d", "mfs_visc_extrapolate3d_workspace_bytes", "mfs_visc_rhs3d", "mfs_vcg3d_create", "mfs_vcg3d_setup",
                   "mfs_vcg3d_bind", "mfs_vcg3d_solve", "mfs_vcg3d_history", "mfs_visc_writeback3d", "mfs_vcg3d_destroy", "mfs_vcg3d_dofs"})])
def test_c_caller_builds_against_the_header_and_the_library(which, symbols, tmp_path):
    exe = _build(tmp_path, which)
    r = subprocess.run(["nm", "-u", exe], capture_output=True, text=True)
    used = {ln.split()[-1].split("@")[0] for ln in r.stdout.splitlines() if "mfs_" in ln}
    assert symbols <= used, used


@pytest.mark.gpu
@pytest.mark.parametrize("which,grid", [("pressure", (20, 24, 16)), ("pressure", (33, 17, 24)), ("pressure", (48, 80, 48)),
                                        ("viscosity", (12, 16, 20)), ("viscosity", (24, 20, 28)), ("viscosity", (48, 80, 48))],
                         ids=lambda v: v if isinstance(v, str) else "x".join(map(str, v)))
def test_c_caller_solves_and_agrees_with_the_oracle(which, grid, tmp_path):
    exe = _build(tmp_path, which)
    r = subprocess.run([exe] + [str(v) for v in grid], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.startswith("OK "), r.stdout
