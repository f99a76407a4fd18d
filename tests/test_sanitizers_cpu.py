"""AddressSanitizer + UBSan over the host side of the C ABI and the host-emulated kernels (SURVEY section 5 hook):
the N=40 solver kernel -- tables, condensation, Riccati sweep with the blocked elimination, line search, write-out --
runs a few interior-point iterations under the sanitizers in a child process.  Any out-of-bounds access to the
emulated LDS object or the workspace, any undefined shift / overflow, and any __syncthreads() that part of the block
never reaches (the emulation aborts on it) fails the test.  GPU sanitizers are not available on the pool."""
import os
import subprocess
import sys

from conftest import ROOT

PKG = os.path.join(ROOT, "landing-controller_amd")
CHILD = r"""
import sys, importlib, numpy as np
sys.path.insert(0, %r)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, K = 40, 2
P, X0, _, _ = problem.make_batch(2, N, 0.6, seed=3)
L = capi.LandingLib(N, lib_path=%r)
o = L.default_opts(); o.max_iter = K
g = L.solve_host(P, X0, o)
assert g["iters"].tolist() == [2 * K, 2 * K], g["iters"]      # K interior-point iterations, then K of the feasibility phase (its elastic row passes run under the sanitizers too)
assert np.isfinite(g["x"]).all() and np.isfinite(g["kkt"]).all()
e = L.eval_host(X0, P, np.ones(2), np.ones((2, L.ng)))
assert all(np.isfinite(v).all() for v in e.values())
L.close()
print("SANITIZED-OK")
"""


def test_emulated_solver_under_asan_ubsan():
    lib = os.path.join(ROOT, "tests", "emu", "liblanding_emu_asan.so")
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu-asan"], check=True, capture_output=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, lib)], capture_output=True, text=True, env=env, timeout=280)
    err = "\n".join(l for l in r.stderr.splitlines() if "doesn't fully support makecontext" not in l)
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, r.stdout[-2000:] + err[-4000:]
    assert "runtime error" not in err and "AddressSanitizer" not in err, err[-4000:]
