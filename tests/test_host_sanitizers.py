"""ASan + UBSan over the library's HOST code (csrc/plan.hip: plan upload / block recycling pool / re-pooling;
csrc/hierarchy.hip: the bi-stride builder, reference graph_wrappers/bsms_graph_wrapper.py:8-154): the sanitized build
(bsms-gnn_amd/build.py: build_host_sanitized) is driven by tests/helpers/host_sanitizer_driver.py in a torch-free process
under LD_PRELOAD of GCC's ASan runtime (g++ compiles the two host-only files as plain C++); any report fails the test.  CPU: the hierarchy builder and the error path of
bsms_plan_create.  GPU box (-m gpu): the whole plan life cycle incl. recycling and six concurrent host threads."""
import importlib.util
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT


def _build():
    spec = importlib.util.spec_from_file_location("_bsms_build", os.path.join(ROOT, "bsms-gnn_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rt = mod.asan_runtime()
    if rt is None:
        pytest.skip("GCC's shared ASan runtime (libasan.so) not found")
    if os.path.exists(mod.ASAN_LIB) and subprocess.run(["which", "g++"], capture_output=True).returncode != 0:
        return mod.ASAN_LIB, rt                         # prebuilt library travelled here, no compiler needed
    return mod.build_host_sanitized(), rt


def _run(extra):
    lib, rt = _build()
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "host_sanitizer_driver.py"), lib, *extra],
                         env=env, capture_output=True, text=True, timeout=550)
    report = out.stdout[-1500:] + "\n" + out.stderr[-4000:]
    assert out.returncode == 0, report
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error:" not in out.stderr, report
    return out.stdout


@pytest.mark.timeout(600)
def test_hierarchy_builder_under_asan_ubsan():
    out = _run([])
    assert "hierarchy:" in out and "no sanitizer report" in out


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_plan_life_cycle_under_asan_ubsan():
    assert torch.cuda.is_available()
    out = _run(["--gpu"])
    assert "plans: create / pool / re-pool" in out
