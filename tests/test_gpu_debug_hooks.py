"""Hooks of include/ccmp_debug.h on the GPU: the cases of tests/debug_hook_cases.py, run ONCE in a process of their own against
lib/libccmp_debug.so (one child process beside the suite's own; the default library, which every other test loads, exports none of
the hooks)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_debug_hook_cases_pass_on_the_debug_library(ccmp_built):
    from closed_chain_motion_planner_amd import _lib

    assert os.path.exists(_lib.DEBUG_LIBPATH), "lib/libccmp_debug.so is built beside lib/libccmp.so (closed_chain_motion_planner_amd/build.py)"
    env = dict(os.environ, CCMP_LIBRARY=_lib.DEBUG_LIBPATH)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "debug_hook_cases.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-500:]
