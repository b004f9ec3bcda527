"""The traversal stack's spill path (entries beyond the LDS part go to HBM) is rare with the
regular 16-entry LDS stack.  A build of the same library with a 4-entry LDS part makes nearly
every ray spill; the ray-casting parity tests must pass on it unchanged."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_parity_with_a_four_entry_lds_stack():
    lib = os.path.join(ROOT, "tests", "native", "_build", "libpt_hip_smallstack.so")
    assert os.path.exists(lib), "build it with `make native-tests`"
    env = dict(os.environ, PT_HIP_LIB=lib)
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_raycast.py"), "-m", "gpu", "-x", "-q",
                          "-k", "closest_hit or shadow_visibility or render_matches_oracle", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert " passed" in res.stdout and "failed" not in res.stdout
