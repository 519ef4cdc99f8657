"""The library's host-only units (csrc/mesh_host.hip: reader / writer / face graph / permutation / partition / halo plans;
csrc/ordering.hip) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (tools/sanitize/run.sh): the reference's
2-D mesh and a tetrahedral box through every entry point, 120 mutated file sets through the reader (accepted or rejected:
never a crash, an overflow, a leak), random permutations and partitions with empty ranks.  GPU sanitizers do not exist on
this pool; this is the part of the library that parses files somebody else wrote."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sanitizers_link(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text("int main() { return 0; }\n")
    return shutil.which("g++") is not None and subprocess.run(
        ["g++", "-fsanitize=address,undefined", str(src), "-o", str(tmp_path / "t")], capture_output=True).returncode == 0


def test_host_mesh_code_is_clean_under_asan_and_ubsan(tmp_path):
    if not _sanitizers_link(tmp_path):
        pytest.skip("g++ with libasan / libubsan is not available here")
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize", "run.sh"), str(tmp_path / "work"), "120", "3"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "sanitizers: clean" in p.stdout and "accepted" in p.stdout
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr and "LeakSanitizer" not in p.stderr, p.stderr[-4000:]
