"""The built library's gfx950 code holds neither of the two hazards the compiler cannot see across an `asm` statement
(tools/asm_hazard_scan.py; round 4 met both in the resident path: a granule stored with a clobbered tag, a polled load
through a base register still being restored) -- and the scanner does see them in a kernel written to have them."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import asm_hazard_scan  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"
needs_tools = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(os.path.join(asm_hazard_scan.LLVM, "llvm-objdump"))),
                                 reason="needs hipcc and llvm-objdump")

BAD = r"""
#include <hip/hip_runtime.h>
__global__ void store_then_clobber(float4 *p) {
  asm volatile("v_mov_b32 v2, 1.0\n\tv_mov_b32 v3, 1.0\n\tv_mov_b32 v4, 1.0\n\tv_mov_b32 v5, 1.0\n\ts_nop 4\n\t"
               "global_store_dwordx4 v[0:1], v[2:5], off sc1\n\ts_nop 0\n\tv_mov_b32 v3, 0" ::: "v0", "v1", "v2", "v3", "v4", "v5", "memory");
}
__global__ void restore_then_load(float4 *p) {
  asm volatile("v_readlane_b32 s8, v9, 0\n\tv_readlane_b32 s9, v9, 1\n\ts_nop 2\n\t"
               "global_load_dwordx4 v[2:5], v1, s[8:9] sc1\n\ts_waitcnt vmcnt(0)" ::: "v1", "v2", "v3", "v4", "v5", "v9", "s8", "s9", "memory");
}
__global__ void both_kept_apart(float4 *p) {
  asm volatile("global_store_dwordx4 v[0:1], v[2:5], off sc1\n\ts_nop 1\n\tv_mov_b32 v3, 0\n\t"
               "v_readlane_b32 s8, v9, 0\n\tv_readlane_b32 s9, v9, 1\n\ts_nop 4\n\t"
               "global_load_dwordx4 v[2:5], v1, s[8:9] sc1\n\ts_waitcnt vmcnt(0)" ::: "v0", "v1", "v2", "v3", "v4", "v5", "v9", "s8", "s9", "memory");
}
"""


def _scan(path):
    """(exit code, report) of the scanner run on a library or object file."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_hazard_scan.py"), path], capture_output=True, text=True, timeout=600)
    return r.returncode, r.stdout


@needs_tools
def test_scanner_sees_a_store_data_hazard_and_a_restored_base_hazard(tmp_path):
    src = tmp_path / "bad.hip"
    src.write_text(BAD)
    obj = tmp_path / "bad.o"
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-c", str(src), "-o", str(obj)], check=True, capture_output=True, timeout=600)
    rc, out = _scan(str(obj))
    lines = [ln for ln in out.splitlines() if ln.startswith("hazard")]
    assert rc == 1
    assert sum("hazard A" in ln and "store_then_clobber" in ln for ln in lines) == 1, out
    assert sum("hazard B" in ln and "restore_then_load" in ln for ln in lines) == 2, out  # (one per half of the base)
    assert not any("both_kept_apart" in ln for ln in lines), out


@needs_tools
def test_built_library_is_free_of_both_hazards():
    lib = os.path.join(ROOT, "stormruler_amd", "libstorm_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    rc, out = _scan(lib)
    assert rc == 0, out
    assert "0 hazards" in out and shutil.which("python3") is not None
