"""The layout in blocks (csrc/rt_lay.h) as far as the host decides it: the
block plan of rt_reserve, ray -> address, and the block segments that
downloads, uploads and the gather walk -- tests/hostemu/blocks_host.cpp,
compiled with hipcc, runs on the CPU (the kernels' side: tests/test_blocks_gpu.py
and the GPU suite under RT_MI355_BLOCK_RAYS)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_block_plan_addresses_and_segments(tmp_path):
    exe = str(tmp_path / "blocks_host")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1",
                           "-std=c++17", "-o", exe,
                           os.path.join(ROOT, "tests", "hostemu",
                                        "blocks_host.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "0 failures" in out.stdout, out.stdout
