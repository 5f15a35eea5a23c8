"""the C ABI used from plain C (tests/c/abi_parity.c): compiled with gcc against include/gigl_hip.h, linked with
libgigl_hip.so and the C oracle, run on the GPU box.  No Python, torch or ctypes between the caller and the library —
what a cgo / JNI shim of the reference would do."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_matches_oracle(tmp_path):
    exe = str(tmp_path / "abi_parity")
    lib_dirs = [os.path.join(ROOT, "gigl_amd"), os.path.join(ROOT, "oracle"), "/opt/rocm/lib"]
    cmd = ["gcc", "-O2", "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests/c/abi_parity.c"),
           "-o", exe, "-L" + lib_dirs[0], "-l:libgigl_hip.so", "-L" + lib_dirs[1], "-l:libgigl_oracle.so",
           "-L" + lib_dirs[2], "-lamdhip64", "-Wl,-rpath," + ":".join(lib_dirs)]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, f"step {out.returncode}: {out.stderr}"
    assert "C ABI parity OK" in out.stdout
