"""The reference's own gtest scenarios (restated in tests/cpp/test_kba_shim.cpp) against the keyframe_bundle_adjustment
C++ shim (limo_amd/kba): CPU tier links the emulated C-ABI, GPU tier links liblimo_hip.so."""
import subprocess

import pytest

import emu_ffi


def run(exe):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0, "C++ shim tests failed:\n" + r.stdout[-4000:]
    assert "0 failed tests" in r.stdout


def test_reference_scenarios_with_emulated_backend():
    run(emu_ffi.build_shim_tests(gpu=False))


@pytest.mark.gpu
def test_reference_scenarios_on_gpu():
    run(emu_ffi.build_shim_tests(gpu=True))
