"""include/AudioSDR_hip.hpp -- the header-only `class AudioSDR : public AudioStream` drop-in over the C ABI -- compiles
against an application-side AudioStream.h (a test-only mock of the HOST APPLICATION's header, tests/mock/) and behaves
like the reference object: setters once, update() per block, missing-input guard (AudioSDR.cpp:48-56)."""
import os
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
EXE = os.path.join(ROOT, "tests", "mock", "adaptor_main")


EXE2 = os.path.join(ROOT, "tests", "mock", "frontend_main")


def _build(A, src="adaptor_main.cpp", exe=EXE):
    lib_dir = os.path.dirname(A.library_path())
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "mock"),
           os.path.join(ROOT, "tests", "mock", src), "-o", exe, "-L", lib_dir, "-lasdr_hip", "-Wl,-rpath," + lib_dir,
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_adaptor_compiles_and_serves_the_control_plane(A):
    _build(A)
    out = subprocess.run([EXE, "-1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "offset 5390.0 lower 5390.0 upper 8390.0 mode 1 agc 1" in out.stdout


@pytest.mark.gpu
def test_adaptor_update_on_gpu(gpu):
    _build(gpu)
    out = subprocess.run([EXE, "0"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "released 17 guard 1" in out.stdout      # 8 updates x 2 releases + the guard's single release


def test_frontend_adaptors_compile_and_serve_the_control_plane(A):
    """include/AudioSDRlib_hip.hpp: AudioSDRpreProcessor / AudioIQgenerator / AudioGrabberComplex256 by their
    reference names; setI2SerrorCompensation cancels auto-detection (AudioSDRpreProcessor.cpp:160-163)."""
    _build(A, "frontend_main.cpp", EXE2)
    out = subprocess.run([EXE2, "-1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "auto 1 corr 0" in out.stdout and "auto 0 corr 1 new 0" in out.stdout


@pytest.mark.gpu
def test_frontend_adaptor_graph_on_gpu(gpu):
    _build(gpu, "frontend_main.cpp", EXE2)
    out = subprocess.run([EXE2, "0"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "grabs 4" in out.stdout and "released 16 16 16" in out.stdout
