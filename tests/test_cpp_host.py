"""The C++ host layer (bevyray_amd/host/raytracing.hpp + demo_main.cpp): builds with plain g++
against the C ABI; without a GPU it must fail loudly, with one its frame must equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

import bevyray_amd as brt
from bevyray_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "bevyray_amd", "host", "bevyray_demo")


def _build():
    _lib.build()
    proc = subprocess.run(["make", "-C", os.path.join(ROOT, "bevyray_amd", "csrc"), "host"], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    return DEMO


def test_cpp_host_builds_and_fails_loudly_without_gpu():
    import torch
    demo = _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    proc = subprocess.run([demo, "32", "18", "1", "2"], capture_output=True, text=True)
    assert proc.returncode == 3 and "no CPU path" in proc.stderr


@pytest.mark.gpu
def test_cpp_host_frame_matches_oracle(tmp_path, oracle):
    demo = _build()
    out = tmp_path / "frame.bin"
    w, h, spp, bounces = 200, 112, 2, 4
    proc = subprocess.run([demo, str(w), str(h), str(spp), str(bounces), "0.5", str(out)], capture_output=True, text=True,
                          env={**os.environ, "BRT_NO_TORCH": "1"})
    assert proc.returncode == 0, proc.stdout + proc.stderr
    got = np.fromfile(out, np.float32).reshape(h, w, 4)
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = brt.cover_camera(w, h, spp, bounces, brt.Raytracing.Pure, 0.5)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert f"{cnt['rays']} rays" in proc.stdout
