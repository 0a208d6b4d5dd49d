"""pyramid_kernel's phase code (r-vio_amd/csrc/pyr_sep.h) walked on the host, thread by thread, against the oracle's cv::pyrDown chain.

The kernel body is written as per-thread phases between workgroup barriers; tests/hostemu/pyr_emu.cpp compiles the SAME header with
g++ and runs blocks x phases x threads in plain loops (a barrier = the end of a thread loop).  What this pins on the CPU: the
separable pass pair, the reflect-101 index tables, the patch geometry and which workgroup stores which tile — at the sizes whose
tiles fit nothing (the GPU suite repeats them on the device: tests/test_gpu_frontend.py, tests/test_gpu_detector.py).  The LDS image
is poisoned with two different patterns: a read of a byte the workgroup never wrote would change the result.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostemu", "pyr_emu.cpp")
HDR = os.path.join(HERE, "..", "r-vio_amd", "csrc", "pyr_sep.h")
LIB = os.path.join(HERE, "hostemu", "libpyr_emu.so")
up = C.POINTER(C.c_ubyte)


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-Wall", SRC, "-o", LIB])
    return C.CDLL(LIB)


def run(emu, img, stride_pad, levels, copy0, poison, reverse=0):
    h, w = img.shape
    stride = w + stride_pad
    buf = np.full((h, stride), 0xA5, np.uint8)
    buf[:, :w] = img
    outs, lw, lh = [], w, h
    for _ in range(4):
        outs.append(np.full((lh, lw), 0x5A, np.uint8))
        lw, lh = (lw + 1) // 2, (lh + 1) // 2
    lds = emu.pyr_emulate(buf.ctypes.data_as(up), w, h, stride, levels, copy0, *[o.ctypes.data_as(up) for o in outs], poison, reverse)
    assert lds <= 20 * 1024          # eight workgroups per CU
    return outs


SIZES = [(752, 480), (200, 136), (757, 483), (1000, 562), (376, 240), (65, 65), (64, 64), (129, 71), (33, 17), (1920, 1080)]


@pytest.mark.parametrize("w,h", SIZES)
def test_phases_equal_the_pyrdown_chain(emu, w, h):
    rng = np.random.default_rng(w * 10007 + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ref = [img]
    for _ in range(3):
        ref.append(O.pyr_down(ref[-1]))
    for poison, reverse in ((0x00, 0), (0xFF, 1)):   # second walk: threads in descending order (no phase may depend on the order)
        outs = run(emu, img, stride_pad=5 if w % 2 else 0, levels=4, copy0=1, poison=poison, reverse=reverse)
        for lv in range(4):
            assert outs[lv].shape == ref[lv].shape
            assert np.array_equal(outs[lv], ref[lv]), "level %d at %dx%d (poison %#x): %d pixels differ" % (
                lv, w, h, poison, int((outs[lv] != ref[lv]).sum()))


@pytest.mark.parametrize("levels", [1, 2, 3])
def test_fewer_levels_and_no_copy(emu, levels):
    """Tracker.nPyramidLevel < 3; copy0 = 0 is the equalised frame that already is level 0 (EnableEqualizer: 1)"""
    rng = np.random.default_rng(levels)
    img = rng.integers(0, 256, (203, 315), dtype=np.uint8)
    ref = [img]
    for _ in range(3):
        ref.append(O.pyr_down(ref[-1]))
    outs = run(emu, img, 3, levels, 0, 0x77)
    assert (outs[0] == 0x5A).all()                       # not written: the caller's image is level 0
    for lv in range(1, 4):
        if lv < levels:
            assert np.array_equal(outs[lv], ref[lv])
        else:
            assert (outs[lv] == 0x5A).all()              # levels the configuration does not have are not touched


def test_saturated_images(emu):
    """the u16 column sums at their maximum (255 * 16 = 4080) and the rounding of (v + 128) >> 8 at both ends"""
    for v in (0, 255):
        img = np.full((97, 131), v, np.uint8)
        outs = run(emu, img, 0, 4, 1, 0x33)
        for o in outs:
            assert (o == v).all()
