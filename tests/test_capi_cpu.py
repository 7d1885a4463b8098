"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol
include/take_hip.h declares, its structs have the layout the Python binding assumes, and — with no GPU in the
process — it refuses to work instead of falling back to a CPU path.  (No compute calls here.)"""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from take_amd import capi
from take_amd import cdefs as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "take_hip.h")


@pytest.fixture(scope="module")
def lib():
    capi.build()
    return capi.lib()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(take_hip_[a-z_]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/take_hip.h but not exported"
    assert sorted(capi.EXPORTS) == names


def test_abi_version(lib):
    assert lib.take_hip_abi_version() == 5


def test_struct_layout_matches_header():
    fields = {
        "TakeTexture": D.TakeTexture, "TakeMaterial": D.TakeMaterial, "TakeImage3": D.TakeImage3,
        "TakeMesh": D.TakeMesh, "TakeSphere": D.TakeSphere, "TakeLight": D.TakeLight, "TakeCamera": D.TakeCamera,
        "TakeSceneDesc": D.TakeSceneDesc, "TakeBuildOpts": D.TakeBuildOpts, "TakeRenderOpts": D.TakeRenderOpts,
        "TakeRayF": D.TakeRayF, "TakeRayD": D.TakeRayD, "TakeHitF": D.TakeHitF, "TakeHitD": D.TakeHitD,
        "TakeCounters": D.TakeCounters, "TakeInstance": D.TakeInstance, "TakePlyLayout": D.TakePlyLayout,
    }
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "take_hip.h"', "int main(void){"]
    for n, cls in fields.items():
        prog.append(f'printf("{n} %zu\\n", sizeof({n}));')
        for fname, _ in cls._fields_:
            prog.append(f'printf("{n}.{fname} %zu\\n", offsetof({n}, {fname}));')
    prog.append("return 0;}")
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write("\n".join(prog))
        exe = os.path.join(td, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, stdout=subprocess.PIPE, text=True).stdout
    want = dict(line.split() for line in out.strip().splitlines())
    for n, cls in fields.items():
        assert C.sizeof(cls) == int(want[n]), n
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == int(want[f"{n}.{fname}"]), f"{n}.{fname}"


def test_no_gpu_means_error_not_fallback(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the no-GPU contract is checked in the CPU container")
    rc = lib.take_hip_device_count()
    assert rc == -3  # TAKE_E_NO_GPU
    assert b"no CPU path" in lib.take_hip_last_error()


def test_scene_create_without_gpu_raises(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from helpers import golden_scene

    with pytest.raises(capi.TakeError) as e:
        capi.Scene(golden_scene("cbox"))
    assert e.value.code == -3


def test_group_and_egress_entry_points_without_gpu(lib):
    """the newer entry points keep the contract: no HIP device -> TAKE_E_NO_GPU, never a fallback"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from helpers import golden_scene

    with pytest.raises(capi.TakeError) as e:
        capi.SceneGroup(golden_scene("cbox"), [0, 0])
    assert e.value.code == -3
    buf = (C.c_uint16 * 16)()
    rc = lib.take_hip_pack_exr_scanlines(C.cast(buf, C.c_void_p), 0, 2, 2, C.cast(buf, C.c_void_p), None)
    assert rc == -3


def test_product_never_imports_the_oracle():
    """the product package must not import, link or execute anything under oracle/ (only tests, smoke and
    bench.py's cpu_baseline leg may)"""
    pkg = os.path.join(ROOT, "take_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), fn
                assert "take_oracle" not in txt and "libtake_oracle" not in txt and "hostsim" not in txt.replace(
                    "tests/hostsim", ""), fn
