"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/shasta_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "shasta_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(shasta_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from shasta_amd import hip
    lib = hip.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "library does not export " + n
        assert n in hip.SYMBOLS, "ctypes binding missing for " + n
    assert lib.shasta_abi_version() == hip.ABI_VERSION == 15
    assert b"gfx950" in lib.shasta_build_info()
    # the library carries the hash of the sources it was built from; hip.load() refuses a stale one
    from shasta_amd import build
    assert lib.shasta_build_info().decode().endswith("src " + build.source_hash())


def test_library_exports_exactly_the_declared_abi_and_reads_no_environment():
    """-fvisibility=hidden + the linker version script: the dynamic symbol table holds the declared C functions and nothing else
    (no kernel handles, no C++ helpers); kernel choices are per-call `options` bits, not getenv switches."""
    import subprocess
    from shasta_amd import hip
    out = subprocess.run(["nm", "-D", "--defined-only", hip.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported == _declared()
    und = subprocess.run(["nm", "-D", "--undefined-only", hip.lib_path()], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    csrc = os.path.join(ROOT, "shasta_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_size_queries_do_not_need_a_gpu():
    from shasta_amd import hip
    lib = hip.load()
    assert lib.shasta_packed_bytes(500, 7, 256) > 0
    assert lib.shasta_packed_bytes(500, 7, 128) == 0  # unsupported feature width is reported, not guessed
    assert lib.shasta_aug_shape_aux_bytes(500, 256, 0) >= 4 * 2000 * 4  # one maximum per first-layer weight row
    assert lib.shasta_aug_shape_aux_bytes(500, 256, 32) >= 4 * 2000 * 128000 * 4  # + the pre-cut piece image: 4 bytes per weight
    assert lib.shasta_forward_workspace_bytes(8, 500, 7, 256) > 8 * 502 * 504 * 4
    assert lib.shasta_voxelize_workspace_bytes(300000, 160000, 10) > 160000 * 10 * 4
    # ... + the cell -> first point hash table: 2^20 (key, index) pairs for 3e5 points - not the reference's 332 MB dense map
    assert lib.shasta_voxelize_workspace_bytes(300000, 160000, 10) < 160000 * 10 * 4 + 3 * 300000 * 4 + (1 << 20) * 8 + (1 << 16)


def test_missing_library_fails_loudly(monkeypatch):
    from shasta_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "_LIB_PATH", "/nonexistent/libshasta_hip.so")
    with pytest.raises(hip.ShastaHipError):
        hip.load()


def test_registry_builds_reference_style_config():
    import shasta_amd
    cfg = dict(type="Shasta", reader=dict(type="VoxelFeatureExtractorV3", num_input_features=5), backbone=None, neck=None,
               bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
               max_obj=20, num_feats=3, num_point=5)
    m = shasta_amd.build_simp_track(cfg, train_cfg=None, test_cfg=dict(x=1))
    assert m.test_cfg == dict(x=1)
    keys = list(m.state_dict().keys())
    assert keys[0] == "shared_conv.0.weight" and "aug_shape.3.2.bias" in keys and "aff.10.weight" in keys
    assert m.state_dict()["aug_shape.0.0.weight"].shape == (20 * 320 // 64, 20 * 320)
    assert m.state_dict()["aug_dets.0.0.weight"].shape == (140 // 32, 140)
    assert m.state_dict()["res_coeff.0.weight"].shape == (32 + 40, 2 * 320 + 6)
    assert [type(c).__name__ for c in m.children()][:3] == ["VoxelFeatureExtractorV3", "BEVFeatureExtractor", "Sequential"]
    with pytest.raises(KeyError):
        shasta_amd.build_simp_track(dict(cfg, type="NoSuchTracker"))
    with pytest.raises(KeyError):  # the spconv backbone is not part of this package: loud, not silent
        shasta_amd.build_simp_track(dict(cfg, backbone=dict(type="SpMiddleResNetFHD")))
