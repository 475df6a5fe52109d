"""Import helpers for the upstream reference at /root/reference (build container ONLY).

Nothing here runs on the GPU box: /root/reference does not exist there.  This module is
used by make_golden.py (fixture generation) and by the optional `-m ref` cross-checks.
It never copies reference source; it imports it in place with the absent third-party
packages stubbed.
"""
import importlib
import importlib.machinery
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("SHASTA_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True

_STUBS = ["torchvision", "torchvision.models", "terminaltables", "spconv", "addict", "numba",
          "pycocotools", "pycocotools.mask", "filterpy", "filterpy.kalman", "shapely",
          "shapely.geometry"]


class _Permissive:
    """Callable dummy: `@stub.jit(nopython=True)` and bare `@stub.jit` become identities."""

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Permissive()

    def __mro_entries__(self, bases):
        return (object,)


def _install_stubs():
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            if importlib.util.find_spec(name) is not None:
                continue
        except (ImportError, ValueError, AttributeError):
            pass
        m = types.ModuleType(name)
        m.__path__ = []
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)

        def _ga(attr, _n=name):
            if attr.startswith("__"):
                raise AttributeError(attr)
            return _Permissive()

        m.__getattr__ = _ga
        sys.modules[name] = m


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "det3d"))


def import_shasta():
    """Returns the reference module det3d.models.tracker.shasta with builders neutralised."""
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    S = importlib.import_module("det3d.models.tracker.shasta")
    S.builder.build_reader = S.builder.build_backbone = S.builder.build_neck = lambda cfg: None
    return S


def build_ref_model(max_obj, num_feats, num_point, share_conv_channel=64, in_channels=512,
                    out_stride=8):
    S = import_shasta()
    m = S.Shasta(reader=None, backbone=None, neck=None,
                 bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54],
                                    voxel_size=[0.075, 0.075], out_stride=out_stride),
                 max_obj=max_obj, num_feats=num_feats, num_point=num_point,
                 share_conv_channel=share_conv_channel, in_channels=in_channels).eval()
    return m


def import_point_cloud_ops():
    _install_stubs()
    p = os.path.join(REF_ROOT, "det3d/ops/point_cloud/point_cloud_ops.py")
    spec = importlib.util.spec_from_file_location("_ref_point_cloud_ops", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_voxel_reader():
    S = import_shasta()
    return importlib.import_module("det3d.models.readers.voxel_encoder")


def import_pub_tracker():
    p = os.path.join(REF_ROOT, "tools/nusc_shasta")
    if p not in sys.path:
        sys.path.insert(0, p)
    return importlib.import_module("pub_tracker")


def import_association():
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return importlib.import_module("mot_3d.association")
