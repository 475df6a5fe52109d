"""`build_*` entry points of det3d/models/builder.py:20-75 for the registries this path uses.

A None config builds nothing (returns None): the spconv backbone and the RPN neck are CenterPoint upstream code outside
this package; a user who has them registers their classes in BACKBONES / NECKS and the
Shasta module calls them exactly like the reference.  A list of configs becomes an nn.Sequential, like the reference."""
from torch import nn

from . import registry as R


def build(cfg, registry, default_args=None):
    if cfg is None:
        return None
    if isinstance(cfg, (list, tuple)):
        return nn.Sequential(*(R.build_from_cfg(one, registry, default_args) for one in cfg))
    return R.build_from_cfg(cfg, registry, default_args)


def _plain(registry):
    return lambda cfg: build(cfg, registry)


def _with_run_cfgs(registry):
    return lambda cfg, train_cfg=None, test_cfg=None: build(cfg, registry, {"train_cfg": train_cfg, "test_cfg": test_cfg})


build_reader = _plain(R.READERS)
build_backbone = _plain(R.BACKBONES)
build_neck = _plain(R.NECKS)
build_second_stage_module = _plain(R.SECOND_STAGE)
build_simp_track = _with_run_cfgs(R.TRACK)
build_track = _with_run_cfgs(R.TRACK)
