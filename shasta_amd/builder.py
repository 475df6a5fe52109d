"""Builders mirroring det3d/models/builder.py:20-75 for the registries this path uses.

`build_backbone` / `build_neck` return None for a None config: the spconv backbone and the RPN neck are
CenterPoint upstream code outside this package.  A user who has them can
register their classes in BACKBONES / NECKS and the Shasta module will call them exactly like the reference.
"""
from torch import nn

from .registry import BACKBONES, NECKS, READERS, SECOND_STAGE, TRACK, build_from_cfg


def build(cfg, registry, default_args=None):
    if cfg is None:
        return None
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_second_stage_module(cfg):
    return build(cfg, SECOND_STAGE)


def build_reader(cfg):
    return build(cfg, READERS)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_simp_track(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, TRACK, dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_track(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, TRACK, dict(train_cfg=train_cfg, test_cfg=test_cfg))
