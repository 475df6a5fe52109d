"""`Shasta`: the affinity network with the reference's model API, computed by hand-written HIP kernels.

Mirror of det3d/models/tracker/shasta.py:9-327 (`Shasta(BaseTrack)`):
  * same constructor keywords (:11-25), same sub-module creation ORDER (so a seeded default init gives the same
    weights and `children()` indices 1,2 are backbone, neck as tools/nusc_shasta/train.py:184-191 expects),
    same `state_dict` key names and shapes (:42-106);
  * `forward(example, train_mode=True) -> (matched1 (B,N,N+2), matched2 (B,N+2,N), example)` with the reference's
    side effects: `example["det_boxes"][:, :, :2]` back-projected in place (:216,270), `example["bev_feature"]`
    set to the NHWC map after shared_conv (:224), `self.newborn/fp/dead_trk/fn` anchor boxes (:260-267).
The nn.Linear modules here only HOLD the parameters; every compute step of rows 4-16 of SURVEY.md 8(a) runs through
the C ABI (include/shasta_hip.h).  There is no PyTorch/CPU fallback: without the built library, or with CPU
tensors, forward raises.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import builder, hip
from .registry import TRACK


# ---- pointer-cache invalidation: torch calls these hooks whenever a Parameter / sub-module is (re-)registered on ANY module ---------
_OWNER = weakref.WeakKeyDictionary()  # sub-module of a Shasta -> weakref to that Shasta
_HOOKED = []


def _on_register(module, name, value):
    ref = _OWNER.get(module)
    owner = ref() if ref is not None else None
    if owner is not None:
        owner._wstruct = None
    return None


def _watch(model):
    """(Re-)enrol every sub-module of `model`; install the two global hooks once."""
    if not _HOOKED:
        from torch.nn.modules import module as M
        _HOOKED.append(M.register_module_parameter_registration_hook(_on_register))
        _HOOKED.append(M.register_module_module_registration_hook(_on_register))
    ref = weakref.ref(model)
    for sub in model.modules():
        _OWNER[sub] = ref


class BaseTrack(nn.Module):
    """det3d/models/tracker/base.py:11-72 (the properties Shasta uses)."""

    def __init__(self):
        super().__init__()
        self.fp16_enabled = False

    @property
    def with_reader(self):
        return hasattr(self, "reader") and self.reader is not None

    @property
    def with_neck(self):
        return hasattr(self, "neck") and self.neck is not None


@TRACK.register_module
class Shasta(BaseTrack):
    def __init__(self, reader, backbone, neck, bev_extractor, train_cfg=None, test_cfg=None, pretrained=None,
                 max_obj=100, num_feats=7, in_channels=512, share_conv_channel=64, num_point=5):
        super().__init__()
        self.reader = builder.build_reader(reader)
        self.backbone = builder.build_backbone(backbone)
        self.neck = builder.build_neck(neck)
        self.bev_extractor = builder.build_second_stage_module(bev_extractor)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.init_weights(pretrained=pretrained)

        self.num_feats = num_feats
        self.max_obj = max_obj
        self.num_point = num_point

        self.shared_conv = nn.Sequential(
            nn.Conv2d(in_channels, share_conv_channel, kernel_size=3, padding=1, bias=True),
            nn.BatchNorm2d(share_conv_channel),
            nn.ReLU(inplace=True),
        )
        self.aug_shape = nn.ModuleList()
        self.aug_shape_input = max_obj * share_conv_channel * num_point
        self.aug_shape_output = share_conv_channel * num_point
        for _ in range(4):
            self.aug_shape.append(nn.Sequential(
                nn.Linear(self.aug_shape_input, self.aug_shape_input // 64),
                nn.ReLU(inplace=True),
                nn.Linear(self.aug_shape_input // 64, self.aug_shape_output),
            ))
        F = self.aug_shape_output
        self.fuse_shape = nn.Sequential(
            nn.Linear(2 * F, F // 8), nn.ReLU(inplace=True),
            nn.Linear(F // 8, F // 16), nn.ReLU(inplace=True),
            nn.Linear(F // 16, F // 32), nn.ReLU(inplace=True),
            nn.Linear(F // 32, 1),
        )
        self.aug_input = max_obj * 7
        self.aug_dets = nn.ModuleList()
        for _ in range(4):
            self.aug_dets.append(nn.Sequential(
                nn.Linear(self.aug_input, self.aug_input // 32),
                nn.ReLU(inplace=True),
                nn.Linear(self.aug_input // 32, 7),
            ))
        self.fuse_det = nn.Sequential(
            nn.Linear(self.num_feats * 2, 32), nn.ReLU(inplace=True),
            nn.Linear(32, 8), nn.ReLU(inplace=True),
            nn.Linear(8, 1),
        )
        self.res_coeff = nn.Sequential(
            nn.Linear(self.num_feats * 2 + F * 2, 32 + F // 8), nn.ReLU(inplace=True),
            nn.Linear(32 + F // 8, 8 + F // 32), nn.ReLU(inplace=True),
            nn.Linear(8 + F // 32, 3),
        )
        self.aff = nn.Sequential(
            nn.Linear(max_obj + 2, 128), nn.ReLU(inplace=True),
            nn.Linear(128, 64), nn.ReLU(inplace=True),
            nn.Linear(64, 32), nn.ReLU(inplace=True),
            nn.Linear(32, 64), nn.ReLU(inplace=True),
            nn.Linear(64, 128), nn.ReLU(inplace=True),
            nn.Linear(128, max_obj + 2),
        )
        self.softmax1 = nn.Softmax(dim=2)
        self.softmax2 = nn.Softmax(dim=1)

        # HIP-side state (not parameters, not in state_dict)
        self._packed = None
        self._packed_key = None
        self._aux = None
        self._aux_key = None
        self._guard_key = None      # weight set (pointers, versions) for which the range guard refused the fp16 weight stream
        self.f16x2_guard = None     # what _ensure_aux measured and decided, once a companion has been built
        self._conv_packed = None
        self._conv_key = None
        self._conv_bank = None
        self._wstruct = None
        self._plist = None
        self._bufs = {}
        self._graph_bufs = []
        # How fp32 products are formed on the matrix cores (shasta_weights.options, include/shasta_hip.h); fp32 operands in HBM and
        # fp32 accumulation in every mode:
        #   "f16x2" (default): as "pieces", but the aug_shape first layer (the 4.1 GB weight stream) cuts every operand into TWO
        #            fp16 pieces, rounded to nearest and range-scaled per row, and forms three products per fp32 product;
        #   "pieces": above 32 frame-pairs per call the aug_shape first layer and, from 8192 table rows, the row-embedding GEMMs
        #            and the aff layers use three exact bf16 pieces per operand, six products per fp32 product;
        #   "f32":   the f32 MFMA kernels everywhere.
        #   "f16grid" (opt-in, NOT fp32-equivalent): as "f16x2", but at F = 256 and from 8192 table rows the pair kernel takes the fp16
        #            pieces of its hidden activations from a fixed grid per MLP (22 bits of the tile's largest sum): a third fewer
        #            vector instructions, errors of `residual` about 2x (max) / 5x (rms) those of the fp32 kernels, 1e-6 of its range.
        self.arithmetic = "f16x2"
        # "f16x2" only: keep the first aug_shape layers ALSO as pre-cut fp16 pieces (+4 bytes per weight resident: 4.1 GB at N=500, built
        # lazily with the row maxima for the first forward of more than 64 frame-pairs) and stream those instead of cutting the fp32
        # tensors inside the kernel (SHASTA_OPT_PRECUT_WEIGHT_STREAM): bit-identical results, weight-stream kernel 2.75 -> 2.56 ms at
        # 512 frame-pairs.  False keeps only the fp32 checkpoint tensors resident.
        self.precut_weight_stream = True
        # train() mode: shared_conv forward (batch-statistics BatchNorm) and backward by the hand-written kernels (shared_conv_train.py);
        # False keeps the module's own nn.Sequential (MIOpen / ATen + autograd), e.g. to time the two against each other
        self.hand_written_train_conv = True
        # training backward (shasta_amd/training.py): "fp32" (parity path, like the reference's train.py:149) or "bf16": the GEMMs of
        # aff and of the pair MLPs' first-layer tables take bf16 operands with fp32 accumulation (BASELINE config 5's reduced-precision
        # option); the pair MLPs' later layers run per pair on chip in fp32 either way, unless dense_pair_backward keeps the round-4
        # formulation (every pair's hidden activations in HBM, strided GEMMs - with bf16 operands under the option)
        self.train_precision = "fp32"
        self.dense_pair_backward = False
        self.keep_intermediates = False  # tests: also return residual / matched via self.last_intermediates
        self.last_intermediates = None

    # ---- reference API -------------------------------------------------------------------------------------
    def init_weights(self, pretrained=None):
        """shasta.py:111-119 swallows every error; here a missing/broken checkpoint is reported loudly but, like the
        reference, does not abort construction."""
        if pretrained is None:
            return
        try:
            checkpoint = torch.load(pretrained, map_location="cpu")
            sd = checkpoint.get("state_dict", checkpoint) if isinstance(checkpoint, dict) else checkpoint
            load_state_dict_permissive(self, sd)
            print("init weight from {}".format(pretrained))
        except Exception as e:  # noqa: BLE001
            print("no pretrained model at {} ({})".format(pretrained, e))

    def extract_feat(self, data):
        """shasta.py:164-210.  With a backbone/neck registered by the user the call sequence is the reference's.
        Without them (this package does not ship the spconv backbone or the RPN) the neck outputs are read from
        `data['bev_map']`, `data['prev_bev_map']` (B, in_channels, H, W)."""
        if self.backbone is None:
            if "bev_map" not in data or "prev_bev_map" not in data:
                raise KeyError("Shasta.extract_feat: no backbone configured and example has no 'bev_map'/'prev_bev_map' "
                               "(neck outputs) nor 'bev_feature'/'prev_bev_feature'")
            return data["bev_map"], None, data["prev_bev_map"], None
        if "voxels" not in data or "prev_voxels" not in data:
            raise KeyError("Shasta.extract_feat: only the hard-voxel branch is supported (the reference's dynamic "
                           "branch never defines prev_input_features, shasta.py:165-176,201-203)")
        input_features = self.reader(data["voxels"], data["num_points"])
        prev_input_features = self.reader(data["prev_voxels"], data["prev_num_points"])
        x, vf = self.backbone(input_features, data["coordinates"], len(data["points"]), data["shape"][0])
        px, pvf = self.backbone(prev_input_features, data["prev_coordinates"], len(data["prev_points"]),
                                data["prev_shape"][0])
        if self.with_neck:
            x = self.neck(x)
            px = self.neck(px)
        return x, vf, px, pvf

    # ---- HIP plumbing --------------------------------------------------------------------------------------
    def _small_params(self):
        mods = [self.fuse_shape[0], self.fuse_shape[2], self.fuse_shape[4], self.fuse_shape[6], self.fuse_det[0],
                self.fuse_det[2], self.fuse_det[4], self.res_coeff[0], self.res_coeff[2], self.res_coeff[4]]
        mods += [self.aff[k] for k in (0, 2, 4, 6, 8, 10)]  # all six: the piece kernels read every aff layer from the packed copy
        return [p for m in mods for p in (m.weight, m.bias)]

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / .float(): parameter storage moves -> drop pointer caches
        self.invalidate_weights_cache()
        return super()._apply(fn, *a, **k)

    def invalidate_weights_cache(self):
        """Forget every kernel-side copy / pointer table derived from the parameters (the shasta_weights struct, the packed small
        weights, the companion of the aug_shape first layers incl. its pre-cut fp16 image, the packed shared_conv): the next forward
        rebuilds them.  The caches follow parameter re-assignment, moves, optimizer steps and load_state_dict by themselves (pointer +
        version checks); call this after writing parameter memory in a way that bumps no version counter (`p.data.copy_()`, a raw-pointer
        write from another library)."""
        self._wstruct = None
        self._plist = None
        self._packed_key = None
        self._aux_key = None
        self._guard_key = None
        self._conv_key = None
        self._conv_raw = None
        if getattr(self, "_conv_bank", None) is not None:
            self._conv_bank._key = None

    def _weights(self):
        """shasta_weights struct over the live parameter storage (no copies).  Cached, because building it walks 34 modules - a visible
        part of a forward at small configurations.  It is re-validated on every call by the data pointers of ALL its tensors (6 us: the
        Parameter objects themselves are cached, which catches `p.data = ...`), and a parameter or sub-module that is REPLACED
        (`m.aff[4].weight = nn.Parameter(...)`, pruning / parametrisation utilities, `m.aff[4] = nn.Linear(...)`) drops the cache through
        torch's registration hooks (_watch).  Optimizer steps and load_state_dict write in place; moves go through _apply."""
        plist = getattr(self, "_plist", None)
        ws = getattr(self, "_wstruct", None)
        if ws is not None and plist is not None:
            probe = tuple(p.data_ptr() for p in plist) + (self.arithmetic, self.precut_weight_stream, int(getattr(self, "extra_options", 0)))
            if ws[1] == probe:
                return ws[0]
        w = self._build_weights()
        _watch(self)
        self._plist = [p for m in self._linear_modules() for p in (m.weight, m.bias)]
        self._wstruct = (w, tuple(p.data_ptr() for p in self._plist) + (self.arithmetic, self.precut_weight_stream, int(getattr(self, "extra_options", 0))))
        return w

    def _linear_modules(self):
        return ([self.aug_shape[i][j] for i in range(4) for j in (0, 2)] + [self.aug_dets[i][j] for i in range(4) for j in (0, 2)] +
                [self.fuse_shape[k] for k in (0, 2, 4, 6)] + [self.fuse_det[k] for k in (0, 2, 4)] + [self.res_coeff[k] for k in (0, 2, 4)] +
                [self.aff[k] for k in (0, 2, 4, 6, 8, 10)])

    def _options(self):
        """shasta_weights.options of Shasta.arithmetic / precut_weight_stream (before the range guard of _ensure_aux)."""
        if self.arithmetic not in ("pieces", "f32", "f16x2", "f16grid"):
            raise ValueError("Shasta.arithmetic must be 'pieces', 'f32', 'f16x2' or 'f16grid'")
        o = {"pieces": 0, "f32": hip.OPT_F32_WEIGHT_STREAM | hip.OPT_F32_EMBED_GEMM | hip.OPT_F32_AFF,
             "f16x2": hip.OPT_F16X2_WEIGHT_STREAM | hip.OPT_F16X2_PAIR | hip.OPT_F16X2_AFF,
             "f16grid": hip.OPT_F16X2_WEIGHT_STREAM | hip.OPT_F16X2_PAIR | hip.OPT_F16GRID_PAIR | hip.OPT_F16X2_AFF}[self.arithmetic]
        if self.arithmetic in ("f16x2", "f16grid") and self.precut_weight_stream:
            o |= hip.OPT_PRECUT_WEIGHT_STREAM
        return o | int(getattr(self, "extra_options", 0))  # (hip.OPT_ONE_PASS_AFF / OPT_TWO_PASS_AFF: the forms of the aff stage)

    def _build_weights(self):
        def lin(m):
            for p in (m.weight, m.bias):
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise hip.ShastaHipError("Shasta parameters must be contiguous fp32 device tensors "
                                             "(call .cuda() first; there is no CPU path)")
            return hip.Linear(m.weight.data_ptr(), m.bias.data_ptr())

        w = hip.Weights()
        w.max_obj, w.num_feats, w.feat_dim = self.max_obj, self.num_feats, self.aug_shape_output
        w.options = self._options()
        for i in range(4):
            w.aug_shape[i][0], w.aug_shape[i][1] = lin(self.aug_shape[i][0]), lin(self.aug_shape[i][2])
            w.aug_dets[i][0], w.aug_dets[i][1] = lin(self.aug_dets[i][0]), lin(self.aug_dets[i][2])
        for j, k in enumerate((0, 2, 4, 6)):
            w.fuse_shape[j] = lin(self.fuse_shape[k])
        for j, k in enumerate((0, 2, 4)):
            w.fuse_det[j] = lin(self.fuse_det[k])
            w.res_coeff[j] = lin(self.res_coeff[k])
        for j, k in enumerate((0, 2, 4, 6, 8, 10)):
            w.aff[j] = lin(self.aff[k])
        return w

    def _ensure_packed(self, w, device):
        """Packed copy of the small pair / aff weights (shasta_pack_weights_f32): rebuilt when one of those tensors changed."""
        # (the Parameter objects from _weights()' cache - it has just validated them - instead of a walk over 16 modules per forward:
        # 16 ms per 800 frames x 7 classes of the configs 2-4 chain)
        plist = getattr(self, "_plist", None)
        small = plist[32:] if plist is not None and len(plist) == 64 else self._small_params()
        key = tuple((p.data_ptr(), p._version) for p in small)
        if self._packed is not None and self._packed_key == key and self._packed.device == device:
            return
        lib = hip.load()
        nbytes = lib.shasta_packed_bytes(self.max_obj, self.num_feats, self.aug_shape_output)
        if nbytes == 0:
            raise hip.ShastaHipError("unsupported feature width F=%d (supported: 64, 256, 320)" % self.aug_shape_output)
        self._packed = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        hip.check(lib.shasta_pack_weights_f32(C.byref(w), hip.ptr(self._packed), nbytes, hip.stream_ptr()),
                  "shasta_pack_weights_f32")
        self._packed_key = key

    def _ensure_aux(self, w, B, device, training=False):
        """Companion of the four aug_shape first-layer matrices (shasta_aug_shape_aux_f32: their row maxima and, with
        precut_weight_stream, their pre-cut fp16 piece image): one pass over 4.1 GB at N=500, so it is built lazily - only for a
        forward that takes the fp16 weight stream - and again only after one of the four matrices changed (an optimizer step bumps
        their versions).  The fp16 stream serves more than 64 frame-pairs per call; with the pre-cut image also every inference call
        of at least 17 (its loop then holds nothing but DMA, LDS reads and MFMAs: 32 / 64 items per weight pass run at the speed of
        the stream; up to 16 the f32 16x16x4 kernel is as fast and needs no activation image).  Training steps (weights change every step) never build the image: affinity_from_bev clears the PRECUT bit for
        them, so they keep the kernels that read the fp32 tensors and only the row maxima are rebuilt.  Without a companion the library recomputes the maxima inside every call that needs them.

        Range guard (inference): the same pass records every row's max|w| / mean|w|; when the largest of them exceeds
        hip.F16X2_MAX_ROW_RATIO (a row that one fp16 scale cannot represent to fp32 accuracy, include/shasta_hip.h) the weight stream of
        THIS weight set falls back to the exact bf16-piece form.  `self.f16x2_guard` says what was measured and decided."""
        plist = getattr(self, "_plist", None)  # (_weights() keeps the parameter objects: nn.Module attribute look-ups cost 10 us each)
        first = [plist[4 * i] for i in range(4)] if plist else [self.aug_shape[i][0].weight for i in range(4)]
        wkey = tuple((p.data_ptr(), p._version) for p in first)
        fp16_bits = hip.OPT_F16X2_WEIGHT_STREAM | hip.OPT_PRECUT_WEIGHT_STREAM
        if self._guard_key is not None and self._guard_key != wkey:  # other weights: decide again (w is a fresh copy with all its bits)
            self._guard_key = None
        if self._guard_key == wkey and self._guard_key is not None:
            w.options &= ~fp16_bits
        small = bool(w.options & hip.OPT_PRECUT_WEIGHT_STREAM) and not training and B >= hip.PRECUT_MIN_BATCH
        need = bool(w.options & hip.OPT_F16X2_WEIGHT_STREAM) and (B > 64 or small) and self.max_obj * self.aug_shape_output >= 64
        if not need:
            w.aug_shape_aux, w.aug_shape_aux_bytes = None, 0
            return
        key = wkey + (w.options,)
        if self._aux is None or self._aux_key != key or self._aux.device != device:
            lib = hip.load()
            nbytes = lib.shasta_aug_shape_aux_bytes(self.max_obj, self.aug_shape_output, w.options)
            self._aux = None
            self._aux = torch.empty(nbytes // 4, dtype=torch.int32, device=device)
            w.aug_shape_aux, w.aug_shape_aux_bytes = None, 0
            hip.check(lib.shasta_aug_shape_aux_f32(C.byref(w), hip.ptr(self._aux), nbytes, hip.stream_ptr()), "shasta_aug_shape_aux_f32")
            self._aux_key = key
            if not training and not torch.cuda.is_current_stream_capturing():  # one host read per weight set (never per step)
                ratio, row = C.c_float(), C.c_int()
                hip.check(lib.shasta_aug_shape_aux_row_ratio(self.max_obj, self.aug_shape_output, hip.ptr(self._aux), nbytes, C.byref(ratio),
                                                             C.byref(row), hip.stream_ptr()), "shasta_aug_shape_aux_row_ratio")
                tripped = not (ratio.value <= hip.F16X2_MAX_ROW_RATIO)  # a NaN trips it
                self.f16x2_guard = dict(max_row_ratio=ratio.value, row=row.value, threshold=hip.F16X2_MAX_ROW_RATIO, tripped=tripped,
                                        weight_stream="pieces" if tripped else "f16x2")
                if tripped:
                    print("shasta_amd: aug_shape first-layer row %d has max|w| / mean|w| = %.3g > %g: the fp16 weight stream is not "
                          "fp32-accurate for it, using the bf16-piece form for this weight set" % (row.value, ratio.value, hip.F16X2_MAX_ROW_RATIO))
                    self._guard_key = wkey
                    self._aux, self._aux_key = None, None
                    w.options &= ~fp16_bits
                    w.aug_shape_aux, w.aug_shape_aux_bytes = None, 0
                    return
        w.aug_shape_aux, w.aug_shape_aux_bytes = self._aux.data_ptr(), self._aux.numel() * 4

    def _work_buffers(self, B, device):
        """Feature / box tables and the stage workspace.  ONE set per device, sized for the largest batch seen so far and
        sliced for smaller ones (the tables are batch-major, the C ABI only needs `workspace_bytes` >= its own figure), so a
        sweep over batch sizes does not pin one full set per size in HBM.  A set that a hipGraph capture has recorded pointers
        of is kept alive for the life of the module."""
        k = str(device)
        cur = self._bufs.get(k)
        if cur is None or cur["B"] < B:
            lib = hip.load()
            N, F = self.max_obj, self.aug_shape_output
            ws = lib.shasta_forward_workspace_bytes(B, N, self.num_feats, F)
            if cur is not None and cur.get("captured"):
                self._graph_bufs.append(cur)
            cur = self._bufs[k] = dict(
                B=B, feat=torch.empty(B, N + 2, F, device=device), prev_feat=torch.empty(B, N + 2, F, device=device),
                det_tab=torch.empty(B, N + 2, 8, device=device), prev_tab=torch.empty(B, N + 2, 8, device=device),
                ws=torch.empty((ws + 3) // 4, dtype=torch.float32, device=device), ws_bytes=ws)
        if torch.cuda.is_current_stream_capturing():
            cur["captured"] = True
        if cur["B"] == B:
            return cur
        return dict(feat=cur["feat"][:B], prev_feat=cur["prev_feat"][:B], det_tab=cur["det_tab"][:B],
                    prev_tab=cur["prev_tab"][:B], ws=cur["ws"], ws_bytes=cur["ws_bytes"])

    def shared_conv_nhwc(self, bev_map, prev_bev_map=None, bound=None):
        """shasta.py:223-228: relu(bn(conv3x3(map))) -> NHWC for the current (and, in the same launch, the previous) neck
        output, by the hand-written implicit-GEMM kernels: csrc/shared_conv_f16.hip (arithmetic "f16x2" / "f16grid": fp32 products
        from two fp16 pieces per operand; maps up to 187 columns wide) or csrc/shared_conv.hip (strict f32 MFMA: "f32", "pieces",
        wider maps).  Eval-mode BatchNorm (running
        statistics) only: in train() mode the module's own nn.Sequential is used so that batch statistics behave like
        the reference.  Returns one tensor, or a pair when prev_bev_map is given.  bound: max |x| of the maps as their producer knows
        it (fp16 form only; SharedConvBank.__call__): the pass over the maps that finds it is skipped."""
        conv, bn = self.shared_conv[0], self.shared_conv[1]
        maps = [t for t in (bev_map, prev_bev_map) if t is not None]
        if not all(t.is_cuda for t in maps):
            raise hip.ShastaHipError("Shasta.shared_conv_nhwc needs device tensors; there is no CPU path")
        needs_grad = torch.is_grad_enabled() and (any(t.requires_grad for t in maps) or
                                                   any(p.requires_grad for p in self.shared_conv.parameters()))
        if self.training or needs_grad:
            # Training (train.py:191 keeps every BN in train() mode: batch statistics; :183-189 freeze backbone and neck only, so
            # shared_conv trains): conv + batch-statistics BatchNorm (+ the exchange of a synchronised one) + ReLU -> NHWC and the
            # backward for its four parameters, hand-written (shared_conv_train.py / csrc/shared_conv_train.hip).  What stays on the
            # module's own nn.Sequential: a map that itself requires grad (somebody trains the neck), an eval-mode BatchNorm under
            # autograd (frozen-BN fine-tuning), a single map, maps wider than the weight-gradient kernel holds (255 columns).
            from . import shared_conv_train as sct
            if (prev_bev_map is not None and self.hand_written_train_conv and sct.supported(self, bev_map) and
                    not prev_bev_map.requires_grad):
                return sct.shared_conv_train(self, bev_map, prev_bev_map)
            outs = [self.shared_conv(t).permute(0, 2, 3, 1).contiguous() for t in maps]
            return outs[0] if prev_bev_map is None else tuple(outs)
        if self.arithmetic in ("f16x2", "f16grid"):  # products from two fp16 pieces per operand (csrc/shared_conv_f16.hip)
            if self._conv_bank is None:
                from .shared_conv import SharedConvBank
                self._conv_bank = SharedConvBank([self])
            if self._conv_bank.supported(bev_map.shape[2], bev_map.shape[3]):
                res = self._conv_bank(bev_map, prev_bev_map, bound=bound)
                return res[0] if prev_bev_map is None else (res[0][0], res[1][0])
        if conv.in_channels % 8 != 0:  # the kernel's K chunks are 8 channels wide: zero-pad the channel axis (exact)
            return self._shared_conv_padded(bev_map, prev_bev_map)
        lib = hip.load()
        tensors = [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        dev = bev_map.device
        if self._conv_packed is None or self._conv_key != key or self._conv_packed.device != dev:
            nbytes = lib.shasta_shared_conv_packed_bytes(conv.in_channels)
            self._conv_packed = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            hip.check(lib.shasta_shared_conv_pack_f32(*[hip.ptr(t.detach()) for t in tensors], float(bn.eps),
                                                      conv.in_channels, hip.ptr(self._conv_packed), nbytes,
                                                      hip.stream_ptr()), "shasta_shared_conv_pack_f32")
            self._conv_key = key
        x = bev_map.float().contiguous()
        B, Cin, H, W = x.shape
        out = torch.empty(B, H, W, conv.out_channels, device=dev)
        xp = outp = None
        if prev_bev_map is not None:
            xp = prev_bev_map.float().contiguous()
            if xp.shape != x.shape:
                raise ValueError("bev_map and prev_bev_map must have the same shape")
            outp = torch.empty_like(out)
        hip.check(lib.shasta_shared_conv_f32(hip.ptr(x), hip.ptr(xp), B, Cin, H, W, hip.ptr(self._conv_packed), hip.ptr(out),
                                             hip.ptr(outp), hip.stream_ptr()), "shasta_shared_conv_f32")
        return out if prev_bev_map is None else (out, outp)

    def _shared_conv_padded(self, bev_map, prev_bev_map):
        """in_channels not a multiple of 8: run the same HIP kernel on zero-padded channels (zeros add exactly)."""
        conv, bn = self.shared_conv[0], self.shared_conv[1]
        lib = hip.load()
        cin = conv.in_channels
        cp = (cin + 7) // 8 * 8
        dev = bev_map.device
        tensors = [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = ("pad",) + tuple((t.data_ptr(), t._version) for t in tensors)
        if self._conv_packed is None or self._conv_key != key or self._conv_packed.device != dev:
            wpad = torch.zeros(conv.out_channels, cp, 3, 3, device=dev)
            wpad[:, :cin] = conv.weight.detach()
            nbytes = lib.shasta_shared_conv_packed_bytes(cp)
            self._conv_packed = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            hip.check(lib.shasta_shared_conv_pack_f32(hip.ptr(wpad), *[hip.ptr(t.detach()) for t in tensors[1:]], float(bn.eps), cp,
                                                      hip.ptr(self._conv_packed), nbytes, hip.stream_ptr()),
                      "shasta_shared_conv_pack_f32")
            self._conv_key = key
        outs = []
        for t in (bev_map, prev_bev_map):
            if t is None:
                continue
            B, _, H, W = t.shape
            x = torch.zeros(B, cp, H, W, device=dev)
            x[:, :cin] = t.float()
            out = torch.empty(B, H, W, conv.out_channels, device=dev)
            hip.check(lib.shasta_shared_conv_f32(hip.ptr(x), None, B, cp, H, W, hip.ptr(self._conv_packed), hip.ptr(out), None,
                                                 hip.stream_ptr()), "shasta_shared_conv_f32")
            outs.append(out)
        return outs[0] if prev_bev_map is None else tuple(outs)

    def affinity_from_bev(self, bev_nhwc, prev_bev_nhwc, det_boxes, prev_det_boxes, l1_events=None, _train_keep=None):
        """Rows 4-16 of SURVEY.md 8(a) (shasta.py:231-325) on device.  bev maps (B,H,W,C) fp32 NHWC, boxes (B,N,>=10)
        fp32 contiguous; det_boxes[:, :, :2] is back-projected in place."""
        lib = hip.load()
        if _train_keep is None and torch.is_grad_enabled() and any(p.requires_grad for p in self._small_params()):
            raise hip.ShastaHipError("Shasta.forward is the inference path: call it under torch.no_grad(), or use "
                                     "shasta_amd.training.affinity_train for a differentiable forward")
        B, N = det_boxes.shape[0], det_boxes.shape[1]
        if N != self.max_obj or prev_det_boxes.shape[1] != N:
            raise ValueError("det_boxes must be padded to max_obj=%d rows (got %d)" % (self.max_obj, N))
        dev = det_boxes.device
        if B == 0:  # nothing to launch
            return torch.empty(0, N, N + 2, device=dev), torch.empty(0, N + 2, N, device=dev)
        # a per-call copy of the cached struct: what this call decides (range guard, companion buffer, the training path's bits)
        # never sticks to the cache
        w = hip.Weights.from_buffer_copy(self._weights())
        if _train_keep is not None and (w.options & hip.OPT_PRECUT_WEIGHT_STREAM):
            # training steps change the weights every step: they keep the kernels that read the fp32 tensors (no 4 GB piece image is
            # rebuilt per step only to be streamed once)
            w.options &= ~hip.OPT_PRECUT_WEIGHT_STREAM
        self._ensure_packed(w, dev)
        self._ensure_aux(w, B, dev, training=_train_keep is not None)
        bufs = self._work_buffers(B, dev)
        m1 = torch.empty(B, N, N + 2, device=dev)
        m2 = torch.empty(B, N + 2, N, device=dev)
        res = mat = None
        if self.keep_intermediates:
            res = torch.empty(B, N + 2, N + 2, device=dev)
            mat = torch.empty(B, N + 2, N + 2, device=dev)
        if _train_keep is not None:  # training.py: same kernels, keeps the residual and the anchor hidden activations
            self.bev_extractor.gather_boxes(bev_nhwc, det_boxes, self.num_point, bufs["feat"])
            self.bev_extractor.gather_boxes(prev_bev_nhwc, prev_det_boxes, self.num_point, bufs["prev_feat"])
            H4 = 4 * (N * self.aug_shape_output // 64)
            _train_keep["residual"] = torch.empty(B, N + 2, N + 2, device=dev)
            _train_keep["shape_hidden"] = torch.empty(B, max(H4, 1), device=dev)
            hip.check(lib.shasta_affinity_forward_train_f32(
                C.byref(w), hip.ptr(self._packed), B, hip.ptr(bufs["feat"]), hip.ptr(bufs["prev_feat"]),
                hip.ptr(det_boxes), hip.ptr(prev_det_boxes), det_boxes.shape[2], hip.ptr(bufs["det_tab"]),
                hip.ptr(bufs["prev_tab"]), hip.ptr(m1), hip.ptr(m2), hip.ptr(_train_keep["residual"]),
                hip.ptr(_train_keep["shape_hidden"]), hip.ptr(bufs["ws"]), bufs["ws_bytes"], hip.stream_ptr()),
                "shasta_affinity_forward_train_f32")
            for k in ("feat", "prev_feat", "det_tab", "prev_tab"):
                _train_keep[k] = bufs[k].clone()
        else:
            # inference: gather + rows 6-16 in ONE call (shasta_affinity_from_bev_f32); l1_events (bench.py): hipEvents around the two
            # heaviest kernels (L1 start / stop, pair start / stop)
            if bev_nhwc.shape != prev_bev_nhwc.shape or bev_nhwc.shape[0] != B or bev_nhwc.shape[3] * self.num_point != self.aug_shape_output:
                raise ValueError("BEV maps must be (B, H, W, C) with C * num_point = %d, both of one shape" % self.aug_shape_output)
            if prev_det_boxes.shape[2] != det_boxes.shape[2]:
                raise ValueError("det_boxes and prev_det_boxes must have the same row width")
            x0, y0, vx, vy, stp = self.bev_extractor._geom()
            evs = None if l1_events is None else (C.c_void_p * 4)(*[e.value for e in l1_events[:4]])
            anch = torch.empty(B, 4, 7, device=dev)  # newborn, fp, dead_trk, fn: a fresh tensor per forward, written by the library
            hip.check(lib.shasta_affinity_from_bev_f32(
                C.byref(w), hip.ptr(self._packed), B, hip.ptr(bev_nhwc), hip.ptr(prev_bev_nhwc), bev_nhwc.shape[1], bev_nhwc.shape[2],
                bev_nhwc.shape[3], x0, y0, vx, vy, stp, hip.ptr(bufs["feat"]), hip.ptr(bufs["prev_feat"]), hip.ptr(det_boxes),
                hip.ptr(prev_det_boxes), det_boxes.shape[2], hip.ptr(bufs["det_tab"]), hip.ptr(bufs["prev_tab"]), hip.ptr(m1), hip.ptr(m2),
                hip.ptr(res), hip.ptr(mat), hip.ptr(anch), hip.ptr(bufs["ws"]), bufs["ws_bytes"], hip.stream_ptr(), evs),
                "shasta_affinity_from_bev_f32")
            self.newborn, self.fp, self.dead_trk, self.fn = anch[:, 0:1], anch[:, 1:2], anch[:, 2:3], anch[:, 3:4]
        self._last_forward = (w, B, bufs["ws"], bufs["ws_bytes"])
        if _train_keep is not None:
            # shasta.py:260-267 leaves the four anchor boxes on the module as fresh tensors: the training entry has no output for them,
            # so they are copied out of the work buffers (two small copies), which the next forward overwrites
            pa, da = bufs["prev_tab"][:, N:, :7].clone(), bufs["det_tab"][:, N:, :7].clone()
            self.newborn, self.fp = pa[:, 0:1], pa[:, 1:2]
            self.dead_trk, self.fn = da[:, 0:1], da[:, 1:2]
        if self.keep_intermediates:
            self.last_intermediates = dict(feature=bufs["feat"], prev_feature=bufs["prev_feat"], residual=res,
                                           matched=mat, det_tab=bufs["det_tab"], prev_tab=bufs["prev_tab"])
        return m1, m2

    def forward_status(self):
        """Asynchronous status of this module's most recent forward (shasta_forward_status): 0, or bit 0 = a row group of the one-pass aff
        kernel timed out waiting for its siblings and wrote NaN into its rows of matched2 (the launch itself had returned SHASTA_OK).
        Synchronises the current stream; raises nothing - callers that read results on the host anyway can afford the check."""
        last = getattr(self, "_last_forward", None)
        if last is None:
            return 0
        w, B, ws, ws_bytes = last
        st = C.c_int(0)
        hip.check(hip.load().shasta_forward_status(C.byref(w), B, hip.ptr(ws), ws_bytes, C.byref(st), hip.stream_ptr()), "shasta_forward_status")
        return st.value

    def forward(self, example, train_mode=True, **kwargs):
        det = example["det_boxes"]
        prev = example["prev_det_boxes"]
        if not det.is_cuda:
            raise hip.ShastaHipError("Shasta.forward needs device tensors (example_to_device); there is no CPU path")
        # BEV maps: precomputed NHWC features (synthetic / cached), or extract_feat + shared_conv like the reference
        if self.backbone is None and "bev_map" not in example and "bev_feature" in example and "prev_bev_feature" in example:
            bev = example["bev_feature"].float().contiguous()
            prev_bev = example["prev_bev_feature"].float().contiguous()
        else:
            bev_map, _, prev_bev_map, _ = self.extract_feat(example)
            bev, prev_bev = self.shared_conv_nhwc(bev_map, prev_bev_map, bound=example.get("bev_map_bound"))
        example["bev_feature"] = bev
        inplace = det.dtype == torch.float32 and det.is_contiguous() and det.shape[2] >= 10
        det_k = det if inplace else det.float().contiguous()
        prev_k = prev if (prev.dtype == torch.float32 and prev.is_contiguous()) else prev.float().contiguous()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._small_params()):
            from .training import affinity_train  # differentiable: same forward kernels + the HIP backward (training.py)
            m1, m2 = affinity_train(self, bev, prev_bev, det_k, prev_k)
        else:
            m1, m2 = self.affinity_from_bev(bev, prev_bev, det_k, prev_k)
        if not inplace:
            det[:, :, :2] = det_k[:, :, :2].to(det.dtype)  # keep the reference's in-place side effect
        return m1, m2, example


def load_state_dict_permissive(module, state_dict, logger=None):
    """det3d/torchie/trainer/checkpoint.py:67-104 semantics (unknown keys and shape mismatches are skipped) but every
    skipped key is reported instead of silently dropped."""
    own = module.state_dict()
    skipped = []
    for name, param in state_dict.items():
        if name.startswith("module."):
            name = name[7:]
        if name not in own:
            skipped.append((name, "unexpected"))
            continue
        if own[name].shape != param.shape:
            skipped.append((name, "shape %s vs %s" % (tuple(param.shape), tuple(own[name].shape))))
            continue
        own[name].copy_(param)
    missing = sorted(set(own.keys()) - {n[7:] if n.startswith("module.") else n for n in state_dict.keys()})
    for name, why in skipped:
        print("load_state_dict: skipped %s (%s)" % (name, why))
    if missing:
        print("load_state_dict: missing keys: %s" % ", ".join(missing))
    return skipped, missing
