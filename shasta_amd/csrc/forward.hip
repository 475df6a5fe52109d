// C-ABI entry points of the affinity forward: argument checks, workspace carving, stage sequencing.
#include "common.hpp"
#include "pair_layout.hpp"

namespace shasta {
size_t anchor_shape_workspace_bytes(int B, int N, int F);
size_t anchor_boxes_workspace_bytes(int B, int N);
size_t pair_workspace_bytes(int B, int N, int F);
size_t aff_workspace_bytes(int B, int N);
const float* anchor_shape_hidden(const void* ws, int B, int N, int F);
int anchor_shape(const shasta_weights* w, int B, float* feat, float* prev_feat, void* ws, size_t ws_bytes, hipStream_t st,
                 hipEvent_t ev0, hipEvent_t ev1, const unsigned* wmax, bool xmax_ready);
bool anchor_shape_uses_xmax(const shasta_weights* w, int B);
unsigned* anchor_shape_xmax(void* ws, int B, int N, int F);
unsigned* anchor_shape_xmax_slots(void* ws, int B, int N, int F);
size_t bev_absmax_slot_bytes(int items);
int launch_absmax_finalize(const unsigned* slots, unsigned* out, int items, hipStream_t st);
int launch_bev_gather(const float* bev, int B, int H, int W, int C, const float* boxes, int N, int box_stride, int box_batch_stride,
                      int num_point, float pc_x0, float pc_y0, float vs_x, float vs_y, float out_stride, float* out, int out_row_stride,
                      int out_batch_stride, unsigned* absmax, hipStream_t st, const float* bev2, const float* boxes2, float* out2,
                      unsigned* absmax2);
int anchor_boxes(const shasta_weights* w, int B, float* det_boxes, const float* prev_det_boxes, int box_stride,
                 float* det_tab, float* prev_tab, float* hid_ws, hipStream_t st, float* anchors_out);
bool anchor_stage_fused_serves(const shasta_weights* w, int B);
int anchor_stage_fused(const shasta_weights* w, int B, float* feat, float* prev_feat, float* det_boxes, const float* prev_det_boxes,
                       int box_stride, float* det_tab, float* prev_tab, void* ws, size_t ws_bytes, hipStream_t st, hipEvent_t ev0,
                       hipEvent_t ev1, const unsigned* wmax, bool xmax_ready, float* anchors_out);
int pair_residual(const shasta_weights* w, const float* packed, int B, const float* feat, const float* prev_feat,
                  const float* det_tab, const float* prev_tab, float* residual, int ld, void* ws, size_t ws_bytes,
                  hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
int aff_status(const shasta_weights* w, int B, int ld, const void* ws, int* status, hipStream_t st);
int aff_softmax(const shasta_weights* w, const float* packed, int B, const float* residual, int ld, float* m1,
                float* m2, float* matched_out, void* ws, size_t ws_bytes, hipStream_t st);
int pack_weights(const shasta_weights* w, float* packed, hipStream_t st);
int launch_w_maxima(const float* const W[4], int H, int K, unsigned* wmax, float* sumabs, hipStream_t st);
size_t precut_image_bytes(int H, int K);
int launch_precut_weights(const float* const W[4], const unsigned* wmax, void* img, int H, int K, hipStream_t st);

static int check_weights(const shasta_weights* w) {
    SHASTA_REQUIRE(w, "null weights");
    SHASTA_REQUIRE(w->max_obj >= 1 && w->max_obj <= 2046, "max_obj out of range (1..2046)");
    SHASTA_REQUIRE(w->num_feats >= 1 && w->num_feats <= 7, "num_feats must be 1..7");
    SHASTA_REQUIRE(w->feat_dim == 64 || w->feat_dim == 256 || w->feat_dim == 320, "feat_dim must be 64, 256 or 320");
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) {
            SHASTA_REQUIRE(w->aug_shape[i][j].weight && w->aug_shape[i][j].bias, "aug_shape weights missing");
            // aug_dets hidden width is 7N//32: zero for N < 5, then only the last bias exists
            SHASTA_REQUIRE(7 * w->max_obj / 32 == 0 || (w->aug_dets[i][j].weight && w->aug_dets[i][j].bias),
                           "aug_dets weights missing");
            SHASTA_REQUIRE(j == 0 || w->aug_dets[i][j].bias, "aug_dets output bias missing");
        }
    for (int i = 0; i < 4; ++i) SHASTA_REQUIRE(w->fuse_shape[i].weight && w->fuse_shape[i].bias, "fuse_shape weights missing");
    for (int i = 0; i < 3; ++i) {
        SHASTA_REQUIRE(w->fuse_det[i].weight && w->fuse_det[i].bias, "fuse_det weights missing");
        SHASTA_REQUIRE(w->res_coeff[i].weight && w->res_coeff[i].bias, "res_coeff weights missing");
    }
    for (int i = 0; i < 6; ++i) SHASTA_REQUIRE(w->aff[i].weight && w->aff[i].bias, "aff weights missing");
    for (int i = 0; i < 4; ++i)
        SHASTA_REQUIRE((uintptr_t)w->aug_shape[i][0].weight % 16 == 0, "aug_shape.*.0.weight must be 16-byte aligned");
    // a companion buffer must have been built with the SAME options (the pre-cut image sits behind the maxima only when it was
    // built with SHASTA_OPT_PRECUT_WEIGHT_STREAM): its size says so
    SHASTA_REQUIRE(!w->aug_shape_aux || w->aug_shape_aux_bytes >= shasta_aug_shape_aux_bytes(w->max_obj, w->feat_dim, w->options),
                   "aug_shape_aux: companion buffer smaller than shasta_aug_shape_aux_bytes(max_obj, feat_dim, options) - built without "
                   "SHASTA_OPT_PRECUT_WEIGHT_STREAM? (set shasta_weights.aug_shape_aux_bytes)");
    return SHASTA_OK;
}

struct FwdWs {
    size_t anchor, boxes, pair, aff, residual, total;
    FwdWs(int B, int N, int F) {
        const int T = N + 2, Dp = (T + 3) / 4 * 4;
        anchor = anchor_shape_workspace_bytes(B, N, F);
        boxes = anchor_boxes_workspace_bytes(B, N);
        pair = pair_workspace_bytes(B, N, F);
        aff = aff_workspace_bytes(B, N);
        residual = align_up((size_t)B * T * Dp * sizeof(float), 256);
        // anchor / pair / aff scratch is live one stage at a time -> shared region (the two anchor stages side by side: small batches run
        // them interleaved, anchor_stage_fused)
        size_t stage = anchor + boxes > pair ? anchor + boxes : pair;
        stage = stage > aff ? stage : aff;
        stage = stage > boxes ? stage : boxes;
        total = stage + residual;
    }
};

}  // namespace shasta

using namespace shasta;

extern "C" size_t shasta_packed_bytes(int max_obj, int num_feats, int feat_dim) {
    if (feat_dim != 64 && feat_dim != 256 && feat_dim != 320) return 0;
    return PackedLayout(max_obj, num_feats, feat_dim).total * sizeof(float);
}

extern "C" int shasta_pack_weights_f32(const shasta_weights* w, void* packed, size_t packed_bytes,
                                       shasta_stream_t stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    SHASTA_REQUIRE(packed && (uintptr_t)packed % 16 == 0, "pack: packed buffer null or not 16-byte aligned");
    if (packed_bytes < shasta_packed_bytes(w->max_obj, w->num_feats, w->feat_dim)) {
        set_error_msg("pack: packed buffer too small");
        return SHASTA_E_WORKSPACE;
    }
    return pack_weights(w, static_cast<float*>(packed), as_stream(stream));
}

extern "C" size_t shasta_aug_shape_aux_bytes(int max_obj, int feat_dim, int options) {
    const size_t H = (size_t)max_obj * feat_dim / 64;
    size_t n = aux_image_offset(H);  // row maxima + row statistics (common.hpp)
    if (options & SHASTA_OPT_PRECUT_WEIGHT_STREAM) n += precut_image_bytes((int)H, max_obj * feat_dim);
    return n;
}

extern "C" int shasta_aug_shape_aux_f32(const shasta_weights* w, void* aux, size_t aux_bytes, shasta_stream_t stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    SHASTA_REQUIRE(aux && (uintptr_t)aux % 16 == 0, "aug_shape_aux: buffer null or not 16-byte aligned");
    if (aux_bytes < shasta_aug_shape_aux_bytes(w->max_obj, w->feat_dim, w->options)) {
        set_error_msg("aug_shape_aux: buffer too small");
        return SHASTA_E_WORKSPACE;
    }
    const float* W[4];
    for (int i = 0; i < 4; ++i) W[i] = w->aug_shape[i][0].weight;
    const int K = w->max_obj * w->feat_dim, H = K / 64;
    float* stats = reinterpret_cast<float*>(static_cast<char*>(aux) + aux_maxima_bytes((size_t)H));
    if ((rc = launch_w_maxima(W, H, K, static_cast<unsigned*>(aux), stats, as_stream(stream)))) return rc;
    if ((w->options & SHASTA_OPT_PRECUT_WEIGHT_STREAM) && precut_image_bytes(H, K))
        rc = launch_precut_weights(W, static_cast<const unsigned*>(aux), static_cast<char*>(aux) + aux_image_offset((size_t)H), H, K, as_stream(stream));
    return rc;
}

extern "C" int shasta_aug_shape_aux_row_ratio(int max_obj, int feat_dim, const void* aux, size_t aux_bytes, float* h_max_ratio, int* h_row,
                                              shasta_stream_t stream) {
    SHASTA_REQUIRE(aux && h_max_ratio && max_obj >= 1 && feat_dim >= 1, "aug_shape_aux_row_ratio: bad argument");
    const size_t H = (size_t)max_obj * feat_dim / 64;
    SHASTA_REQUIRE(aux_bytes >= aux_image_offset(H), "aug_shape_aux_row_ratio: not a companion buffer of this shape");
    float s[2];
    const char* src = static_cast<const char*>(aux) + aux_maxima_bytes(H) + 4 * H * sizeof(float);
    hipError_t e = hipMemcpyAsync(s, src, sizeof(s), hipMemcpyDeviceToHost, as_stream(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    if (e != hipSuccess) {
        set_error("aug_shape_aux_row_ratio", e);
        return SHASTA_E_LAUNCH;
    }
    *h_max_ratio = s[0];
    if (h_row) *h_row = (int)s[1];
    return SHASTA_OK;
}

extern "C" size_t shasta_forward_workspace_bytes(int B, int max_obj, int num_feats, int feat_dim) {
    (void)num_feats;
    return FwdWs(B, max_obj, feat_dim).total;
}

extern "C" int shasta_anchor_shape_f32(const shasta_weights* w, int B, float* feat, float* prev_feat, void* workspace,
                                       size_t workspace_bytes, shasta_stream_t stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    SHASTA_REQUIRE(B >= 0 && feat && prev_feat && workspace, "anchor_shape: bad argument");
    SHASTA_REQUIRE(((uintptr_t)feat | (uintptr_t)prev_feat) % 16 == 0, "anchor_shape: tables must be 16-byte aligned");
    return anchor_shape(w, B, feat, prev_feat, workspace, workspace_bytes, as_stream(stream), nullptr, nullptr,
                        static_cast<const unsigned*>(w->aug_shape_aux), false);
}

extern "C" int shasta_anchor_boxes_f32(const shasta_weights* w, int B, float* det_boxes, const float* prev_det_boxes,
                                       int box_stride, float* det_tab, float* prev_tab, void* workspace,
                                       size_t workspace_bytes, shasta_stream_t stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    SHASTA_REQUIRE(B >= 0 && det_boxes && prev_det_boxes && det_tab && prev_tab && workspace, "anchor_boxes: bad argument");
    SHASTA_REQUIRE(box_stride >= 10, "anchor_boxes: box rows need [x,y,z,w,l,h,yaw,vx,vy,dt]");
    if (workspace_bytes < anchor_boxes_workspace_bytes(B, w->max_obj)) {
        set_error_msg("anchor_boxes: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    return anchor_boxes(w, B, det_boxes, prev_det_boxes, box_stride, det_tab, prev_tab, static_cast<float*>(workspace),
                        as_stream(stream), nullptr);
}

extern "C" int shasta_pair_residual_f32(const shasta_weights* w, const void* packed, int B, const float* feat,
                                        const float* prev_feat, const float* det_tab, const float* prev_tab,
                                        float* residual, int ld_residual, void* workspace, size_t workspace_bytes,
                                        shasta_stream_t stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    SHASTA_REQUIRE(B >= 0 && packed && feat && prev_feat && det_tab && prev_tab && residual && workspace,
                   "pair_residual: bad argument");
    SHASTA_REQUIRE(ld_residual >= w->max_obj + 2, "pair_residual: ld_residual < N+2");
    SHASTA_REQUIRE(((uintptr_t)feat | (uintptr_t)prev_feat | (uintptr_t)packed) % 16 == 0, "pair_residual: alignment");
    return pair_residual(w, static_cast<const float*>(packed), B, feat, prev_feat, det_tab, prev_tab, residual,
                         ld_residual, workspace, workspace_bytes, as_stream(stream), nullptr, nullptr);
}

extern "C" int shasta_aff_softmax_f32(const shasta_weights* w, const void* packed, int B, const float* residual,
                                      int ld_residual, float* matched1, float* matched2, float* matched_out,
                                      void* workspace, size_t workspace_bytes, shasta_stream_t stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    SHASTA_REQUIRE(B >= 0 && packed && residual && matched1 && matched2 && workspace, "aff_softmax: bad argument");
    SHASTA_REQUIRE(ld_residual >= w->max_obj + 2, "aff_softmax: ld_residual < N+2");
    return aff_softmax(w, static_cast<const float*>(packed), B, residual, ld_residual, matched1, matched2, matched_out,
                       workspace, workspace_bytes, as_stream(stream));
}

extern "C" int shasta_aff_status(const shasta_weights* w, int B, int ld_residual, const void* workspace, size_t workspace_bytes,
                                 int* status, shasta_stream_t stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    SHASTA_REQUIRE(B >= 0 && workspace && status, "aff_status: bad argument");
    if (workspace_bytes < aff_workspace_bytes(B, w->max_obj)) {
        set_error_msg("aff_status: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    return aff_status(w, B, ld_residual, workspace, status, as_stream(stream));
}

extern "C" int shasta_forward_status(const shasta_weights* w, int B, const void* workspace, size_t workspace_bytes, int* status,
                                     shasta_stream_t stream) {
    int rc = check_weights(w);
    if (rc) return rc;
    SHASTA_REQUIRE(B >= 0 && workspace && status, "forward_status: bad argument");
    const int N = w->max_obj, T = N + 2, Dp = (T + 3) / 4 * 4;
    const FwdWs L(B, N, w->feat_dim);
    if (workspace_bytes < L.total) {
        set_error_msg("forward_status: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    return aff_status(w, B, Dp, static_cast<const char*>(workspace) + L.residual, status, as_stream(stream));
}

struct BevSource {
    const float *bev, *prev_bev;
    int H, W, C;
    float pc_x0, pc_y0, vs_x, vs_y, out_stride;
    float* anchor_boxes_out;
};

static int forward_impl(const shasta_weights* w, const void* packed, int B, float* feat, float* prev_feat,
                        float* det_boxes, const float* prev_det_boxes, int box_stride, float* det_tab, float* prev_tab,
                        float* matched1, float* matched2, float* residual_out, float* matched_out, void* workspace,
                        size_t workspace_bytes, shasta_stream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                        float* shape_hidden_out = nullptr, hipEvent_t ev_pair0 = nullptr, hipEvent_t ev_pair1 = nullptr,
                        const BevSource* src = nullptr) {
    int rc = check_weights(w);
    if (rc) return rc;
    SHASTA_REQUIRE(B >= 0 && packed && feat && prev_feat && det_boxes && prev_det_boxes && det_tab && prev_tab &&
                       matched1 && matched2 && workspace,
                   "forward: null pointer");
    SHASTA_REQUIRE(box_stride >= 10, "forward: box rows need [x,y,z,w,l,h,yaw,vx,vy,dt]");
    SHASTA_REQUIRE(((uintptr_t)feat | (uintptr_t)prev_feat | (uintptr_t)packed | (uintptr_t)workspace) % 16 == 0,
                   "forward: feat/prev_feat/packed/workspace must be 16-byte aligned");
    const int N = w->max_obj, F = w->feat_dim, T = N + 2, Dp = (T + 3) / 4 * 4;
    const FwdWs L(B, N, F);
    if (workspace_bytes < L.total) {
        set_error_msg("forward: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    if (B == 0) return SHASTA_OK;
    hipStream_t st = as_stream(stream);
    float* residual = static_cast<float*>(workspace);
    void* stage = static_cast<char*>(workspace) + L.residual;
    const size_t stage_bytes = L.total - L.residual;
    const float* pk = static_cast<const float*>(packed);
    bool xmax_ready = false;
    if (src) {
        // rows [0, N) of both tables straight from the BEV maps; when the fp16 weight stream follows, the gather also leaves the
        // largest magnitude of every batch item where anchor_shape looks for it (no separate pass over the tables)
        unsigned* slots = nullptr;
        if (anchor_shape_uses_xmax(w, B)) {
            slots = anchor_shape_xmax_slots(stage, B, N, F);
            if (hipMemsetAsync(slots, 0, bev_absmax_slot_bytes(2 * B), st) != hipSuccess) return SHASTA_E_LAUNCH;
            xmax_ready = true;
        }
        const int np = F / src->C;
        // both frames in one launch (grid.y = 2)
        if ((rc = launch_bev_gather(src->bev, B, src->H, src->W, src->C, det_boxes, N, box_stride, N * box_stride, np, src->pc_x0, src->pc_y0,
                                    src->vs_x, src->vs_y, src->out_stride, feat, F, T * F, slots, st, src->prev_bev, prev_det_boxes, prev_feat,
                                    slots ? slots + bev_absmax_slot_bytes(B) / sizeof(unsigned) : nullptr)))
            return rc;
        if (slots && (rc = launch_absmax_finalize(slots, anchor_shape_xmax(stage, B, N, F), 2 * B, st))) return rc;
    }
    if (!shape_hidden_out && anchor_stage_fused_serves(w, B) && !anchor_shape_uses_xmax(w, B)) {
        // one or a few frame-pairs: three launches less (the training path wants the hidden activations materialised: the long form)
        if ((rc = anchor_stage_fused(w, B, feat, prev_feat, det_boxes, prev_det_boxes, box_stride, det_tab, prev_tab, stage, stage_bytes, st, ev0,
                                     ev1, static_cast<const unsigned*>(w->aug_shape_aux), xmax_ready, src ? src->anchor_boxes_out : nullptr)))
            return rc;
    } else {
        // row maxima of the first-layer weights (fp16 form of the weight stream): the caller's companion buffer, or recomputed per call
        if ((rc = anchor_shape(w, B, feat, prev_feat, stage, stage_bytes, st, ev0, ev1, static_cast<const unsigned*>(w->aug_shape_aux), xmax_ready)))
            return rc;
        if (shape_hidden_out) {  // training: the backward re-uses the hidden activations instead of re-streaming the weights
            const size_t H = (size_t)N * F / 64;
            hipError_t e = hipMemcpyAsync(shape_hidden_out, anchor_shape_hidden(stage, B, N, F), (size_t)B * 4 * H * sizeof(float),
                                          hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) {
                set_error("forward: copy anchor hidden", e);
                return SHASTA_E_LAUNCH;
            }
        }
        if ((rc = anchor_boxes(w, B, det_boxes, prev_det_boxes, box_stride, det_tab, prev_tab, static_cast<float*>(stage), st,
                               src ? src->anchor_boxes_out : nullptr)))
            return rc;
    }
    if ((rc = pair_residual(w, pk, B, feat, prev_feat, det_tab, prev_tab, residual, Dp, stage, stage_bytes, st, ev_pair0, ev_pair1)))
        return rc;
    if (residual_out) {
        hipError_t e = hipMemcpy2DAsync(residual_out, (size_t)T * sizeof(float), residual, (size_t)Dp * sizeof(float),
                                        (size_t)T * sizeof(float), (size_t)B * T, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) {
            set_error("forward: copy residual", e);
            return SHASTA_E_LAUNCH;
        }
    }
    return aff_softmax(w, pk, B, residual, Dp, matched1, matched2, matched_out, stage, stage_bytes, st);
}

extern "C" int shasta_affinity_forward_f32(const shasta_weights* w, const void* packed, int B, float* feat,
                                           float* prev_feat, float* det_boxes, const float* prev_det_boxes,
                                           int box_stride, float* det_tab, float* prev_tab, float* matched1,
                                           float* matched2, float* residual_out, float* matched_out, void* workspace,
                                           size_t workspace_bytes, shasta_stream_t stream) {
    return forward_impl(w, packed, B, feat, prev_feat, det_boxes, prev_det_boxes, box_stride, det_tab, prev_tab, matched1,
                        matched2, residual_out, matched_out, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int shasta_affinity_forward_train_f32(const shasta_weights* w, const void* packed, int B, float* feat,
                                                 float* prev_feat, float* det_boxes, const float* prev_det_boxes,
                                                 int box_stride, float* det_tab, float* prev_tab, float* matched1,
                                                 float* matched2, float* residual_out, float* shape_hidden_out,
                                                 void* workspace, size_t workspace_bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(residual_out && shape_hidden_out, "forward_train: null output");
    return forward_impl(w, packed, B, feat, prev_feat, det_boxes, prev_det_boxes, box_stride, det_tab, prev_tab, matched1,
                        matched2, residual_out, nullptr, workspace, workspace_bytes, stream, nullptr, nullptr,
                        shape_hidden_out);
}

extern "C" int shasta_affinity_forward_timed_f32(const shasta_weights* w, const void* packed, int B, float* feat,
                                                 float* prev_feat, float* det_boxes, const float* prev_det_boxes,
                                                 int box_stride, float* det_tab, float* prev_tab, float* matched1,
                                                 float* matched2, void* workspace, size_t workspace_bytes,
                                                 shasta_stream_t stream, void* ev_l1_start, void* ev_l1_stop,
                                                 void* ev_pair_start, void* ev_pair_stop) {
    SHASTA_REQUIRE(ev_l1_start && ev_l1_stop && ev_pair_start && ev_pair_stop, "forward_timed: null event");
    return forward_impl(w, packed, B, feat, prev_feat, det_boxes, prev_det_boxes, box_stride, det_tab, prev_tab, matched1,
                        matched2, nullptr, nullptr, workspace, workspace_bytes, stream,
                        static_cast<hipEvent_t>(ev_l1_start), static_cast<hipEvent_t>(ev_l1_stop), nullptr,
                        static_cast<hipEvent_t>(ev_pair_start), static_cast<hipEvent_t>(ev_pair_stop));
}

extern "C" int shasta_affinity_from_bev_f32(const shasta_weights* w, const void* packed, int B, const float* bev, const float* prev_bev,
                                            int H, int W, int C, float pc_x0, float pc_y0, float vs_x, float vs_y, float out_stride,
                                            float* feat, float* prev_feat, float* det_boxes, const float* prev_det_boxes, int box_stride,
                                            float* det_tab, float* prev_tab, float* matched1, float* matched2, float* residual_out,
                                            float* matched_out, float* anchor_boxes_out, void* workspace, size_t workspace_bytes,
                                            shasta_stream_t stream, void* const* h_events4) {
    SHASTA_REQUIRE(w && bev && prev_bev, "affinity_from_bev: null pointer");
    SHASTA_REQUIRE(H > 0 && W > 0 && C > 0 && w->feat_dim % C == 0, "affinity_from_bev: feat_dim must be num_point * C");
    const int np = w->feat_dim / C;
    SHASTA_REQUIRE(np == 1 || np == 4 || np == 5, "affinity_from_bev: num_point = feat_dim / C must be 1, 4 or 5");
    SHASTA_REQUIRE(box_stride >= 10, "affinity_from_bev: box rows need [x,y,z,w,l,h,yaw,vx,vy,dt]");
    const BevSource src{bev, prev_bev, H, W, C, pc_x0, pc_y0, vs_x, vs_y, out_stride, anchor_boxes_out};
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    if (h_events4)
        for (int i = 0; i < 4; ++i) {
            SHASTA_REQUIRE(h_events4[i], "affinity_from_bev: null event");
            e[i] = static_cast<hipEvent_t>(h_events4[i]);
        }
    return forward_impl(w, packed, B, feat, prev_feat, det_boxes, prev_det_boxes, box_stride, det_tab, prev_tab, matched1, matched2,
                        residual_out, matched_out, workspace, workspace_bytes, stream, e[0], e[1], nullptr, e[2], e[3], &src);
}

extern "C" int shasta_event_create(void** ev) {
    SHASTA_REQUIRE(ev, "event_create: null");
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) {
        set_error("event_create", rc);
        return SHASTA_E_LAUNCH;
    }
    *ev = e;
    return SHASTA_OK;
}

extern "C" int shasta_event_destroy(void* ev) {
    if (ev) (void)hipEventDestroy(static_cast<hipEvent_t>(ev));
    return SHASTA_OK;
}

extern "C" int shasta_event_elapsed_ms(void* start, void* stop, float* h_ms) {
    SHASTA_REQUIRE(start && stop && h_ms, "event_elapsed: null");
    hipError_t rc = hipEventElapsedTime(h_ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop));
    if (rc != hipSuccess) {
        set_error("event_elapsed", rc);
        return SHASTA_E_LAUNCH;
    }
    return SHASTA_OK;
}
