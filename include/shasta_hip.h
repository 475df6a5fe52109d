/*
 * shasta_hip.h -- C ABI of the MI355X (gfx950) ShaSTA affinity hot path.
 *
 * The reference (tsadja/ShaSTA) has no native plugin on this path: its boundary is the Python
 * `nn.Module.forward` of det3d/models/tracker/shasta.py and the ATen ops below it.  This header
 * is the C boundary a maintainer binds instead (ctypes / pybind / cgo alike): plain pointers and
 * sizes, no torch types, every call asynchronous on the caller's `hipStream_t`, no allocation
 * inside, no globals and no environment switches (kernel choices that change arithmetic are per-call
 * `options` bits of shasta_weights), `int` status (0 = ok, <0 = SHASTA_E_*), never exit()/abort().
 * The shared library exports exactly the functions declared here (built with -fvisibility=hidden).
 * All pointers are DEVICE pointers unless a parameter name starts with `h_`.
 * Arithmetic: fp32 operands in HBM and fp32 accumulation everywhere (the reference runs apex O0 = fp32,
 * tools/nusc_shasta/train.py:149).  HOW an fp32 product is formed on the matrix cores is a per-call choice
 * (shasta_weights.options below): by default (options = 0) large batches form it from six products of three exact bf16
 * pieces per operand; SHASTA_OPT_F16X2_* form it from three products of two range-scaled, round-to-nearest fp16 pieces
 * per operand (a 23-bit operand representation, |error| <= 2^-24 per operand: one rounding more than a true fp32
 * multiply); SHASTA_OPT_F32_* use the f32 MFMA instructions only (strict fp32 products).
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference root).
 */
#ifndef SHASTA_HIP_H
#define SHASTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#pragma GCC visibility push(default)

typedef void* shasta_stream_t; /* a hipStream_t; NULL = the null stream */

#define SHASTA_OK 0
#define SHASTA_E_ARG (-1)       /* bad size / null pointer / unsupported shape */
#define SHASTA_E_WORKSPACE (-2) /* workspace too small */
#define SHASTA_E_LAUNCH (-3)    /* hipGetLastError() != hipSuccess after a launch */
#define SHASTA_E_ALIGN (-4)     /* pointer or leading dimension not aligned as documented */
#define SHASTA_E_UNSUPPORTED (-5) /* the device refuses a resource the kernel needs (LDS per workgroup): take the documented other path */

/* Library/ABI version (bumped on any signature change) and a human readable build string. */
int shasta_abi_version(void);
const char* shasta_build_info(void);
/* Last hip error string seen by a failed call on this thread (never NULL). */
const char* shasta_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K1  point -> voxel scatter (+ per-voxel mean)
 * replaces det3d/ops/point_cloud/point_cloud_ops.py:112-184 (`points_to_voxel`, kernel :7-55,
 * reverse_index=True as called from det3d/core/input/voxel_generator.py:19-30) and
 * det3d/models/readers/voxel_encoder.py:18-28 (`VoxelFeatureExtractorV3.forward`).
 * Output order and contents are bit-identical to the serial reference loop: voxels are numbered
 * in first-touch order of the input points, new voxels past `max_voxels` are dropped, each voxel
 * keeps its first `max_points` points in input order.
 *
 *  points      (P, ndim) fp32, ndim >= 3, xyz first
 *  range6      h_: x0,y0,z0,x1,y1,z1 ; voxel3 h_: vx,vy,vz   (fp32, as numpy float32 in the ref)
 *  voxels      (max_voxels, max_points, ndim) fp32   -- zero padded
 *  coors       (max_voxels, 3) int32 z,y,x
 *  num_points  (max_voxels,) int32
 *  mean        (max_voxels, ndim) fp32 or NULL        -- sum over slots / count
 *  num_voxels  (1,) int32 device scalar (V); rows >= V of every output are not written (the reference returns the first V rows)
 *  workspace   shasta_voxelize_workspace_bytes(P, max_voxels, max_points) bytes.  It holds, besides the per-point scratch, the
 *              cell -> first point map: the reference allocates a dense (gz, gy, gx) int32 map per call (331 MB for the nuScenes
 *              grid, point_cloud_ops.py:150); a cloud of P points touches at most P cells, so here it is an open-addressing hash
 *              table of 2^ceil(log2(2 P)) entries (8 MB for 3e5 points), cleared by one memset inside the call.
 * ------------------------------------------------------------------------------------------ */
size_t shasta_voxelize_workspace_bytes(int num_points, int max_voxels, int max_points);
int shasta_voxelize_mean_f32(const float* points, int num_points, int ndim, const float* h_range6,
                             const float* h_voxel3, int max_points, int max_voxels, float* voxels,
                             int32_t* coors, int32_t* num_points_per_voxel, float* mean,
                             int32_t* num_voxels, void* workspace, size_t workspace_bytes, shasta_stream_t stream);

/* A batch of clouds in ONE chain of launches: the reference voxelises the current and the previous cloud of every sample
 * (datasets/pipelines/preprocess.py:179-208: `Voxelization.__call__` runs the generator on res["lidar"]["points"] and on
 * res["lidar"]["prev_points"]), one cloud per DataLoader worker call.  `points` holds the clouds back to back ((sum P, ndim) fp32),
 * h_offsets[0..num_clouds] (HOST) their first rows; num_clouds <= 32.  Every output has a leading cloud axis:
 * voxels (num_clouds, max_voxels, max_points, ndim), coors (num_clouds, max_voxels, 3), num_points (num_clouds, max_voxels),
 * mean (num_clouds, max_voxels, ndim) or NULL, num_voxels (num_clouds,) int32 ON THE DEVICE (no host read inside the call).
 * workspace: shasta_voxelize_batch_workspace_bytes (per-point scratch + one hash table per cloud, sized for the largest cloud).
 * Cloud c's results are bit for bit those of shasta_voxelize_mean_f32 on that cloud alone. */
size_t shasta_voxelize_batch_workspace_bytes(const int* h_offsets, int num_clouds, int max_voxels, int max_points);
int shasta_voxelize_mean_batch_f32(const float* points, const int* h_offsets, int num_clouds, int ndim, const float* h_range6,
                                   const float* h_voxel3, int max_points, int max_voxels, float* voxels, int32_t* coors,
                                   int32_t* num_points_per_voxel, float* mean, int32_t* num_voxels, void* workspace,
                                   size_t workspace_bytes, shasta_stream_t stream);

/* The reader alone, for callers that already hold voxelised input (det3d/models/readers/voxel_encoder.py:18-28 as called from
 * Shasta.extract_feat, det3d/models/tracker/shasta.py:178-179): out (V, num_features) = sum over the max_points slots of
 * voxels (V, max_points, ndim)[..., :num_features] / num_points.  num_points_f32: (V,) fp32 - example_to_device
 * (det3d/torchie/apis/train_track.py:60) has already cast the counts to float, as in the reference. */
int shasta_voxel_mean_f32(const float* voxels, const float* num_points_f32, int num_voxels, int max_points, int ndim,
                          int num_features, float* out, shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K2  BEV bilinear gather at box centre / edge mid-points
 * replaces Shasta.get_box_center (det3d/models/tracker/shasta.py:121-161),
 * center_to_corner_box2d (det3d/core/bbox/box_torch_ops.py:184-203),
 * BEVFeatureExtractor.forward (det3d/models/second_stage/bird_eye_view.py:18-41) and
 * bilinear_interpolate_torch (det3d/core/utils/center_utils.py:92-121).
 *
 *  bev     (B,H,W,C) fp32 NHWC
 *  boxes   (B,N,>=7) fp32 rows [x,y,z,w,l,h,yaw,...], `box_stride` floats per row,
 *          `box_batch_stride` floats per batch item
 *  out     feature table: row n of batch b at out + b*out_batch_stride + n*out_row_stride,
 *          num_point*C floats [pt0 C | pt1 C | ...]; num_point in {1,4,5}
 * ------------------------------------------------------------------------------------------ */
int shasta_bev_gather_f32(const float* bev, int B, int H, int W, int C, const float* boxes, int N,
                          int box_stride, int box_batch_stride, int num_point, float pc_x0,
                          float pc_y0, float vs_x, float vs_y, float out_stride, float* out,
                          int out_row_stride, int out_batch_stride, shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K0  shared_conv: Conv2d(Cin->64, 3x3, pad 1, bias) + BatchNorm2d(64) (eval) + ReLU -> NHWC
 * replaces det3d/models/tracker/shasta.py:42-47 as applied at :223-228 (`self.shared_conv(bev_map)` followed by
 * `.permute(0, 2, 3, 1).contiguous()`), i.e. produces example['bev_feature'].
 *
 *  weight (64, Cin, 3, 3), bias (64), bn_* (64) : the tensors shared_conv.0.* / shared_conv.1.* of the state_dict
 *  packed : shasta_shared_conv_packed_bytes(Cin) bytes, 16-byte aligned; re-pack when any of the six tensors changes
 *  x      : (B, Cin, H, W) fp32 NCHW (neck output), Cin a multiple of 8
 *  out    : (B, H, W, 64) fp32 NHWC
 *  x_prev, out_prev : the previous frame's map and output, processed in the same launch (both NULL to skip)
 * ------------------------------------------------------------------------------------------ */
size_t shasta_shared_conv_packed_bytes(int in_channels);
int shasta_shared_conv_pack_f32(const float* weight, const float* bias, const float* bn_weight,
                                const float* bn_bias, const float* bn_mean, const float* bn_var,
                                float bn_eps, int in_channels, void* packed, size_t packed_bytes,
                                shasta_stream_t stream);
int shasta_shared_conv_f32(const float* x, const float* x_prev, int B, int in_channels, int H, int W,
                           const void* packed, float* out, float* out_prev, shasta_stream_t stream);

/* K0 on the fp16 matrix path, one launch for up to 8 class heads (the per-class models of tools/nusc_shasta/eval.py:86-101,
 * official_val.sh each hold their own shared_conv.0 / .1 and convolve the SAME neck output): fp32 maps in, fp32 NHWC maps out, fp32
 * accumulation; every fp32 product is formed from three products of two range-scaled, round-to-nearest fp16 pieces per operand (the
 * SHASTA_OPT_F16X2_* arithmetic: one power-of-two scale per output channel, fixed at pack time, and one per image, taken from a
 * max-reduction pass over the map inside the call).  The map is read from HBM once for all heads.
 *
 *  shasta_shared_conv_f16x2_supported : 1 when the fp16 kernel serves (in_channels, H, W): in_channels % 16 == 0 and a map at most
 *          187 columns wide (the staged tile of 256 pixels + one image row either side must fit 640 LDS slots); other shapes stay
 *          on shasta_shared_conv_f32
 *  packed : `heads` images of shasta_shared_conv_f16x2_packed_bytes(Cin) bytes, `head_stride_bytes` apart (multiple of 16), each
 *           written by shasta_shared_conv_pack_f16x2 from that head's six tensors; re-pack when one of them changes
 *  x, x_prev : (B, Cin, H, W) fp32 NCHW; x_prev / h_out_prev NULL together to skip the previous frame
 *  h_out, h_out_prev : HOST arrays of `heads` device pointers, each (B, H, W, 64) fp32 NHWC
 *  workspace : shasta_shared_conv_multi_workspace_bytes(B) bytes (the image maxima)
 * Non-finite inputs: an image that holds an Inf or a NaN is cut with scale 1; where torch would give an Inf the result is a NaN. */
int shasta_shared_conv_f16x2_supported(int in_channels, int H, int W);
size_t shasta_shared_conv_f16x2_packed_bytes(int in_channels);
int shasta_shared_conv_pack_f16x2(const float* weight, const float* bias, const float* bn_weight,
                                  const float* bn_bias, const float* bn_mean, const float* bn_var,
                                  float bn_eps, int in_channels, void* packed, size_t packed_bytes,
                                  shasta_stream_t stream);
/* The same with the largest magnitude of the maps supplied by the caller (x_absmax_bound > 0: the producer of the neck output knows a
 * bound of it) instead of found by a pass over them: every image is cut under the bound's scale.  The bound need not be tight - an fp16
 * piece pair keeps 22 significant bits of every element within 2^-17 of it; an element beyond 4 x the bound overflows fp16 and comes
 * out as NaN / Inf, never as a wrong finite number. */
int shasta_shared_conv_multi_bounded_f32(const float* x, const float* x_prev, int B, int in_channels, int H, int W,
                                         const void* packed, size_t head_stride_bytes, int heads, float* const* h_out,
                                         float* const* h_out_prev, void* workspace, size_t workspace_bytes,
                                         float x_absmax_bound, shasta_stream_t stream);
size_t shasta_shared_conv_multi_workspace_bytes(int B);
/* The workspace that lets a call of this shape (B frame pairs - two_maps != 0 - or B single maps, `heads` class heads) take its fastest
 * form: from three heads over enough maps on, the input is cut ONCE into an fp16 piece image (68 MB per 512 x 180 x 180 map, behind the
 * image maxima) that every head's workgroups fetch by LDS-DMA.  A call given less (at least shasta_shared_conv_multi_workspace_bytes(B))
 * cuts the input tile per workgroup; same results bit for bit. */
size_t shasta_shared_conv_multi_workspace_bytes_for(int B, int in_channels, int H, int W, int heads, int two_maps);
int shasta_shared_conv_multi_f32(const float* x, const float* x_prev, int B, int in_channels, int H, int W,
                                 const void* packed, size_t head_stride_bytes, int heads, float* const* h_out,
                                 float* const* h_out_prev, void* workspace, size_t workspace_bytes,
                                 shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K0 in train() mode (csrc/shared_conv_train.hip): what autograd does for `self.shared_conv(bev_map)` when the reference trains it -
 * tools/nusc_shasta/train.py:183-191 freezes children 1, 2 (backbone, neck) only and keeps every BatchNorm in train mode; the
 * optimizer (:143-147) holds shared_conv.0.{weight,bias} and shared_conv.1.{weight,bias}.  The neck is frozen: no input gradient.
 *
 * Forward of one BatchNorm call (= one map: det3d/models/tracker/shasta.py:223-228 calls shared_conv on the current map, then on the
 * previous one):
 *   1. y = conv(x) + bias, NHWC: shasta_shared_conv_multi_f32 / shasta_shared_conv_f32 on a RAW pack (shasta_shared_conv_pack_raw_*:
 *      the epilogue writes conv + bias as is - no BatchNorm, no ReLU); the fp16 form leaves the image maxima at the head of its
 *      workspace (2B uint32 bit patterns), which the weight gradient takes over
 *   2. shasta_bn_stats_f32: mean_m2[0:64] = batch mean, [64:128] = sum of squared deviations (float64 accumulation) over the
 *      M = maps x H x W pixels of y; a synchronised BatchNorm merges these over the ranks (shasta_amd/sync_bn.py) before step 3
 *   3. shasta_bn_finalize_f32: stat[0:64] = mean, [64:128] = 1 / sqrt(M2 / n + eps); running statistics (NULL to skip) updated as
 *      nn.BatchNorm2d does (momentum, unbiased variance), *num_batches_tracked += 1
 *   4. shasta_bn_relu_apply_f32: out = relu((y - mean) invstd gamma + beta)
 * Backward, per BatchNorm call, given gout = d loss / d out (M, 64):
 *   5. shasta_bn_relu_bwd_reduce_f32: sums[0:64] = sum g', [64:128] = sum g' xhat (g' = gout where out > 0; = dbeta, dgamma of this
 *      rank), [128:192] = max |g'|, [192:256] = max |xhat|; a synchronised BatchNorm all-reduces sums[0:128] into sums_global
 *   6. shasta_bn_relu_bwd_dy_f16x2: dy = gamma invstd (g' - sums_global[0] / n - xhat sums_global[1] / n) for images
 *      [img0, img0 + nimg) of the `nimg_total` maps of the step, written as range-scaled fp16 pieces in the fragment order of the
 *      weight-gradient kernel into `dy` (shasta_conv_dy_bytes(nimg_total, H, W)); edy[0:64] = this call's scale exponents; dbias[0:64]
 *      (+)= sum dy
 *   7. shasta_conv_wgrad_f16x2, once for all maps of the step: dweight (64, Cin, 3, 3) = sum over pixels and maps of dy x shifted x, three
 *      fp16 piece products per fp32 product, fp32 accumulation, fixed summation order (same bits on every run).  xmax: the image maxima
 *      of step 1 in the order [current maps of the B frame pairs, previous maps]; edy: [2][64] (current call, previous call).
 * shasta_conv_train_supported: maps up to 255 columns wide (any Cin, H). */
int shasta_conv_train_supported(int in_channels, int H, int W);
int shasta_shared_conv_pack_raw_f32(const float* weight, const float* bias, int in_channels, void* packed, size_t packed_bytes,
                                    shasta_stream_t stream);
int shasta_shared_conv_pack_raw_f16x2(const float* weight, const float* bias, int in_channels, void* packed, size_t packed_bytes,
                                      shasta_stream_t stream);
size_t shasta_bn_workspace_bytes(void);
int shasta_bn_stats_f32(const float* y, long M, float* mean_m2, void* workspace, size_t workspace_bytes, shasta_stream_t stream);
int shasta_bn_finalize_f32(const float* mean_m2, double n, float eps, float momentum, float* stat, float* running_mean,
                           float* running_var, long* num_batches_tracked, shasta_stream_t stream);
int shasta_bn_relu_apply_f32(const float* y, long M, const float* stat, const float* gamma, const float* beta, float* out,
                             shasta_stream_t stream);
int shasta_bn_relu_bwd_reduce_f32(const float* y, const float* gout, long M, const float* stat, const float* gamma,
                                  const float* beta, float* sums, void* workspace, size_t workspace_bytes, shasta_stream_t stream);
size_t shasta_conv_dy_bytes(int nimg, int H, int W);
int shasta_bn_relu_bwd_dy_f16x2(const float* y, const float* gout, int nimg, int img0, int nimg_total, int H, int W, const float* stat,
                                const float* gamma, const float* beta, const float* sums_global, const float* sums_local,
                                double n_global, void* dy, size_t dy_bytes, float* edy, float* dbias, int accumulate_dbias,
                                shasta_stream_t stream);
size_t shasta_conv_wgrad_workspace_bytes(int nimg, int in_channels, int H, int W);
int shasta_conv_wgrad_f16x2(const float* x, const float* x_prev, int B, int in_channels, int H, int W, const unsigned* xmax,
                            const void* dy, const float* edy, float* dweight, void* workspace, size_t workspace_bytes,
                            shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Weights of the affinity network, raw nn.Linear layout (out_features, in_features) row major,
 * exactly the tensors of the reference state_dict (det3d/models/tracker/shasta.py:49-106).
 * ------------------------------------------------------------------------------------------ */
typedef struct shasta_linear {
    const float* weight; /* (out, in) */
    const float* bias;   /* (out,) */
} shasta_linear;

/* shasta_weights.options: every bit keeps fp32 operands and fp32 accumulation; they select HOW the fp32 products are formed.
 * Default (0): above 32 frame-pairs per call the aug_shape first layer, and from 8192 table rows the row-embedding GEMMs of the
 * pair stage and the six aff layers, form each fp32 product from six exact bf16 piece products on the bf16 matrix path
 * (anchor_split.hip, gemm_pieces.hip, aff_pieces.hip; error at the level of the fp32 FMA's own rounding). */
#define SHASTA_OPT_F32_WEIGHT_STREAM 1 /* aug_shape first layer on v_mfma_f32_32x32x2_f32 for every batch size */
#define SHASTA_OPT_F32_EMBED_GEMM 2    /* row-embedding GEMMs on v_mfma_f32_32x32x2_f32 for every row count */
#define SHASTA_OPT_F32_AFF 4           /* the six aff layers on v_mfma_f32_16x16x4_f32 for every row count */
#define SHASTA_OPT_F16X2_WEIGHT_STREAM 8 /* aug_shape first layer above 64 frame-pairs from two range-scaled fp16 pieces per operand
                                            (round to nearest, three products per fp32 product) instead of three bf16 pieces (six) */
#define SHASTA_OPT_F16X2_PAIR 16         /* second layers of the three pair MLPs in the same two-piece fp16 form (feat_dim 256: 16x16x32 tiles, 320: 32x32x16 tiles) */
#define SHASTA_OPT_PRECUT_WEIGHT_STREAM 32 /* with SHASTA_OPT_F16X2_WEIGHT_STREAM: stream the aug_shape first-layer weights as pre-cut fp16
                                              pieces from the companion buffer (shasta_aug_shape_aux_f32 built with this bit: + 4 bytes per
                                              weight resident) instead of cutting the fp32 tensors on the fly; same arithmetic, same results */

#define SHASTA_OPT_F16GRID_PAIR 64 /* with SHASTA_OPT_F16X2_PAIR, from 8192 table rows: the fp16 pieces of the per-pair hidden activations on a
                                      fixed grid per MLP (22 bits of the tile's largest sum) instead of cut per pair from the fp32 sum:
                                      a third fewer vector instructions in the pair kernel, errors of the residual 2x (max) / 5x (rms)
                                      those of the fp32 kernels (about 1e-6 of its range): NOT fp32-equivalent, opt-in */

#define SHASTA_OPT_TWO_PASS_AFF 128 /* from 8192 table rows: keep the two-kernel form of the aff stage (six layers + row softmax, then a
                                       column-softmax pass over the stored logits) instead of the single pass whose workgroups exchange
                                       per-column partials; same layer arithmetic, softmax sums in another order */

#define SHASTA_OPT_F16X2_AFF 256 /* from 8192 table rows: the six aff layers from two range-scaled fp16 pieces per operand (three products
                                    per fp32 product; one scale per weight row, per activation row and layer, in layer 1 per row and
                                    32-column chunk) instead of three bf16 pieces (six) */

#define SHASTA_OPT_ONE_PASS_AFF 512 /* the one-pass form of the aff stage (layers + both softmaxes, bf16 or - with SHASTA_OPT_F16X2_AFF -
                                       fp16 pieces) for EVERY row count, not only from 8192 table rows (where it is the default);
                                       small batches are otherwise served by 16-row workgroups on the f32 matrix path */

typedef struct shasta_weights {
    int max_obj;   /* N */
    int num_feats; /* nf: 1..7 */
    int feat_dim;  /* F = share_conv_channel * num_point; supported: 64, 256, 320 */
    int options;   /* SHASTA_OPT_* bits, 0 = default */
    shasta_linear aug_shape[4][2]; /* aug_shape.{i}.{0,2}: (N*F/64, N*F), (F, N*F/64) */
    shasta_linear aug_dets[4][2];  /* aug_dets.{i}.{0,2}:  (7N/32, 7N), (7, 7N/32)     */
    shasta_linear fuse_shape[4];   /* fuse_shape.{0,2,4,6}: 2F->F/8->F/16->F/32->1      */
    shasta_linear fuse_det[3];     /* fuse_det.{0,2,4}:     2nf->32->8->1               */
    shasta_linear res_coeff[3];    /* res_coeff.{0,2,4}:    2F+2nf->32+F/8->8+F/32->3   */
    shasta_linear aff[6];          /* aff.{0,2,4,6,8,10}:   N+2->128->64->32->64->128->N+2 */
    const void* aug_shape_aux;     /* shasta_aug_shape_aux_f32 output for the CURRENT aug_shape.{i}.0.weight, or NULL */
    size_t aug_shape_aux_bytes;    /* its size: every call that takes the struct rejects (SHASTA_E_ARG) a companion smaller than
                                      shasta_aug_shape_aux_bytes(max_obj, feat_dim, options), e.g. one built without the PRECUT bit */
} shasta_weights;

/* Packed (kernel-ready) copy of the SMALL weights: MFMA fragments of the pair MLPs (fuse_shape, fuse_det, res_coeff),
 * their factorised first layers, the six aff layers as piece fragments.  It depends on those tensors only - not on the
 * aug_shape / aug_dets matrices and not on `options` - and must be re-packed whenever one of them changes. */
size_t shasta_packed_bytes(int max_obj, int num_feats, int feat_dim);
int shasta_pack_weights_f32(const shasta_weights* w, void* packed, size_t packed_bytes,
                            shasta_stream_t stream);

/* Companion of the four aug_shape.{i}.0.weight matrices (N*F/64, N*F), which are used in place and never copied (4.1 GB at
 * N=500, F=256): the largest magnitude of every weight row = the range exponents of the fp16 form of the weight stream
 * (SHASTA_OPT_F16X2_WEIGHT_STREAM, more than 64 frame-pairs per call) and, when w->options holds
 * SHASTA_OPT_PRECUT_WEIGHT_STREAM, the pre-cut fp16 piece image of the matrices behind them (a forward may only use that option with
 * a companion built with it).  One pass over the matrices.  Hand it to the forward as
 * shasta_weights.aug_shape_aux and recompute it whenever one of the four matrices changes; with aug_shape_aux == NULL a forward
 * that needs it recomputes it into its workspace on every call (correct, one extra pass over the weights per call). */
size_t shasta_aug_shape_aux_bytes(int max_obj, int feat_dim, int options);
int shasta_aug_shape_aux_f32(const shasta_weights* w, void* aux, size_t aux_bytes, shasta_stream_t stream);

/* Range guard of the two-piece fp16 weight stream.  That form is block floating point: ONE power-of-two scale per weight row puts the
 * row's largest magnitude into (2^13, 2^14], so every weight is represented with an ABSOLUTE error of at most 2^-38 of the row maximum.
 * That is below fp32's own per-element rounding (2^-24 relative) as long as the elements that carry the row are within 2^14 of its
 * maximum.  shasta_aug_shape_aux_f32 therefore also records, per row, max|w| / mean|w| of the row's OTHER entries; this call returns the largest over the
 * 4 * N*F/64 rows (and its row, MLP-major) to the HOST - it synchronises `stream` - so that the caller can take the fp16 form only
 * when the figure is at most SHASTA_F16X2_MAX_ROW_RATIO and fall back to the exact bf16-piece form (options without
 * SHASTA_OPT_F16X2_WEIGHT_STREAM) otherwise.  Default uniform init: ~2; Gaussian rows: ~6; log-normal (sigma 2) rows: ~1e3; a row with
 * one entry 2^20 x the others: ~1e5 (refused).  A NaN figure means non-finite weights. */
#define SHASTA_F16X2_MAX_ROW_RATIO 16384.0f
int shasta_aug_shape_aux_row_ratio(int max_obj, int feat_dim, const void* aux, size_t aux_bytes, float* h_max_ratio, int* h_row,
                                   shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K3-K6  affinity forward after the gather: Shasta.forward, det3d/models/tracker/shasta.py:240-325
 *   anchors (aug_shape :241-247, aug_dets :260-267), back-projection (:270, IN PLACE on
 *   det_boxes), anchor concat (:273-274), hand-designed residuals (:277-283), fuse_shape
 *   (:286-290), fuse_det (:293-307), res_coeff (:310-316), combine (:319), aff + softmaxes
 *   (:323-325).
 *
 *  feat, prev_feat  (B, N+2, F) fp32 tables; rows [0,N) filled by shasta_bev_gather_f32 (out_row_stride=F,
 *                   out_batch_stride=(N+2)*F); rows N, N+1 are written here (anchor shape embeddings:
 *                   feat gets dead_trk_geom, fn_geom; prev_feat gets newborn_geom, fp_geom)
 *  det_boxes        (B,N,box_stride>=10) fp32 [x,y,z,w,l,h,yaw,vx,vy,dt,...]; x,y are back-projected in place
 *  prev_det_boxes   (B,N,box_stride>=7)
 *  det_tab, prev_tab (B, N+2, 8) fp32 out: the 7-vectors after back-projection with the anchor boxes
 *                   appended (prev_tab rows N,N+1 = newborn, fp ; det_tab rows N,N+1 = dead_trk, fn)
 *  matched1         (B, N, N+2) fp32 out  = softmax over dim 2 of matched[:, :-2, :]
 *  matched2         (B, N+2, N) fp32 out  = softmax over dim 1 of matched[:, :, :-2]
 *  residual_out     (B, N+2, N+2) fp32 or NULL: the aff input (for parity tests)
 *  matched_out      (B, N+2, N+2) fp32 or NULL: the aff output before the softmaxes
 * ------------------------------------------------------------------------------------------ */
size_t shasta_forward_workspace_bytes(int B, int max_obj, int num_feats, int feat_dim);
int shasta_affinity_forward_f32(const shasta_weights* w, const void* packed, int B, float* feat,
                                float* prev_feat, float* det_boxes, const float* prev_det_boxes,
                                int box_stride, float* det_tab, float* prev_tab, float* matched1,
                                float* matched2, float* residual_out, float* matched_out,
                                void* workspace, size_t workspace_bytes, shasta_stream_t stream);

/* The same forward INCLUDING the gather (K2): rows [0,N) of feat / prev_feat are filled from the NHWC maps bev / prev_bev (B,H,W,C),
 * C = feat_dim / num_point, with the boxes' geometry arguments of shasta_bev_gather_f32 (the gather reads det_boxes BEFORE the
 * back-projection, as shasta.py:231-239 does) - i.e. all of Shasta.forward behind shared_conv (shasta.py:231-325) in one call.  Same
 * results as shasta_bev_gather_f32 x 2 + shasta_affinity_forward_f32, bit for bit; when the fp16 weight stream follows, the gather
 * also produces the activation row maxima that stream needs, which otherwise take a pass of their own over the tables.
 * anchor_boxes_out: NULL, or (B, 4, 7): the aug_dets anchor boxes newborn, fp, dead_trk, fn (shasta.py:260-267 leaves them on the
 * module as fresh tensors; they are also rows N, N+1 of prev_tab / det_tab).
 * h_events4: NULL, or four events from shasta_event_create recorded around the weight-stream kernel [0],[1] and the pair kernel
 * [2],[3] (bench.py's live roofline). */
int shasta_affinity_from_bev_f32(const shasta_weights* w, const void* packed, int B, const float* bev, const float* prev_bev,
                                 int H, int W, int C, float pc_x0, float pc_y0, float vs_x, float vs_y, float out_stride,
                                 float* feat, float* prev_feat, float* det_boxes, const float* prev_det_boxes, int box_stride,
                                 float* det_tab, float* prev_tab, float* matched1, float* matched2, float* residual_out,
                                 float* matched_out, float* anchor_boxes_out, void* workspace, size_t workspace_bytes,
                                 shasta_stream_t stream, void* const* h_events4);

/* Training forward: the same kernels and values as shasta_affinity_forward_f32; additionally keeps what the backward
 * needs and the inference path throws away: residual_out (B, N+2, N+2) and shape_hidden_out (B, 4*H), H = N*F/64, the ReLU
 * outputs of aug_shape.{0..3}.0 (row b = [mlp0 H | mlp1 H | mlp2 H | mlp3 H]), so that the 4 x (H, N*F) first-layer weights
 * are not streamed a second time to recompute them. */
int shasta_affinity_forward_train_f32(const shasta_weights* w, const void* packed, int B, float* feat, float* prev_feat,
                                      float* det_boxes, const float* prev_det_boxes, int box_stride, float* det_tab,
                                      float* prev_tab, float* matched1, float* matched2, float* residual_out,
                                      float* shape_hidden_out, void* workspace, size_t workspace_bytes,
                                      shasta_stream_t stream);
/* Measurement variant of the forward: identical work, plus hipEventRecord on `stream` around the launches of the two
 * kernels that carry the step - ev_l1_* around the aug_shape first-layer weight stream (anchor_l1_kernel /
 * anchor_l1_mfma_kernel / anchor_l1_split_kernel), ev_pair_* around the pair kernel (pair_mfma4_kernel) - so that
 * bench.py can read both durations live inside its timed region and report the roofline of whichever is the longer.
 * Events come from shasta_event_create (they are plain hipEvent_t); shasta_event_elapsed_ms needs both events
 * completed (synchronise the stream first). */
int shasta_affinity_forward_timed_f32(const shasta_weights* w, const void* packed, int B, float* feat,
                                      float* prev_feat, float* det_boxes, const float* prev_det_boxes,
                                      int box_stride, float* det_tab, float* prev_tab, float* matched1,
                                      float* matched2, void* workspace, size_t workspace_bytes,
                                      shasta_stream_t stream, void* ev_l1_start, void* ev_l1_stop,
                                      void* ev_pair_start, void* ev_pair_stop);
int shasta_event_create(void** ev);
int shasta_event_destroy(void* ev);
int shasta_event_elapsed_ms(void* start, void* stop, float* h_ms);

/* Stage entry points (the forward above is exactly these in sequence; exposed for per-stage parity
 * tests and for callers that own the schedule).  Layouts as documented above. */
int shasta_anchor_shape_f32(const shasta_weights* w, int B, float* feat, float* prev_feat, void* workspace,
                            size_t workspace_bytes, shasta_stream_t stream);
int shasta_anchor_boxes_f32(const shasta_weights* w, int B, float* det_boxes, const float* prev_det_boxes,
                            int box_stride, float* det_tab, float* prev_tab, void* workspace,
                            size_t workspace_bytes, shasta_stream_t stream);
int shasta_pair_residual_f32(const shasta_weights* w, const void* packed, int B, const float* feat,
                             const float* prev_feat, const float* det_tab, const float* prev_tab,
                             float* residual, int ld_residual, void* workspace, size_t workspace_bytes,
                             shasta_stream_t stream);
int shasta_aff_softmax_f32(const shasta_weights* w, const void* packed, int B, const float* residual,
                           int ld_residual, float* matched1, float* matched2, float* matched_out,
                           void* workspace, size_t workspace_bytes, shasta_stream_t stream);

/* Status of the most recent aff stage launched on a workspace - asynchronous failures the launching call could not report (it had
 * returned SHASTA_OK before the kernel ran).  *status = 0, or bit 0: a row group of the one-pass aff kernel (layers + both softmaxes
 * of shasta.py:323-325 in one launch, from 8192 table rows) gave up waiting for the column statistics of its sibling row groups;
 * its rows of matched2 are NaN.  Both calls synchronise on `stream`.  shasta_aff_status: `workspace` / `ld_residual` as passed to
 * shasta_aff_softmax_f32; shasta_forward_status: `workspace` as passed to shasta_affinity_forward_f32 / _from_bev_f32 / _train_f32
 * (valid until the next forward on that workspace). */
int shasta_aff_status(const shasta_weights* w, int B, int ld_residual, const void* workspace, size_t workspace_bytes,
                      int* status, shasta_stream_t stream);
int shasta_forward_status(const shasta_weights* w, int B, const void* workspace, size_t workspace_bytes, int* status,
                          shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Rotated 3-D IoU / GIoU distance matrix (float64, like the reference)
 * replaces mot_3d/association.py:108-120 (`compute_iou_distance`): dist[d][t] = 1 - iou3d(det_d, trk_t)[1]
 * (mot_3d/utils/geometry.py:161-176) or 1 - giou3d(det_d, trk_t) (:208-231), boxes [x, y, z, o, l, w, h]
 * (mot_3d/data_protos/bbox.py:27-33), `box_stride` doubles per row.
 * ------------------------------------------------------------------------------------------ */
int shasta_iou3d_distance_f64(const double* dets, int num_dets, const double* tracks, int num_tracks,
                              int box_stride, int giou, double* dist, shasta_stream_t stream);

/* Generic fp32 MFMA GEMM used by the stages above: C[m][n] = act(sum_k A[m][k]*W[n][k] + bias[n]).
 * act: 0 none, 1 relu, 2 abs.  lda/ldw multiples of 4 and 16-byte aligned bases. */
int shasta_gemm_nt_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C,
                       int ldc, int M, int N, int K, int act, shasta_stream_t stream);
/* The same contract on the bf16 matrix path: every fp32 product is the sum of six exact bf16 piece products (three
 * 8-bit pieces per operand, fp32 accumulation), error below the fp32 FMA's own rounding; 2.7x fewer matrix cycles.
 * Used for the row-embedding GEMMs of the pair stage from 8192 rows up. */
int shasta_gemm_nt_pieces_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C,
                              int ldc, int M, int N, int K, int act, shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batched decode decisions (consumer of the affinity matrices)
 * replaces the per-element .item() loop of tools/nusc_shasta/eval.py:127-173 (== validate.py:68-114) by compact
 * per-row / per-column decisions; thresholds 0.5 / 0.7 are the reference's hard-coded ones (eval.py:137,141,161,163).
 *  n_prev, n_cur (B,) int32 : number of real previous / current detections of each frame pair
 *  prev_class (B,N) int32   : 0 keep, 1 dead track, 2 false negative, -1 row >= n_prev
 *  prev_score (B,N) fp32    : matched1[n, dead column]; ref_detection_score of a propagated FN box is 1 - this
 *  det_flags  (B,N) int32   : bit 0 kept (not a false positive), bit 1 newborn, -1 column >= n_cur
 *  det_score  (B,N) fp32    : matched2[FP row, k]; ref_detection_score of a kept detection is 1 - this
 * ------------------------------------------------------------------------------------------ */
int shasta_decode_flags_f32(const float* matched1, const float* matched2, const int32_t* n_prev,
                            const int32_t* n_cur, int B, int max_obj, int32_t* prev_class, float* prev_score,
                            int32_t* det_flags, float* det_score, shasta_stream_t stream);

/* Strided form (training path: dX = dY.W and dW = dY^T.X of every nn.Linear, torch autograd's addmm backward):
 *   C[m][n] = act(sum_k A[m*sa_m + k*sa_k] * W[n*sw_n + k*sw_k] + bias[n]) * (relu_mask[m][n] > 0)
 * act: 0 none, 1 relu, 2 abs, +4: accumulate into C, +8: bf16 operands (A and W are rounded to bf16, nearest even, on chip and
 * multiplied on the bf16 matrix path; accumulation, bias, epilogue and C stay fp32 - the reduced-precision option of the
 * training GEMMs, fp32 master weights).  splitk_ws (optional): scratch for a deterministic split of long reductions
 * (weight gradients). */
int shasta_gemm_strided_f32(const float* A, long sa_m, long sa_k, const float* W, long sw_n, long sw_k,
                            const float* bias, const float* relu_mask, int ldmask, float* C, int ldc, int M,
                            int N, int K, int act, void* splitk_ws, size_t splitk_ws_bytes,
                            shasta_stream_t stream);

/* `count` (<= 8) such products of ONE shape in one launch (two with a split reduction): the four aug_shape / aug_dets MLPs of the anchor
 * backward (det3d/models/tracker/shasta.py:49-57, 69-76 have four of each, equal in shape), the two sides of a pair MLP's first layer.
 * A, W, bias, relu_mask, C: HOST arrays of `count` device pointers (bias / relu_mask: NULL or an array whose entries may be NULL alike
 * for all members); strides, sizes and `act` are shared.  Member i is computed exactly as shasta_gemm_strided_f32 would compute it
 * alone with splitk_ws_bytes / count bytes of scratch (same tiles, same reduction slices, same bits).  Members must not overlap in C. */
int shasta_gemm_strided_group_f32(int count, const float* const* A, const float* const* W, const float* const* bias,
                                  const float* const* relu_mask, float* const* C, long sa_m, long sa_k, long sw_n, long sw_k,
                                  int ldmask, int ldc, int M, int N, int K, int act, void* splitk_ws, size_t splitk_ws_bytes,
                                  shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Rotated BEV NMS
 * replaces det3d/ops/iou3d_nms: `nms_gpu` (src/iou3d_nms.cpp:100-143 with nms_kernel, src/iou3d_nms_kernel.cu:267-311 and
 * iou_bev :226-233) as called from det3d/ops/iou3d_nms/iou3d_nms_utils.py:74-89 and
 * det3d/core/bbox/box_torch_ops.py:248-276 (`rotate_nms_pcdet`).
 *  boxes_sorted (N,7) fp32 [x, y, z, dx, dy, dz, heading], already ordered by descending score; N <= 32768
 *  keep (N,) int32: indices (into boxes_sorted) of the kept boxes in score order; num_keep (1,) int32 device scalar
 *  workspace: shasta_nms_workspace_bytes(N) bytes (the N x ceil(N/64) suppression bitmask)
 * ------------------------------------------------------------------------------------------ */
size_t shasta_nms_workspace_bytes(int num_boxes);
int shasta_nms_rotated_f32(const float* boxes_sorted, int num_boxes, float thresh, void* workspace, size_t workspace_bytes,
                           int32_t* keep, int32_t* num_keep, shasta_stream_t stream);

/* The same with the axis-aligned IoU of the BEV footprints (heading ignored): `nms_normal_gpu`
 * (src/iou3d_nms.cpp:146-188, nms_normal_kernel + iou_normal src/iou3d_nms_kernel.cu:313-372, called from
 * det3d/ops/iou3d_nms/iou3d_nms_utils.py:93-106).  Same arguments and workspace as shasta_nms_rotated_f32. */
int shasta_nms_normal_f32(const float* boxes_sorted, int num_boxes, float thresh, void* workspace, size_t workspace_bytes,
                          int32_t* keep, int32_t* num_keep, shasta_stream_t stream);

/* Pairwise BEV matrices of det3d/ops/iou3d_nms: `boxes_overlap_bev_gpu` / `boxes_iou_bev_gpu` (src/iou3d_nms.cpp:56-97,
 * boxes_overlap_kernel / boxes_iou_bev_kernel src/iou3d_nms_kernel.cu:236-265) and the 3-D IoU that
 * det3d/ops/iou3d_nms/iou3d_nms_utils.py:35-72 (`boxes_iou3d_gpu`) builds on the overlap.
 *  boxes_a (num_a, 7), boxes_b (num_b, 7) fp32 [x, y, z, dx, dy, dz, heading]; out (num_a, num_b) fp32
 *  mode 0: overlap area of the rotated footprints; 1: BEV IoU = overlap / max(sa + sb - overlap, 1e-8);
 *  mode 2: 3-D IoU = overlap * overlap_h / max(vol_a + vol_b - overlap * overlap_h, 1e-6)
 * The overlap is the exact intersection area (float64 convex clip, rounded to fp32); the reference's fp32 routine additionally
 * counts corners up to 1e-2 outside the other box as inside (iou3d_nms_kernel.cu:51-61), so it can exceed the exact area for
 * nearly touching boxes. */
int shasta_boxes_bev_f32(const float* boxes_a, int num_a, const float* boxes_b, int num_b, int mode, float* out,
                         shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Public tracker step, device part: centre-distance matrix + greedy assignment for `scenes` independent scenes
 * replaces tools/nusc_shasta/pub_tracker.py:94-108 (`dist`, `invalid`, `dist + invalid * 1e18`) and
 * tools/nusc_shasta/track_utils.py:3-14 (`greedy_assignment`), same float32 / float64 arithmetic and tie-breaking.
 *  det_xy (S,Nmax,2) fp32 = ct + tracking of every detection, trk_xy (S,Mmax,2), det_cat / trk_cat int32 class ids,
 *  max_diff (S,Nmax) fp32 per-detection gate, n / m (S,) int32 valid counts (device), Mmax <= 4096
 *  dist (S,Nmax,Mmax) float64 or NULL (rows >= n, columns >= m are not written), match (S,Nmax) int32: track index or -1
 *  row_any (S,Nmax) / col_any (S,Mmax) int32 or NULL: 1 when the row / column holds a pair inside the gate, i.e. the value
 *  of `(dist[i, :] <= gate).sum() > 0` / `(dist[:, j] <= gate).sum() > 0` that pub_tracker.py:156,178 test
 * ------------------------------------------------------------------------------------------ */
int shasta_center_greedy_f32(const float* det_xy, const float* trk_xy, const int32_t* det_cat, const int32_t* trk_cat,
                             const float* max_diff, const int32_t* n, const int32_t* m, int scenes, int Nmax, int Mmax,
                             double* dist, int32_t* match, int32_t* row_any, int32_t* col_any, shasta_stream_t stream);

/* The merged tracker of whole scenes in ONE launch (replaces the per-frame loop of tools/nusc_shasta/pub_test.py:88-162 over
 * tools/nusc_shasta/pub_tracker_merged.py:57-225 `PubTrackerMerged.step_centertrack`, one tracker per scene, greedy association,
 * TRK_REF confidence refinement, `newborn` / `dead` suppression, coasting up to max_age).  Detections of all scenes lie in one set
 * of arrays, frame f of scene s = rows frame_off[s][f] .. frame_off[s][f+1]-1 in file order ((scenes, Fmax+1) offsets, frames past
 * n_frames[s] unused); det_cls = index into the class tables (-1: not a tracking class), det_flags bit 0 = 'newborn' in the
 * detection dict, bit 1 = 'dead'; frame_lag (scenes, Fmax) = time_lag of step_centertrack.  Class tables (n_cls <= 8, HOST arrays):
 * gate = NUSCENE_CLS_VELOCITY_ERROR, ref / alpha / beta = TRK_REF.  Outputs per detection: out_status 0 = not in the frame's result
 * (suppressed or unknown class), 1 = took over a track, 2 = new track; out_id = tracking_id (per scene, from 1); out_ref = refined
 * ref_detection_score = the tracking_score of the result row (rows with active == 0 - coasting tracks - are never emitted by
 * pub_test.py).  Row order of a frame's result: class by class, matched detections then new ones, each in file order.  out_err
 * (scenes,): 0, 1 = a frame holds more than 512 detections, 2 = more than 768 tracks alive (the caller falls back to the per-frame
 * path).  plain != 0: the single-list tracker of tools/nusc_shasta/pub_tracker.py:35-210 (eval.py:226-300) instead - all tracking
 * classes form one group (result order: matched detections, then new ones), cls_ref[0] / cls_alpha[0] / cls_beta[0] are its
 * refine_confidence / alpha / beta, a new track's score is its detection_score, coasting leaves scores alone.  Bit-identical to the
 * host trackers of shasta_amd.pub_tracker (float64 arithmetic in the reference's operation order). */
int shasta_track_merged_f64(const double* det_xy, const double* det_vel, const int32_t* det_cls, const double* det_score,
                            const double* det_ref, const int32_t* det_flags, const int32_t* frame_off, const double* frame_lag,
                            const int32_t* n_frames, int scenes, int Fmax, int n_cls, const float* cls_gate, const int32_t* cls_ref,
                            const double* cls_alpha, const double* cls_beta, int max_age, int plain, int32_t* out_status,
                            int32_t* out_id, double* out_ref, int32_t* out_err, shasta_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training path, backward helpers (the nn.Linear layers run on shasta_gemm_strided_f32; the first layer of each pair MLP
 * is factorised over the table rows).  Replaces what torch autograd derives from det3d/models/tracker/shasta.py:241-325 in
 * tools/nusc_shasta/train.py:198-213.
 * ------------------------------------------------------------------------------------------ */
/* factorised first layer of a pair MLP: H[(b,t,d)][:E] = relu(UP[(b,t)][:E] + UC[(b,d)][:E]) (H is (B*T*D, E) dense) */
int shasta_pair_hidden_f32(const float* UP, int ldp, const float* UC, int ldc, int B, int T, int D, int E, float* H,
                           shasta_stream_t stream);
/* its transpose: gUP[(b,t)] = sum_d gZ[(b,t,d)], gUC[(b,d)] = sum_t gZ[(b,t,d)]  (dense (B*T, E) outputs, fixed order) */
int shasta_pair_reduce_f32(const float* gZ, int B, int T, int D, int E, float* gUP, float* gUC, shasta_stream_t stream);
/* One pair MLP behind its factorised first layer, per pair on chip (pair_bwd.hip; det3d/models/tracker/shasta.py:59-92, 286-316):
 * kind 0 = fuse_shape, 1 = fuse_det, 2 = res_coeff; UP (B*T, E1), UC (B*D, E1) dense = the first layer's two table products (bias in
 * UC), E1 = F/8 | 32 | 32 + F/8.  wt = six device pointers {W2, b2, W3, b3, W4, b4} - the later layers' weights and biases as
 * nn.Linear holds them ((out, in) row-major), W4 / b4 NULL for the three-layer MLPs.  forward: out (B*T*D, n_out) with n_out = 1 | 1 | 3.  backward: gout
 * (B*T*D, n_out) -> gUP (B*T, E1), gUC (B*D, E1) (gradients of the first layer's pre-activation summed over the detections / the
 * tracks) and gW = the later layers' gradients as one flat image [gW2 (out, in) | gb2 | gW3 | gb3 | gW4 | gb4] of
 * shasta_pair_mlp_grad_floats() floats; fixed summation order.  feat_dim 64, 256, 320 (shasta_pair_mlp_supported); others:
 * SHASTA_E_UNSUPPORTED (callers keep the dense formulation: shasta_pair_hidden_f32 + shasta_gemm_strided_f32). */
int shasta_pair_mlp_supported(int feat_dim);
int shasta_pair_mlp_grad_floats(int kind, int feat_dim);
size_t shasta_pair_mlp_workspace_bytes(int kind, int feat_dim, int B, int T, int D);
int shasta_pair_mlp_forward_f32(int kind, int feat_dim, const float* UP, const float* UC, const float* const* wt, int B, int T, int D,
                                float* out, shasta_stream_t stream);
int shasta_pair_mlp_backward_f32(int kind, int feat_dim, const float* UP, const float* UC, const float* const* wt, const float* gout,
                                 int B, int T, int D, float* gUP, float* gUC, float* gW, float* ws, size_t ws_bytes,
                                 shasta_stream_t stream);
/* hand-designed residual (shasta.py:277-283) materialised: dist (B,T,ld), denom (>= 2*B*D floats: column norms, scratch) */
int shasta_hand_dist_f32(const float* prev_tab, const float* det_tab, int B, int T, int D, int nf, float* dist, int ld,
                         float* denom, shasta_stream_t stream);
/* gradient of dist w.r.t. table rows row0 .. row0+nrows-1 of both box tables (the anchor rows), accumulated */
int shasta_hand_dist_bwd_f32(const float* gdist, int ldg, const float* prev_tab, const float* det_tab, float* denom, int B,
                             int T, int D, int nf, int row0, int nrows, float* dprev_tab, float* ddet_tab,
                             shasta_stream_t stream);
/* gradient of residual = alpha*fused + beta*dist + omega*shape (shasta.py:319) */
int shasta_combine_bwd_f32(const float* gres, const float* coeff, int ldc, const float* fused, int ldf, const float* shape,
                           int lds, const float* dist, int B, int T, int D, int ld, float* gcoeff, float* gfused,
                           float* gshape, float* gdist, shasta_stream_t stream);
/* The training loss of tools/nusc_shasta/train.py:200-211 on m1 (B, N, N+2), m2 (B, N+2, N), gt (B, N+2, N+2), all dense:
 * sums[0..3] = sum(gt1 . -log(m1 + 1e-10)), sum(gt1), sum(gt2 . -log(m2 + 1e-10)), sum(gt2); sums[4] = the loss; ws: 4 B (N+2) floats.
 * _bwd: g1 = dloss/dm1, g2 = dloss/dm2 scaled by the device scalar gloss[0].  Fixed summation order. */
int shasta_affinity_loss_f32(const float* m1, const float* m2, const float* gt, int B, int N, float* ws, float* sums,
                             shasta_stream_t stream);
int shasta_affinity_loss_bwd_f32(const float* m1, const float* m2, const float* gt, const float* sums, const float* gloss, int B, int N,
                                 float* g1, float* g2, shasta_stream_t stream);
/* gradient of the two softmaxes (shasta.py:324-325) w.r.t. matched (B, N+2, ld) */
int shasta_softmax_bwd_f32(const float* m1, const float* g1, const float* m2, const float* g2, int B, int N, float* gmatched,
                           int ld, shasta_stream_t stream);
/* out[n] = sum_m Y[m][n] (bias gradients); ws (optional, up to 1024*N floats used) lets the rows be split over the chip */
int shasta_colsum_f32(const float* Y, int ldy, int M, int N, float* out, float* ws, size_t ws_bytes, shasta_stream_t stream);
/* columns [c0,c1) of rows of width `cols`: forward out = |x|, backward out = g*sign(x); other columns pass through */
int shasta_abs_f32(const float* x, const float* g, float* out, long n, int cols, int c0, int c1, int backward,
                   shasta_stream_t stream);
/* gradient of shasta_bev_gather_f32 w.r.t. the BEV map, ADDED to dbev (zeroed by the caller).  H * W < 2^17 (the reference's 180 x 180):
 * the terms of a pixel are summed in a fixed order - one workgroup per batch item sorts its N * num_point * 4 contributions by pixel in
 * LDS - so the result is bit-reproducible; larger maps: scatter-add with float atomics (equal to fp32 rounding only). */
int shasta_bev_gather_bwd_f32(const float* dfeat, int B, int H, int W, int C, const float* boxes, int N, int box_stride,
                              int box_batch_stride, int num_point, float pc_x0, float pc_y0, float vs_x, float vs_y,
                              float out_stride, int row_stride, int batch_stride, float* dbev, shasta_stream_t stream);

/* The two 1 GB streams of the anchor backward at small (world x) batch R <= 16 (K % 4 == 0, 16-byte aligned):
 *   shasta_lowrank_outer_f32: dW (H, K) = G^T X, G (R, ldg >= H), X (R, ldx >= K)   -- torch autograd of nn.Linear weight
 *   shasta_smallm_nn_f32:     Y (R rows, stride ldy) (+)= G W, W (H, K)             -- ... of its input
 * fixed-order sums; shasta_smallm_nn_workspace_bytes(R, H, K) bytes of scratch for the second one. */
int shasta_lowrank_outer_f32(const float* G, int ldg, const float* X, int ldx, int R, int H, int K, float* dW, shasta_stream_t stream);
size_t shasta_smallm_nn_workspace_bytes(int R, int H, int K);
int shasta_smallm_nn_f32(const float* G, int ldg, const float* W, int R, int H, int K, float* Y, long ldy, int accumulate, void* workspace,
                         size_t workspace_bytes, shasta_stream_t stream);

/* x *= alpha (gradient averaging over data-parallel ranks) */
int shasta_scale_f32(float* x, long n, float alpha, shasta_stream_t stream);

/* One fused Adam update of a parameter tensor (torch.optim.Adam semantics of tools/nusc_shasta/train.py:147,215: L2
 * weight decay folded into the gradient, bias correction with `step` (1-based), no amsgrad).  16-byte aligned pointers.
 * d_dyn (all four Adam entry points; NULL for the plain form): four DEVICE floats {lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step), beta1,
 * beta2} that replace `lr`, the betas and `step` - a training step replayed from a captured hipGraph cannot take the step number and
 * the scheduler's learning rate / momentum (OneCycleLR cycles both, tools/nusc_shasta/train.py:172) as launch arguments (frozen at
 * capture time).  shasta_adam_prepare_f32 advances a device-side step counter and writes them from d_hyper = {lr, beta1, beta2}
 * (device), once per optimizer step (training.FusedAdam(..., capturable=True)). */
int shasta_adam_prepare_f32(int* d_step, const float* d_hyper, float* d_dyn, shasta_stream_t stream);
int shasta_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int step, const float* d_dyn, shasta_stream_t stream);

/* The same update for `count` tensors (host arrays of device pointers and element counts) that share lr / betas / eps / weight_decay /
 * step: one launch per 48 tensors (the small weight and bias tensors of a model); no alignment requirement. */
int shasta_adam_multi_f32(int count, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                          const long* n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                          const float* d_dyn, shasta_stream_t stream);

/* The same update for an (H, K) matrix whose gradient is the rank-R product g[h][k] = sum_r G[r][h] X[r][k] (G: (R, H) at ldg, X: (R, K) at
 * ldx; the first aug_shape layers of tools/nusc_shasta/train.py:198-218: G = gradient of the hidden activations - already divided by
 * the world size when the factors were gathered over the ranks -, X = the layer's inputs, R = frame-pairs of the step over all ranks,
 * 1..64).  The gradient is formed in registers inside the pass and never touches memory.  K and ldx multiples of 4. */
int shasta_adam_lowrank_f32(float* param, float* exp_avg, float* exp_avg_sq, int H, int K, const float* G, int ldg, const float* X, int ldx,
                            int R, float lr, float beta1, float beta2, float eps, float weight_decay, int step, const float* d_dyn,
                            shasta_stream_t stream);

/* shasta_adam_lowrank_f32 and, in the same pass over the matrix, Y (+)= Gdx . W with the weights as they are BEFORE the update (Gdx: (Rdx, H)
 * at ldgdx, Y: (Rdx, K) at ldy; the backward's dx = ghid . W1 of a first aug_shape layer, which otherwise reads the 1 GB matrix once more:
 * shasta_smallm_nn_f32).  For an optimizer that steps these matrices inside the backward (training.FusedAdam(..., in_backward=True)).
 * 1 <= R <= 64, 1 <= Rdx <= 16; workspace: shasta_adam_lowrank_dx_workspace_bytes(H, K, Rdx). */
size_t shasta_adam_lowrank_dx_workspace_bytes(int H, int K, int Rdx);
int shasta_adam_lowrank_dx_f32(float* param, float* exp_avg, float* exp_avg_sq, int H, int K, const float* G, int ldg, const float* X, int ldx,
                               int R, const float* Gdx, int ldgdx, int Rdx, float* Y, long ldy, int accumulate, void* workspace,
                               size_t workspace_bytes, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               const float* d_dyn, shasta_stream_t stream);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* SHASTA_HIP_H */
