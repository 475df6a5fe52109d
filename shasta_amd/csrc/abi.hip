// ABI bookkeeping: version, build string, thread-local last error.
#include "common.hpp"
#include <string.h>

namespace shasta {
static thread_local char g_err[256] = "";
void set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
void set_error_msg(const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); }
}  // namespace shasta

extern "C" int shasta_abi_version(void) { return 15; }  // 15: shasta_gemm_strided_group_f32 (up to 8 products of one shape per launch: the anchor backward); 14: the Adam entry points take d_dyn (step-dependent factors from device memory, shasta_adam_prepare_f32: a training step replayed from a hipGraph); 13: K0 train mode (shasta_bn_*, shasta_conv_wgrad_f16x2, raw conv packs), shasta_voxelize_mean_batch_f32, the voxeliser's cell map is a hash table inside the workspace (cell_map arguments and shasta_voxelize_cell_map_* gone); 12: shasta_aff_status / shasta_forward_status (status word of the one-pass aff kernel, tiles by ticket); 11: shasta_pair_mlp_* (a pair MLP recomputed and back-propagated per pair on chip), shasta_adam_multi_f32, shasta_adam_lowrank_dx_f32, shasta_affinity_loss_f32 / _bwd_f32; 10: shasta_adam_lowrank_f32, SHASTA_OPT_F16X2_AFF / TWO_PASS_AFF / ONE_PASS_AFF, SHASTA_E_UNSUPPORTED, packed buffer + the fp16 aff section; 9: shasta_track_merged_f64 (a scene's merged tracker in one launch); 8: shasta_shared_conv_multi_f32 + the fp16 pack of K0; 7: shasta_weights.aug_shape_aux + shasta_aug_shape_aux_* (the row maxima left the packed buffer); 6: shasta_weights.options replaces the environment switches, hidden visibility; 5: shasta_gemm_nt_pieces_f32; 4: forward_timed takes the pair-kernel events; 3: shasta_center_greedy_f32 row/column flags; 2: training / tracker entry points
#ifndef SHASTA_SOURCE_HASH  // shasta_amd/build.py: sha256 over the sources and the header this library was built from
#define SHASTA_SOURCE_HASH "unknown"
#endif
extern "C" const char* shasta_build_info(void) {
    return "shasta_hip gfx950 fp32 (hipcc " __VERSION__ ") src " SHASTA_SOURCE_HASH;
}
extern "C" const char* shasta_last_error(void) { return shasta::g_err; }
