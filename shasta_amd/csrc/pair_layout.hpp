// Layout of the packed pair-MLP weights, shared by the pack kernel and the pair kernel.
//
// The three pair MLPs of the reference (det3d/models/tracker/shasta.py:59-67 fuse_shape, :78-84 fuse_det,
// :86-92 res_coeff) are applied to every (track t, detection d) pair on an input that is a CONCATENATION of a
// per-track part and a per-detection part.  Their first Linear is therefore separable:
//     W1 . [prev_t ; cur_d] + b1  =  (W1[:, prev cols] . prev_t)  +  (W1[:, cur cols] . cur_d + b1)
// The two halves ("row embeddings" UP[t], UC[d], ET floats per table row) are produced once per table row by the
// generic GEMM; the pair kernel only adds them, applies ReLU and runs the remaining small layers on the matrix
// cores.  The (B, T*D, 2F) pair tensor of the reference (516 MB at N=500) is never materialised.
//
// Later layers run as  out^T[feat][pair] = W[feat][k] . h^T[k][pair]  with v_mfma_f32_16x16x4_f32:
//   A operand = weight fragment (lane l: row i = l&15, k-slot kq = l>>4), B operand = activations of pair l&15,
//   D: lane (pair = l&15, kq = l>>4) register r holds output row 4*kq + r.
// Output feature f of a 16-row block is stored in row 4*(f%4) + f/4, so that lane kq register r holds feature
// kq + 4r: the accumulator registers of one layer are then, unmoved, the B operands of the next layer's k-steps
// (step r covers features 4r..4r+3), and a block with c valid features needs only ceil(c/4) steps.
// The last layer of each MLP uses identity rows (feature f in row f) so its outputs sit in registers 0..2 of the
// kq == 0 lanes.
#pragma once

namespace shasta {

#if defined(__HIPCC__)
#define SH_HD __host__ __device__
#else
#define SH_HD
#endif

SH_HD constexpr int nblk(int h) { return (h + 15) / 16; }
SH_HD constexpr int blk_count(int h, int b) { return (h - 16 * b) < 16 ? (h - 16 * b) : 16; }
SH_HD constexpr int chained_steps(int hprev) { return (hprev / 16) * 4 + ((hprev % 16) + 3) / 4; }

struct PairDims {
    int F, H1, H2, H3, R1, R2, ET;
    SH_HD constexpr PairDims(int f)
        : F(f), H1(f / 8), H2(f / 16), H3(f / 32), R1(32 + f / 8), R2(8 + f / 32), ET(f / 8 + 32 + f / 8 + 32) {}
};

// layer ids in packed order
enum { L_FS2 = 0, L_FS3, L_FS4, L_RC2, L_RC3, L_FD2, L_FD3, L_COUNT };

struct LayerDesc {
    int hout;     // output features
    int kin;      // input features
    int chained;  // 0: input built by VALU (k = kq*S + s), 1: input = previous layer's accumulators
    int final_;   // identity output rows
    SH_HD constexpr int steps() const { return chained ? chained_steps(kin) : kin / 4; }
    SH_HD constexpr int frags() const { return nblk(hout) * steps(); }
};

SH_HD constexpr LayerDesc layer_desc(int F, int l) {
    const PairDims d(F);
    return l == L_FS2   ? LayerDesc{d.H2, d.H1, 0, 0}
           : l == L_FS3 ? LayerDesc{d.H3, d.H2, 1, 0}
           : l == L_FS4 ? LayerDesc{1, d.H3, 1, 1}
           : l == L_RC2 ? LayerDesc{d.R2, d.R1, 0, 0}
           : l == L_RC3 ? LayerDesc{3, d.R2, 1, 1}
           : l == L_FD2 ? LayerDesc{8, 32, 0, 0}
                        : LayerDesc{1, 8, 1, 1};
}

// offset (in 64-float fragments) of layer l's first weight fragment / first bias fragment group
SH_HD constexpr int frag_offset(int F, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += layer_desc(F, i).frags();
    return o;
}
SH_HD constexpr int total_frags(int F) { return frag_offset(F, L_COUNT); }
SH_HD constexpr int bias_offset(int F, int l) {  // in units of 256 floats ([64 lanes][4 regs])
    int o = 0;
    for (int i = 0; i < l; ++i) o += nblk(layer_desc(F, i).hout);
    return o;
}
SH_HD constexpr int total_bias_blocks(int F) { return bias_offset(F, L_COUNT); }

// VALU formulation (pair_valu_kernel): layer l is stored transposed and padded, Wt[kin][HP] followed by bias[HP],
// HP = hout rounded up to 4, so that the HP weights that multiply one input feature are one aligned scalar load.
SH_HD constexpr int vw_hp(int F, int l) { return (layer_desc(F, l).hout + 3) / 4 * 4; }
SH_HD constexpr int vw_size(int F, int l) { return (layer_desc(F, l).kin + 1) * vw_hp(F, l); }
SH_HD constexpr int vw_offset(int F, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += vw_size(F, i);
    return o;
}
SH_HD constexpr int vw_total(int F) { return (vw_offset(F, L_COUNT) + 3) / 4 * 4; }

// 4x4x1 formulation (pair_mfma4_kernel): layer l as A operands of v_mfma_f32_4x4x1_16B_f32.  Per output block ob (4
// features) and k-group kg (4 inputs): 16 floats [i][kk] = W[4*ob + i][4*kg + kk] (zero padded); after all blocks, the
// bias as [ob][i].  Lane l reads the float4 of row i = l & 3.
SH_HD constexpr int a4_nob(int F, int l) { return (layer_desc(F, l).hout + 3) / 4; }
SH_HD constexpr int a4_kg(int F, int l) { return (layer_desc(F, l).kin + 3) / 4; }
SH_HD constexpr int a4_size(int F, int l) { return a4_nob(F, l) * a4_kg(F, l) * 16 + a4_nob(F, l) * 4; }
SH_HD constexpr int a4_offset(int F, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += a4_size(F, i);
    return o;
}
SH_HD constexpr int a4_total(int F) { return a4_offset(F, L_COUNT); }

// Packed buffer sections (float offsets).  Dp = padded aff width (multiple of 4).
struct PackedLayout {
    int F, nf, N, D, Dp, E12, ET;
    size_t frags, biasf, vw, a4, wemb_prev, wemb_cur, bemb_cur, wbox_prev, wbox_cur, bbox_cur, aff0, total;
    SH_HD PackedLayout(int max_obj, int num_feats, int f) {
        const PairDims d(f);
        F = f;
        nf = num_feats;
        N = max_obj;
        D = max_obj + 2;
        Dp = (D + 3) / 4 * 4;
        E12 = d.H1 + d.R1;
        ET = d.ET;
        size_t o = 0;
        frags = o;      o += (size_t)total_frags(f) * 64;
        biasf = o;      o += (size_t)total_bias_blocks(f) * 256;
        vw = o;         o += (size_t)vw_total(f);      // transposed / padded layers 2-4 for the VALU pair kernel
        a4 = o;         o += (size_t)a4_total(f);      // 4x4x1 MFMA A operands of layers 2-4
        wemb_prev = o;  o += (size_t)E12 * f;          // [E12][F]   fuse_shape.0 / res_coeff.0, prev feature cols
        wemb_cur = o;   o += (size_t)E12 * f;          // [E12][F]   ... cur feature cols
        bemb_cur = o;   o += (size_t)((E12 + 3) / 4 * 4);  // [E12]  fuse_shape.0.bias | res_coeff.0.bias
        wbox_prev = o;  o += (size_t)(d.R1 + 32) * 8;  // [R1+32][8] res_coeff.0 / fuse_det.0 prev box cols (nf used)
        wbox_cur = o;   o += (size_t)(d.R1 + 32) * 8;  // [R1+32][8] ... cur box cols
        bbox_cur = o;   o += (size_t)32;               // [32]       fuse_det.0.bias
        aff0 = o;       o += (size_t)128 * Dp;         // aff.0.weight zero padded to (128, Dp)
        total = o;
    }
};

}  // namespace shasta
