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
// Later layers run with one LANE per pair on v_mfma_f32_4x4x1_16B_f32 (pair.hip): the four result registers of a lane are
// four output features of its own pair, so a layer's accumulators are, unmoved, the next layer's inputs.
#pragma once

namespace shasta {

#if defined(__HIPCC__)
#define SH_HD __host__ __device__
#else
#define SH_HD
#endif

struct PairDims {
    int F, H1, H2, H3, R1, R2, ET;
    SH_HD constexpr PairDims(int f)
        : F(f), H1(f / 8), H2(f / 16), H3(f / 32), R1(32 + f / 8), R2(8 + f / 32), ET(f / 8 + 32 + f / 8 + 32) {}
};

// layer ids in packed order
enum { L_FS2 = 0, L_FS3, L_FS4, L_RC2, L_RC3, L_FD2, L_FD3, L_COUNT };

struct LayerDesc {
    int hout;  // output features
    int kin;   // input features
};

SH_HD constexpr LayerDesc layer_desc(int F, int l) {
    const PairDims d(F);
    return l == L_FS2   ? LayerDesc{d.H2, d.H1}
           : l == L_FS3 ? LayerDesc{d.H3, d.H2}
           : l == L_FS4 ? LayerDesc{1, d.H3}
           : l == L_RC2 ? LayerDesc{d.R2, d.R1}
           : l == L_RC3 ? LayerDesc{3, d.R2}
           : l == L_FD2 ? LayerDesc{8, 32}
                        : LayerDesc{1, 8};
}

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

// aff layers as bf16 piece fragments (aff_pieces.hip): k steps (16 wide) and feature blocks (32 wide) of the six layers for a
// table of D = N + 2 columns; one fragment = [64 lanes][16 B] = 256 dwords, three pieces per (feature block, k step)
SH_HD constexpr int ap_ksteps(int layer, int D) {
    return layer == 0 ? (D + 15) / 16 : layer == 1 ? 8 : layer == 2 ? 4 : layer == 3 ? 2 : layer == 4 ? 4 : 8;
}
SH_HD constexpr int ap_fblocks(int layer, int D) {
    return layer == 0 ? 4 : layer == 1 ? 2 : layer == 2 ? 1 : layer == 3 ? 2 : layer == 4 ? 4 : (D + 31) / 32;
}
SH_HD constexpr int ap_kin(int layer, int D) { return layer == 0 ? D : layer == 1 ? 128 : layer == 2 ? 64 : layer == 3 ? 32 : layer == 4 ? 64 : 128; }
SH_HD constexpr int ap_out(int layer, int D) { return layer == 0 ? 128 : layer == 1 ? 64 : layer == 2 ? 32 : layer == 3 ? 64 : layer == 4 ? 128 : D; }
SH_HD constexpr size_t ap_layer_offset(int layer, int D) {  // in fragments
    size_t o = 0;
    for (int l = 0; l < layer; ++l) o += (size_t)ap_fblocks(l, D) * ap_ksteps(l, D) * 3;
    return o;
}

// the same layers as fp16 piece fragments (aff_f16.hip): two pieces per (feature block, k step), every output row scaled by a
// power of two; behind the fragments one float per output feature and layer (padded to whole feature blocks): the factor 2^-e that
// undoes the scale.  Offsets in floats from the start of the section.
SH_HD constexpr size_t ap16_frag_offset(int layer, int D) {  // in fragments of 256 floats
    size_t o = 0;
    for (int l = 0; l < layer; ++l) o += (size_t)ap_fblocks(l, D) * ap_ksteps(l, D) * 2;
    return o;
}
SH_HD constexpr size_t ap16_scale_offset(int layer, int D) {
    size_t o = (ap16_frag_offset(6, D) + 2) * 256;  // two spare fragments: the phantom k step behind an odd layer-1 width reads them
    for (int l = 0; l < layer; ++l) o += (size_t)ap_fblocks(l, D) * 32;
    return o;
}
SH_HD constexpr size_t ap16_total(int D) { return ap16_scale_offset(6, D); }

// Packed buffer sections (float offsets).  Dp = padded aff width (multiple of 4).
struct PackedLayout {
    int F, nf, N, D, Dp, E12, ET;
    size_t a4, wemb_prev, wemb_cur, bemb_cur, wbox_prev, wbox_cur, bbox_cur, aff0, affp, p16, p16w, embp, aff16, total;
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
        a4 = o;         o += (size_t)((a4_total(f) + 3) / 4 * 4);  // 4x4x1 MFMA A operands of layers 2-4
        wemb_prev = o;  o += (size_t)E12 * f;          // [E12][F]   fuse_shape.0 / res_coeff.0, prev feature cols
        wemb_cur = o;   o += (size_t)E12 * f;          // [E12][F]   ... cur feature cols
        bemb_cur = o;   o += (size_t)((E12 + 3) / 4 * 4);  // [E12]  fuse_shape.0.bias | res_coeff.0.bias
        wbox_prev = o;  o += (size_t)(d.R1 + 32) * 8;  // [R1+32][8] res_coeff.0 / fuse_det.0 prev box cols (nf used)
        wbox_cur = o;   o += (size_t)(d.R1 + 32) * 8;  // [R1+32][8] ... cur box cols
        bbox_cur = o;   o += (size_t)32;               // [32]       fuse_det.0.bias
        aff0 = o;       o += (size_t)128 * Dp;         // aff.0.weight zero padded to (128, Dp)
        affp = o;       o += ap_layer_offset(6, D) * 256;  // the six aff layers as bf16 piece fragments (aff_pieces.hip)
        p16 = o;        o += (size_t)(2 * 4 * 64 * 4 + 4);  // second layers of the pair MLPs as fp16 piece fragments + 3 exponents (pair_f16.hip)
        p16w = o;       o += (size_t)(2 * 16 * 64 * 4 + 4);  // the same as 32x32x16 fragments (pair_f16w.hip: up to 16 fragments x 2 pieces x 1 KB)
        // wemb_prev / wemb_cur as bf16 piece fragments [side][feature block][k step][piece][64 lanes] x 16 B (embed_rows.hip)
        embp = o;       o += (size_t)2 * ((E12 + 31) / 32) * (f / 16) * 3 * 256;
        aff16 = o;      o += ap16_total(D);  // the six aff layers as fp16 piece fragments + their descale factors (aff_f16.hip)
        total = o;
    }
};

#if defined(__HIPCC__)
// The hand-designed residual of one pair (det3d/models/tracker/shasta.py:277-283) from the hand rows of its track (hp, 16 floats)
// and of its detection (hd: slots 0 - 6 and 8 - 12 of that row), the column norm dnm = max(||.||, 1e-12) of the detection and
// rdn = 1.0f / dnm (IEEE, once per lane).  row_prep writes zeros into the box slots >= num_feats of both rows, so the sum of squares
// runs over all seven slots in the reference's order (x + 0.0 is exact) without a per-slot select.  The division by the
// loop-invariant dnm is a multiplication by rdn with one residual correction (q = d2 rdn; q += fma(-q, dnm, d2) rdn: the correctly
// rounded quotient whenever rdn is the correctly rounded reciprocal - Markstein - in 3 instructions instead of the 11 of the generic
// IEEE sequence); the square root is the hardware's (1 ulp; its operand comes from this library's own cosf / sinf of the yaws).
__device__ __forceinline__ float hand_dist(const float (&hp)[16], const float (&hd)[12], float dnm, float rdn) {
    typedef float hpf2 __attribute__((ext_vector_type(2)));
    hpf2 d01 = hpf2{hp[0], hp[1]} - hpf2{hd[0], hd[1]}, d23 = hpf2{hp[2], hp[3]} - hpf2{hd[2], hd[3]},
         d45 = hpf2{hp[4], hp[5]} - hpf2{hd[4], hd[5]};
    const float d6 = hp[6] - hd[6];
    d01 *= d01;
    d23 *= d23;
    d45 *= d45;
    const float d2 = (((((d01[0] + d01[1]) + d23[0]) + d23[1]) + d45[0]) + d45[1]) + d6 * d6;
    const float q = d2 * rdn;
    const float r = __builtin_fmaf(__builtin_fmaf(-q, dnm, d2), rdn, q);
    const float dim = (__builtin_fabsf(hp[8] - hd[7]) + __builtin_fabsf(hp[9] - hd[8])) + __builtin_fabsf(hp[10] - hd[9]);
    hpf2 cs = hpf2{hp[11], hp[12]} - hpf2{hd[10], hd[11]};
    cs *= cs;
    return (r + dim) + __builtin_amdgcn_sqrtf(cs[0] + cs[1]);
}
#endif

}  // namespace shasta
