// Next row f-2: rotated 3-D IoU / GIoU matrices for the `mot_3d.association` modes asso='iou' / 'giou'.
// Restates mot_3d/utils/geometry.py:161-176 (iou3d), :208-231 (giou3d), :234-237 (PolyArea2D) and
// mot_3d/data_protos/bbox.py:70-84 (box2corners2d) for all (detection, track) pairs at once; the reference evaluates
// them one pair at a time in Python with shapely (mot_3d/association.py:108-120).  float64 like the reference.
// One thread per pair: corners -> Sutherland-Hodgman clip of quad A by quad B (<= 8 vertices) -> shoelace area;
// GIoU additionally the convex hull of the 8 corners (monotone chain) and its area.
#include "common.hpp"
#include "geom2d.hpp"

namespace shasta {

__device__ __forceinline__ void corners2d(const double* b, P2* c) {
    const double x = b[0], y = b[1], o = b[3], l = b[4], w = b[5];
    const double cs = cos(o), sn = sin(o);
    c[0] = {x + cs * l / 2 + sn * w / 2, y + sn * l / 2 - cs * w / 2};
    c[1] = {x + cs * l / 2 - sn * w / 2, y + sn * l / 2 + cs * w / 2};
    c[2] = {2 * x - c[0].x, 2 * y - c[0].y};
    c[3] = {2 * x - c[1].x, 2 * y - c[1].y};
}

__device__ double hull_area8(const P2* a4, const P2* b4) {
    P2 p[8];
    for (int i = 0; i < 4; ++i) {
        p[i] = a4[i];
        p[4 + i] = b4[i];
    }
    for (int i = 1; i < 8; ++i) {  // insertion sort by (x, y)
        const P2 v = p[i];
        int j = i - 1;
        while (j >= 0 && (p[j].x > v.x || (p[j].x == v.x && p[j].y > v.y))) {
            p[j + 1] = p[j];
            --j;
        }
        p[j + 1] = v;
    }
    P2 h[17];
    int k = 0;
    for (int i = 0; i < 8; ++i) {  // lower hull
        while (k >= 2 && (h[k - 1].x - h[k - 2].x) * (p[i].y - h[k - 2].y) - (h[k - 1].y - h[k - 2].y) * (p[i].x - h[k - 2].x) <= 0) --k;
        h[k++] = p[i];
    }
    const int lo = k + 1;
    for (int i = 6; i >= 0; --i) {  // upper hull
        while (k >= lo && (h[k - 1].x - h[k - 2].x) * (p[i].y - h[k - 2].y) - (h[k - 1].y - h[k - 2].y) * (p[i].x - h[k - 2].x) <= 0) --k;
        h[k++] = p[i];
    }
    return fabs(shoelace(h, k - 1));
}

// out[d][t] = 1 - iou3d (mode 0) or 1 - giou3d (mode 1); boxes are [x, y, z, o, l, w, h]
__global__ void iou3d_matrix_kernel(const double* __restrict__ dets, const double* __restrict__ trks, int nd, int nt,
                                    int stride, int mode, double* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nd * nt) return;
    const int d = idx / nt, t = idx - d * nt;
    const double* a = dets + (size_t)d * stride;
    const double* b = trks + (size_t)t * stride;
    P2 ca[4], cb[4];
    corners2d(a, ca);
    corners2d(b, cb);
    const double inter = clip_area(ca, cb);
    const double za = a[2], zb = b[2], ha = a[6], hb = b[6];
    const double d1 = (za + ha / 2) - (zb - hb / 2), d2 = (zb + hb / 2) - (za - ha / 2);
    const double oh = fmax(0.0, fmin(d1, d2));
    const double vola = a[5] * a[4] * ha, volb = b[5] * b[4] * hb;
    double v;
    if (mode == 0) {
        const double ov = inter * oh;
        v = ov / ((vola + volb - ov) + 1e-5);
    } else {
        const double uh = fmax(d1, d2);
        const double I = inter * oh;
        const double U = vola + volb - I;
        const double Cc = hull_area8(ca, cb) * uh;
        v = I / U - (Cc - U) / Cc;
    }
    out[idx] = 1.0 - v;
}

}  // namespace shasta

extern "C" int shasta_iou3d_distance_f64(const double* dets, int num_dets, const double* tracks, int num_tracks, int box_stride,
                                         int giou, double* dist, shasta_stream_t stream) {
    using namespace shasta;
    SHASTA_REQUIRE(num_dets >= 0 && num_tracks >= 0 && box_stride >= 7, "iou3d: bad size");
    if (num_dets == 0 || num_tracks == 0) return SHASTA_OK;
    SHASTA_REQUIRE(dets && tracks && dist, "iou3d: null pointer");
    SHASTA_REQUIRE((long)num_dets * num_tracks < (1L << 31), "iou3d: matrix too large");
    const int total = num_dets * num_tracks;
    hipLaunchKernelGGL(iou3d_matrix_kernel, dim3(cdiv(total, 128)), dim3(128), 0, as_stream(stream), dets, tracks, num_dets,
                       num_tracks, box_stride, giou ? 1 : 0, dist);
    return check_launch("iou3d_matrix");
}
