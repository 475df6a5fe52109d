// Planar geometry shared by the rotated-IoU matrix (iou3d.hip) and the rotated NMS (nms.hip): float64 throughout.
#pragma once

namespace shasta {

struct P2 {
    double x, y;
};

__device__ __forceinline__ double shoelace(const P2* p, int n) {
    if (n < 3) return 0.0;
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const P2 a = p[i], b = p[(i + 1) % n];
        s += a.x * b.y - a.y * b.x;
    }
    return s * 0.5;
}

__device__ inline double clip_area(const P2* subj, const P2* clip) {
    P2 buf0[10], buf1[10];
    P2* in = buf0;
    P2* out = buf1;
    int n = 4;
    for (int i = 0; i < 4; ++i) in[i] = subj[i];
    const double sgn = shoelace(clip, 4) >= 0 ? 1.0 : -1.0;
    for (int e = 0; e < 4 && n > 0; ++e) {
        const P2 a = clip[e], b = clip[(e + 1) & 3];
        const double ex = b.x - a.x, ey = b.y - a.y;
        int m = 0;
        for (int j = 0; j < n; ++j) {
            const P2 p = in[j], q = in[(j + 1) % n];
            const double sp = sgn * (ex * (p.y - a.y) - ey * (p.x - a.x));
            const double sq = sgn * (ex * (q.y - a.y) - ey * (q.x - a.x));
            if (sp >= 0) out[m++] = p;
            if ((sp >= 0) != (sq >= 0)) {
                const double t = sp / (sp - sq);
                out[m++] = {p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)};
            }
        }
        P2* tmp = in;
        in = out;
        out = tmp;
        n = m;
    }
    return fabs(shoelace(in, n));
}

}  // namespace shasta
