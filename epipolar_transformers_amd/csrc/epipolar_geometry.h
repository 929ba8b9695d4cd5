// Per-pixel / per-sample geometry of the epipolar sampler, shared by every
// kernel (and by the host-side test hook).  float32 with ONE rounding per
// reference ATen op -- this translation unit is compiled with
// -ffp-contract=off and spells out the two places where the reference's CPU
// kernels were measured to fuse (the K=3/K=4 GEMMs and torch.cross).  The
// discontinuous parts (edge validity tests, floor) make bit-faithful
// arithmetic the only way to stay inside the 1e-5 sample-location tolerance.
//
// Restates: modeling/layers/epipolar.py:338-414 (grid2sample_locs),
// vision/multiview.py:25-37,159-163 (normalize, coord2pix) and the
// unnormalise/floor/weight step of ATen's CPU grid_sampler_2d (what
// F.grid_sample at epipolar.py:199,210 runs).
#pragma once

#include <hip/hip_runtime.h>

#include "epipolar_amd.h"

#define ET_HD __host__ __device__ __forceinline__

namespace et {

struct Segment {
    float sx, sy;  // first valid rectangle intersection (image coords)
    float vx, vy;  // second - first
};

ET_HD float sign_of(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// epipolar.py:338-407 for one reference pixel centre (gx, gy).
// cam: ET_CAM_STRIDE floats (P1inv 4x3 | P2 3x4 | e2).
ET_HD Segment epipolar_segment(const EtLayerDesc &d, const float *cam, float gx, float gy)
{
    const float *p1inv = cam, *p2 = cam + 12, *e2 = cam + 24;
    // X = P1inv @ [gx, gy, 1]^T  : ascending-k FMA chain (epipolar.py:338)
    float X[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = p1inv[i * 3 + 0] * gx;
        acc = fmaf(p1inv[i * 3 + 1], gy, acc);
        acc = fmaf(p1inv[i * 3 + 2], 1.0f, acc);
        X[i] = acc;
    }
    // x2 = P2 @ X  (epipolar.py:340), then x2 /= x2.z (epipolar.py:342)
    float x2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc = p2[j * 4 + 0] * X[0];
        acc = fmaf(p2[j * 4 + 1], X[1], acc);
        acc = fmaf(p2[j * 4 + 2], X[2], acc);
        acc = fmaf(p2[j * 4 + 3], X[3], acc);
        x2[j] = acc;
    }
    const float z = x2[2];
    const float ax = x2[0] / z, ay = x2[1] / z, az = x2[2] / z;
    // l = e2 x x2 (epipolar.py:350); a*b - c*d is evaluated as fma(a, b, -(c*d))
    const float ex = e2[0], ey = e2[1], ez = e2[2];
    const float l0 = fmaf(ey, az, -(ez * ay));
    const float l1 = fmaf(ez, ax, -(ex * az));
    const float l2 = fmaf(ex, ay, -(ey * ax));
    // intersections with the four rectangle edges, clamped denominators (epipolar.py:369-373)
    const float d1 = sign_of(l1) * fmaxf(fabsf(l1), d.eps);
    const float d0 = sign_of(l0) * fmaxf(fabsf(l0), d.eps);
    const float by1 = -(d.xmin * l0 + l2) / d1;
    const float by2 = -(d.xmax * l0 + l2) / d1;
    const float bx0 = -(d.ymin * l1 + l2) / d0;
    const float bx3 = -(d.ymax * l1 + l2) / d0;
    // validity on half-open eps-shrunk ranges (epipolar.py:388-393)
    const bool m0 = (bx0 >= d.xmin + d.eps) && (bx0 < d.xmax - d.eps);
    const bool m1 = (by1 > d.ymin + d.eps) && (by1 <= d.ymax - d.eps);
    const bool m2 = (by2 >= d.ymin + d.eps) && (by2 < d.ymax - d.eps);
    const bool m3 = (bx3 > d.xmin + d.eps) && (bx3 <= d.xmax - d.eps);
    const int cnt = (int)m0 + (int)m1 + (int)m2 + (int)m3;
    Segment s;
    if (cnt < 2) {
        // no segment: both ends parked far outside (epipolar.py:50-53,395-403)
        s.sx = d.xmin - 10000.f;
        s.sy = d.ymin - 10000.f;
        s.vx = 0.f;
        s.vy = 0.f;
        return s;
    }
    // first two valid candidates in edge order [bx0@ymin, by1@xmin, by2@xmax, bx3@ymax]
    // (epipolar.py:375-386,402-407); a third valid one (line through a corner) is ignored.
    float ax0, ay0, bx_, by_;
    if (m0) {
        ax0 = bx0; ay0 = d.ymin;
        if (m1)      { bx_ = d.xmin; by_ = by1; }
        else if (m2) { bx_ = d.xmax; by_ = by2; }
        else         { bx_ = bx3;    by_ = d.ymax; }
    } else if (m1) {
        ax0 = d.xmin; ay0 = by1;
        if (m2) { bx_ = d.xmax; by_ = by2; }
        else    { bx_ = bx3;    by_ = d.ymax; }
    } else {
        ax0 = d.xmax; ay0 = by2;
        bx_ = bx3;    by_ = d.ymax;
    }
    s.sx = ax0;
    s.sy = ay0;
    s.vx = bx_ - ax0;
    s.vy = by_ - ay0;
    return s;
}

// Dividing by a power of two is the same float32 operation as multiplying by its (exact) reciprocal,
// bit for bit, and costs one instruction instead of ~10.  The resize / downsample factors of every
// reference config are powers of two (1, 2, 4); kernels that care test this once per launch.
//
// The division by the map size of `normalize` (multiview.py:30-35: / (size - 1), or / size) is NOT by a power of two,
// but it is by a constant c: with r = RN(1 / c) (one true division per thread),
//     q0 = RN(a r),   e = a - q0 c  (exact: one fma),   q = RN(q0 + e r)
// is the correctly rounded quotient RN(a / c) -- Markstein's correction step; checked exhaustively on the CPU
// (tests/test_geometry_division_cpu.py: every float a with an exponent within +-40, every divisor 2 .. 1024 the map sizes
// produce, zero mismatches).  Three instructions instead of the ~13 of the IEEE division sequence, in the per-sample
// chain every kernel evaluates 2 K times per reference pixel.
struct Pow2Recips {
    float resize, predict, down;  // reciprocals, valid when ok
    float div_w, div_h;           // the divisors of normalize for x and y: size - 1 (USE_CORRECT_NORMALIZE) or size
    float rcp_w, rcp_h;           // RN(1 / div_w), RN(1 / div_h)
    bool ok;
};

ET_HD bool is_pow2(float v)
{
    int e;
    return v > 0.f && frexpf(v, &e) == 0.5f && e > -100 && e < 100;
}

ET_HD Pow2Recips pow2_recips(const EtLayerDesc &d)
{
    Pow2Recips r;
    r.ok = is_pow2(d.image_resize) && is_pow2(d.predict_resize) && is_pow2(d.downsample);
    r.resize = 1.f / d.image_resize;
    r.predict = 1.f / d.predict_resize;
    r.down = 1.f / d.downsample;
    r.div_w = d.correct_normalize ? (float)(d.W - 1) : (float)d.W;
    r.div_h = d.correct_normalize ? (float)(d.H - 1) : (float)d.H;
    r.rcp_w = 1.f / r.div_w;
    r.rcp_h = 1.f / r.div_h;
    // (the correction step assumes a normal quotient and divisor: map sizes of 2 .. 16384 and coordinates of at most
    //  ~1e4 pixels are far inside that; a 1-pixel map divides by zero in the reference as well)
    r.ok = r.ok && r.div_w >= 1.f && r.div_h >= 1.f;
    return r;
}

// RN(a / c) for the constant c, rc = RN(1 / c)
ET_HD float div_by_const(float a, float c, float rc)
{
    const float q0 = a * rc;
    const float e = fmaf(-q0, c, a);
    return fmaf(e, rc, q0);
}

// epipolar.py:411-414: /resize, coord2pix (multiview.py:163), normalize (multiview.py:30-35)
template <bool P2>
ET_HD float to_normalized_t(const EtLayerDesc &d, float v, int size, const Pow2Recips &pr, float dv = 0.f, float rc = 0.f)
{
    if (P2) {
        v = v * pr.resize;
        v = v * pr.predict;
        v = (v + 0.5f - d.downsample / 2.0f) * pr.down;
        if (d.correct_normalize) return -1.f + div_by_const(2.f * v, dv, rc);
        return -1.f + div_by_const(2.f * (v + 0.5f), dv, rc);
    } else {
        v = v / d.image_resize;
        v = v / d.predict_resize;
        v = (v + 0.5f - d.downsample / 2.0f) / d.downsample;
    }
    if (d.correct_normalize) return -1.f + 2.f * v / (float)(size - 1);
    return -1.f + 2.f * (v + 0.5f) / (float)size;
}

ET_HD float to_normalized(const EtLayerDesc &d, float v, int size)
{
    return to_normalized_t<false>(d, v, size, Pow2Recips());
}

// de_normalize (multiview.py:50-57), used for corr_pos
ET_HD float de_normalize(const EtLayerDesc &d, float v, int size)
{
    if (d.correct_normalize) return (v + 1.f) * (float)(size - 1) / 2.f;
    return (v + 1.f) * (float)size / 2.f - 0.5f;
}

// ATen CPU grid sampler un-normalisation
ET_HD float unnormalize(float v, int size, int align_corners)
{
    if (align_corners) return (v + 1.f) * ((float)(size - 1) / 2.f);
    return (v + 1.f) * ((float)size / 2.f) - 0.5f;
}

// One sample along the segment.  The four bilinear taps of a sample are kept
// in four "tap registers" addressed by coordinate parity: tap (x, y) lives in
// register r = ((y & 1) << 1) | (x & 1).  The 2x2 footprint of any sample maps
// onto four distinct registers, and consecutive samples (spacing < 1 px) find
// most of their taps already resident -- an exact 2x2 direct-mapped cache that
// removes the re-reads of shared taps.
//
// tap[r] is the linear pixel index (y * W + x) of the tap routed to register r,
// or -1 when that tap falls outside the image; weight[r] is its bilinear
// weight, already zeroed for out-of-image taps (zero padding).
struct SampleSetup {
    int tap[4];
    float weight[4];
    float nx, ny;
};

// normalised location of one sample along the segment (epipolar.py:409-414)
template <bool P2>
ET_HD void sample_location(const EtLayerDesc &d, const Segment &s, float step, const Pow2Recips &pr, float &nx, float &ny)
{
    // start + vec * step (epipolar.py:409): product rounded, then the sum
    const float lx = s.sx + s.vx * step;
    const float ly = s.sy + s.vy * step;
    nx = to_normalized_t<P2>(d, lx, d.W, pr, pr.div_w, pr.rcp_w);
    ny = to_normalized_t<P2>(d, ly, d.H, pr, pr.div_h, pr.rcp_h);
}

ET_HD SampleSetup sample_setup(const EtLayerDesc &d, const Segment &s, float step)
{
    SampleSetup o;
    sample_location<false>(d, s, step, Pow2Recips(), o.nx, o.ny);
    const float x = unnormalize(o.nx, d.W, d.align_corners);
    const float y = unnormalize(o.ny, d.H, d.align_corners);
    const float xw = floorf(x), yn = floorf(y);
    // bilinear weights nw = s*e, ne = s*w, sw = n*e, se = n*w (ATen naming)
    const float w = x - xw, e = 1.f - w, n = y - yn, so = 1.f - n;
    // clamp far-away coordinates before the int conversion; every tap of a
    // clamped cell is still outside the image
    const int x0 = (int)fminf(fmaxf(xw, -2.f), (float)d.W);
    const int y0 = (int)fminf(fmaxf(yn, -2.f), (float)d.H);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int py = r >> 1, px = r & 1;
        const int ty = ((y0 & 1) != py), tx = ((x0 & 1) != px);
        const int xx = x0 + tx, yy = y0 + ty;
        const bool ok = ((unsigned)xx < (unsigned)d.W) && ((unsigned)yy < (unsigned)d.H);
        const float wy = ty ? n : so, wx = tx ? w : e;  // selects, not an indexed array (no scratch)
        o.weight[r] = ok ? wy * wx : 0.f;
        o.tap[r] = ok ? yy * d.W + xx : -1;
    }
    return o;
}

}  // namespace et
