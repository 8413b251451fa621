// Prompt-unified Encoder (PuE) Gaussian vectors and click disk maps: integer bookkeeping that must be
// bit-exact with the reference (isegm/model/is_vpu_model.py:189-291, isegm/model/ops.py:39-202, 347-379,
// isegm/model/is_model.py:97-121).  Bandwidth-bound pointwise kernels (172.6 KB/img and 1.6 MB/img written).
#include "vpu_common.h"
#include "../../include/vpu_hip.h"

#define ST reinterpret_cast<hipStream_t>(stream)

namespace {

__device__ __forceinline__ bool in_img(int x, int y, int w, int h) { return !(x < 0 || x > w || y < 0 || y > h); }
// python floor division (b > 0)
__device__ __forceinline__ int fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// One block per output row (b, slot).  All row-level decisions are made redundantly by every thread from the same
// scalars, so they are wave-uniform.
template <typename T>
__global__ __launch_bounds__(256) void pue_encode_kernel(const float* __restrict__ points, const int* __restrict__ boxes,
                                                         const float* __restrict__ lut, T* __restrict__ out,
                                                         double* __restrict__ out64, int n, int num_max, int img, int ld) {
    const int slot = blockIdx.x % (2 * num_max), b = blockIdx.x / (2 * num_max);
    const int pol = slot / num_max, within = slot % num_max;
    const int E = 2 * img + 3;
    // kind: 0 not-a-point, 1 click, 2 box
    int kind = 0, x = 0, y = 0, rx = 9, ry = 9, sx = 3, sy = 3, label = pol;
    bool empty = false;  // vectors all zero (corner-drop rule / degenerate box)
    if (within < n) {
        const int i = pol * n + within;
        const float* p = points + ((int64_t)b * 2 * n + i) * 3;
        if (p[2] != -1.0f) {
            kind = 1;
            x = (int)p[0];  // astype(int32): truncation toward zero (ops.py:81)
            y = (int)p[1];
        }
        if (boxes && boxes[b * 5 + 4] == i) {  // is_vpu_model.py:276-277: the box row replaces whatever was there
            const int* bx = boxes + b * 5;
            kind = 2;
            x = bx[0]; y = bx[1];
            label = i < n ? 0 : 1;
            const int w = bx[2], h = bx[3];
            if (bx[0] + bx[1] + w + h == 0) empty = true;
            const int kw = fdiv(w, 2) * 2 - 1, kh = fdiv(h, 2) * 2 - 1;
            rx = fdiv(kw - 1, 2); ry = fdiv(kh - 1, 2);
            sx = fdiv(rx, 3); sy = fdiv(ry, 3);
            // sigma == 0 -> zero vectors (ops.py:150,161).  A negative sigma (w or h in {0,1}) is NOT caught by the
            // reference: that axis comes out empty through the slice arithmetic while the other is still drawn.
            if (sx == 0 || sy == 0) empty = true;
        }
        if (kind != 0 && !empty) {
            const int ulx = x - rx, uly = y - ry, brx = x + rx + 1, bry = y + ry + 1;
            if (!in_img(ulx, uly, img, img) && !in_img(brx, bry, img, img)) empty = true;
        }
    }
    T* o = out + ((int64_t)b * 2 * num_max + slot) * ld;
    double* o64 = out64 ? out64 + ((int64_t)b * 2 * num_max + slot) * E : nullptr;
    for (int j = threadIdx.x; j < ld; j += 256) {
        float v = 0.f;
        if (kind == 0) {
            v = (j == E - 1) ? 1.f : 0.f;
        } else if (j < 2 * img) {
            if (!empty) {
                const bool isx = j < img;
                const int pos = isx ? j : j - img;
                const int c = isx ? x : y, r = isx ? rx : ry, s = isx ? sx : sy;
                const int d = pos - c;
                if (d >= -r && d <= r) {
                    if (kind == 1) v = lut[d + 9];
                    else v = expf(-((float)(d * d)) / (float)(2 * s * s));
                }
            }
        } else if (j < E) {
            v = (j - 2 * img == label) ? 1.f : 0.f;
        }
        o[j] = from_f32<T>(v);
        if (o64 && j < E) o64[j] = (double)v;
    }
}

// One thread per pixel, both polarity channels.  fp32 with separately rounded sub / mul / add (no FMA
// contraction) so that the <= r^2 comparison is bit-exact with torch (ops.py:359-375).
__global__ __launch_bounds__(256) void disk_maps_kernel(const float* __restrict__ points, float* __restrict__ out, int n, int H,
                                                        int W, float r2) {
    extern __shared__ float sp[];  // [2n][2] + 2 counts
    const int b = blockIdx.y;
    int* cnt = reinterpret_cast<int*>(sp + 4 * n);
    if (2 * n <= 64) {
        // the valid clicks of each polarity are compacted to the front of their half (typically 1-5 of 24 slots are in
        // use: the pixel loop below walked all 48 slots, 40 us at ViT-B bs 12); min over the same set -> same bits
        if (threadIdx.x < 64) {
            const int i = threadIdx.x;
            float pr = -1.f, pc = -1.f;
            if (i < 2 * n) {
                pr = points[((int64_t)b * 2 * n + i) * 3 + 0];
                pc = points[((int64_t)b * 2 * n + i) * 3 + 1];
            }
            const bool valid = i < 2 * n && !(fmaxf(pr, pc) < 0.f);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const bool vg = valid && (i >= g * n) && (i < (g + 1) * n);
                const unsigned long long m = __ballot(vg);
                const int idx = __popcll(m & ((1ull << i) - 1ull));
                if (vg) { sp[2 * (g * n + idx)] = pr; sp[2 * (g * n + idx) + 1] = pc; }
                if (i == 0) cnt[g] = __popcll(m);
            }
        }
    } else {
        for (int i = threadIdx.x; i < 2 * n; i += 256) {
            sp[2 * i] = points[((int64_t)b * 2 * n + i) * 3 + 0];
            sp[2 * i + 1] = points[((int64_t)b * 2 * n + i) * 3 + 1];
        }
        if (threadIdx.x < 2) cnt[threadIdx.x] = n;
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int r = pix / W, c = pix % W;
    const float fr = (float)r, fc = (float)c;
    for (int g = 0; g < 2; ++g) {
        float best = 1e6f;
        const int ng = cnt[g];
        for (int i = g * n; i < g * n + ng; ++i) {
            const float pr = sp[2 * i], pc = sp[2 * i + 1];
            if (fmaxf(pr, pc) < 0.f) continue;
            const float dr = __fsub_rn(fr, pr), dc = __fsub_rn(fc, pc);
            const float d = __fadd_rn(__fmul_rn(dr, dr), __fmul_rn(dc, dc));
            best = fminf(best, d);
        }
        out[(((int64_t)b * 2 + g) * H + r) * W + c] = best <= r2 ? 1.f : 0.f;
    }
}


// ------------------------------------------------------------------------------------------------
// Scribble prompts (prompt type 2).
// (1) _guassinvector_scribble (is_vpu_model.py:294-352): after the click rows, the LAST valid positive row of a sample is
//     replaced by that sample's scribble vectors (x profile | y profile | label one-hot 0).  The profiles themselves come
//     from the host: GaussianVector_scribble (ops.py:244-296) is a sequential walk that deletes rows of the point list and
//     draws from Python's global `random` -- its results only match the reference when drawn from that very generator.
// (2) ISModel.draw_scribble (is_model.py:123-146): open poly-line of thickness 3 OR-ed into the positive disk channel.
//     cv2.polylines is not available to pin against; rule (exact integers, shared with the oracle's rasteriser): a pixel is
//     set iff its squared distance to a segment is <= 1.
template <typename T>
__global__ __launch_bounds__(256) void pue_scribble_rows_kernel(const float* __restrict__ points, const double* __restrict__ vec,
                                                                T* __restrict__ out, double* __restrict__ out64, int n,
                                                                int num_max, int img, int ld) {
    const int b = blockIdx.x;
    __shared__ int row_s;
    if (threadIdx.x == 0) {
        int row = -1;
        for (int i = 0; i < n; ++i)
            if (points[((int64_t)b * 2 * n + i) * 3 + 2] != -1.0f) row = i;
        row_s = row;
    }
    __syncthreads();
    const int row = row_s;
    if (row < 0) return;                 // no valid positive click: nothing is replaced (is_vpu_model.py:337-339)
    const int E = 2 * img + 3;
    T* o = out + ((int64_t)b * 2 * num_max + row) * ld;
    double* o64 = out64 ? out64 + ((int64_t)b * 2 * num_max + row) * E : nullptr;
    const double* v = vec + (int64_t)b * 2 * img;
    for (int j = threadIdx.x; j < ld; j += 256) {
        double x = 0.0;
        if (j < 2 * img) x = v[j];
        else if (j == 2 * img) x = 1.0;
        o[j] = from_f32<T>((float)x);
        if (o64 && j < E) o64[j] = x;
    }
}

// ---- OpenCV's ThickLine (thickness 3, LINE_8, shift 0), one block per (segment, sample) ------------------------------------
// cv2.rectangle / cv2.polylines with thickness 3 (is_model.py:109,129) = PolyLine -> ThickLine per segment (modules/imgproc/
// src/drawing.cpp; restated function for function in oracle/vpu_oracle.py, which this code equals bit for bit): the filled
// quadrilateral p +- dp (dp: the perpendicular of length 2^17 / |p1 - p0| in 16.16 fixed point, cvRound of doubles) through
// FillConvexPoly -- its outline by Line2, its inside by the two-edge scan conversion -- plus a filled Circle of radius 2 at the
// flagged ends.  The drawing routines are sequential; here every thread decides ONE pixel of the segment's bounding box (grown
// by 3) with their closed forms: a DDA position is start + k * step, a scan row's edge positions are x0 + (y - y0) * dx inside
// a PHASE (the rows between two edge set-ups; thread 0 runs the set-up state machine over the <= 5 phases of a quadrilateral).
constexpr int TL_SHIFT = 16;
constexpr long long TL_ONE = 1LL << TL_SHIFT;

__device__ __forceinline__ long long tl_cdiv(long long a, long long b) { return a / b; }   // C++ division truncates: as OpenCV's

struct TlEdge {          // Line2 after clipping and ordering
    int ok, xmajor, ecount, ex, ey;    // ex, ey: the rounded end point (drawn first)
    long long s0, t0, step;            // x-major: pixel (s0 + k, (t0 + k step) >> 16); y-major: ((t0 + k step) >> 16, s0 + k)
};
struct TlPhase { int y0, y1; long long x0, dx0, x1, dx1; };
struct TlSeg {
    int quad, nphase, flags, cx[2], cy[2];
    TlEdge e[4];
    TlPhase ph[6];
};

__device__ void tl_clip(long long width, long long height, long long& x1, long long& y1, long long& x2, long long& y2, int& vis) {
    const long long right = width - 1, bottom = height - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    vis = (c1 | c2) == 0;
}

__device__ void tl_edge_setup(TlEdge& e, long long x1, long long y1, long long x2, long long y2, int H, int W) {
    int vis;
    tl_clip((long long)W << TL_SHIFT, (long long)H << TL_SHIFT, x1, y1, x2, y2, vis);
    e.ok = vis;
    if (!vis) return;
    long long dx = x2 - x1, dy = y2 - y1;
    const long long ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    e.xmajor = ax > ay;
    if (ax > ay) {
        if (dx < 0) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; dy = -dy; }
        e.step = tl_cdiv(dy * TL_ONE, ax | 1);
        e.ecount = (int)((x2 - x1) >> TL_SHIFT);
    } else {
        if (dy < 0) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; dx = -dx; }
        e.step = tl_cdiv(dx * TL_ONE, ay | 1);
        e.ecount = (int)((y2 - y1) >> TL_SHIFT);
    }
    x1 += TL_ONE >> 1;
    y1 += TL_ONE >> 1;
    e.ex = (int)((x2 + (TL_ONE >> 1)) >> TL_SHIFT);
    e.ey = (int)((y2 + (TL_ONE >> 1)) >> TL_SHIFT);
    if (e.xmajor) { e.s0 = x1 >> TL_SHIFT; e.t0 = y1; }
    else { e.s0 = y1 >> TL_SHIFT; e.t0 = x1; }
}

// thread 0: everything of ThickLine(p0, p1, thickness 3) that does not depend on the pixel
__device__ void tl_seg_setup(TlSeg& sg, int px0, int py0, int px1, int py1, int flags, int H, int W) {
    const long long p0x = (long long)px0 << TL_SHIFT, p0y = (long long)py0 << TL_SHIFT;
    const long long p1x = (long long)px1 << TL_SHIFT, p1y = (long long)py1 << TL_SHIFT;
    sg.flags = flags;
    sg.cx[0] = px0; sg.cy[0] = py0; sg.cx[1] = px1; sg.cy[1] = py1;      // (p + 2^15) >> 16 of an integer point
    const double inv = 1.0 / (double)TL_ONE;
    const double dx = (double)(p0x - p1x) * inv, dy = (double)(p1y - p0y) * inv;
    double r = dx * dx + dy * dy;
    sg.quad = 0; sg.nphase = 0;
    for (int i = 0; i < 4; ++i) sg.e[i].ok = 0;
    if (!(fabs(r) > 2.220446049250313e-16)) return;      // DBL_EPSILON: a repeated point draws its end caps only
    const long long th = 3LL << (TL_SHIFT - 1);
    r = ((double)th + (double)TL_ONE * 0.5) / sqrt(r);   // (thickness 3 is odd)
    const long long dpx = (long long)rint(dy * r), dpy = (long long)rint(dx * r);     // cvRound: half to even
    long long vx[4] = {p0x + dpx, p0x - dpx, p1x - dpx, p1x + dpx};
    long long vy[4] = {p0y + dpy, p0y - dpy, p1y - dpy, p1y + dpy};
    sg.quad = 1;
    // FillConvexPoly (npts 4, shift 16): outline edges v3->v0, v0->v1, v1->v2, v2->v3
    const long long delta = TL_ONE >> 1;
    long long xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
    int imin = 0;
    for (int i = 0; i < 4; ++i) {
        if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
        ymax = vy[i] > ymax ? vy[i] : ymax;
        xmax = vx[i] > xmax ? vx[i] : xmax;
        xmin = vx[i] < xmin ? vx[i] : xmin;
        const int j = (i + 3) & 3;
        tl_edge_setup(sg.e[i], vx[j], vy[j], vx[i], vy[i], H, W);
    }
    xmin = (xmin + delta) >> TL_SHIFT; xmax = (xmax + delta) >> TL_SHIFT;
    ymin = (ymin + delta) >> TL_SHIFT; ymax = (ymax + delta) >> TL_SHIFT;
    if (xmax < 0 || ymax < 0 || xmin >= W || ymin >= H) return;
    if (ymax > H - 1) ymax = H - 1;
    int e_idx[2] = {imin, imin}, e_di[2] = {1, 3}, e_ye[2] = {(int)ymin, (int)ymin};
    long long e_x[2] = {-TL_ONE, -TL_ONE}, e_dx[2] = {0, 0};
    int edges = 4, y = (int)ymin;
    while (true) {
        for (int i = 0; i < 2; ++i) {
            if (y >= e_ye[i]) {
                int idx0 = e_idx[i];
                const int di = e_di[i];
                int idx = idx0 + di;
                if (idx >= 4) idx -= 4;
                for (; edges-- > 0;) {
                    const int ty = (int)((vy[idx] + delta) >> TL_SHIFT);
                    if (ty > y) {
                        const long long xs = vx[idx0], xe = vx[idx];
                        e_ye[i] = ty;
                        e_dx[i] = tl_cdiv((xe - xs) * 2 + (ty - y), 2LL * (ty - y));
                        e_x[i] = xs;
                        e_idx[i] = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= 4) idx -= 4;
                }
            }
        }
        if (edges < 0) break;
        // rows [y, yn) share this set-up: the next one happens at the smaller ye (both > y now, or the edge walker is spent)
        int yn = e_ye[0] < e_ye[1] ? e_ye[0] : e_ye[1];
        if (yn <= y) yn = y + 1;
        if (yn > (int)ymax + 1) yn = (int)ymax + 1;
        TlPhase& p = sg.ph[sg.nphase++];
        p.y0 = y; p.y1 = yn; p.x0 = e_x[0]; p.dx0 = e_dx[0]; p.x1 = e_x[1]; p.dx1 = e_dx[1];
        e_x[0] += e_dx[0] * (yn - y);
        e_x[1] += e_dx[1] * (yn - y);
        y = yn;
        if (y > (int)ymax || sg.nphase >= 6) break;
    }
}

__device__ __forceinline__ bool tl_hit(const TlSeg& sg, int x, int y, int W) {
    // end caps: Circle(center, 2, fill): rows cy +- dy hold |x - cx| <= dx, rows cy +- dx hold |x - cx| <= dy, over the midpoint
    // iterations (dx, dy) = (2, 0), (1, 1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (sg.flags & (i + 1)) {
            const int ax = abs(x - sg.cx[i]), ay = abs(y - sg.cy[i]);
            int err = 0, dx = 2, dy = 0, plus = 1, minus = 3;
            while (dx >= dy) {
                if ((ay == dy && ax <= dx) || (ay == dx && ax <= dy)) return true;
                ++dy; err += plus; plus += 2;
                if (err > 0) { err -= minus; --dx; minus -= 2; }
            }
        }
    if (!sg.quad) return false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const TlEdge& e = sg.e[i];
        if (!e.ok) continue;
        if (x == e.ex && y == e.ey) return true;
        const long long k = (e.xmajor ? x : y) - e.s0;
        if (k >= 0 && k <= e.ecount && ((e.t0 + k * e.step) >> TL_SHIFT) == (e.xmajor ? y : x)) return true;
    }
    for (int i = 0; i < sg.nphase; ++i) {
        const TlPhase& p = sg.ph[i];
        if (y >= p.y0 && y < p.y1) {
            if (y < 0) return false;
            const long long a = p.x0 + p.dx0 * (y - p.y0), b = p.x1 + p.dx1 * (y - p.y0);
            const long long l = a > b ? b : a, r = a > b ? a : b;
            int xx1 = (int)((l + (TL_ONE >> 1)) >> TL_SHIFT), xx2 = (int)((r + (TL_ONE >> 1)) >> TL_SHIFT);
            if (!(xx2 >= 0 && xx1 < W)) return false;
            return x >= xx1 && x <= xx2;
        }
    }
    return false;
}

// the threads of a block walk the segment's bounding box grown by 3 pixels (half width 2 + the roundings)
__device__ void tl_draw_segment(TlSeg& sg, int x0, int y0, int x1, int y1, int flags, float* __restrict__ plane, int H, int W) {
    if (threadIdx.x == 0) tl_seg_setup(sg, x0, y0, x1, y1, flags, H, W);
    __syncthreads();
    const long long lx = (long long)min(x0, x1) - 3, hx = (long long)max(x0, x1) + 3;
    const long long ly = (long long)min(y0, y1) - 3, hy = (long long)max(y0, y1) + 3;
    const int bx0 = (int)max(lx, 0LL), bx1 = (int)min(hx, (long long)W - 1);
    const int by0 = (int)max(ly, 0LL), by1 = (int)min(hy, (long long)H - 1);
    if (bx1 < bx0 || by1 < by0) return;
    const int bw = bx1 - bx0 + 1, npx = bw * (by1 - by0 + 1);
    for (int i = threadIdx.x; i < npx; i += blockDim.x) {
        const int x = bx0 + i % bw, y = by0 + i / bw;
        if (tl_hit(sg, x, y, W)) plane[(int64_t)y * W + x] = 1.0f;
    }
}

// cv2.polylines(image, [curve], False, 255, 3): segment seg = points seg, seg + 1; end caps at both ends of the first
// segment, at the end of every other one (PolyLine's flags).  A one-point curve draws nothing.
__global__ __launch_bounds__(256) void draw_polyline_kernel(const int* __restrict__ curve, float* __restrict__ disks, int P,
                                                            int H, int W) {
    __shared__ TlSeg sg;
    const int seg = blockIdx.x, b = blockIdx.y;
    if (seg + 1 >= P) return;
    const int* c = curve + ((int64_t)b * P + seg) * 2;
    // channel 0: always the positive channel (is_model.py:124,145)
    tl_draw_segment(sg, c[0], c[1], c[2], c[3], seg == 0 ? 3 : 2, disks + (int64_t)b * 2 * H * W, H, W);
}

// cv2.rectangle((x0, y0), (x1, y1), 255, 3) = the closed poly-line (x0,y0) (x1,y0) (x1,y1) (x0,y1): segment s runs from corner
// s - 1 to corner s with an end cap at corner s (is_model.py:97-121; python floor division for the half sizes)
__global__ __launch_bounds__(256) void draw_box_kernel(const int* __restrict__ boxes, float* __restrict__ disks, int n, int H, int W) {
    __shared__ TlSeg sg;
    const int seg = blockIdx.x, b = blockIdx.y;
    const int* bx = boxes + b * 5;
    const int ch = bx[4] < n ? 0 : 1;
    const int hw = fdiv(bx[2], 2), hh = fdiv(bx[3], 2);
    const int x0 = bx[0] - hw, x1 = bx[0] + hw, y0 = bx[1] - hh, y1 = bx[1] + hh;
    const int cx[4] = {x0, x1, x1, x0}, cy[4] = {y0, y0, y1, y1};
    const int j = (seg + 3) & 3;
    tl_draw_segment(sg, cx[j], cy[j], cx[seg], cy[seg], 2, disks + ((int64_t)b * 2 + ch) * H * W, H, W);
}

// ------------------------------------------------------------------------------------------------
// NoBRS loop bookkeeping on the device (isegm/inference/transforms/zoom_in.py:30-165, isegm/inference/clicker.py:29-56):
// the previous prediction never leaves the GPU; these two reductions replace the host passes over it.
// (1) bounding box of {prob > thr} joined with the positive clicks (the reference sets those pixels in a copy of the mask
//     before taking the box: same box): out[b] = {count of mask pixels, rmin, rmax, cmin, cmax}.
// (2) arg-max with numpy's tie rule (first index in raster order) of dist * keep over each of P planes, as ONE 64-bit key
//     per plane: (float bits of the maximum) << 32 | (0xFFFFFFFF - linear index) -- distances are >= 0, so their bit
//     patterns order like the values, and the larger low half is the smaller index.
__global__ __launch_bounds__(256) void mask_bbox_kernel(const float* __restrict__ prob, float thr, int* __restrict__ out,
                                                        int H, int W) {
    const int b = blockIdx.y;
    const float* p = prob + (int64_t)b * H * W;
    int cnt = 0, r0 = H, r1 = -1, c0 = W, c1 = -1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        if (p[i] > thr) {
            const int r = i / W, c = i - r * W;
            ++cnt; r0 = min(r0, r); r1 = max(r1, r); c0 = min(c0, c); c1 = max(c1, c);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, 64);
        r0 = min(r0, __shfl_xor(r0, o, 64)); r1 = max(r1, __shfl_xor(r1, o, 64));
        c0 = min(c0, __shfl_xor(c0, o, 64)); c1 = max(c1, __shfl_xor(c1, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && cnt > 0) {
        int* o5 = out + b * 5;
        atomicAdd(o5, cnt); atomicMin(o5 + 1, r0); atomicMax(o5 + 2, r1); atomicMin(o5 + 3, c0); atomicMax(o5 + 4, c1);
    }
}
__global__ void mask_bbox_init_kernel(int* __restrict__ out, const int* __restrict__ clicks, int nclicks, int B, int H, int W) {
    const int b = threadIdx.x;
    if (b >= B) return;
    int r0 = H, r1 = -1, c0 = W, c1 = -1;
    for (int i = 0; i < nclicks; ++i) {        // positive clicks of this image: (row, col) pairs
        const int r = clicks[2 * i], c = clicks[2 * i + 1];
        r0 = min(r0, r); r1 = max(r1, r); c0 = min(c0, c); c1 = max(c1, c);
    }
    int* o5 = out + b * 5;
    o5[0] = 0; o5[1] = r0; o5[2] = r1; o5[3] = c0; o5[4] = c1;
}

// false-negative / false-positive masks of one prediction against the ground truth, restricted to the labelled pixels
// (clicker.py:30-31): out[0] = gt & !pred & valid, out[1] = !gt & pred & valid
__global__ __launch_bounds__(256) void error_masks_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ gt,
                                                          const uint8_t* __restrict__ valid, uint8_t* __restrict__ out, int HW) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const bool p = pred[i] != 0, g = gt[i] != 0, v = valid[i] != 0;
    out[i] = (g && !p && v) ? 1 : 0;
    out[HW + i] = (!g && p && v) ? 1 : 0;
}

__global__ __launch_bounds__(256) void masked_argmax_kernel(const float* __restrict__ dist, const uint8_t* __restrict__ keep,
                                                            unsigned long long* __restrict__ out, int HW) {
    const int pl = blockIdx.y;
    const float* d = dist + (int64_t)pl * HW;
    unsigned long long best = 0ull;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const float v = keep[i] ? d[i] : 0.0f;                       // (dist * not_clicked_map: a clicked pixel counts as 0)
        const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (0xFFFFFFFFu - (unsigned)i);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(best, o, 64);
        best = t > best ? t : best;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out + pl, best);
}

// ------------------------------------------------------------------------------------------------
// Exact Euclidean distance transform: distance of every non-zero pixel to the nearest zero pixel (0 at zero pixels) --
// what the click simulators take the arg-max of (isegm/inference/clicker.py:29-56 cv2.distanceTransform(DIST_L2, 0),
// isegm/engine/trainer.py:628-629, 673-674, 736-737; scipy.ndimage.distance_transform_edt in this build's mirror).
// Two separable passes on integers (exact): column pass g = vertical distance to the nearest zero, row pass
// d2[x] = min_x' (x - x')^2 + g[x']^2 by direct search (W <= 1024: ~90 M integer ops for a 448 x 448 mask);
// dist = float(sqrt(double(d2))) -- bit-identical to the float64 transform cast to float32.
// border: the image is surrounded by zero pixels (the callers' np.pad(mask, 1) without materialising it).
constexpr int EDT_INF = 1 << 28;

__global__ __launch_bounds__(256) void edt_cols_kernel(const uint8_t* __restrict__ mask, int* __restrict__ g, int H, int W,
                                                       int border) {
    const int x = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (x >= W) return;
    const uint8_t* m = mask + (int64_t)b * H * W + x;
    int* gg = g + (int64_t)b * H * W + x;
    int d = border ? 0 : EDT_INF;
    for (int y = 0; y < H; ++y) {
        d = m[(int64_t)y * W] ? (d >= EDT_INF ? EDT_INF : d + 1) : 0;
        gg[(int64_t)y * W] = d;
    }
    d = border ? 0 : EDT_INF;
    for (int y = H - 1; y >= 0; --y) {
        d = m[(int64_t)y * W] ? (d >= EDT_INF ? EDT_INF : d + 1) : 0;
        const int o = gg[(int64_t)y * W];
        gg[(int64_t)y * W] = d < o ? d : o;
    }
}

__global__ __launch_bounds__(256) void edt_rows_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ g,
                                                       float* __restrict__ dist, int H, int W, int border) {
    extern __shared__ int gsq[];   // [W]
    const int y = blockIdx.x, b = blockIdx.y;
    const int64_t row = ((int64_t)b * H + y) * W;
    for (int x = threadIdx.x; x < W; x += 256) {
        const int v = g[row + x];
        gsq[x] = v >= 32768 ? EDT_INF : v * v;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        float out = 0.f;
        if (mask[row + x]) {
            int best = EDT_INF;
            if (border) {
                const int l = (x + 1) * (x + 1), r = (W - x) * (W - x);
                best = l < r ? l : r;
            }
            for (int xp = 0; xp < W; ++xp) {
                const int dx = x - xp;
                const int c = dx * dx + gsq[xp];
                best = c < best ? c : best;
            }
            out = best >= EDT_INF ? INFINITY : (float)sqrt((double)best);
        }
        dist[row + x] = out;
    }
}


// ------------------------------------------------------------------------------------------------
// 8-connected component labelling of B masks by union-find on the pixel grid: root[i] = smallest linear index of the
// component of foreground pixel i (-1 on background).  Components ordered by their root are in the raster order of their
// first pixel, i.e. the label order of scipy.ndimage.label / skimage.measure.label that max_connected_regions
// (isegm/engine/trainer.py:1175-1190) scans.  The result does not depend on the order in which the unions happen.
__device__ __forceinline__ int cc_find(int* __restrict__ lab, int i) {
    // path halving: every visited node is re-pointed to its grandparent (parents only ever decrease towards the root,
    // so a concurrent writer can at worst install another valid ancestor)
    int r = i;
    while (true) {
        const int p = __hip_atomic_load(lab + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == r) return r;
        const int gp = __hip_atomic_load(lab + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_store(lab + r, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r = p;
    }
}
__device__ __forceinline__ void cc_union(int* __restrict__ lab, int a, int b) {
    while (true) {
        a = cc_find(lab, a);
        b = cc_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }          // a > b: hang the larger root under the smaller
        const int old = atomicMin(lab + a, b);
        if (old == a) return;                                  // a was still a root: linked
        a = old;                                               // somebody re-parented a meanwhile: merge that parent with b
    }
}
// one block per image row: every foreground pixel starts as a child of the FIRST pixel of its horizontal run (inclusive
// max-scan of "index after the last background pixel"), so the union step only has to join runs of adjacent rows
__global__ __launch_bounds__(256) void cc_init_kernel(const uint8_t* __restrict__ mask, int* __restrict__ lab, int H, int W) {
    __shared__ int sc[256];
    const int64_t row = ((int64_t)blockIdx.y * H + blockIdx.x) * W;
    int carry = 0;                                             // run start candidate carried over from the previous chunk
    for (int x0 = 0; x0 < W; x0 += 256) {
        const int x = x0 + threadIdx.x;
        const bool fg = x < W && mask[row + x];
        int v = (x < W && !fg) ? x + 1 : 0;                    // a background pixel at x: later runs start at >= x + 1
        sc[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int t = threadIdx.x >= o ? sc[threadIdx.x - o] : 0;
            __syncthreads();
            v = v > t ? v : t;
            sc[threadIdx.x] = v;
            __syncthreads();
        }
        v = v > carry ? v : carry;
        if (x < W) lab[row + x] = fg ? (int)(row + v) : -1;
        carry = sc[255] > carry ? sc[255] : carry;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void cc_merge_kernel(const uint8_t* __restrict__ mask, int* __restrict__ lab, int H, int W) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= W || y == 0) return;
    const int64_t base = (int64_t)b * H * W;
    const int i = (int)(base + (int64_t)y * W + x);
    if (!mask[i]) return;
    // join this pixel's run with the runs of the row above it touches (N, NW, NE); a run of the row above is joined once
    // per run of this row: skip a neighbour whose left neighbour (same upper run) was already seen from this run
    const bool left_fg = x > 0 && mask[i - 1];
    const bool n = mask[i - W], nw = x > 0 && mask[i - W - 1], ne = x + 1 < W && mask[i - W + 1];
    if (n && !(left_fg && nw)) cc_union(lab, i, i - W);
    else if (!n && nw && !left_fg) cc_union(lab, i, i - W - 1);
    if (ne && !n) cc_union(lab, i, i - W + 1);
}
__global__ __launch_bounds__(256) void cc_flatten_kernel(int* __restrict__ lab, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && lab[i] >= 0) {
        int r = (int)i;
        while (lab[r] != r) r = lab[r];     // (read-only walk: the unions are complete)
        lab[i] = r;
    }
}

// ---- per-component table over the labels of cc_roots: (root, pixel count, ymin, ymax, xmin, xmax) per component, in
// no particular order (the host sorts the few rows by root = label order).  Integer atomics only: exact, order-free.
__global__ __launch_bounds__(256) void cc_table_roots_kernel(const int* __restrict__ roots, int* __restrict__ slots,
                                                             int* __restrict__ table, int* __restrict__ counter, int64_t n, int kmax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && roots[i] == (int)i) {
        const int s = atomicAdd(counter, 1);
        slots[i] = s < kmax ? s : -1;
        if (s < kmax) {
            int* t = table + 6 * s;
            t[0] = (int)i; t[1] = 0; t[2] = 0x7FFFFFFF; t[3] = -1; t[4] = 0x7FFFFFFF; t[5] = -1;
        }
    }
}
// one wave per 64 consecutive pixels of a row: a run of one component (the common case) costs five atomics per wave
__global__ __launch_bounds__(256) void cc_table_accum_kernel(const int* __restrict__ roots, const int* __restrict__ slots,
                                                             int* __restrict__ table, int H, int W) {
    const int y = blockIdx.y, b = blockIdx.z, x = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const int64_t i = ((int64_t)b * H + y) * W + x;
    const int r = x < W ? roots[i] : -1;
    const int s = r >= 0 ? slots[r] : -1;
    const unsigned long long fg = __builtin_amdgcn_ballot_w64(s >= 0);
    if (fg == 0) return;
    const int first = __builtin_ctzll(fg), last = 63 - __builtin_clzll(fg);
    const int s0 = __shfl(s, first, 64);
    const bool uniform = __builtin_amdgcn_ballot_w64(s >= 0 && s != s0) == 0;
    if (uniform) {
        if (lane == first) {
            int* t = table + 6 * s0;
            atomicAdd(t + 1, __builtin_popcountll(fg));
            atomicMin(t + 2, y); atomicMax(t + 3, y);
            atomicMin(t + 4, x); atomicMax(t + 5, x - first + last);
        }
    } else if (s >= 0) {
        int* t = table + 6 * s;
        atomicAdd(t + 1, 1);
        atomicMin(t + 2, y); atomicMax(t + 3, y);
        atomicMin(t + 4, x); atomicMax(t + 5, x);
    }
}

}  // namespace

extern "C" int vpu_pue_encode(const float* points, const int32_t* boxes, const float* lut, void* out, double* out64,
                              int32_t B, int32_t n, int32_t num_max, int32_t img, int32_t ld, int32_t dtype,
                              void* stream) {
    vpu_clear_stale_error();
    if (n > num_max || ld < 2 * img + 3 || B <= 0) { vpu_set_error("pue_encode: n <= num_max, ld >= 2*img+3"); return VPU_ERR_ARG; }
    const unsigned grid = (unsigned)(B * 2 * num_max);
    if (dtype == VPU_BF16)
        pue_encode_kernel<bf16_t><<<grid, 256, 0, ST>>>(points, boxes, lut, (bf16_t*)out, out64, n, num_max, img, ld);
    else if (dtype == VPU_F32)
        pue_encode_kernel<float><<<grid, 256, 0, ST>>>(points, boxes, lut, (float*)out, out64, n, num_max, img, ld);
    else { vpu_set_error("pue_encode: dtype"); return VPU_ERR_ARG; }
    return vpu_check_launch("vpu_pue_encode");
}

extern "C" int vpu_pue_scribble_rows(const float* points, const double* vec, void* out, double* out64, int32_t B, int32_t n,
                                     int32_t num_max, int32_t img, int32_t ld, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (!points || !vec || !out || n > num_max || ld < 2 * img + 3 || B <= 0) {
        vpu_set_error("pue_scribble_rows: non-null operands, n <= num_max, ld >= 2*img+3");
        return VPU_ERR_ARG;
    }
    if (dtype == VPU_BF16) pue_scribble_rows_kernel<bf16_t><<<B, 256, 0, ST>>>(points, vec, (bf16_t*)out, out64, n, num_max, img, ld);
    else if (dtype == VPU_F32) pue_scribble_rows_kernel<float><<<B, 256, 0, ST>>>(points, vec, (float*)out, out64, n, num_max, img, ld);
    else { vpu_set_error("pue_scribble_rows: dtype"); return VPU_ERR_ARG; }
    return vpu_check_launch("vpu_pue_scribble_rows");
}

extern "C" int vpu_draw_polyline(const int32_t* curve, float* disks, int32_t B, int32_t P, int32_t H, int32_t W, void* stream) {
    vpu_clear_stale_error();
    if (!curve || !disks || B <= 0 || P <= 0 || P > 65535 || H <= 0 || W <= 0) { vpu_set_error("draw_polyline: sizes"); return VPU_ERR_ARG; }
    if (P < 2) return VPU_OK;        // (PolyLine draws segments: a single point draws nothing)
    draw_polyline_kernel<<<dim3((unsigned)(P - 1), (unsigned)B), 256, 0, ST>>>(curve, disks, P, H, W);
    return vpu_check_launch("vpu_draw_polyline");
}

extern "C" int vpu_mask_bbox(const float* prob, float thr, const int32_t* pos_clicks, int32_t nclicks, int32_t* out, int32_t B,
                             int32_t H, int32_t W, void* stream) {
    vpu_clear_stale_error();
    if (!prob || !out || B < 1 || B > 64 || H < 1 || W < 1 || nclicks < 0 || (nclicks > 0 && !pos_clicks)) {
        vpu_set_error("mask_bbox: sizes (1 <= B <= 64)");
        return VPU_ERR_ARG;
    }
    mask_bbox_init_kernel<<<1, 64, 0, ST>>>(out, pos_clicks, nclicks, B, H, W);
    const int nb = (H * W + 2047) / 2048;
    mask_bbox_kernel<<<dim3((unsigned)(nb < 128 ? nb : 128), (unsigned)B), 256, 0, ST>>>(prob, thr, out, H, W);
    return vpu_check_launch("vpu_mask_bbox");
}

extern "C" int vpu_error_masks(const uint8_t* pred, const uint8_t* gt, const uint8_t* valid, uint8_t* out, int32_t H, int32_t W,
                               void* stream) {
    vpu_clear_stale_error();
    if (!pred || !gt || !valid || !out || H < 1 || W < 1) { vpu_set_error("error_masks: sizes"); return VPU_ERR_ARG; }
    error_masks_kernel<<<(H * W + 255) / 256, 256, 0, ST>>>(pred, gt, valid, out, H * W);
    return vpu_check_launch("vpu_error_masks");
}

extern "C" int vpu_masked_argmax(const float* dist, const uint8_t* keep, uint64_t* out, int32_t planes, int32_t H, int32_t W,
                                 void* stream) {
    vpu_clear_stale_error();
    if (!dist || !keep || !out || planes < 1 || H < 1 || W < 1) { vpu_set_error("masked_argmax: sizes"); return VPU_ERR_ARG; }
    if (hipMemsetAsync(out, 0, sizeof(uint64_t) * planes, ST) != hipSuccess) { vpu_set_error("masked_argmax: memset"); return VPU_ERR_LAUNCH; }
    const int nb = (H * W + 2047) / 2048;
    masked_argmax_kernel<<<dim3((unsigned)(nb < 128 ? nb : 128), (unsigned)planes), 256, 0, ST>>>(
        dist, keep, reinterpret_cast<unsigned long long*>(out), H * W);
    return vpu_check_launch("vpu_masked_argmax");
}

extern "C" int vpu_disk_maps(const float* points, const int32_t* boxes, float* out, int32_t B, int32_t n, int32_t H,
                             int32_t W, float radius, void* stream) {
    vpu_clear_stale_error();
    if (B <= 0 || n <= 0 || n > 1024) { vpu_set_error("disk_maps: sizes"); return VPU_ERR_ARG; }
    dim3 grid((H * W + 255) / 256, B);
    disk_maps_kernel<<<grid, 256, 2 * n * 2 * sizeof(float) + 2 * sizeof(int), ST>>>(points, out, n, H, W, radius * radius);
    if (boxes) draw_box_kernel<<<dim3(4, (unsigned)B), 256, 0, ST>>>(boxes, out, n, H, W);    // cv2.rectangle(..., 3) on top
    return vpu_check_launch("vpu_disk_maps");
}

extern "C" int vpu_edt(const uint8_t* mask, int32_t* scratch, float* dist, int32_t B, int32_t H, int32_t W,
                       int32_t zero_border, void* stream) {
    vpu_clear_stale_error();
    if (!mask || !scratch || !dist || B < 1 || H < 1 || W < 1 || W > 8192 || H > 32767) {
        vpu_set_error("edt: null pointer, or size out of range (W <= 8192, H <= 32767)");
        return VPU_ERR_ARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    edt_cols_kernel<<<dim3((W + 255) / 256, B), 256, 0, s>>>(mask, scratch, H, W, zero_border);
    edt_rows_kernel<<<dim3(H, B), 256, (size_t)W * sizeof(int), s>>>(mask, scratch, dist, H, W, zero_border);
    return vpu_check_launch("vpu_edt");
}

// ------------------------------------------------------------------------------------------------
// 5 x 5 chamfer distance transform: cv2.distanceTransform(mask, DIST_L2, 5) of the TRAINING simulators
// (isegm/engine/trainer.py:628-629, 673-674, 736-737), restated from OpenCV's published two-pass algorithm
// (distanceTransform_5x5: 16-bit fixed point, a = 65536, b = 91750, c = 143976; see oracle/vpu_oracle.py::chamfer_l2_5x5, which
// this kernel equals bit for bit).  One workgroup per mask, one thread per column; the rows of a pass are sequential, a row
// is t[j] = min(cand[j], t[j-1] + a) = a j + prefix-min(cand[k] - a k): a block-wide min-scan per row.
// ------------------------------------------------------------------------------------------------
constexpr int CH_A = 65536, CH_B = 91750, CH_C = 143976, CH_INF = 0x3FFFFFFF;

__device__ __forceinline__ int block_scan_min(int v, int* wsum /* [16] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v = v < t ? v : t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int pre = CH_INF * 2;
    for (int w = 0; w < wave; ++w) pre = pre < wsum[w] ? pre : wsum[w];
    __syncthreads();
    return v < pre ? v : pre;
}

// rows live in LDS as [Wp + 4] ints with two "infinite" columns on either side
__global__ __launch_bounds__(1024) void chamfer5_kernel(const uint8_t* __restrict__ mask, int* __restrict__ scratch,
                                                        float* __restrict__ dist, int H, int W, int pad) {
    extern __shared__ int sm[];          // r1 | r2 | cur, each Wp + 4
    __shared__ int wsum[16];
    const int Hp = H + 2 * pad, Wp = W + 2 * pad, LD = Wp + 4;
    int* r1 = sm;                        // the previous row of the pass (i - 1 forward, i + 1 backward)
    int* r2 = sm + LD;                   // the one before it
    int* cur = sm + 2 * LD;
    const int b = blockIdx.x, j = threadIdx.x;
    const uint8_t* m = mask + (int64_t)b * H * W;
    int* T = scratch + (int64_t)b * Hp * Wp;
    for (int i = j; i < 3 * LD; i += blockDim.x) sm[i] = CH_INF;
    __syncthreads();
    // ---- forward pass
    for (int i = 0; i < Hp; ++i) {
        int v = CH_INF * 2;              // (threads beyond the row do not disturb the scan)
        const bool live = j < Wp;
        if (live) {
            const int y = i - pad, x = j - pad;
            const bool fg = y >= 0 && y < H && x >= 0 && x < W && m[(int64_t)y * W + x] != 0;
            int cand = 0;
            if (fg) {
                const int* p1 = r1 + 2 + j;
                const int* p2 = r2 + 2 + j;
                int t = p2[-1] + CH_C, u = p2[1] + CH_C;
                t = t < u ? t : u; u = p1[-2] + CH_C; t = t < u ? t : u; u = p1[2] + CH_C; t = t < u ? t : u;
                u = p1[-1] + CH_B; t = t < u ? t : u; u = p1[1] + CH_B; t = t < u ? t : u; u = p1[0] + CH_A; t = t < u ? t : u;
                cand = t;
            }
            v = cand - CH_A * j;
        }
        v = block_scan_min(v, wsum);
        if (live) {
            const int t = v + CH_A * j;
            cur[2 + j] = t;
            T[(int64_t)i * Wp + j] = t;
        }
        __syncthreads();
        int* tmp = r2; r2 = r1; r1 = cur; cur = tmp;          // rotate: cur becomes the previous row
    }
    // ---- backward pass (rows bottom-up, columns right-to-left: thread j handles column Wp - 1 - j)
    for (int i = threadIdx.x; i < 3 * LD; i += blockDim.x) sm[i] = CH_INF;
    __syncthreads();
    r1 = sm; r2 = sm + LD; cur = sm + 2 * LD;
    for (int i = Hp - 1; i >= 0; --i) {
        int v = CH_INF * 2;
        const bool live = j < Wp;
        const int col = Wp - 1 - j;
        if (live) {
            const int* p1 = r1 + 2 + col;
            const int* p2 = r2 + 2 + col;
            int t = T[(int64_t)i * Wp + col], u = p2[-1] + CH_C;
            t = t < u ? t : u; u = p2[1] + CH_C; t = t < u ? t : u; u = p1[-2] + CH_C; t = t < u ? t : u; u = p1[2] + CH_C; t = t < u ? t : u;
            u = p1[-1] + CH_B; t = t < u ? t : u; u = p1[1] + CH_B; t = t < u ? t : u; u = p1[0] + CH_A; t = t < u ? t : u;
            v = t - CH_A * j;            // suffix scan over columns = prefix scan over j: t[col] = min(cand, t[col+1] + a)
        }
        v = block_scan_min(v, wsum);
        if (live) {
            const int t = v + CH_A * j;
            cur[2 + col] = t;
            const int y = i - pad, x = col - pad;
            // (no zero pixel anywhere: OpenCV's DIST_MAX saturation, UINT_MAX - c, as the oracle has it)
            const unsigned tu = t >= CH_INF ? 0xFFFFFFFFu - (unsigned)CH_C : (unsigned)t;
            if (y >= 0 && y < H && x >= 0 && x < W) dist[(int64_t)b * H * W + (int64_t)y * W + x] = (float)tu * (1.0f / 65536.0f);
        }
        __syncthreads();
        int* tmp = r2; r2 = r1; r1 = cur; cur = tmp;
    }
}

extern "C" int vpu_chamfer5(const uint8_t* mask, int32_t* scratch, float* dist, int32_t B, int32_t H, int32_t W,
                            int32_t zero_border, void* stream) {
    vpu_clear_stale_error();
    const int pad = zero_border ? 1 : 0;
    if (!mask || !scratch || !dist || B < 1 || H < 1 || W < 1 || W + 2 * pad > 1024 || H > 16000) {
        vpu_set_error("chamfer5: null pointer, or size out of range (W + border <= 1024, H <= 16000)");
        return VPU_ERR_ARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int Wp = W + 2 * pad;
    const int threads = ((Wp + 63) / 64) * 64;
    chamfer5_kernel<<<B, threads, (size_t)3 * (Wp + 4) * sizeof(int), s>>>(mask, scratch, dist, H, W, pad);
    return vpu_check_launch("vpu_chamfer5");
}

extern "C" int vpu_cc_roots(const uint8_t* mask, int32_t* roots, int32_t B, int32_t H, int32_t W, void* stream) {
    vpu_clear_stale_error();
    const int64_t n = (int64_t)B * H * W;
    if (!mask || !roots || B < 1 || H < 1 || W < 1 || n >= 0x7FFFFFFFLL) {
        vpu_set_error("cc_roots: null pointer or B*H*W >= 2^31");
        return VPU_ERR_ARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = (unsigned)((n + 255) / 256);
    cc_init_kernel<<<dim3(H, B), 256, 0, s>>>(mask, roots, H, W);
    cc_merge_kernel<<<dim3((W + 255) / 256, H, B), 256, 0, s>>>(mask, roots, H, W);
    cc_flatten_kernel<<<g, 256, 0, s>>>(roots, n);
    return vpu_check_launch("vpu_cc_roots");
}

extern "C" int vpu_cc_table(const int32_t* roots, int32_t* slots, int32_t* table, int32_t kmax, int32_t B, int32_t H, int32_t W,
                            void* stream) {
    vpu_clear_stale_error();
    const int64_t n = (int64_t)B * H * W;
    if (!roots || !slots || !table || kmax < 1 || B < 1 || H < 1 || W < 1 || n >= 0x7FFFFFFFLL || B > 65535 || H > 65535) {
        vpu_set_error("cc_table: null pointer, kmax < 1, or sizes out of range");
        return VPU_ERR_ARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int32_t* counter = table + 6 * (int64_t)kmax;     // the component count lives behind the last row
    if (hipMemsetAsync(counter, 0, sizeof(int32_t), s) != hipSuccess) { vpu_set_error("cc_table: memset"); return VPU_ERR_LAUNCH; }
    cc_table_roots_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(roots, slots, table, counter, n, kmax);
    cc_table_accum_kernel<<<dim3((W + 255) / 256, H, B), 256, 0, s>>>(roots, slots, table, H, W);
    return vpu_check_launch("vpu_cc_table");
}
