// f2: ingest front-end for sources that are not image_size x image_size -- expand2square (mm_utils.py:257-268)
// + the image processor's bicubic shortest-edge resize + centre crop (HF CLIPImageProcessor -> PIL ImagingResample, 8 bpc),
// uint8 in, uint8 out, BIT-EXACT with PIL: the coefficient tables are built on the host in double exactly as Resample.c
// does (precompute_coeffs / normalize_coeffs_8bpc, 22 fractional bits) and the two passes (horizontal, then vertical, each
// rounding to uint8) are integer arithmetic.  HBM/L2-bound byte work: one thread per output pixel, taps read through L1/L2.
#include <stdlib.h>
#include <map>
#include <math.h>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"
#include "host.h"

namespace {
constexpr int PREC = 32 - 8 - 2;

struct Plan {            // one resize axis: out positions x (first tap, tap count, ksize int32 coefficients)
    int in = 0, out = 0, ksize = 0;
    int* d_first = nullptr;
    int* d_count = nullptr;
    int* d_kk = nullptr;
};

double bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

std::mutex g_plan_mu;
std::map<std::pair<int, int>, Plan> g_plans;

int get_plan(int in, int out, Plan& p) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find({in, out});
    if (it != g_plans.end()) { p = it->second; return SM_OK; }
    const double scale = (double)in / out;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * fs;
    p.in = in; p.out = out; p.ksize = (int)ceil(support) * 2 + 1;
    std::vector<int> first(out), count(out), kk((size_t)out * p.ksize, 0);
    std::vector<double> k(p.ksize);
    for (int xx = 0; xx < out; ++xx) {
        const double center = (xx + 0.5) * scale, ss = 1.0 / fs;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in) xmax = in;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { k[x] = bicubic((x + xmin - center + 0.5) * ss); ww += k[x]; }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            kk[(size_t)xx * p.ksize + x] = (int)(k[x] < 0 ? -0.5 + k[x] * (1 << PREC) : 0.5 + k[x] * (1 << PREC));
        }
        first[xx] = xmin; count[xx] = xmax;
    }
    SM_HIP(hipMalloc(&p.d_first, out * sizeof(int)));
    SM_HIP(hipMalloc(&p.d_count, out * sizeof(int)));
    SM_HIP(hipMalloc(&p.d_kk, kk.size() * sizeof(int)));
    SM_HIP(hipMemcpy(p.d_first, first.data(), out * sizeof(int), hipMemcpyHostToDevice));
    SM_HIP(hipMemcpy(p.d_count, count.data(), out * sizeof(int), hipMemcpyHostToDevice));
    SM_HIP(hipMemcpy(p.d_kk, kk.data(), kk.size() * sizeof(int), hipMemcpyHostToDevice));
    g_plans[{in, out}] = p;
    return SM_OK;
}

struct Geo { int Hp, Wp, py, px, oh, ow, top, left; };
Geo geometry(int H, int W, int pad, int out) {
    Geo g;
    g.Hp = H; g.Wp = W; g.py = g.px = 0;
    if (pad && H != W) {
        const int L = H > W ? H : W;
        g.py = W > H ? (W - H) / 2 : 0;
        g.px = H > W ? (H - W) / 2 : 0;
        g.Hp = g.Wp = L;
    }
    if (g.Hp <= g.Wp) { g.oh = out; g.ow = (int)((double)out * g.Wp / g.Hp); }
    else { g.oh = (int)((double)out * g.Hp / g.Wp); g.ow = out; }
    g.top = (g.oh - out) / 2; g.left = (g.ow - out) / 2;
    return g;
}
}  // namespace

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PREC;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass over the (virtually padded) source: tmp[b][y][xx][c], y in [0, Hp), xx in [0, ow)
__global__ void ingest_hpass_kernel(const uint8_t* __restrict__ src, int H, int W, int Hp, int py, int px, uint32_t bg,
                                    const int* __restrict__ first, const int* __restrict__ count, const int* __restrict__ kk,
                                    int ksize, int ow, uint8_t* __restrict__ tmp, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int xx = (int)(t % ow);
    const size_t r = t / ow;
    const int y = (int)(r % Hp), b = (int)(r / Hp);
    const int sy = y - py;
    if (sy < 0 || sy >= H) {
        // a row of the pad canvas: every tap is the background colour and the coefficients sum to 2^22 +- a few units, so
        // (bg * sum + 2^21) >> 22 == bg exactly (|bg * delta| < 2^21) -- what PIL's arithmetic yields, without the taps
        uint8_t* o = tmp + t * 3;
        o[0] = bg & 255; o[1] = (bg >> 8) & 255; o[2] = (bg >> 16) & 255;
        return;
    }
    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
    const int x0 = first[xx], n = count[xx];
    const int* k = kk + (size_t)xx * ksize;
    const bool row_in = true;
    const uint8_t* row = src + ((size_t)b * H + sy) * W * 3;
    for (int j = 0; j < n; ++j) {
        const int sx = x0 + j - px;
        int p0 = bg & 255, p1 = (bg >> 8) & 255, p2 = (bg >> 16) & 255;
        if (row_in && sx >= 0 && sx < W) { p0 = row[sx * 3]; p1 = row[sx * 3 + 1]; p2 = row[sx * 3 + 2]; }
        const int w = k[j];
        a0 += p0 * w; a1 += p1 * w; a2 += p2 * w;
    }
    uint8_t* o = tmp + t * 3;
    o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
}

// vertical pass restricted to the centre-crop window: dst[b][oy][ox][c] = resized[b][oy + top][ox + left][c]
__global__ void ingest_vpass_kernel(const uint8_t* __restrict__ tmp, int Hp, int ow, const int* __restrict__ first,
                                    const int* __restrict__ count, const int* __restrict__ kk, int ksize, int top, int left,
                                    int out, uint8_t* __restrict__ dst, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int ox = (int)(t % out);
    const size_t r = t / out;
    const int oy = (int)(r % out), b = (int)(r / out);
    const int yy = oy + top, xx = ox + left;
    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
    const int y0 = first[yy], n = count[yy];
    const int* k = kk + (size_t)yy * ksize;
    const uint8_t* col = tmp + (((size_t)b * Hp + y0) * ow + xx) * 3;
    for (int j = 0; j < n; ++j) {
        const int w = k[j];
        a0 += col[0] * w; a1 += col[1] * w; a2 += col[2] * w;
        col += (size_t)ow * 3;
    }
    uint8_t* o = dst + t * 3;
    o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
}

// vertical pass, 4 consecutive output BYTES per thread (the vertical filter is independent per byte): one aligned dword load
// per tap instead of one byte load per channel and tap.  Needs 4-byte aligned rows and crop offset (336 * 3 = 1008 is).
__global__ void ingest_vpass4_kernel(const uint8_t* __restrict__ tmp, int Hp, int row_bytes, const int* __restrict__ first,
                                     const int* __restrict__ count, const int* __restrict__ kk, int ksize, int top, int left_bytes,
                                     int out, int out_row_bytes, uint8_t* __restrict__ dst, size_t total) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int q = out_row_bytes >> 2;
    const int i4 = (int)(t % q);
    const size_t r = t / q;
    const int oy = (int)(r % out), b = (int)(r / out);
    const int yy = oy + top;
    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0, a3 = a0;
    const int y0 = first[yy], n = count[yy];
    const int* k = kk + (size_t)yy * ksize;
    const uint8_t* col = tmp + ((size_t)b * Hp + y0) * row_bytes + left_bytes + i4 * 4;
    for (int j = 0; j < n; ++j) {
        const uint32_t v = *(const uint32_t*)col;
        const int w = k[j];
        a0 += (int)(v & 255) * w; a1 += (int)((v >> 8) & 255) * w; a2 += (int)((v >> 16) & 255) * w; a3 += (int)(v >> 24) * w;
        col += row_bytes;
    }
    // the shift and the clamp are kept apart by an empty asm: fused, hipcc (ROCm 7.2) selects v_ashr_pk_u8_i32 for the first two
    // bytes and ORs the other two into a destination whose upper half still holds accumulator bits (bytes 2/3 came out wrong)
    int s0 = a0 >> PREC, s1 = a1 >> PREC, s2 = a2 >> PREC, s3 = a3 >> PREC;
    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    s0 = min(max(s0, 0), 255); s1 = min(max(s1, 0), 255); s2 = min(max(s2, 0), 255); s3 = min(max(s3, 0), 255);
    *(uint32_t*)(dst + ((size_t)b * out + oy) * out_row_bytes + i4 * 4) =
        (uint32_t)s0 | ((uint32_t)s1 << 8) | ((uint32_t)s2 << 16) | ((uint32_t)s3 << 24);
}

// horizontal pass with the source row segment staged in LDS: a block = one (frame, row) x 256 consecutive output columns; the
// ~(256 * scale + taps) source pixels it needs are fetched with coalesced byte loads (pad columns filled with the background),
// the taps then come from LDS.  Falls back to the direct kernel when the segment does not fit.
#define HP_LDS_BYTES 16384
__global__ __launch_bounds__(256) void ingest_hpass_lds_kernel(const uint8_t* __restrict__ src, int H, int W, int Hp, int py, int px,
                                                               uint32_t bg, const int* __restrict__ first, const int* __restrict__ count,
                                                               const int* __restrict__ kk, int ksize, int ow, int xblocks,
                                                               uint8_t* __restrict__ tmp) {
    __shared__ __attribute__((aligned(16))) uint8_t seg[HP_LDS_BYTES];
    const int xb = blockIdx.x % xblocks;
    const int r = blockIdx.x / xblocks;
    const int y = r % Hp, b = r / Hp;
    const int xx0 = xb * 256, xx1 = min(xx0 + 256, ow);
    const int xx = xx0 + threadIdx.x;
    const int sy = y - py;
    uint8_t* orow = tmp + ((size_t)b * Hp + y) * ow * 3;
    if (sy < 0 || sy >= H) {            // pad canvas row: the result is the background colour (see ingest_hpass_kernel)
        if (xx < xx1) { orow[xx * 3] = bg & 255; orow[xx * 3 + 1] = (bg >> 8) & 255; orow[xx * 3 + 2] = (bg >> 16) & 255; }
        return;
    }
    const int p0 = first[xx0], p1 = first[xx1 - 1] + count[xx1 - 1];       // padded-canvas pixel range [p0, p1)
    const uint8_t* row = src + ((size_t)b * H + sy) * W * 3;
    // bytes [p0*3, p1*3) of the canvas row -> seg[0 ..); inside the image they are row[(p - px)*3 + c]
    const int nbytes = (p1 - p0) * 3;
    const long img0 = (long)(p0 - px) * 3;                                   // byte offset in the image row of canvas byte p0*3
    const long img_bytes = (long)W * 3;
    for (int sidx = threadIdx.x; sidx < nbytes; sidx += 256) {        // consecutive threads -> consecutive bytes (coalesced)
        const long gb = img0 + sidx;
        uint8_t val;
        if (gb >= 0 && gb < img_bytes) val = row[gb];
        else val = (bg >> (8 * (sidx % 3))) & 255;                        // seg[0] starts on a pixel boundary: channel = sidx % 3
        seg[sidx] = val;
    }
    __syncthreads();
    if (xx >= xx1) return;
    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
    const int x0 = first[xx], n = count[xx];
    const int* k = kk + (size_t)xx * ksize;
    const uint8_t* sp = seg + (x0 - p0) * 3;
    for (int j = 0; j < n; ++j) {
        const int w = k[j];
        a0 += sp[0] * w; a1 += sp[1] * w; a2 += sp[2] * w;
        sp += 3;
    }
    orow[xx * 3] = clip8(a0); orow[xx * 3 + 1] = clip8(a1); orow[xx * 3 + 2] = clip8(a2);
}

extern "C" size_t sm_ingest_tmp_bytes(int B, int H, int W, int pad_square, int out_size) {
    if (B <= 0 || H <= 0 || W <= 0 || out_size <= 0) return 0;
    const Geo g = geometry(H, W, pad_square, out_size);
    return (size_t)B * g.Hp * g.ow * 3;
}

extern "C" int sm_ingest_frames(const uint8_t* src, int B, int H, int W, int pad_square, const uint8_t* pad_rgb_host,
                                int out_size, uint8_t* dst, uint8_t* tmp, void* stream) {
    SM_REQUIRE(src && dst && tmp && B > 0 && H > 0 && W > 0 && out_size > 0, "sm_ingest_frames: bad args");
    SM_REQUIRE(!pad_square || pad_rgb_host, "sm_ingest_frames: pad colour missing");
    const Geo g = geometry(H, W, pad_square, out_size);
    Plan ph, pv;
    int rc;
    if ((rc = get_plan(g.Wp, g.ow, ph))) return rc;
    if ((rc = get_plan(g.Hp, g.oh, pv))) return rc;
    const uint32_t bg = pad_rgb_host ? (pad_rgb_host[0] | (pad_rgb_host[1] << 8) | (pad_rgb_host[2] << 16)) : 0;
    hipStream_t st = (hipStream_t)stream;
    // widest source segment a 256-column block of the horizontal pass can need: 256 * scale + 2 * support (+ slack)
    const double hscale = (double)g.Wp / g.ow;
    const int seg_px = (int)(256 * hscale + 2 * ph.ksize + 8);
    if (seg_px * 3 + 8 <= HP_LDS_BYTES) {
        const int xblocks = (g.ow + 255) / 256;
        ingest_hpass_lds_kernel<<<(unsigned)((size_t)B * g.Hp * xblocks), 256, 0, st>>>(src, H, W, g.Hp, g.py, g.px, bg, ph.d_first, ph.d_count,
                                                                                       ph.d_kk, ph.ksize, g.ow, xblocks, tmp);
    } else {
        const size_t n1 = (size_t)B * g.Hp * g.ow;
        ingest_hpass_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(src, H, W, g.Hp, g.py, g.px, bg, ph.d_first, ph.d_count, ph.d_kk,
                                                                          ph.ksize, g.ow, tmp, n1);
    }
    SM_LAUNCH_CHECK();
    const int row_bytes = g.ow * 3, out_row_bytes = out_size * 3, left_bytes = g.left * 3;
    if ((row_bytes & 3) == 0 && (out_row_bytes & 3) == 0 && (left_bytes & 3) == 0 && ((uintptr_t)tmp & 3) == 0 && ((uintptr_t)dst & 3) == 0) {
        const size_t n2 = (size_t)B * out_size * (out_row_bytes >> 2);
        ingest_vpass4_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(tmp, g.Hp, row_bytes, pv.d_first, pv.d_count, pv.d_kk, pv.ksize, g.top,
                                                                           left_bytes, out_size, out_row_bytes, dst, n2);
    } else {
        const size_t n2 = (size_t)B * out_size * out_size;
        ingest_vpass_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(tmp, g.Hp, g.ow, pv.d_first, pv.d_count, pv.d_kk, pv.ksize, g.top,
                                                                          g.left, out_size, dst, n2);
    }
    SM_LAUNCH_CHECK();
    return SM_OK;
}
