// f2 ingest front-end: baseline JPEG (Motion-JPEG AVI frames, stills) -> RGB u8 in HBM without PIL.
// Replaces the decord / PIL decode in front of the streaming loop (/root/reference/streammind/eval/video_score_stream_demo.py:212-225,
// mm_utils.py:399-435) for the one codec this image can decode at all (no rocDecode / VCN library, no ffmpeg).
//
// Split at the only serial part: the entropy-coded segment is a bit-serial Huffman stream, so the HOST parses the markers and
// decodes it to quantised DCT coefficients (sm_jpeg_decode_coefs: one frame per call, thread-safe, callers run frames in
// parallel); everything after that is data-parallel and runs on the GPU from the coefficient image:
//     jpeg_idct_kernel   dequantise + 8x8 inverse DCT per block                  (one thread per block column / row, LDS transpose)
//     jpeg_rgb_kernel    chroma upsampling + YCbCr -> RGB, u8 HWC                (one thread per pixel)
// The arithmetic is libjpeg's, integer for integer -- the accurate integer IDCT (jidctint.c "islow": 13-bit constants, two passes,
// PASS1_BITS = 2), the "fancy" triangle-filter upsampling of jdsample.c (h2v1: 3/4 + 1/4; h2v2: 9/16, 3/16, 3/16, 1/16 with the
// alternating 8 / 7 rounding bias), the 16-bit fixed-point colour tables of jdcolor.c -- restated from the published algorithm
// (IJG libjpeg 6b / libjpeg-turbo, the decoder PIL links: third-party, not in the reference tree), so the output is BIT-EXACT against
// PIL (libjpeg-turbo defaults: JDCT_ISLOW, do_fancy_upsampling) -- tests/test_gpu_jpeg.py compares every byte.
// Supported: baseline sequential DCT (SOF0; SOF1 with 8-bit samples and <= 2 tables per class also decodes), 8-bit, 1 or 3 components,
// sampling 4:4:4 / 4:2:2 (h2v1) / 4:2:0 (h2v2), restart intervals, interleaved or per-component scans.  Everything else (progressive,
// arithmetic, CMYK, 4:4:0, 12-bit) returns SM_EINVAL naming the reason and the caller falls back to its host decoder.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "common.h"
#include "host.h"

// ------------------------------------------------------------------------------------------------ host: markers + Huffman
namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__device__ const uint8_t kZigzagDev[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0}, vals[256] = {0};
    // canonical decode tables (ITU T.81 Annex F.2.2.3): codes of length l lie in [mincode[l], maxcode[l]], valptr[l] indexes vals
    int mincode[17], maxcode[18], valptr[17];
    uint16_t fast[512];                       // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
    // false: the code-length counts over-subscribe the code space (a length-l code would exceed 2^l: no prefix code has these
    // counts -- jdhuff.c rejects the table the same way) or name more than 256 symbols.  An over-subscribed table would index
    // `fast` far beyond its 512 entries (bits[1] = 255: ~130 KB past the array).
    bool build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            if (code > (1 << l) || k > 256) return false;
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memset(fast, 0, sizeof(fast));
        code = 0; k = 0;
        for (int l = 1; l <= 9; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                const int lo = code << (9 - l);
                for (int j = 0; j < (1 << (9 - l)) && lo + j < 512; ++j) fast[lo + j] = (uint16_t)((l << 8) | vals[k]);
            }
            code <<= 1;
        }
        return true;
    }
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t acc = 0; int n = 0;              // `n` valid bits at the top of acc (left-aligned in the low 64 - ... we keep them right-aligned)
    bool hit_marker = false;
    BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
    void fill() {
        while (n <= 48) {
            uint8_t b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;              // stuffed zero
                    else { hit_marker = true; b = 0; }                    // a marker: feed zeros, leave p on it
                } else ++p;
            }
            acc = (acc << 8) | b;
            n += 8;
        }
    }
    inline int peek(int k) { if (n < k) fill(); return (int)((acc >> (n - k)) & ((1u << k) - 1)); }
    inline void skip(int k) { n -= k; }
    inline int get(int k) { if (!k) return 0; int v = peek(k); n -= k; return v; }
    void restart() { acc = 0; n = 0; hit_marker = false; }                 // byte-align, drop the look-ahead
};

inline int huff_decode(BitReader& br, const Huff& h) {
    int look = br.peek(9);
    const uint16_t f = h.fast[look];
    if (f) { br.skip(f >> 8); return f & 0xff; }
    int code = br.peek(16), l = 10;
    for (; l <= 16; ++l) {
        const int c = code >> (16 - l);
        if (c <= h.maxcode[l] && h.maxcode[l] >= 0 && c >= h.mincode[l]) {
            br.skip(l);
            return h.vals[h.valptr[l] + c - h.mincode[l]];
        }
    }
    return -1;
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Comp { int id = 0, hs = 1, vs = 1, tq = 0, td = 0, ta = 0; };

struct Parsed {
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    Comp comp[3];
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    bool sof = false;
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

}  // namespace

static void fill_info(const Parsed& P, sm_jpeg_info_t* info) {
    memset(info, 0, sizeof(*info));
    info->width = P.width; info->height = P.height; info->ncomp = P.ncomp;
    info->mcu_w = 8 * P.hmax; info->mcu_h = 8 * P.vmax;
    info->mcus_x = (P.width + info->mcu_w - 1) / info->mcu_w;
    info->mcus_y = (P.height + info->mcu_h - 1) / info->mcu_h;
    size_t off = 0;
    for (int c = 0; c < P.ncomp; ++c) {
        info->hs[c] = P.comp[c].hs; info->vs[c] = P.comp[c].vs;
        info->blocks_x[c] = info->mcus_x * P.comp[c].hs;
        info->blocks_y[c] = info->mcus_y * P.comp[c].vs;
        info->coef_offset[c] = (int)off;
        off += (size_t)info->blocks_x[c] * info->blocks_y[c] * 64;
    }
    info->coef_count = (int)off;
}

// markers up to (and including) the first SOS header; *scan = first entropy-coded byte.  Later scans are parsed by the decode loop.
static int parse_segments(const uint8_t* d, size_t len, size_t& pos, Parsed& P, bool stop_at_sof, int* scan_comps, int* scan_n) {
    while (pos + 4 <= len) {
        if (d[pos] != 0xFF) { ++pos; continue; }
        const int m = d[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        if (m == 0x00) { pos += 2; continue; }                                    // a stuffed byte left over behind a scan
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { pos += 2; continue; }
        if (m == 0xD9) SM_FAIL(SM_EINVAL, "jpeg: EOI before any scan");
        const int L = be16(d + pos + 2);
        if (L < 2 || pos + 2 + L > len) SM_FAIL(SM_EINVAL, "jpeg: truncated segment 0x%02X", m);
        const uint8_t* s = d + pos + 4;
        const int n = L - 2;
        if (m == 0xC0 || m == 0xC1) {
            // one frame header per image: a second SOF between the scans of a multi-scan file would change the geometry under the
            // coefficient layout fixed after the first SOS (block counts, offsets)
            SM_REQUIRE(!P.sof, "jpeg: second frame header (SOF%d) inside one image", m - 0xC0);
            SM_REQUIRE(n >= 6 && s[0] == 8, "jpeg: %d-bit samples (8-bit only)", n >= 1 ? s[0] : 0);
            P.height = be16(s + 1); P.width = be16(s + 3); P.ncomp = s[5];
            SM_REQUIRE(P.width > 0 && P.height > 0, "jpeg: empty frame (%d x %d)", P.width, P.height);
            SM_REQUIRE(P.ncomp == 1 || P.ncomp == 3, "jpeg: %d components (1 or 3 supported)", P.ncomp);
            SM_REQUIRE(n >= 6 + 3 * P.ncomp, "jpeg: short SOF");
            for (int c = 0; c < P.ncomp; ++c) {
                Comp& k = P.comp[c];
                k.id = s[6 + 3 * c]; k.hs = s[7 + 3 * c] >> 4; k.vs = s[7 + 3 * c] & 15; k.tq = s[8 + 3 * c];
                SM_REQUIRE(k.tq < 4, "jpeg: quantisation table id %d", k.tq);
            }
            if (P.ncomp == 1) { P.comp[0].hs = P.comp[0].vs = 1; }                 // a single component is never subsampled (T.81 A.2.2)
            P.hmax = P.comp[0].hs; P.vmax = P.comp[0].vs;
            if (P.ncomp == 3) {
                SM_REQUIRE(P.comp[1].hs == 1 && P.comp[1].vs == 1 && P.comp[2].hs == 1 && P.comp[2].vs == 1 && (P.hmax == 1 || P.hmax == 2) &&
                           (P.vmax == 1 || P.vmax == 2) && !(P.hmax == 1 && P.vmax == 2),
                           "jpeg: sampling %dx%d,%dx%d,%dx%d (4:4:4, 4:2:2 and 4:2:0 supported)", P.comp[0].hs, P.comp[0].vs, P.comp[1].hs, P.comp[1].vs,
                           P.comp[2].hs, P.comp[2].vs);
            }
            P.sof = true;
            if (stop_at_sof) { pos += 2 + L; return SM_OK; }
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            SM_FAIL(SM_EINVAL, "jpeg: SOF%d (progressive / lossless / arithmetic) is not baseline", m - 0xC0);
        } else if (m == 0xC4) {
            int o = 0;
            while (o + 17 <= n) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                SM_REQUIRE(tc < 2 && th < 4, "jpeg: Huffman table class %d id %d", tc, th);
                Huff& h = tc ? P.ac[th] : P.dc[th];
                int cnt = 0;
                h.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) { h.bits[i] = s[o + i]; cnt += s[o + i]; }
                SM_REQUIRE(cnt <= 256 && o + 17 + cnt <= n, "jpeg: bad Huffman table");
                memcpy(h.vals, s + o + 17, cnt);
                h.present = true;
                SM_REQUIRE(h.build(), "jpeg: Huffman table class %d id %d over-subscribes its code space", tc, th);
                o += 17 + cnt;
            }
        } else if (m == 0xDB) {
            int o = 0;
            while (o < n) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                SM_REQUIRE(tq < 4 && o + 1 + 64 * (pq ? 2 : 1) <= n, "jpeg: bad quantisation table");
                for (int i = 0; i < 64; ++i) P.qt[tq][kZigzag[i]] = (uint16_t)(pq ? be16(s + o + 1 + 2 * i) : s[o + 1 + i]);
                P.qt_present[tq] = true;
                o += 1 + 64 * (pq ? 2 : 1);
            }
        } else if (m == 0xDD) {
            SM_REQUIRE(n >= 2, "jpeg: short DRI");
            P.restart = be16(s);
        } else if (m == 0xDA) {
            SM_REQUIRE(P.sof, "jpeg: SOS before SOF");
            const int ns = s[0];
            SM_REQUIRE(ns >= 1 && ns <= P.ncomp && n >= 1 + 2 * ns + 3, "jpeg: bad SOS");
            for (int i = 0; i < ns; ++i) {
                int ci = -1;
                for (int c = 0; c < P.ncomp; ++c) if (P.comp[c].id == s[1 + 2 * i]) ci = c;
                SM_REQUIRE(ci >= 0, "jpeg: scan names an unknown component");
                P.comp[ci].td = s[2 + 2 * i] >> 4; P.comp[ci].ta = s[2 + 2 * i] & 15;
                SM_REQUIRE(P.comp[ci].td < 4 && P.comp[ci].ta < 4, "jpeg: Huffman table id");
                scan_comps[i] = ci;
            }
            SM_REQUIRE(s[1 + 2 * ns] == 0 && s[2 + 2 * ns] == 63, "jpeg: spectral selection %d..%d (progressive scan)", s[1 + 2 * ns], s[2 + 2 * ns]);
            *scan_n = ns;
            pos += 2 + L;
            return SM_OK;
        }
        pos += 2 + L;
    }
    SM_FAIL(SM_EINVAL, stop_at_sof ? "jpeg: no SOF0 frame header" : "jpeg: no scan");
}

extern "C" int sm_jpeg_info(const uint8_t* data, size_t len, sm_jpeg_info_t* info) {
    SM_REQUIRE(data && info && len >= 4 && data[0] == 0xFF && data[1] == 0xD8, "sm_jpeg_info: not a JPEG (no SOI)");
    Parsed P;
    size_t pos = 2;
    int sc[3], sn = 0;
    int rc = parse_segments(data, len, pos, P, true, sc, &sn);
    if (rc) return rc;
    fill_info(P, info);
    return SM_OK;
}

// The standard (ITU T.81 Annex K.3) Huffman tables, for Motion-JPEG frames stored without DHT segments (the AVI1 convention).
static const uint8_t kStdDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kStdDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kStdDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kStdAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kStdAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kStdAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kStdAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static void std_table(Huff& h, const uint8_t* bits, const uint8_t* vals, int nv) {
    h.bits[0] = 0;
    for (int i = 0; i < 16; ++i) h.bits[i + 1] = bits[i];
    memcpy(h.vals, vals, nv);
    h.present = true;
    (void)h.build();        // Annex K.3 tables: valid by construction
}

// One frame: markers + every scan's entropy-coded segment -> quantised coefficients, natural (row-major) order inside a block,
// blocks row-major per component plane (planes padded to whole MCUs), components back to back (sm_jpeg_info_t.coef_offset), and the
// components' quantisation tables qt[ncomp][64] (natural order).  Host memory in, host memory out (pinned memory if the caller wants
// the upload to overlap); no HIP call: callers decode the frames of a batch on as many host threads as they like.
extern "C" int sm_jpeg_decode_coefs(const uint8_t* data, size_t len, const sm_jpeg_info_t* want, int16_t* coefs, uint16_t* qt) {
    SM_REQUIRE(data && coefs && qt && len >= 4 && data[0] == 0xFF && data[1] == 0xD8, "sm_jpeg_decode_coefs: not a JPEG (no SOI)");
    Parsed P;
    size_t pos = 2;
    int sc[3], sn = 0;
    int rc = parse_segments(data, len, pos, P, false, sc, &sn);
    if (rc) return rc;
    sm_jpeg_info_t I;
    fill_info(P, &I);
    if (want)
        SM_REQUIRE(want->width == I.width && want->height == I.height && want->ncomp == I.ncomp && want->hs[0] == I.hs[0] && want->vs[0] == I.vs[0] &&
                   want->coef_count == I.coef_count, "sm_jpeg_decode_coefs: frame is %dx%d/%d comps/%dx%d sampling, the batch was opened as %dx%d/%d/%dx%d",
                   I.width, I.height, I.ncomp, I.hs[0], I.vs[0], want->width, want->height, want->ncomp, want->hs[0], want->vs[0]);
    memset(coefs, 0, (size_t)I.coef_count * sizeof(int16_t));
    bool done[3] = {false, false, false};
    for (;;) {
        // tables: what the stream defined, else the standard ones (Motion-JPEG frames without DHT)
        for (int i = 0; i < sn; ++i) {
            const Comp& k = P.comp[sc[i]];
            if (!P.dc[k.td].present) { SM_REQUIRE(k.td < 2, "jpeg: DC table %d undefined", k.td); std_table(P.dc[k.td], k.td ? kStdDcChrBits : kStdDcLumBits, kStdDcVals, 12); }
            if (!P.ac[k.ta].present) { SM_REQUIRE(k.ta < 2, "jpeg: AC table %d undefined", k.ta); std_table(P.ac[k.ta], k.ta ? kStdAcChrBits : kStdAcLumBits, k.ta ? kStdAcChrVals : kStdAcLumVals, 162); }
            SM_REQUIRE(P.qt_present[k.tq], "jpeg: quantisation table %d undefined", k.tq);
        }
        // geometry of this scan: interleaved = whole MCUs; a single-component scan walks that component's own 8x8 blocks (T.81 A.2.3)
        int mx, my;
        if (sn > 1) { mx = I.mcus_x; my = I.mcus_y; }
        else {
            const Comp& k = P.comp[sc[0]];
            mx = ((P.width * k.hs + P.hmax - 1) / P.hmax + 7) / 8;
            my = ((P.height * k.vs + P.vmax - 1) / P.vmax + 7) / 8;
        }
        BitReader br(data + pos, data + len);
        int pred[3] = {0, 0, 0};
        int until_restart = P.restart, next_rst = 0;
        for (int y = 0; y < my; ++y)
            for (int x = 0; x < mx; ++x) {
                if (P.restart && until_restart == 0) {
                    // byte-align, expect RSTn
                    br.restart();
                    const uint8_t* q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                        if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) break;      // some other marker: corrupt, give up below
                        ++q;
                    }
                    SM_REQUIRE(q + 1 < br.end && q[0] == 0xFF && q[1] == 0xD0 + next_rst, "jpeg: restart marker RST%d not found", next_rst);
                    br.p = q + 2;
                    next_rst = (next_rst + 1) & 7;
                    until_restart = P.restart;
                    pred[0] = pred[1] = pred[2] = 0;
                }
                for (int i = 0; i < sn; ++i) {
                    const int ci = sc[i];
                    const Comp& k = P.comp[ci];
                    const int bh = sn > 1 ? k.hs : 1, bv = sn > 1 ? k.vs : 1;
                    for (int v = 0; v < bv; ++v)
                        for (int h = 0; h < bh; ++h) {
                            const int bx = x * bh + h, by = y * bv + v;
                            SM_REQUIRE(bx < I.blocks_x[ci] && by < I.blocks_y[ci], "jpeg: block (%d, %d) outside the component's %d x %d blocks", bx, by, I.blocks_x[ci], I.blocks_y[ci]);
                            int16_t* blk = coefs + I.coef_offset[ci] + ((size_t)by * I.blocks_x[ci] + bx) * 64;
                            int s = huff_decode(br, P.dc[k.td]);
                            SM_REQUIRE(s >= 0 && s <= 11, "jpeg: bad DC code");
                            if (s) pred[ci] += extend(br.get(s), s);
                            blk[0] = (int16_t)pred[ci];
                            for (int kk = 1; kk < 64;) {
                                const int rs = huff_decode(br, P.ac[k.ta]);
                                SM_REQUIRE(rs >= 0, "jpeg: bad AC code");
                                const int r = rs >> 4, sz = rs & 15;
                                if (sz == 0) {
                                    if (r != 15) break;          // EOB
                                    kk += 16;                    // ZRL
                                    continue;
                                }
                                kk += r;
                                SM_REQUIRE(kk < 64, "jpeg: AC run past the block");
                                blk[kZigzag[kk]] = (int16_t)extend(br.get(sz), sz);
                                ++kk;
                            }
                        }
                }
                if (P.restart) --until_restart;
            }
        for (int i = 0; i < sn; ++i) done[sc[i]] = true;
        bool all = true;
        for (int c = 0; c < P.ncomp; ++c) all &= done[c];
        if (all) break;
        // next scan (per-component baseline files): the reader stopped on the marker that ends this segment
        pos = (size_t)(br.p - data);
        rc = parse_segments(data, len, pos, P, false, sc, &sn);
        if (rc) return rc;
    }
    for (int c = 0; c < P.ncomp; ++c) memcpy(qt + 64 * c, P.qt[P.comp[c].tq], 64 * sizeof(uint16_t));
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------ entropy decode on the GPU (restart intervals)
// Markers only: the frame's geometry, tables and where its ONE scan's entropy-coded bytes lie.  The decode itself is jpeg_huff_kernel below.
extern "C" int sm_jpeg_scan_prepare(const uint8_t* data, size_t len, const sm_jpeg_info_t* want, sm_jpeg_scan_t* out) {
    SM_REQUIRE(data && out && len >= 4 && data[0] == 0xFF && data[1] == 0xD8, "sm_jpeg_scan_prepare: not a JPEG (no SOI)");
    Parsed P;
    size_t pos = 2;
    int sc[3], sn = 0;
    int rc = parse_segments(data, len, pos, P, false, sc, &sn);
    if (rc) return rc;
    sm_jpeg_info_t I;
    fill_info(P, &I);
    if (want)
        SM_REQUIRE(want->width == I.width && want->height == I.height && want->ncomp == I.ncomp && want->hs[0] == I.hs[0] && want->vs[0] == I.vs[0] &&
                   want->coef_count == I.coef_count, "sm_jpeg_scan_prepare: frame is %dx%d/%d comps/%dx%d sampling, the batch was opened as %dx%d/%d/%dx%d",
                   I.width, I.height, I.ncomp, I.hs[0], I.vs[0], want->width, want->height, want->ncomp, want->hs[0], want->vs[0]);
    SM_REQUIRE(sn == P.ncomp, "jpeg: %d of %d components in the first scan (per-component scans: host path)", sn, P.ncomp);
    for (int i = 0; i < sn; ++i) SM_REQUIRE(sc[i] == i, "jpeg: scan component order (host path)");
    memset(out, 0, sizeof(*out));
    for (int c = 0; c < P.ncomp; ++c) {
        const Comp& k = P.comp[c];
        if (!P.dc[k.td].present) { SM_REQUIRE(k.td < 2, "jpeg: DC table %d undefined", k.td); std_table(P.dc[k.td], k.td ? kStdDcChrBits : kStdDcLumBits, kStdDcVals, 12); }
        if (!P.ac[k.ta].present) { SM_REQUIRE(k.ta < 2, "jpeg: AC table %d undefined", k.ta); std_table(P.ac[k.ta], k.ta ? kStdAcChrBits : kStdAcLumBits, k.ta ? kStdAcChrVals : kStdAcLumVals, 162); }
        SM_REQUIRE(P.qt_present[k.tq], "jpeg: quantisation table %d undefined", k.tq);
        memcpy(out->qt[c], P.qt[k.tq], 64 * sizeof(uint16_t));
        for (int t = 0; t < 2; ++t) {
            const Huff& h = t ? P.ac[k.ta] : P.dc[k.td];
            sm_jpeg_huff_t& d = t ? out->ac[c] : out->dc[c];
            memcpy(d.fast, h.fast, sizeof(d.fast));
            for (int l = 1; l <= 16; ++l) { d.maxcode[l] = h.maxcode[l]; d.valoff[l] = h.valptr[l] - h.mincode[l]; }
            d.maxcode[0] = -1; d.maxcode[17] = 0x7fffffff; d.valoff[0] = 0;
            memcpy(d.vals, h.vals, 256);
        }
    }
    out->scan_offset = (uint32_t)pos; out->scan_len = (uint32_t)(len - pos);
    out->restart = P.restart; out->ncomp = P.ncomp;
    out->n_intervals = P.restart > 0 ? (I.mcus_x * I.mcus_y + P.restart - 1) / P.restart : 0;       // 0: no DRI -- one serial stream, the self-synchronising decode
    SM_REQUIRE(out->n_intervals <= 4096, "jpeg: %d restart intervals per frame (the GPU index keeps 4096: host path)", out->n_intervals);
    return SM_OK;
}

#define JH_MAX_INT 4096            // restart intervals per frame the index kernel keeps (sm_jpeg_scan_prepare refuses frames with more)
// Start of every restart interval: interval 0 at the scan's first byte, interval k behind the k-th RSTn marker.  One block per frame; every thread
// counts the markers of its contiguous chunk, an LDS scan ranks them, a second walk writes the offsets in order.  Also checks the marker numbering
// (RST(k mod 8) ends interval k) and that exactly n_intervals - 1 markers exist.
__global__ __launch_bounds__(1024) void jpeg_rst_index_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ offsets,
                                                              const sm_jpeg_scan_t* __restrict__ scans, uint32_t* __restrict__ starts, int32_t* __restrict__ status) {
    __shared__ int cnt[1024];
    const int f = blockIdx.x, tid = threadIdx.x;
    const sm_jpeg_scan_t& sc = scans[f];
    const uint8_t* d = bytes + offsets[f] + sc.scan_offset;
    const uint32_t n = sc.scan_len;
    const uint32_t per = (n + 1023) / 1024, lo = tid * per, hi = lo + per < n ? lo + per : n;
    int c = 0;
    for (uint32_t p = lo; p < hi; ++p)
        if (d[p] == 0xFF && p + 1 < n && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7) ++c;
    cnt[tid] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                      // inclusive scan
        const int v = tid >= o ? cnt[tid - o] : 0;
        __syncthreads();
        cnt[tid] += v;
        __syncthreads();
    }
    int rank = cnt[tid] - c;                                  // markers in front of this thread's chunk
    uint32_t* st = starts + (size_t)f * JH_MAX_INT;
    if (tid == 0) st[0] = 0;
    bool bad = false;
    for (uint32_t p = lo; p < hi; ++p)
        if (d[p] == 0xFF && p + 1 < n && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7) {
            if (d[p + 1] != 0xD0 + (rank & 7)) bad = true;
            ++rank;
            if (rank < JH_MAX_INT) st[rank] = p + 2;
        }
    if (bad || (tid == 1023 && cnt[1023] != sc.n_intervals - 1)) atomicCAS(&status[f], 0, 3);
}

// Unstuffing, one WAVE per restart interval: the interval's bytes (from its start to the marker that ends it) are copied without the zero that follows every
// data 0xFF, in place of the raw bytes' own range of a second buffer (the clean image is never longer than the raw one); 64 bytes per step, ranks by ballot.
// lens[f][iv] = clean length.  The decode lanes then read a plain bit stream: no per-byte marker / stuffing checks in their loop.
__global__ __launch_bounds__(256) void jpeg_unstuff_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ offsets, const sm_jpeg_scan_t* __restrict__ scans,
                                                           const uint32_t* __restrict__ starts, uint8_t* __restrict__ clean, uint32_t* __restrict__ lens,
                                                           const int32_t* __restrict__ status) {
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int iv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const sm_jpeg_scan_t& sc = scans[f];
    if (iv >= sc.n_intervals || iv >= JH_MAX_INT) return;
    // a frame whose markers do not number its intervals (a stream cut short, a wrong RSTn) has starts[] entries nobody wrote: nothing of it is read
    // (found by tools/jpeg_fuzz.py: a 720p file cut at two thirds faulted here on a stale offset)
    if (status[f] != 0) { if (lane == 0) lens[(size_t)f * JH_MAX_INT + iv] = 0; return; }
    const uint32_t* st = starts + (size_t)f * JH_MAX_INT;
    uint32_t lo = st[iv], hi = iv + 1 < sc.n_intervals ? st[iv + 1] - 2 : sc.scan_len;       // (the RSTn marker itself is not data)
    hi = hi <= sc.scan_len ? hi : sc.scan_len;
    lo = lo <= hi ? lo : hi;
    const uint8_t* src = bytes + offsets[f] + sc.scan_offset;
    uint8_t* dst = clean + offsets[f] + sc.scan_offset + lo;
    uint32_t out = 0;
    bool ended = false;
    for (uint32_t p0 = lo; p0 < hi && !ended; p0 += 64) {
        const uint32_t p = p0 + lane;
        const bool in = p < hi;
        const uint32_t b = in ? src[p] : 0u;
        const uint32_t prev = (in && p > lo) ? src[p - 1] : 0u;
        const uint32_t next = (in && p + 1 < hi) ? src[p + 1] : 0u;
        // a marker inside the range (0xFF followed by neither 0x00 nor the end of the range: EOI behind the last interval) ends the data
        const bool is_marker = in && b == 0xFF && p + 1 < hi && next != 0x00;
        const unsigned long long mk = __ballot(is_marker);
        const int first_mk = mk ? __ffsll((long long)mk) - 1 : 64;
        const bool keep = in && lane < first_mk && !(b == 0x00 && prev == 0xFF);
        const unsigned long long kb = __ballot(keep);
        if (keep) dst[out + __popcll(kb & ((1ull << lane) - 1))] = (uint8_t)b;
        out += __popcll(kb);
        ended = mk != 0;
    }
    if (lane == 0) lens[(size_t)f * JH_MAX_INT + iv] = out;
}

// One lane per restart interval (T.81 F.2.2: DC difference + AC run/size codes, receive + extend), tables of the frame in LDS.  ONE uniform loop: every
// trip decodes one Huffman symbol (+ its value bits) for every live lane -- no nested data-dependent loops, so the lanes of a wave, which sit at different
// coefficients of different blocks, still execute the same instructions (the first form, a transliteration of the host decoder with its refill / AC /
// long-code loops, spent ~5500 clk per symbol on one wave per SIMD: 31.5 ms for 28 frames of 720p).  The bit window is refilled 32 bits at a time from the
// unstuffed image (one unaligned dword, prefetched one word ahead); codes longer than 9 bits take an unrolled scan of the length classes.
struct JhGeom { int mcus_x, mcus_y, ncomp, hs[3], vs[3], blocks_x[3], coef_offset[3], coef_count; };
__global__ __launch_bounds__(64) void jpeg_huff_kernel(const uint8_t* __restrict__ clean, const uint32_t* __restrict__ offsets, const sm_jpeg_scan_t* __restrict__ scans,
                                                       const uint32_t* __restrict__ starts, const uint32_t* __restrict__ lens, JhGeom g, int16_t* __restrict__ coefs,
                                                       uint16_t* __restrict__ qt, int32_t* __restrict__ status) {
    __shared__ sm_jpeg_huff_t tab[6];                          // dc[0..2], ac[0..2] of this frame
    __shared__ uint8_t zz[64];
    const int f = blockIdx.y, lane = threadIdx.x;
    const sm_jpeg_scan_t& sc = scans[f];
    if (sc.n_intervals <= 0 && blockIdx.x == 0 && lane == 0) atomicCAS(&status[f], 0, 3);      // no DRI: sm_jpeg_entropy_decode_sync is the entry for this frame
    if ((int)blockIdx.x * 64 >= sc.n_intervals) return;         // (the grid covers the largest interval count of the geometry)
    {
        const uint32_t* src = (const uint32_t*)sc.dc;          // dc[3] and ac[3] are contiguous in sm_jpeg_scan_t
        uint32_t* dst = (uint32_t*)tab;
        for (int w = lane; w < (int)(6 * sizeof(sm_jpeg_huff_t) / 4); w += 64) dst[w] = src[w];
        zz[lane] = kZigzagDev[lane];
        if (blockIdx.x == 0)
            for (int w = lane; w < 3 * 64; w += 64) qt[(size_t)f * 192 + w] = sc.qt[w / 64][w % 64];
    }
    __syncthreads();
    const int iv = blockIdx.x * 64 + lane;
    bool live = iv < sc.n_intervals && iv < JH_MAX_INT && status[f] != 3;
    const uint8_t* src = clean + offsets[f] + sc.scan_offset + (live ? starts[(size_t)f * JH_MAX_INT + iv] : 0u);
    const uint32_t len = live ? lens[(size_t)f * JH_MAX_INT + iv] : 0u;
    int16_t* cf = coefs + (size_t)f * g.coef_count;
    const int total = g.mcus_x * g.mcus_y;
    int mcu = iv * sc.restart;
    const int mcu_end = mcu + sc.restart < total ? mcu + sc.restart : total;
    live = live && mcu < mcu_end;
    // blocks of an MCU in scan order: component ci, block b of its bh x bv
    const int nb0 = g.ncomp > 1 ? g.hs[0] * g.vs[0] : 1, nbm = g.ncomp > 1 ? nb0 + 2 : 1;       // blocks per MCU (chroma components are 1 x 1)
    int bi = 0;                                                 // block index inside the MCU
    int k = 0;                                                  // next coefficient (zig-zag); 0 = the DC symbol comes next
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int16_t* blk = cf;
    int ci = 0;
    // MCU position kept incrementally (no integer division in the loop); luma blocks of an MCU: bh in {1, 2} columns
    int mx = live ? mcu % g.mcus_x : 0, my = live ? mcu / g.mcus_x : 0;
    const int bh0 = g.ncomp > 1 ? g.hs[0] : 1, bv0 = g.ncomp > 1 ? g.vs[0] : 1;
    const sm_jpeg_huff_t* hdc = &tab[0];
    const sm_jpeg_huff_t* hac = &tab[3];
    auto locate = [&]() {
        ci = bi < nb0 ? 0 : bi - nb0 + 1;
        const int bcol = ci == 0 ? (bh0 == 2 ? (bi & 1) : 0) : 0, brow = ci == 0 ? (bh0 == 2 ? (bi >> 1) : bi) : 0;
        const int bx = ci == 0 ? mx * bh0 + bcol : mx, by = ci == 0 ? my * bv0 + brow : my;
        const int bxn = ci == 0 ? g.blocks_x[0] : (ci == 1 ? g.blocks_x[1] : g.blocks_x[2]);
        const int co = ci == 0 ? g.coef_offset[0] : (ci == 1 ? g.coef_offset[1] : g.coef_offset[2]);
        blk = cf + co + ((size_t)by * bxn + bx) * 64;
        hdc = &tab[ci]; hac = &tab[3 + ci];
    };
    if (live) locate();
    // bit window: `n` valid bits at the bottom of acc; words are consumed big-endian from the clean bytes, zeros behind the interval's end
    uint64_t acc = 0;
    int n = 0;
    uint32_t bp = 0;
    auto load_word = [&](uint32_t at) -> uint32_t {
        uint32_t v = 0;
        if (at < len) {
            v = __builtin_bswap32(*(const uint32_t*)(src + at));        // unaligned dword; reads <= 3 bytes past the interval: inside the (padded) image
            if (at + 4 > len) v &= 0xFFFFFFFFu << (8 * (at + 4 - len));
        }
        return v;
    };
    uint32_t nextw = load_word(0);
    int err = 0;
    while (__any(live)) {
        if (live) {
            if (n < 32) { acc = (acc << 32) | nextw; n += 32; bp += 4; nextw = load_word(bp); }
            const sm_jpeg_huff_t& h = *(k == 0 ? hdc : hac);
            const uint32_t look = (uint32_t)(acc >> (n - 16)) & 0xFFFFu;
            const uint32_t fe = h.fast[look >> 7];
            int clen = fe >> 8, sym = fe & 0xFF;
            if (fe == 0) {
                // a code of 10..16 bits: the seven length classes' largest codes are read together (independent LDS reads), the shortest matching
                // length wins, then ONE offset and ONE symbol read (as dependent reads per length this was ~2000 clk whenever any lane took it)
                int mc[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) mc[q] = h.maxcode[10 + q];
                clen = 17;
#pragma unroll
                for (int q = 6; q >= 0; --q) {
                    const int c = (int)(look >> (6 - q));
                    if (mc[q] >= 0 && c <= mc[q]) clen = 10 + q;
                }
                if (clen <= 16) sym = h.vals[((int)(look >> (16 - clen)) + h.valoff[clen]) & 255];
            }
            if (clen > 16) { err = 1; live = false; }
            n -= clen;
            const int size = k == 0 ? sym : (sym & 15), run = k == 0 ? 0 : (sym >> 4);
            int val = 0;
            if (size) {
                val = (int)((acc >> (n - size)) & ((1u << size) - 1));
                n -= size;
                val = val < (1 << (size - 1)) ? val - (1 << size) + 1 : val;
            }
            if (k == 0) {
                if (sym > 11) { err = 1; live = false; }
                int pr = ci == 0 ? pred0 : (ci == 1 ? pred1 : pred2);
                pr += val;
                if (ci == 0) pred0 = pr; else if (ci == 1) pred1 = pr; else pred2 = pr;
                blk[0] = (int16_t)pr;
                k = 1;
            } else if (size == 0) {
                k = run == 15 ? k + 16 : 64;
            } else {
                k += run;
                if (k > 63) { err = 2; live = false; }
                else blk[zz[k]] = (int16_t)val;
                ++k;
            }
            if (k >= 64 && live) {                              // next block / MCU
                k = 0;
                if (++bi == nbm) {
                    bi = 0;
                    if (++mcu == mcu_end) live = false;
                    if (++mx == g.mcus_x) { mx = 0; ++my; }
                }
                if (live) locate();
            }
        }
    }
    if (err) atomicCAS(&status[f], 0, err);
}

// ------------------------------------------------------------------------------------------------ entropy decode without restart markers
// One serial Huffman stream per frame, decoded in parallel by SELF-SYNCHRONISATION (Klein & Wiseman 2003; Weissenberger & Schmidt, "Massively parallel
// Huffman decoding on GPUs" 2018 and its JPEG follow-up 2021 -- restated from the published idea): the unstuffed stream is cut into subsequences of JS_BITS
// bits, one lane each.  A lane's decoder state is (bit position, block inside the MCU, next zig-zag index); lane 0 knows its state, every other lane GUESSES
// (start of a block, first bit of its subsequence) and decodes to the end of its subsequence, leaving its exit state.  Round r: every lane takes the exit
// state its predecessor left in round r - 1 as its entry state and decodes again -- unless that entry state did not change.  A prefix code pulls a wrong
// decoder onto the right bit boundaries within a few symbols and the block structure pulls the rest of the state along, so the exit states stop changing after
// a few rounds (lane s is right after round s at the latest).  Then the lanes' completed-block counts are prefix-summed (every lane knows which block it
// starts in), a last pass writes the coefficients -- the DC DIFFERENCE in place of the DC -- and a per-component scan over the blocks in scan order turns
// the differences into values.  Same decode step as the restart-interval kernel (one uniform loop: one symbol per trip for every live lane).
#ifndef JS_LOG
#define JS_LOG 10
#endif
#define JS_BITS (1 << JS_LOG)                                   // bits per subsequence
#define JS_WLOG (JS_LOG - 5)                                    // log2 of its dwords
#define JS_ROUNDS 16                                            // at most; a frame whose records chain leaves the later rounds at once
struct JsGeom { int mcus_x, mcus_y, ncomp, bh0, bv0, nb0, nbm, blocks_x[3], coef_offset[3], coef_count, total_blocks; };
struct JsArr {
    uint64_t* rec;                                              // [lanes] one record per subsequence (layout below)
    uint32_t* base;                                             // [lanes] exclusive prefix sum of the records' completed blocks
    uint32_t* clean_len;                                        // [frames] bytes of the unstuffed stream
    int32_t* raw_end;                                           // [frames] raw position of the first marker behind the data
    uint32_t* tile_cnt;                                         // [frames][tiles_max] bytes every 4096-byte tile of the raw segment keeps
    int32_t* changed;                                           // [frames][JS_ROUNDS + 1] lanes that had work in round r; [JS_ROUNDS] = rounds until none had
};
__device__ __forceinline__ uint32_t js_lane0(const uint32_t* offsets, int f) { return (offsets[f] >> (JS_LOG - 3)) + (uint32_t)f; }      // first lane record of frame f

// Unstuffing of a whole scan, two launches over (tiles of 4096 bytes, frames): every tile counts the bytes it keeps (everything but the 0x00 behind a data
// 0xFF) and reports the first marker it sees; then every tile sums the counts of the tiles before it, ranks its own threads and writes its bytes compacted
// at that place.  The first marker ends the data (the tile that holds it writes the stream's length).  The clean stream starts 16-byte aligned.
#define JS_TILE 4096
__device__ __forceinline__ uint32_t js_clean_at(const uint32_t* offsets, const sm_jpeg_scan_t& sc, int f) { return offsets[f] + (sc.scan_offset & ~15u); }
__device__ __forceinline__ void js_tile_bytes(const uint8_t* src, uint32_t n, uint32_t lo, uint8_t (&b)[18]) {                   // b[e] = byte lo + e - 1
#pragma unroll
    for (int e = 0; e < 18; ++e) { const uint32_t p = lo + e; b[e] = (p >= 1 && p - 1 < n) ? src[p - 1] : 0; }
}
__global__ __launch_bounds__(256) void jpeg_unstuff_count_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ offsets, const sm_jpeg_scan_t* __restrict__ scans,
                                                                 JsArr A, int tiles_max) {
    __shared__ int cnt[4];
    const int f = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const sm_jpeg_scan_t& sc = scans[f];
    const uint32_t n = sc.scan_len;
    if ((uint32_t)t * JS_TILE >= n) return;
    const uint8_t* src = bytes + offsets[f] + sc.scan_offset;
    const uint32_t lo = (uint32_t)t * JS_TILE + tid * 16;
    uint8_t b[18];
    js_tile_bytes(src, n, lo, b);
    int keep = 0, my_end = 0x7fffffff;
#pragma unroll
    for (int e = 15; e >= 0; --e) {
        const uint32_t p = lo + e;
        if (p + 1 < n && b[e + 1] == 0xFF && b[e + 2] != 0x00) my_end = (int)p;
        if (p < n && !(b[e + 1] == 0x00 && b[e] == 0xFF)) ++keep;
    }
    if (my_end != 0x7fffffff) atomicMin(&A.raw_end[f], my_end);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) keep += __shfl_xor(keep, o);
    if ((tid & 63) == 0) cnt[tid >> 6] = keep;
    __syncthreads();
    if (tid == 0) A.tile_cnt[(size_t)f * tiles_max + t] = (uint32_t)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
}
__global__ __launch_bounds__(256) void jpeg_unstuff_write_kernel(const uint8_t* __restrict__ bytes, const uint32_t* __restrict__ offsets, const sm_jpeg_scan_t* __restrict__ scans,
                                                                 uint8_t* __restrict__ clean, JsArr A, int tiles_max) {
    __shared__ int part[4];
    __shared__ int wsum[4];
    const int f = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const sm_jpeg_scan_t& sc = scans[f];
    const uint32_t n = sc.scan_len;
    const int end = A.raw_end[f] < (int)n ? A.raw_end[f] : (int)n;                  // raw bytes [0, end) are data
    if ((int)((uint32_t)t * JS_TILE) >= end && !(end == 0 && t == 0)) return;
    const uint8_t* src = bytes + offsets[f] + sc.scan_offset;
    uint8_t* dst = clean + js_clean_at(offsets, sc, f);
    int before = 0;                                                                // kept bytes of the tiles before this one (all of them wholly data)
    for (int i = tid; i < t; i += 256) before += (int)A.tile_cnt[(size_t)f * tiles_max + i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o);
    if ((tid & 63) == 0) part[tid >> 6] = before;
    const uint32_t lo = (uint32_t)t * JS_TILE + tid * 16;
    uint8_t b[18];
    js_tile_bytes(src, n, lo, b);
    int keep = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int p = (int)lo + e;
        if (p < end && !(b[e + 1] == 0x00 && b[e] == 0xFF)) ++keep;
    }
    int incl = keep;                                                               // inclusive scan inside the wave, then over the four waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if ((tid & 63) >= o) incl += v; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    before = part[0] + part[1] + part[2] + part[3];
    int wbase = 0;
    for (int w = 0; w < (tid >> 6); ++w) wbase += wsum[w];
    int out = before + wbase + incl - keep;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int p = (int)lo + e;
        if (p < end && !(b[e + 1] == 0x00 && b[e] == 0xFF)) dst[out++] = b[e + 1];
    }
    if (tid == 255 && (int)((uint32_t)(t + 1) * JS_TILE) >= end) A.clean_len[f] = (uint32_t)(before + wsum[0] + wsum[1] + wsum[2] + wsum[3]);      // the last tile of data
}

// the shared decode step: one symbol (+ value bits) per call.  `h` tables in LDS; returns false on a code no table holds (only a wrong guess or the padding
// behind the last block produces one: the caller decides what that means)
struct JsDec {
    const uint32_t* src; uint32_t len;         // unstuffed stream of the frame (16-byte aligned), its length in bytes
    const uint32_t* win; uint32_t w0, nw;      // the block's window of it in LDS: dwords [w0, w0 + nw), big-endian already, zeros behind the stream's end,
                                               // one pad word per subsequence (lanes read at a stride of one subsequence: the skew spreads them over the banks)
    uint64_t acc; int n; uint32_t wn, nextw;   // bit window: n valid bits at the bottom of acc, filled up to dword wn; nextw = dword wn
    int bi, k;
    __device__ __forceinline__ uint32_t word(uint32_t w) const {
        const uint32_t i = w - w0;
        if (i < nw) return win[i + (i >> JS_WLOG)];
        uint32_t v = 0;                        // (a lane that followed the stream out of the window)
        if (w * 4 < len) {
            v = __builtin_bswap32(src[w]);
            if (w * 4 + 4 > len) v &= 0xFFFFFFFFu << (8 * (w * 4 + 4 - len));
        }
        return v;
    }
    __device__ __forceinline__ void start(uint32_t p, int bi_, int k_) {
        const uint32_t w = p >> 5;
        acc = word(w); n = 32 - (int)(p & 31); wn = w + 1; nextw = word(wn);
        bi = bi_; k = k_;
    }
    __device__ __forceinline__ uint32_t pos() const { return 32u * wn - (uint32_t)n; }
};
// the block's window of the clean stream into LDS: dwords [w0, w0 + nw) of `src`
__device__ __forceinline__ void js_stage(uint32_t* win, const uint32_t* src, uint32_t len, uint32_t w0, uint32_t nw, int tid, int nthreads) {
    for (uint32_t i = tid; i < nw; i += nthreads) {
        const uint32_t w = w0 + i;
        uint32_t v = 0;
        if (w * 4 < len) {
            v = __builtin_bswap32(src[w]);
            if (w * 4 + 4 > len) v &= 0xFFFFFFFFu << (8 * (w * 4 + 4 - len));
        }
        win[i + (i >> JS_WLOG)] = v;
    }
}
// decodes one symbol; out: is_dc, zig-zag index written (or -1), value; advances (bi, k); `done_block` when the block completed
template <class TAB>
__device__ __forceinline__ bool js_step(JsDec& d, const TAB* tab, int nb0, int nbm, int& widx, int& val, bool& done_block) {
    if (d.n < 32) { d.acc = (d.acc << 32) | d.nextw; d.n += 32; d.wn += 1; d.nextw = d.word(d.wn); }
    const int ci = d.bi < nb0 ? 0 : d.bi - nb0 + 1;
    const sm_jpeg_huff_t& h = tab[(d.k == 0 ? 0 : 3) + ci];
    const uint32_t look = (uint32_t)(d.acc >> (d.n - 16)) & 0xFFFFu;
    const uint32_t fe = h.fast[look >> 7];
    int clen = fe >> 8, sym = fe & 0xFF;
    bool ok = true;
    if (fe == 0) {
        int mc[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) mc[q] = h.maxcode[10 + q];
        clen = 17;
#pragma unroll
        for (int q = 6; q >= 0; --q) {
            const int c = (int)(look >> (6 - q));
            if (mc[q] >= 0 && c <= mc[q]) clen = 10 + q;
        }
        if (clen <= 16) sym = h.vals[((int)(look >> (16 - clen)) + h.valoff[clen]) & 255];
        else { ok = false; clen = 16; sym = 0; }               // progress anyway: 16 bits, "size 0 / end of block"
    }
    d.n -= clen;
    int size = d.k == 0 ? sym : (sym & 15);
    const int run = d.k == 0 ? 0 : (sym >> 4);
    if (d.k == 0 && size > 11) { ok = false; size = 0; }
    val = 0;
    if (size) {
        val = (int)((d.acc >> (d.n - size)) & ((1u << size) - 1));
        d.n -= size;
        val = val < (1 << (size - 1)) ? val - (1 << size) + 1 : val;
    }
    widx = -1;
    if (d.k == 0) { widx = 0; d.k = 1; }
    else if (size == 0) d.k = run == 15 ? d.k + 16 : 64;
    else {
        d.k += run;
        if (d.k > 63) { ok = false; d.k = 64; }
        else { widx = d.k; ++d.k; }
    }
    done_block = d.k >= 64;
    if (done_block) { d.k = 0; d.bi = d.bi + 1 == nbm ? 0 : d.bi + 1; }
    return ok;
}

// One record per subsequence, ONE 64-bit word (written and read whole, so a record is always a true statement "decoding this subsequence from ENTRY leaves
// EXIT after NBLK completed blocks"):  bits 0-14 exit state, 15-24 blocks completed, 25-39 entry state; a state = bit offset past the subsequence's first bit
// (0..30: a symbol is at most 31 bits) | block inside the MCU << 5 | next zig-zag index << 8.  All ones = no record yet.
// The frame is decoded when the records CHAIN: record[0] enters at (0, 0, 0) and record[s] enters where record[s - 1] exits.
#define JS_CHAIN 64                                             // subsequences a lane follows downstream in one round
__device__ __forceinline__ uint32_t js_exit(uint64_t r) { return (uint32_t)r & 0x7FFFu; }
// (blocks completed: a block is at least 2 bits -- tables built for a flat image give the DC difference 0 and the end-of-block ONE bit each --, so up to
//  ~530 blocks end inside 1024 + 30 bits: ten bits; nine were one too few for exactly such images)
__device__ __forceinline__ uint32_t js_nblk(uint64_t r) { return (uint32_t)(r >> 15) & 0x3FFu; }
__device__ __forceinline__ uint32_t js_entry(uint64_t r) { return (uint32_t)(r >> 25) & 0x7FFFu; }
__device__ __forceinline__ uint64_t js_pack(uint32_t entry, uint32_t exit, uint32_t nb) { return (uint64_t)exit | ((uint64_t)(nb & 0x3FFu) << 15) | ((uint64_t)entry << 25); }
__device__ __forceinline__ uint64_t js_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void js_store(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void jpeg_sync_init_kernel(JsArr A, size_t n_lanes, int n_frames) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_lanes) A.rec[i] = ~0ull;
    if (i < (size_t)n_frames) { A.raw_end[i] = 0x7fffffff; A.clean_len[i] = 0; }
    if (i < (size_t)n_frames * (JS_ROUNDS + 1)) A.changed[i] = 0;
}

// One relaxation round.  Lane s looks at its record: if it enters where record[s - 1] exits, nothing to do.  Otherwise it decodes subsequence s from that
// state, stores the record, and FOLLOWS the stream into s + 1, s + 2, ... (the decoder simply keeps going), replacing their records, until the exit state it
// reaches is the one already on record there -- from that point the downstream records were derived from the same state -- or JS_CHAIN subsequences.
// Round 0: no records, every lane starts from the guess (start of an MCU at its first bit) and follows into its successor, which is where most guesses have
// synchronised.  Lanes race on the records they replace; every record is true by itself, only the chaining can be left broken, and the next round's lanes
// see exactly that.  A round in which no lane had anything to do proves the chain (jpeg_blockscan_kernel and the writing pass check it once more).
#define JS_TPB 256                                              // lanes (subsequences) per block
#define JS_AHEAD 32                                             // subsequences behind the block's own that its window holds (a lane following the stream)
__global__ __launch_bounds__(JS_TPB) void jpeg_sync_kernel(const uint8_t* __restrict__ clean, const uint32_t* __restrict__ offsets, const sm_jpeg_scan_t* __restrict__ scans,
                                                           JsArr A, JsGeom g, int round) {
    __shared__ sm_jpeg_huff_t tab[6];
    __shared__ uint32_t win[(JS_TPB + JS_AHEAD) * (JS_BITS / 32 + 1)];
    const int f = blockIdx.y, tid = threadIdx.x;
    const sm_jpeg_scan_t& sc = scans[f];
    const uint32_t len = A.clean_len[f];
    const uint32_t S = (len * 8 + JS_BITS - 1) / JS_BITS;
    if (blockIdx.x * (uint32_t)JS_TPB >= S) return;
    if (round > 0 && A.changed[f * (JS_ROUNDS + 1) + round - 1] == 0) return;       // the round before found the chain whole
    const uint32_t s = blockIdx.x * JS_TPB + tid;
    uint64_t* rec = A.rec + js_lane0(offsets, f);
    bool live = s < S;
    uint32_t entry = 0;
    if (live && s > 0) entry = round == 0 ? 0u : js_exit(js_load(rec + s - 1));
    if (live && round > 0 && js_entry(js_load(rec + s)) == entry) live = false;
    if (!__syncthreads_or(live)) return;
    JsDec d;
    d.src = (const uint32_t*)(clean + js_clean_at(offsets, sc, f)); d.len = len;
    d.win = win; d.w0 = blockIdx.x * JS_TPB * (JS_BITS / 32); d.nw = (JS_TPB + JS_AHEAD) * (JS_BITS / 32);
    {
        const uint32_t* src = (const uint32_t*)sc.dc;
        uint32_t* dst = (uint32_t*)tab;
        for (int w = tid; w < (int)(6 * sizeof(sm_jpeg_huff_t) / 4); w += JS_TPB) dst[w] = src[w];
        js_stage(win, d.src, len, d.w0, d.nw, tid, JS_TPB);
    }
    __syncthreads();
    if (live) atomicAdd(&A.changed[f * (JS_ROUNDS + 1) + round], 1);
    uint32_t j = s, nb = 0, chain = 0;
    uint32_t end = (j + 1) * JS_BITS < len * 8 ? (j + 1) * JS_BITS : len * 8;
    if (live) d.start(s * JS_BITS + (entry & 31), (int)((entry >> 5) & 7), (int)(entry >> 8));
    while (__any(live)) {
        if (live) {
            int widx, val; bool done;
            (void)js_step(d, tab, g.nb0, g.nbm, widx, val, done);
            nb += done ? 1u : 0u;
            if (d.pos() >= end) {                                // subsequence j is behind the decoder
                const uint32_t exit = (d.pos() - end) | ((uint32_t)d.bi << 5) | ((uint32_t)d.k << 8);
                const uint64_t old = js_load(rec + j);
                // (its own subsequence: if the record it started from has been replaced meanwhile, the result is stale -- leave it to the lane that did that)
                const bool stale = j == s && s > 0 && round > 0 && js_exit(js_load(rec + s - 1)) != entry;
                if (!stale) js_store(rec + j, js_pack(entry, exit, nb));
                if (stale || js_exit(old) == exit || j + 1 >= S || ++chain >= JS_CHAIN) live = false;
                else {
                    ++j; entry = exit; nb = 0;                   // the exit state counts from the next subsequence's first bit: its entry state as it is
                    end = (j + 1) * JS_BITS < len * 8 ? (j + 1) * JS_BITS : len * 8;
                }
            }
        }
    }
}

// exclusive prefix sum of the records' completed-block counts (one block per frame); is the chain whole, does it hold all the frame's blocks
__global__ __launch_bounds__(1024) void jpeg_blockscan_kernel(const uint32_t* __restrict__ offsets, JsArr A, JsGeom g, uint32_t lanes_launched,
                                                              int32_t* __restrict__ status) {
    __shared__ uint32_t cnt[1024];
    __shared__ uint32_t carry;
    __shared__ int broken;
    const int f = blockIdx.x, tid = threadIdx.x;
    const uint32_t len = A.clean_len[f], S = (len * 8 + JS_BITS - 1) / JS_BITS, L0 = js_lane0(offsets, f);
    if (tid == 0) { carry = 0; broken = 0; }
    __syncthreads();
    if (S > lanes_launched) {                                   // a file longer than the caller said: its tail was never decoded
        if (tid == 0) atomicCAS(&status[f], 0, 5);
        return;
    }
    for (uint32_t s0 = 0; s0 < S; s0 += 1024) {
        const uint32_t s = s0 + tid;
        uint32_t v = 0;
        if (s < S) {
            const uint64_t r = A.rec[L0 + s];
            v = js_nblk(r);
            if (js_entry(r) != (s == 0 ? 0u : js_exit(A.rec[L0 + s - 1]))) broken = 1;
        }
        cnt[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t t = tid >= o ? cnt[tid - o] : 0;
            __syncthreads();
            cnt[tid] += t;
            __syncthreads();
        }
        if (s < S) A.base[L0 + s] = carry + cnt[tid] - v;
        __syncthreads();
        if (tid == 1023) carry += cnt[1023];
        __syncthreads();
    }
    if (tid == 0) {
        int r = 0;
        while (r < JS_ROUNDS && A.changed[f * (JS_ROUNDS + 1) + r] != 0) ++r;
        A.changed[f * (JS_ROUNDS + 1) + JS_ROUNDS] = r;                                    // (sm_jpeg_sync_rounds reads it)
        if (broken) atomicCAS(&status[f], 0, 5);                                           // JS_ROUNDS rounds did not mend the chain
        else if (carry < (uint32_t)g.total_blocks) atomicCAS(&status[f], 0, 4);            // the stream ends before the last block
    }
}

// the writing pass: every lane decodes its subsequence once more from its record's entry state, blocks numbered from the prefix sum; what it finds has to be
// the record again (exit state, blocks)
__global__ __launch_bounds__(JS_TPB) void jpeg_write_kernel(const uint8_t* __restrict__ clean, const uint32_t* __restrict__ offsets, const sm_jpeg_scan_t* __restrict__ scans,
                                                            JsArr A, JsGeom g, int16_t* __restrict__ coefs, uint16_t* __restrict__ qt, int32_t* __restrict__ status) {
    __shared__ sm_jpeg_huff_t tab[6];
    __shared__ uint32_t win[(JS_TPB + 1) * (JS_BITS / 32 + 1)];
    __shared__ uint8_t zz[64];
    const int f = blockIdx.y, tid = threadIdx.x;
    const sm_jpeg_scan_t& sc = scans[f];
    const uint32_t len = A.clean_len[f];
    const uint32_t S = (len * 8 + JS_BITS - 1) / JS_BITS;
    if (blockIdx.x * (uint32_t)JS_TPB >= S || status[f] != 0) return;
    JsDec d;
    d.src = (const uint32_t*)(clean + js_clean_at(offsets, sc, f)); d.len = len;
    d.win = win; d.w0 = blockIdx.x * JS_TPB * (JS_BITS / 32); d.nw = (JS_TPB + 1) * (JS_BITS / 32);
    {
        const uint32_t* src = (const uint32_t*)sc.dc;
        uint32_t* dst = (uint32_t*)tab;
        for (int w = tid; w < (int)(6 * sizeof(sm_jpeg_huff_t) / 4); w += JS_TPB) dst[w] = src[w];
        if (tid < 64) zz[tid] = kZigzagDev[tid];
        if (blockIdx.x == 0 && tid < 3 * 64) qt[(size_t)f * 192 + tid] = sc.qt[tid / 64][tid % 64];
        js_stage(win, d.src, len, d.w0, d.nw, tid, JS_TPB);
    }
    __syncthreads();
    const uint32_t s = blockIdx.x * JS_TPB + tid;
    const uint32_t L = js_lane0(offsets, f) + s;
    bool live = s < S;
    const uint64_t mine = live ? A.rec[L] : 0ull;
    const uint32_t entry = js_entry(mine);
    const uint32_t end = (s + 1) * JS_BITS < len * 8 ? (s + 1) * JS_BITS : len * 8;
    int G = live ? (int)A.base[L] : 0;                          // the block in progress at the entry
    const bool check = live;                                    // (lanes behind the frame's last block have nothing to write, their records still hold)
    live = live && G < g.total_blocks;
    if (live) d.start(s * JS_BITS + (entry & 31), (int)((entry >> 5) & 7), (int)(entry >> 8));
    int16_t* cf = coefs + (size_t)f * g.coef_count;
    int mcu = live ? G / g.nbm : 0, mx = live ? mcu % g.mcus_x : 0, my = live ? mcu / g.mcus_x : 0;
    int16_t* blk = cf;
    auto locate = [&]() {
        const int bi = d.bi;
        const int ci = bi < g.nb0 ? 0 : bi - g.nb0 + 1;
        const int bcol = ci == 0 ? (g.bh0 == 2 ? (bi & 1) : 0) : 0, brow = ci == 0 ? (g.bh0 == 2 ? (bi >> 1) : bi) : 0;
        const int bx = ci == 0 ? mx * g.bh0 + bcol : mx, by = ci == 0 ? my * g.bv0 + brow : my;
        const int bxn = ci == 0 ? g.blocks_x[0] : (ci == 1 ? g.blocks_x[1] : g.blocks_x[2]);
        const int co = ci == 0 ? g.coef_offset[0] : (ci == 1 ? g.coef_offset[1] : g.coef_offset[2]);
        blk = cf + co + ((size_t)by * bxn + bx) * 64;
    };
    int err = 0;
    uint32_t nb = 0;
    bool whole = false;                                         // decoded to the subsequence's end
    if (live) {
        if (d.bi != G % g.nbm) { err = 5; live = false; }      // the state chain and the block count disagree
        else locate();
    }
    while (__any(live)) {
        if (live) {
            int widx, val; bool done;
            const bool ok = js_step(d, tab, g.nb0, g.nbm, widx, val, done);
            if (!ok) { err = 1; live = false; }
            else {
                if (widx >= 0) blk[zz[widx]] = (int16_t)val;     // widx == 0: the DC DIFFERENCE (jpeg_dc_kernel sums them)
                if (done) {
                    ++G; ++nb;
                    if (d.bi == 0) { if (++mx == g.mcus_x) { mx = 0; ++my; } }
                    if (d.pos() >= end) { live = false; whole = true; }
                    else if (G >= g.total_blocks) live = false;
                    else locate();
                } else if (d.pos() >= end) { live = false; whole = true; }
            }
        }
    }
    if (check && whole && !err) {
        const uint32_t exit = (d.pos() - end) | ((uint32_t)d.bi << 5) | ((uint32_t)d.k << 8);
        if (exit != js_exit(mine) || nb != js_nblk(mine)) err = 5;
    }
    if (err) atomicCAS(&status[f], 0, err);
}

// DC differences -> DC values: per (component, frame) an inclusive scan over the component's blocks in scan order (1024 threads x 4 blocks per tile)
__global__ __launch_bounds__(1024) void jpeg_dc_kernel(JsGeom g, int16_t* __restrict__ coefs, const int32_t* __restrict__ status) {
    __shared__ int cnt[1024];
    __shared__ int carry;
    const int ci = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    if (ci >= g.ncomp || status[f] != 0) return;
    const int nbc = ci == 0 ? g.nb0 : 1, bh = ci == 0 ? g.bh0 : 1;
    const int N = g.mcus_x * g.mcus_y * nbc;
    int16_t* cf = coefs + (size_t)f * g.coef_count + g.coef_offset[ci];
    const int bxn = g.blocks_x[ci];
    auto addr = [&](int j) -> int16_t* {
        const int mcu = j / nbc, b = j - mcu * nbc;
        const int mx = mcu % g.mcus_x, my = mcu / g.mcus_x;
        const int bcol = bh == 2 ? (b & 1) : 0, brow = bh == 2 ? (b >> 1) : b;
        const int bvs = ci == 0 ? g.bv0 : 1;
        return cf + ((size_t)(my * bvs + brow) * bxn + mx * bh + bcol) * 64;
    };
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int j0 = 0; j0 < N; j0 += 4096) {
        int v[4], sum = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int j = j0 + tid * 4 + e; v[e] = j < N ? (int)addr(j)[0] : 0; sum += v[e]; }
        cnt[tid] = sum;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int t = tid >= o ? cnt[tid - o] : 0;
            __syncthreads();
            cnt[tid] += t;
            __syncthreads();
        }
        int run = carry + cnt[tid] - sum;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int j = j0 + tid * 4 + e; run += v[e]; if (j < N) addr(j)[0] = (int16_t)run; }
        __syncthreads();
        if (tid == 1023) carry += cnt[1023];
        __syncthreads();
    }
}

__global__ void jpeg_zero_kernel(u32x4* __restrict__ p, size_t n16, int32_t* __restrict__ status, int n_frames) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n16) p[t] = u32x4{0, 0, 0, 0};
    if (t < (size_t)n_frames) status[t] = 0;
}

// per-HIP-stream index of interval starts (n_frames x JH_MAX_INT words, grown on demand)
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>
static std::mutex g_jh_mu;
struct JhWs { uint32_t* starts = nullptr; uint32_t* lens = nullptr; size_t frames = 0; uint8_t* clean = nullptr; size_t clean_bytes = 0; };
static std::map<hipStream_t, JhWs> g_jh_ws;
extern "C" int sm_jpeg_entropy_decode(const uint8_t* bytes, size_t bytes_total, const uint32_t* offsets, const sm_jpeg_scan_t* scans, const sm_jpeg_info_t* info, int n_frames,
                                      int16_t* coefs, uint16_t* qt, int32_t* status, void* stream) {
    SM_REQUIRE(bytes && bytes_total > 0 && offsets && scans && info && coefs && qt && status && n_frames >= 1, "sm_jpeg_entropy_decode: null arg / no frames");
    SM_REQUIRE((info->ncomp == 1 || info->ncomp == 3) && info->coef_count > 0 && (((size_t)info->coef_count * 2) % 16) == 0 && ((uintptr_t)coefs & 15) == 0,
               "sm_jpeg_entropy_decode: bad info / unaligned coefficient image");
    hipStream_t st = (hipStream_t)stream;
    JhWs ws;
    {
        std::lock_guard<std::mutex> lk(g_jh_mu);
        JhWs& e = g_jh_ws[st];
        if (e.frames < (size_t)n_frames) {
            if (e.starts) { SM_HIP(hipStreamSynchronize(st)); (void)hipFree(e.starts); (void)hipFree(e.lens); e.starts = e.lens = nullptr; e.frames = 0; }
            SM_HIP(hipMalloc((void**)&e.starts, (size_t)n_frames * JH_MAX_INT * sizeof(uint32_t)));
            SM_HIP(hipMalloc((void**)&e.lens, (size_t)n_frames * JH_MAX_INT * sizeof(uint32_t)));
            e.frames = n_frames;
        }
        if (e.clean_bytes < bytes_total + 64) {
            if (e.clean) { SM_HIP(hipStreamSynchronize(st)); (void)hipFree(e.clean); e.clean = nullptr; e.clean_bytes = 0; }
            SM_HIP(hipMalloc((void**)&e.clean, bytes_total + 64));
            e.clean_bytes = bytes_total + 64;
        }
        ws = e;
    }
    const size_t n16 = (size_t)n_frames * info->coef_count * 2 / 16;
    jpeg_zero_kernel<<<(unsigned)((n16 + 255) / 256), 256, 0, st>>>((u32x4*)coefs, n16, status, n_frames);
    SM_LAUNCH_CHECK();
    jpeg_rst_index_kernel<<<n_frames, 1024, 0, st>>>(bytes, offsets, scans, ws.starts, status);
    SM_LAUNCH_CHECK();
    // the interval count of a frame is in its scan struct (device); the grids cover the largest possible count for this geometry: one interval per MCU
    const int max_iv = info->mcus_x * info->mcus_y < JH_MAX_INT ? info->mcus_x * info->mcus_y : JH_MAX_INT;
    jpeg_unstuff_kernel<<<dim3(cdiv(max_iv, 4), n_frames), 256, 0, st>>>(bytes, offsets, scans, ws.starts, ws.clean, ws.lens, status);
    SM_LAUNCH_CHECK();
    JhGeom g;
    memset(&g, 0, sizeof(g));
    g.mcus_x = info->mcus_x; g.mcus_y = info->mcus_y; g.ncomp = info->ncomp; g.coef_count = info->coef_count;
    for (int c = 0; c < info->ncomp; ++c) { g.hs[c] = info->hs[c]; g.vs[c] = info->vs[c]; g.blocks_x[c] = info->blocks_x[c]; g.coef_offset[c] = info->coef_offset[c]; }
    jpeg_huff_kernel<<<dim3(cdiv(max_iv, 64), n_frames), 64, 0, st>>>(ws.clean, offsets, scans, ws.starts, ws.lens, g, coefs, qt, status);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// frames WITHOUT restart markers (sm_jpeg_scan_t.restart == 0 for every frame of the batch): the self-synchronising decode above.  Same arguments and
// results as sm_jpeg_entropy_decode; status 5 = the records did not chain after JS_ROUNDS rounds (the caller falls back to the host decoder).
struct JsWs { uint32_t* lanes = nullptr; size_t n_lanes = 0; uint32_t* per_frame = nullptr; size_t frames = 0; int last_frames = 0; };
static std::map<hipStream_t, JsWs> g_js_ws;
extern "C" int sm_jpeg_entropy_decode_sync(const uint8_t* bytes, size_t bytes_total, size_t max_file_bytes, const uint32_t* offsets, const sm_jpeg_scan_t* scans,
                                           const sm_jpeg_info_t* info, int n_frames, int16_t* coefs, uint16_t* qt, int32_t* status, void* stream) {
    SM_REQUIRE(bytes && bytes_total > 0 && offsets && scans && info && coefs && qt && status && n_frames >= 1, "sm_jpeg_entropy_decode_sync: null arg / no frames");
    SM_REQUIRE(max_file_bytes > 0 && max_file_bytes <= bytes_total, "sm_jpeg_entropy_decode_sync: max_file_bytes %zu outside (0, bytes_total]", max_file_bytes);
    SM_REQUIRE(bytes_total < ((size_t)1 << 31) && max_file_bytes < ((size_t)1 << 28), "sm_jpeg_entropy_decode_sync: %zu bytes per batch / %zu per file (offsets are 32-bit, bit positions 31-bit)",
               bytes_total, max_file_bytes);
    SM_REQUIRE((info->ncomp == 1 || info->ncomp == 3) && info->coef_count > 0 && (((size_t)info->coef_count * 2) % 16) == 0 && ((uintptr_t)coefs & 15) == 0,
               "sm_jpeg_entropy_decode_sync: bad info / unaligned coefficient image");
    hipStream_t st = (hipStream_t)stream;
    const size_t n_lanes = bytes_total / (JS_BITS / 8) + (size_t)n_frames + 64;
    const int tiles_max = (int)((max_file_bytes + JS_TILE - 1) / JS_TILE);
    const size_t per_frame_words = (size_t)n_frames * (JS_ROUNDS + 3 + (size_t)tiles_max);       // clean_len, raw_end, changed[JS_ROUNDS + 1], tile_cnt[tiles_max]
    JhWs ws; JsWs js;
    {
        std::lock_guard<std::mutex> lk(g_jh_mu);
        JhWs& e = g_jh_ws[st];
        if (e.clean_bytes < bytes_total + 64) {
            if (e.clean) { SM_HIP(hipStreamSynchronize(st)); (void)hipFree(e.clean); e.clean = nullptr; e.clean_bytes = 0; }
            SM_HIP(hipMalloc((void**)&e.clean, bytes_total + 64));
            e.clean_bytes = bytes_total + 64;
        }
        ws = e;
        JsWs& j = g_js_ws[st];
        if (j.n_lanes < n_lanes) {
            if (j.lanes) { SM_HIP(hipStreamSynchronize(st)); (void)hipFree(j.lanes); j.lanes = nullptr; j.n_lanes = 0; }
            SM_HIP(hipMalloc((void**)&j.lanes, n_lanes * 3 * sizeof(uint32_t)));
            j.n_lanes = n_lanes;
        }
        if (j.frames < per_frame_words) {
            if (j.per_frame) { SM_HIP(hipStreamSynchronize(st)); (void)hipFree(j.per_frame); j.per_frame = nullptr; j.frames = 0; }
            SM_HIP(hipMalloc((void**)&j.per_frame, per_frame_words * sizeof(uint32_t)));
            j.frames = per_frame_words;
        }
        js = j;
    }
    JsArr A;
    A.rec = (uint64_t*)js.lanes; A.base = js.lanes + 2 * n_lanes;
    A.clean_len = js.per_frame; A.raw_end = (int32_t*)(js.per_frame + n_frames); A.changed = (int32_t*)(js.per_frame + 2 * n_frames);
    A.tile_cnt = js.per_frame + (size_t)n_frames * (JS_ROUNDS + 3);
    JsGeom g;
    memset(&g, 0, sizeof(g));
    g.mcus_x = info->mcus_x; g.mcus_y = info->mcus_y; g.ncomp = info->ncomp; g.coef_count = info->coef_count;
    g.bh0 = info->ncomp > 1 ? info->hs[0] : 1; g.bv0 = info->ncomp > 1 ? info->vs[0] : 1;
    g.nb0 = g.bh0 * g.bv0; g.nbm = info->ncomp > 1 ? g.nb0 + 2 : 1;
    g.total_blocks = info->mcus_x * info->mcus_y * g.nbm;
    for (int c = 0; c < info->ncomp; ++c) { g.blocks_x[c] = info->blocks_x[c]; g.coef_offset[c] = info->coef_offset[c]; }
    const size_t n16 = (size_t)n_frames * info->coef_count * 2 / 16;
    jpeg_zero_kernel<<<(unsigned)((n16 + 255) / 256), 256, 0, st>>>((u32x4*)coefs, n16, status, n_frames);
    SM_LAUNCH_CHECK();
    {
        const size_t n_init = std::max(n_lanes, (size_t)n_frames * (JS_ROUNDS + 1));
        jpeg_sync_init_kernel<<<(unsigned)((n_init + 255) / 256), 256, 0, st>>>(A, n_lanes, n_frames);
        SM_LAUNCH_CHECK();
    }
    jpeg_unstuff_count_kernel<<<dim3(tiles_max, n_frames), 256, 0, st>>>(bytes, offsets, scans, A, tiles_max);
    SM_LAUNCH_CHECK();
    jpeg_unstuff_write_kernel<<<dim3(tiles_max, n_frames), 256, 0, st>>>(bytes, offsets, scans, ws.clean, A, tiles_max);
    SM_LAUNCH_CHECK();
    // lanes per frame: the clean stream is no longer than the file; grid.x covers the longest file of the batch (a longer one reports status 5)
    const unsigned gx = (unsigned)cdiv((int)((max_file_bytes * 8 + JS_BITS - 1) / JS_BITS), JS_TPB);
    for (int r = 0; r < JS_ROUNDS; ++r) {
        jpeg_sync_kernel<<<dim3(gx, n_frames), JS_TPB, 0, st>>>(ws.clean, offsets, scans, A, g, r);
        SM_LAUNCH_CHECK();
    }
    jpeg_blockscan_kernel<<<n_frames, 1024, 0, st>>>(offsets, A, g, gx * (unsigned)JS_TPB, status);
    SM_LAUNCH_CHECK();
    jpeg_write_kernel<<<dim3(gx, n_frames), JS_TPB, 0, st>>>(ws.clean, offsets, scans, A, g, coefs, qt, status);
    SM_LAUNCH_CHECK();
    jpeg_dc_kernel<<<dim3(3, n_frames), 1024, 0, st>>>(g, coefs, status);
    SM_LAUNCH_CHECK();
    {
        std::lock_guard<std::mutex> lk(g_jh_mu);
        g_js_ws[st].last_frames = n_frames;
    }
    return SM_OK;
}
// rounds until the states of each frame of the stream's LAST sm_jpeg_entropy_decode_sync call stopped moving (waits for the stream; a diagnostic)
extern "C" int sm_jpeg_sync_rounds(void* stream, int32_t* rounds, int n_frames) {
    hipStream_t st = (hipStream_t)stream;
    JsWs js;
    {
        std::lock_guard<std::mutex> lk(g_jh_mu);
        auto it = g_js_ws.find(st);
        SM_REQUIRE(it != g_js_ws.end() && rounds && n_frames >= 1 && n_frames <= it->second.last_frames, "sm_jpeg_sync_rounds: no decode of >= %d frames on this stream", n_frames);
        js = it->second;
    }
    SM_HIP(hipStreamSynchronize(st));
    std::vector<int32_t> all((size_t)js.last_frames * (JS_ROUNDS + 1));
    SM_HIP(hipMemcpy(all.data(), js.per_frame + 2 * (size_t)js.last_frames, all.size() * 4, hipMemcpyDeviceToHost));
    for (int f = 0; f < n_frames; ++f) rounds[f] = all[(size_t)f * (JS_ROUNDS + 1) + JS_ROUNDS];
    return SM_OK;
}

// ------------------------------------------------------------------------------------------------ device: IDCT
// jidctint.c (islow): CONST_BITS 13, PASS1_BITS 2; FIX(x) = round(x * 2^13)
#define JF_0_298631336 2446
#define JF_0_390180644 3196
#define JF_0_541196100 4433
#define JF_0_765366865 6270
#define JF_0_899976223 7373
#define JF_1_175875602 9633
#define JF_1_501321110 12299
#define JF_1_847759065 15137
#define JF_1_961570560 16069
#define JF_2_053119869 16819
#define JF_2_562915447 20995
#define JF_3_072711026 25172

// one 1-D pass of the islow IDCT on eight values; SHIFT: descale of the pass
template <int SHIFT>
__device__ __forceinline__ void idct8(const int (&in)[8], int (&out)[8]) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * JF_0_541196100;
    int tmp2 = z1 + z3 * (-JF_1_847759065);
    int tmp3 = z1 + z2 * JF_0_765366865;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) << 13;
    int tmp1 = (z2 - z3) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * JF_1_175875602;
    tmp0 *= JF_0_298631336; tmp1 *= JF_2_053119869; tmp2 *= JF_3_072711026; tmp3 *= JF_1_501321110;
    z1 *= -JF_0_899976223; z2 *= -JF_2_562915447; z3 *= -JF_1_961570560; z4 *= -JF_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    constexpr int R = 1 << (SHIFT - 1);
    out[0] = (tmp10 + tmp3 + R) >> SHIFT; out[7] = (tmp10 - tmp3 + R) >> SHIFT;
    out[1] = (tmp11 + tmp2 + R) >> SHIFT; out[6] = (tmp11 - tmp2 + R) >> SHIFT;
    out[2] = (tmp12 + tmp1 + R) >> SHIFT; out[5] = (tmp12 - tmp1 + R) >> SHIFT;
    out[3] = (tmp13 + tmp0 + R) >> SHIFT; out[4] = (tmp13 - tmp0 + R) >> SHIFT;
}

// the range-limit table of jdmaster.c behind the IDCT (index masked to 10 bits, centred on 128): identical for every value a
// decodable stream produces, and for the wild ones too
__device__ __forceinline__ uint8_t idct_limit(int v) {
    const int i = v & 0x3FF;
    return (uint8_t)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}

struct JpegGeom {
    int width, height, ncomp, hmax, vmax;
    int blocks_x[3], blocks_y[3], coef_offset[3], plane_offset[3], hs[3], vs[3];
    int coef_count, plane_bytes;
};

// 8 threads per block: pass 1 a thread owns a column (dequantise + 1-D IDCT down the column), LDS transpose, pass 2 a thread owns a row
// and writes its 8 samples as one 8-byte store.  256 threads = 32 blocks per workgroup; grid.y = frame.
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const int16_t* __restrict__ coefs, const uint16_t* __restrict__ qt, JpegGeom g, uint8_t* __restrict__ planes) {
    __shared__ int ws[32][64 + 8];
    const int f = blockIdx.y;
    const int lb = threadIdx.x >> 3, t = threadIdx.x & 7;
    int total = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) if (c < g.ncomp) total += g.blocks_x[c] * g.blocks_y[c];
    const int b = blockIdx.x * 32 + lb;
    const bool live = b < total;
    int c = 0, bi = b;
    if (live) {
        while (c + 1 < g.ncomp && bi >= g.blocks_x[c] * g.blocks_y[c]) { bi -= g.blocks_x[c] * g.blocks_y[c]; ++c; }
        const int16_t* blk = coefs + (size_t)f * g.coef_count + g.coef_offset[c] + (size_t)bi * 64;
        const uint16_t* q = qt + ((size_t)f * 3 + c) * 64;
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int)blk[r * 8 + t] * (int)q[r * 8 + t];
        idct8<11>(in, out);                     // CONST_BITS - PASS1_BITS
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb][r * 8 + t] = out[r];
    }
    __syncthreads();
    if (!live) return;
    int in[8], out[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = ws[lb][t * 8 + k];
    idct8<18>(in, out);                         // CONST_BITS + PASS1_BITS + 3
    const int bx = bi % g.blocks_x[c], by = bi / g.blocks_x[c];
    const int pw = g.blocks_x[c] * 8;
    uint8_t* dst = planes + (size_t)f * g.plane_bytes + g.plane_offset[c] + (size_t)(by * 8 + t) * pw + bx * 8;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { lo |= (uint32_t)idct_limit(out[k]) << (8 * k); hi |= (uint32_t)idct_limit(out[4 + k]) << (8 * k); }
    *(uint2*)dst = make_uint2(lo, hi);
}

// ------------------------------------------------------------------------------------------------ device: upsample + colour
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// chroma sample at full resolution (x, y) of a plane `p` (row length pw, real size cw x ch), libjpeg's fancy upsampling
template <int HS, int VS>     // HS, VS: luma-to-chroma ratio (1 or 2)
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int pw, int cw, int ch, int x, int y) {
    if (HS == 1 && VS == 1) return p[(size_t)y * pw + x];
    const int cx = x >> 1;
    if (VS == 1) {
        // h2v1_fancy_upsample: 3/4 nearer + 1/4 further, bias 1 (left output) / 2 (right output); the row's two end samples are copies
        const int v = p[(size_t)y * pw + cx];
        if (x & 1) return cx == cw - 1 ? v : (v * 3 + p[(size_t)y * pw + cx + 1] + 2) >> 2;
        return cx == 0 ? v : (v * 3 + p[(size_t)y * pw + cx - 1] + 1) >> 2;
    }
    // h2v2_fancy_upsample: column sums 3 * nearer row + further row, then the same 3 : 1 mix across columns, bias 8 / 7
    const int cy = y >> 1;
    int cy2 = (y & 1) ? cy + 1 : cy - 1;
    cy2 = cy2 < 0 ? 0 : (cy2 > ch - 1 ? ch - 1 : cy2);          // context rows above the first / below the last REAL row are copies (jdmainct.c)
    const uint8_t* r0 = p + (size_t)cy * pw;
    const uint8_t* r1 = p + (size_t)cy2 * pw;
    const int cur = r0[cx] * 3 + r1[cx];
    if (x & 1) {
        if (cx == cw - 1) return (cur * 4 + 7) >> 4;
        return (cur * 3 + (r0[cx + 1] * 3 + r1[cx + 1]) + 7) >> 4;
    }
    if (cx == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + (r0[cx - 1] * 3 + r1[cx - 1]) + 8) >> 4;
}

template <int HS, int VS>
__global__ __launch_bounds__(256) void jpeg_rgb_kernel(const uint8_t* __restrict__ planes, JpegGeom g, uint8_t* __restrict__ rgb) {
    const int f = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)g.width * g.height) return;
    const int y = (int)(t / g.width), x = (int)(t % g.width);
    const uint8_t* base = planes + (size_t)f * g.plane_bytes;
    const int Y = base[g.plane_offset[0] + (size_t)y * (g.blocks_x[0] * 8) + x];
    uint8_t* o = rgb + ((size_t)f * g.width * g.height + t) * 3;
    if (g.ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; return; }
    const int cw = (g.width + HS - 1) / HS, ch = (g.height + VS - 1) / VS;
    const int cb = chroma_at<HS, VS>(base + g.plane_offset[1], g.blocks_x[1] * 8, cw, ch, x, y) - 128;
    const int cr = chroma_at<HS, VS>(base + g.plane_offset[2], g.blocks_x[2] * 8, cw, ch, x, y) - 128;
    // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
    const int r = Y + ((91881 * cr + 32768) >> 16);
    const int gg = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    const int b = Y + ((116130 * cb + 32768) >> 16);
    o[0] = (uint8_t)clamp255(r); o[1] = (uint8_t)clamp255(gg); o[2] = (uint8_t)clamp255(b);
}

extern "C" size_t sm_jpeg_planes_bytes(const sm_jpeg_info_t* info, int n_frames) {
    if (!info) return 0;
    size_t b = 0;
    for (int c = 0; c < info->ncomp; ++c) b += (size_t)info->blocks_x[c] * info->blocks_y[c] * 64;
    return b * (size_t)(n_frames > 0 ? n_frames : 0);
}

// coefs int16 [n][coef_count], qt uint16 [n][3][64] (both on the device) -> rgb u8 [n][height][width][3]; planes: scratch of
// sm_jpeg_planes_bytes(info, n) bytes (the component sample planes between the two kernels)
extern "C" int sm_jpeg_reconstruct(const int16_t* coefs, const uint16_t* qt, const sm_jpeg_info_t* info, int n_frames, uint8_t* planes, uint8_t* rgb, void* stream) {
    SM_REQUIRE(coefs && qt && info && planes && rgb && n_frames >= 1, "sm_jpeg_reconstruct: null arg / no frames");
    SM_REQUIRE((info->ncomp == 1 || info->ncomp == 3) && info->width > 0 && info->height > 0, "sm_jpeg_reconstruct: bad info");
    JpegGeom g;
    memset(&g, 0, sizeof(g));
    g.width = info->width; g.height = info->height; g.ncomp = info->ncomp; g.hmax = info->hs[0]; g.vmax = info->vs[0];
    int total = 0, poff = 0;
    for (int c = 0; c < info->ncomp; ++c) {
        g.blocks_x[c] = info->blocks_x[c]; g.blocks_y[c] = info->blocks_y[c]; g.coef_offset[c] = info->coef_offset[c]; g.hs[c] = info->hs[c]; g.vs[c] = info->vs[c];
        g.plane_offset[c] = poff;
        poff += info->blocks_x[c] * info->blocks_y[c] * 64;
        total += info->blocks_x[c] * info->blocks_y[c];
    }
    g.coef_count = info->coef_count; g.plane_bytes = poff;
    hipStream_t st = (hipStream_t)stream;
    jpeg_idct_kernel<<<dim3(cdiv(total, 32), n_frames), 256, 0, st>>>(coefs, qt, g, planes);
    SM_LAUNCH_CHECK();
    const dim3 grid(cdiv(info->width * info->height, 256), n_frames);
    const int hs = info->ncomp == 3 ? info->hs[0] : 1, vs = info->ncomp == 3 ? info->vs[0] : 1;
    if (hs == 1 && vs == 1) jpeg_rgb_kernel<1, 1><<<grid, 256, 0, st>>>(planes, g, rgb);
    else if (hs == 2 && vs == 1) jpeg_rgb_kernel<2, 1><<<grid, 256, 0, st>>>(planes, g, rgb);
    else if (hs == 2 && vs == 2) jpeg_rgb_kernel<2, 2><<<grid, 256, 0, st>>>(planes, g, rgb);
    else SM_FAIL(SM_EINVAL, "sm_jpeg_reconstruct: sampling %dx%d", hs, vs);
    SM_LAUNCH_CHECK();
    return SM_OK;
}
