// jpeg_decode.h -- JPEG decoder for the host program.
//
// The reference reads its target with prlib / stb_image (main.cpp:253-259); its two inputs are JPEGs
// (squirrel_cls_mini.jpg: progressive 4:4:4, squirrel_cls.jpg: baseline 4:2:0).  This is a from-scratch decoder for
// 8-bit Huffman JPEGs, baseline (SOF0/SOF1) and progressive (SOF2), 1 or 3 components, any sampling factors up to
// 2x2, restart intervals.  It restates the published algorithms of the IJG / libjpeg-turbo decoder that PIL uses --
// the "islow" 13-bit integer inverse DCT (Loeffler-Ligtenberg-Moschytz), "fancy" triangle-filter chroma upsampling
// and the 16-bit fixed-point YCbCr->RGB conversion -- so that its output can be compared pixel for pixel with
// the decoded fixtures (tests/test_host_io_cpu.py).  Host-side C++ only; nothing here is on the training path.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

namespace s2dio {

struct JpegDecoder {
    struct Huff {
        bool present = false;
        uint8_t bits[17] = {0};
        uint8_t vals[256] = {0};
        int mincode[17], maxcode[18], valptr[17];
        void build()
        {
            int code = 0, k = 0;
            for (int l = 1; l <= 16; l++) {
                valptr[l] = k;
                mincode[l] = code;
                code += bits[l];
                k += bits[l];
                maxcode[l] = bits[l] ? code - 1 : -1;
                code <<= 1;
            }
            maxcode[17] = 0x7fffffff;
        }
    };
    struct Comp {
        int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
        int bw = 0, bh = 0;          // blocks per row / column, padded to whole MCUs
        int pred = 0;
        std::vector<int16_t> coef;   // bw * bh * 64, zig-zag already undone (natural order)
        std::vector<uint8_t> plane;  // (bw*8) x (bh*8) samples after the IDCT
    };

    const uint8_t* d = nullptr;
    size_t n = 0, p = 0;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Comp> comps;
    int width = 0, height = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart_interval = 0;
    bool progressive = false;
    // entropy decoder state
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    int eobrun = 0;
    bool hit_marker = false;

    static const uint8_t* zigzag()
    {
        static const uint8_t z[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                      30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        return z;
    }

    // ---- bit reader (stuffed bytes, markers end the segment and feed zeros) ----
    void reset_bits() { bitbuf = 0; bitcnt = 0; hit_marker = false; }
    void fill()
    {
        while (bitcnt <= 24) {
            uint32_t b = 0;
            if (!hit_marker && p < n) {
                b = d[p];
                if (b == 0xFF) {
                    const uint8_t m = p + 1 < n ? d[p + 1] : 0xD9;
                    if (m == 0x00) p += 2;
                    else { hit_marker = true; b = 0; }
                } else p++;
            }
            bitbuf |= b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int getbits(int k)
    {
        if (k == 0) return 0;
        if (bitcnt < k) fill();
        const int v = (int)(bitbuf >> (32 - k));
        bitbuf <<= k;
        bitcnt -= k;
        return v;
    }
    int getbit() { return getbits(1); }
    int decode(const Huff& h)
    {
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | getbit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        return -1; // corrupt
    }
    static int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
    int receive_extend(int t) { return t ? extend(getbits(t), t) : 0; }

    // ---- marker segments ----
    int be16(size_t at) const { return (d[at] << 8) | d[at + 1]; }
    bool parse_dqt(size_t at, int len)
    {
        size_t q = at, end = at + len;
        while (q < end) {
            const int pq = d[q] >> 4, tq = d[q] & 15;
            q++;
            if (tq > 3 || q + (pq ? 128 : 64) > end) return false;
            for (int i = 0; i < 64; i++) {
                qt[tq][zigzag()[i]] = pq ? (uint16_t)be16(q) : d[q];
                q += pq ? 2 : 1;
            }
            qt_present[tq] = true;
        }
        return true;
    }
    bool parse_dht(size_t at, int len)
    {
        size_t q = at, end = at + len;
        while (q < end) {
            const int tc = d[q] >> 4, th = d[q] & 15;
            q++;
            if (tc > 1 || th > 3 || q + 16 > end) return false;
            Huff& h = tc ? ac[th] : dc[th];
            int total = 0;
            h.bits[0] = 0;
            for (int l = 1; l <= 16; l++) { h.bits[l] = d[q++]; total += h.bits[l]; }
            if (total > 256 || q + total > end) return false;
            std::memcpy(h.vals, d + q, (size_t)total);
            q += total;
            h.present = true;
            h.build();
        }
        return true;
    }
    bool parse_sof(size_t at, int len)
    {
        if (len < 6 || d[at] != 8) return false; // 8-bit samples only
        height = be16(at + 1);
        width = be16(at + 3);
        const int nc = d[at + 5];
        if (width <= 0 || height <= 0 || (nc != 1 && nc != 3) || len < 6 + 3 * nc) return false;
        comps.assign((size_t)nc, Comp());
        hmax = vmax = 1;
        for (int i = 0; i < nc; i++) {
            Comp& c = comps[(size_t)i];
            c.id = d[at + 6 + 3 * i];
            c.h = d[at + 7 + 3 * i] >> 4;
            c.v = d[at + 7 + 3 * i] & 15;
            c.tq = d[at + 8 + 3 * i];
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return false;
            if (c.h > hmax) hmax = c.h;
            if (c.v > vmax) vmax = c.v;
        }
        mcux = (width + 8 * hmax - 1) / (8 * hmax);
        mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (Comp& c : comps) {
            c.bw = mcux * c.h;
            c.bh = mcuy * c.v;
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        return true;
    }

    // ---- one block, baseline / progressive ----
    bool block_baseline(Comp& c, int16_t* b)
    {
        const int t = decode(dc[c.td]);
        if (t < 0 || t > 11) return false;
        c.pred += receive_extend(t);
        b[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            const int rs = decode(ac[c.ta]);
            if (rs < 0) return false;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) { k += 16; continue; }
                break;
            }
            k += r;
            if (k > 63) return false;
            b[zigzag()[k]] = (int16_t)receive_extend(s);
            k++;
        }
        return true;
    }
    bool block_dc_first(Comp& c, int16_t* b, int al)
    {
        const int t = decode(dc[c.td]);
        if (t < 0 || t > 11) return false;
        c.pred += receive_extend(t);
        b[0] = (int16_t)(c.pred * (1 << al));
        return true;
    }
    void block_dc_refine(int16_t* b, int al)
    {
        if (getbit()) b[0] = (int16_t)(b[0] | (1 << al));
    }
    bool block_ac_first(Comp& c, int16_t* b, int ss, int se, int al)
    {
        if (eobrun > 0) { eobrun--; return true; }
        for (int k = ss; k <= se;) {
            const int rs = decode(ac[c.ta]);
            if (rs < 0) return false;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += getbits(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return false;
            b[zigzag()[k]] = (int16_t)(receive_extend(s) * (1 << al));
            k++;
        }
        return true;
    }
    bool block_ac_refine(Comp& c, int16_t* b, int ss, int se, int al)
    {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun <= 0) {
            for (; k <= se;) {
                const int rs = decode(ac[c.ta]);
                if (rs < 0) return false;
                int r = rs >> 4;
                const int s = rs & 15;
                int val = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = 1 << r;
                        if (r) eobrun += getbits(r);
                        break; // the rest of this block: refine only
                    }
                    // r == 15: skip 16 zero-history coefficients, refining the non-zero ones passed
                } else {
                    if (s != 1) return false;
                    val = getbit() ? p1 : m1;
                }
                while (k <= se) {
                    int16_t& co = b[zigzag()[k]];
                    if (co != 0) {
                        if (getbit() && (co & p1) == 0) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
                    } else {
                        if (r == 0) {
                            if (val) co = (int16_t)val;
                            k++;
                            break;
                        }
                        r--;
                    }
                    k++;
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) {
                int16_t& co = b[zigzag()[k]];
                if (co != 0 && getbit() && (co & p1) == 0) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
            }
            eobrun--;
        }
        return true;
    }

    // ---- one scan ----
    bool restart()
    {
        // the bit reader stopped at a marker; it must be RSTn
        reset_bits();
        while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) {
            if (d[p] == 0xFF && d[p + 1] != 0 && d[p + 1] != 0xFF) return false;
            p++;
        }
        if (p + 1 >= n) return false;
        p += 2;
        for (Comp& c : comps) c.pred = 0;
        eobrun = 0;
        return true;
    }
    bool parse_sos(size_t at, int len)
    {
        const int ns = d[at];
        if (ns < 1 || ns > (int)comps.size() || len < 4 + 2 * ns) return false;
        std::vector<Comp*> sc;
        for (int i = 0; i < ns; i++) {
            Comp* c = nullptr;
            for (Comp& x : comps)
                if (x.id == d[at + 1 + 2 * i]) c = &x;
            if (!c) return false;
            c->td = d[at + 2 + 2 * i] >> 4;
            c->ta = d[at + 2 + 2 * i] & 15;
            if (c->td > 3 || c->ta > 3) return false;
            sc.push_back(c);
        }
        const int ss = d[at + 1 + 2 * ns], se = d[at + 2 + 2 * ns], ah = d[at + 3 + 2 * ns] >> 4, al = d[at + 3 + 2 * ns] & 15;
        if (progressive ? (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) : (ss != 0 || se != 63)) return false;
        p = at + (size_t)len;
        reset_bits();
        for (Comp& c : comps) c.pred = 0;
        eobrun = 0;
        auto do_block = [&](Comp& c, int bx, int by) -> bool {
            int16_t* b = &c.coef[((size_t)by * c.bw + bx) * 64];
            if (!progressive) return dc[c.td].present && ac[c.ta].present && block_baseline(c, b);
            if (ss == 0) {
                if (ah == 0) return dc[c.td].present && block_dc_first(c, b, al);
                block_dc_refine(b, al);
                return true;
            }
            if (!ac[c.ta].present) return false;
            return ah == 0 ? block_ac_first(c, b, ss, se, al) : block_ac_refine(c, b, ss, se, al);
        };
        int todo = restart_interval;
        if (ns == 1) { // non-interleaved: the component's own blocks, only those that contain image samples
            Comp& c = *sc[0];
            const int nbx = ((width * c.h + hmax - 1) / hmax + 7) / 8;
            const int nby = ((height * c.v + vmax - 1) / vmax + 7) / 8;
            for (int by = 0; by < nby; by++)
                for (int bx = 0; bx < nbx; bx++) {
                    if (restart_interval && todo == 0) { if (!restart()) return false; todo = restart_interval; }
                    if (!do_block(c, bx, by)) return false;
                    todo--;
                }
        } else {
            for (int my = 0; my < mcuy; my++)
                for (int mx = 0; mx < mcux; mx++) {
                    if (restart_interval && todo == 0) { if (!restart()) return false; todo = restart_interval; }
                    for (Comp* c : sc)
                        for (int v = 0; v < c->v; v++)
                            for (int h = 0; h < c->h; h++)
                                if (!do_block(*c, mx * c->h + h, my * c->v + v)) return false;
                    todo--;
                }
        }
        // leave p at the marker that ended the entropy-coded segment
        while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] != 0x00 && !(d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7))) p++;
        return true;
    }

    // ---- inverse DCT: the IJG "islow" algorithm (jidctint.c), 13-bit constants, two passes ----
    // (64-bit intermediates: on a valid stream no product leaves 32 bits and the results are jidctint.c's to the bit; on a corrupt
    // one -- coefficients up to 2^15 times quantiser steps up to 2^16 -- nothing overflows a signed integer either)
    static void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride)
    {
        const int CB = 13, P1 = 2;
        const int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                      F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
        int64_t ws[64];
        for (int c = 0; c < 8; c++) { // pass 1: columns
            const int16_t* ip = in + c;
            const uint16_t* qp = q + c;
            int64_t* wp = ws + c;
            if (!ip[8] && !ip[16] && !ip[24] && !ip[32] && !ip[40] && !ip[48] && !ip[56]) {
                const int64_t dcv = ((int64_t)ip[0] * qp[0]) * (1 << P1);
                for (int r = 0; r < 8; r++) wp[8 * r] = dcv;
                continue;
            }
            int64_t z2 = (int64_t)ip[16] * qp[16], z3 = (int64_t)ip[48] * qp[48];
            int64_t z1 = (z2 + z3) * F0_541;
            int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
            z2 = (int64_t)ip[0] * qp[0];
            z3 = (int64_t)ip[32] * qp[32];
            int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = (int64_t)ip[56] * qp[56];
            tmp1 = (int64_t)ip[40] * qp[40];
            tmp2 = (int64_t)ip[24] * qp[24];
            tmp3 = (int64_t)ip[8] * qp[8];
            z1 = tmp0 + tmp3;
            z2 = tmp1 + tmp2;
            z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            const int sh = CB - P1, rnd = 1 << (sh - 1);
            wp[0] = (tmp10 + tmp3 + rnd) >> sh;  wp[56] = (tmp10 - tmp3 + rnd) >> sh;
            wp[8] = (tmp11 + tmp2 + rnd) >> sh;  wp[48] = (tmp11 - tmp2 + rnd) >> sh;
            wp[16] = (tmp12 + tmp1 + rnd) >> sh; wp[40] = (tmp12 - tmp1 + rnd) >> sh;
            wp[24] = (tmp13 + tmp0 + rnd) >> sh; wp[32] = (tmp13 - tmp0 + rnd) >> sh;
        }
        for (int r = 0; r < 8; r++) { // pass 2: rows
            const int64_t* wp = ws + 8 * r;
            uint8_t* op = out + (size_t)r * stride;
            const int sh = CB + P1 + 3, rnd = 1 << (sh - 1);
            auto clamp = [](int64_t v) { v += 128; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
            if (!wp[1] && !wp[2] && !wp[3] && !wp[4] && !wp[5] && !wp[6] && !wp[7]) {
                const uint8_t dcv = clamp((wp[0] + (1 << (P1 + 2))) >> (P1 + 3));
                for (int c = 0; c < 8; c++) op[c] = dcv;
                continue;
            }
            int64_t z2 = wp[2], z3 = wp[6];
            int64_t z1 = (z2 + z3) * F0_541;
            int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
            int64_t tmp0 = (wp[0] + wp[4]) * (1 << CB), tmp1 = (wp[0] - wp[4]) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            op[0] = clamp((tmp10 + tmp3 + rnd) >> sh); op[7] = clamp((tmp10 - tmp3 + rnd) >> sh);
            op[1] = clamp((tmp11 + tmp2 + rnd) >> sh); op[6] = clamp((tmp11 - tmp2 + rnd) >> sh);
            op[2] = clamp((tmp12 + tmp1 + rnd) >> sh); op[5] = clamp((tmp12 - tmp1 + rnd) >> sh);
            op[3] = clamp((tmp13 + tmp0 + rnd) >> sh); op[4] = clamp((tmp13 - tmp0 + rnd) >> sh);
        }
    }

    // ---- "fancy" (triangle filter) chroma upsampling, jdsample.c ----
    // in: cw x ch samples (row stride cs); out: full-resolution plane ow x oh
    static void upsample(const Comp& c, int hmax_, int vmax_, int ow, int oh, std::vector<uint8_t>* out)
    {
        const int cs = c.bw * 8;
        const int cw = (ow * c.h + hmax_ - 1) / hmax_, ch = (oh * c.v + vmax_ - 1) / vmax_; // downsampled size
        out->assign((size_t)ow * oh, 0);
        const int fx = hmax_ / c.h, fy = vmax_ / c.v;
        auto in = [&](int x, int y) -> int {
            x = x < 0 ? 0 : (x >= cw ? cw - 1 : x);
            y = y < 0 ? 0 : (y >= ch ? ch - 1 : y);
            return c.plane[(size_t)y * cs + x];
        };
        for (int y = 0; y < oh; y++)
            for (int x = 0; x < ow; x++) {
                int v;
                if (fx == 1 && fy == 1) v = in(x, y);
                else if (fx == 2 && fy == 1) { // h2v1: 3/4 nearer + 1/4 further, alternating rounding
                    const int i = x >> 1;
                    // (the first and last output columns of jdsample.c's special cases fall out of the edge clamp:
                    //  (3a + a + 1 or 2) >> 2 == a)
                    v = (x & 1) ? (3 * in(i, y) + in(i + 1, y) + 2) >> 2 : (3 * in(i, y) + in(i - 1, y) + 1) >> 2;
                } else if (fx == 2 && fy == 2) { // h2v2: vertical 3:1 then horizontal 3:1 on the column sums
                    const int i = x >> 1, j = y >> 1, jn = (y & 1) ? j + 1 : j - 1;
                    auto colsum = [&](int xi) { return 3 * in(xi, j) + in(xi, jn); };
                    v = (x & 1) ? (3 * colsum(i) + colsum(i + 1) + 7) >> 4 : (3 * colsum(i) + colsum(i - 1) + 8) >> 4;
                } else if (fx == 1 && fy == 2) { // h1v2: vertical triangle filter
                    const int j = y >> 1, jn = (y & 1) ? j + 1 : j - 1;
                    v = (3 * in(x, j) + in(x, jn) + ((y & 1) ? 2 : 1)) >> 2;
                } else v = in(x / fx, y / fy);
                (*out)[(size_t)y * ow + x] = (uint8_t)v;
            }
    }

    bool decode_file(const uint8_t* data, size_t size, int* w, int* h, std::vector<uint8_t>* rgb)
    {
        d = data;
        n = size;
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
        p = 2;
        bool have_sof = false;
        while (p + 4 <= n) {
            if (d[p] != 0xFF) { p++; continue; }
            const uint8_t m = d[p + 1];
            if (m == 0xFF) { p++; continue; }
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
            const int len = be16(p + 2);
            if (len < 2 || p + 2 + (size_t)len > n) return false;
            const size_t at = p + 4;
            const int body = len - 2;
            p += 2 + (size_t)len;
            if (m == 0xDB) { if (!parse_dqt(at, body)) return false; }
            else if (m == 0xC4) { if (!parse_dht(at, body)) return false; }
            else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                progressive = (m == 0xC2);
                if (have_sof || !parse_sof(at, body)) return false;
                have_sof = true;
            } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) return false; // lossless / arithmetic
            else if (m == 0xDD) { if (body < 2) return false; restart_interval = be16(at); }
            else if (m == 0xDA) {
                if (!have_sof || !parse_sos(at, body)) return false; // leaves p at the next marker; scans repeat until EOI
            }
        }
        if (!have_sof) return false;
        for (Comp& c : comps) {
            if (!qt_present[c.tq]) return false;
            c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
            for (int by = 0; by < c.bh; by++)
                for (int bx = 0; bx < c.bw; bx++)
                    idct_islow(&c.coef[((size_t)by * c.bw + bx) * 64], qt[c.tq], &c.plane[((size_t)by * 8) * c.bw * 8 + (size_t)bx * 8],
                               c.bw * 8);
        }
        *w = width;
        *h = height;
        rgb->resize((size_t)width * height * 3);
        if (comps.size() == 1) {
            const Comp& c = comps[0];
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) {
                    const uint8_t v = c.plane[(size_t)y * c.bw * 8 + x];
                    uint8_t* o = &(*rgb)[((size_t)y * width + x) * 3];
                    o[0] = o[1] = o[2] = v;
                }
            return true;
        }
        std::vector<uint8_t> Y, Cb, Cr;
        upsample(comps[0], hmax, vmax, width, height, &Y);
        upsample(comps[1], hmax, vmax, width, height, &Cb);
        upsample(comps[2], hmax, vmax, width, height, &Cr);
        // YCbCr -> RGB, jdcolor.c: 16-bit fixed point
        for (size_t i = 0; i < (size_t)width * height; i++) {
            const int y = Y[i], cb = Cb[i] - 128, cr = Cr[i] - 128;
            auto rs = [](int64_t v) { return (int)(v >> 16); }; // arithmetic shift: floor
            auto cl = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
            const int r = y + rs((int64_t)91881 * cr + 32768);
            const int g = y + rs((int64_t)-22554 * cb + (int64_t)-46802 * cr + 32768);
            const int b = y + rs((int64_t)116130 * cb + 32768);
            uint8_t* o = &(*rgb)[i * 3];
            o[0] = cl(r); o[1] = cl(g); o[2] = cl(b);
        }
        return true;
    }
};

inline bool load_jpeg(const std::vector<uint8_t>& data, int* w, int* h, std::vector<uint8_t>* rgb)
{
    JpegDecoder dec;
    return dec.decode_file(data.data(), data.size(), w, h, rgb);
}

} // namespace s2dio
