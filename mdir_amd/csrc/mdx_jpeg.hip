// JPEG decoding split between host and device.
//
// Replaces `Image.open(f).convert('RGB')` of pil_loader (mdir/external/cirtorch/datasets/datahelpers.py:24-31) for
// baseline JPEG files: Pillow hands the file to libjpeg(-turbo) with its defaults -- Huffman entropy decoding, the
// accurate integer inverse DCT (jidctint.c "islow"), "fancy" (triangle) chroma upsampling (jdsample.c) and the
// fixed-point YCbCr -> RGB conversion (jdcolor.c).  Entropy decoding is a serial walk over a bit stream and stays on
// the host (a loader thread calls mdx_jpeg_coefficients; what crosses PCIe is the quantised coefficients, 2 bytes each,
// no more than the decoded pixels); everything after it is per-block / per-pixel integer arithmetic and runs here,
// bit for bit as libjpeg does it: mdx_jpeg_pixels = dequantisation + IDCT (one thread per 8x8 block), then upsampling +
// colour conversion (one thread per output pixel).
//
// Covered: 8-bit Huffman-coded files -- baseline / extended sequential (one interleaved scan or one scan per component)
// and progressive (spectral selection + successive approximation, jdphuff.c) -- grey or YCbCr, luma at full resolution and
// chroma at 1x1, 2x1 or 2x2 (4:4:4, 4:2:2, 4:2:0).  Everything else (arithmetic coding, 12-bit, lossless, CMYK / YCCK,
// RGB-coded, other sampling factors, images narrower than 16 pixels) is reported as unsupported by mdx_jpeg_probe and stays
// with the host decoder.
#include <stdlib.h>

#include "mdx_common.h"

namespace mdx {

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool defined = false;
    uint8_t bits[17];
    uint8_t vals[256];
    // canonical decoding: codes of length l are mincode[l] .. maxcode[l]; valptr[l] = index of the first of them
    int32_t mincode[18], maxcode[18], valptr[18];
    // 9-bit lookahead: (length << 8) | symbol, 0 = longer than 9 bits
    uint16_t fast[512];
    // AC tables, sequential scans: when code AND magnitude bits fit the 9-bit window, the whole coefficient at once:
    // (value << 8) | (run << 4) | total bits, 0 = take the long way
    int16_t fast_ac[512];
};

static bool build_table(HuffTable &t)
{
    int code = 0, k = 0;
    for (int i = 0; i < 512; ++i) t.fast[i] = 0;
    for (int l = 1; l <= 16; ++l) {
        t.valptr[l] = k;
        t.mincode[l] = code;
        for (int i = 0; i < t.bits[l]; ++i, ++k, ++code) {
            // an l-bit code must fit l bits (jdhuff.c jpeg_make_d_derived_tbl: "if (code >= (1 << si)) ERREXIT") -- checked per
            // symbol BEFORE the lookahead table is written: `first + j` below indexes fast[512] only when code < 2^l
            if (k >= 256 || code >= (1 << l)) return false;
            if (l <= 9) {
                const int first = code << (9 - l), n = 1 << (9 - l);
                for (int j = 0; j < n; ++j) t.fast[first + j] = (uint16_t)((l << 8) | t.vals[k]);
            }
        }
        t.maxcode[l] = t.bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    t.maxcode[17] = 0x7FFFFFFF;
    for (int i = 0; i < 512; ++i) {
        t.fast_ac[i] = 0;
        const int e = t.fast[i];
        if (!e) continue;
        const int len = e >> 8, rs = e & 255, run = rs >> 4, mag = rs & 15;
        if (mag && len + mag <= 9) {
            int v = (i >> (9 - len - mag)) & ((1 << mag) - 1);
            if (v < (1 << (mag - 1))) v -= (1 << mag) - 1;
            if (v >= -128 && v <= 127) t.fast_ac[i] = (int16_t)(v * 256 + run * 16 + len + mag);
        }
    }
    return true;
}

static inline int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

struct JpegHeader {
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart = 0;
    int cid[3], h[3], v[3], tq[3], td[3], ta[3];
    uint16_t quant[4][64];          // natural order
    bool qdef[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    int64_t scan_data = -1;         // offset of the entropy-coded segment of the first scan
    int64_t first_sos = -1;         // offset of the first SOS marker's length field
    bool progressive = false;
    bool supported = false;
    const char *why = "no start of frame";
};

// DQT / DHT / DRI segment (they may also come between the scans of a file)
static bool table_segment(JpegHeader &hd, int m, const uint8_t *s, int n)
{
    if (m == 0xDB) {                        // DQT
        int o = 0;
        while (o < n) {
            const int pq = s[o] >> 4, tq = s[o] & 15;
            ++o;
            if (tq > 3 || o + (pq ? 128 : 64) > n) { hd.why = "bad quantisation table"; return false; }
            for (int i = 0; i < 64; ++i) hd.quant[tq][ZIGZAG[i]] = pq ? (uint16_t)be16(s + o + 2 * i) : s[o + i];
            o += pq ? 128 : 64;
            hd.qdef[tq] = true;
        }
    } else if (m == 0xC4) {                 // DHT
        int o = 0;
        while (o + 17 <= n) {
            const int tc = s[o] >> 4, th = s[o] & 15;
            if (tc > 1 || th > 3) { hd.why = "bad Huffman table id"; return false; }
            HuffTable &t = tc ? hd.ac[th] : hd.dc[th];
            int total = 0;
            t.bits[0] = 0;
            for (int i = 1; i <= 16; ++i) { t.bits[i] = s[o + i]; total += t.bits[i]; }
            o += 17;
            if (total > 256 || o + total > n) { hd.why = "bad Huffman table"; return false; }
            for (int i = 0; i < total; ++i) {
                t.vals[i] = s[o + i];
                if (tc == 0 && t.vals[i] > 15) { hd.why = "bad Huffman table"; return false; }     // libjpeg: a fatal error
            }
            o += total;
            if (!build_table(t)) { hd.why = "bad Huffman code lengths"; return false; }
            t.defined = true;
        }
    } else if (m == 0xDD) {                 // DRI
        if (n >= 2) hd.restart = be16(s);
    }
    return true;
}

// headers up to the first scan
static bool parse_header(const uint8_t *d, int64_t size, JpegHeader &hd)
{
    if (size < 4 || d[0] != 0xFF || d[1] != 0xD8) { hd.why = "not a JPEG file"; return false; }
    int64_t p = 2;
    bool have_frame = false, adobe = false;
    int adobe_transform = -1;
    while (p + 4 <= size) {
        if (d[p] != 0xFF) { hd.why = "marker expected"; return false; }
        while (p < size && d[p] == 0xFF) ++p;       // fill bytes
        if (p >= size) break;
        const int m = d[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (p + 2 > size) break;
        const int len = be16(d + p);
        if (len < 2 || p + len > size) { hd.why = "truncated segment"; return false; }
        const uint8_t *s = d + p + 2;
        const int n = len - 2;
        if (m == 0xDB || m == 0xC4 || m == 0xDD) {
            if (!table_segment(hd, m, s, n)) return false;
        } else if (m == 0xEE) {             // APP14 Adobe
            if (n >= 12 && !memcmp(s, "Adobe", 5)) { adobe = true; adobe_transform = s[11]; }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {    // SOF0 / SOF1: sequential Huffman; SOF2: progressive Huffman
            hd.progressive = m == 0xC2;
            if (n < 6) { hd.why = "bad frame header"; return false; }
            if (s[0] != 8) { hd.why = "not 8 bits per sample"; return false; }
            hd.height = be16(s + 1);
            hd.width = be16(s + 3);
            hd.ncomp = s[5];
            if (hd.ncomp != 1 && hd.ncomp != 3) { hd.why = "not a grey or three-component image"; return false; }
            if (n < 6 + 3 * hd.ncomp || hd.width <= 0 || hd.height <= 0) { hd.why = "bad frame header"; return false; }
            for (int c = 0; c < hd.ncomp; ++c) {
                hd.cid[c] = s[6 + 3 * c];
                hd.h[c] = s[7 + 3 * c] >> 4;
                hd.v[c] = s[7 + 3 * c] & 15;
                hd.tq[c] = s[8 + 3 * c];
                if (hd.tq[c] > 3) { hd.why = "bad quantisation table id"; return false; }
            }
            have_frame = true;
        } else if ((m >= 0xC3 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            hd.why = "lossless, hierarchical or arithmetic-coded";
            return false;
        } else if (!(m == 0xFE || (m >= 0xE0 && m <= 0xEF) || m == 0xDA)) {
            hd.why = "unexpected marker";
            return false;
        } else if (m == 0xDA) {             // SOS: the scans are walked by decode_scans
            if (!have_frame) { hd.why = "scan before frame"; return false; }
            hd.first_sos = p;
            hd.scan_data = p + len;
            break;
        }
        p += len;
    }
    if (hd.scan_data < 0) { if (have_frame) hd.why = "no scan"; return false; }
    // colour space as libjpeg guesses it (jdapimin.c default_decompress_parms)
    if (hd.ncomp == 3) {
        if (adobe && adobe_transform != 1) { hd.why = "Adobe marker: not YCbCr"; return false; }
        if (!adobe && hd.cid[0] == 'R' && hd.cid[1] == 'G' && hd.cid[2] == 'B') { hd.why = "RGB-coded"; return false; }
        if (hd.h[1] != 1 || hd.v[1] != 1 || hd.h[2] != 1 || hd.v[2] != 1) { hd.why = "chroma sampling factors"; return false; }
        if (!((hd.h[0] == 1 && hd.v[0] == 1) || (hd.h[0] == 2 && hd.v[0] == 1) || (hd.h[0] == 2 && hd.v[0] == 2))) {
            hd.why = "luma sampling factors";
            return false;
        }
        hd.hmax = hd.h[0];
        hd.vmax = hd.v[0];
        if (hd.width < 16 || hd.height < 2) { hd.why = "too small"; return false; }
    } else {
        hd.h[0] = hd.v[0] = 1;              // a single component is never interleaved: MCU = one block
        hd.hmax = hd.vmax = 1;
    }
    for (int c = 0; c < hd.ncomp; ++c)
        if (!hd.qdef[hd.tq[c]]) { hd.why = "missing quantisation table"; return false; }
    // Pillow refuses images beyond twice its MAX_IMAGE_PIXELS (a decompression bomb): leave those to it
    if ((int64_t)hd.width * hd.height > 2 * (int64_t)89478485) { hd.why = "too many pixels"; return false; }
    // The frame header is untrusted and the caller sizes its coefficient buffer (128 B per block) from it: every block costs
    // the first scan that covers it at least one Huffman code, i.e. one bit, so a file shorter than blocks / 8 bytes cannot
    // hold the picture it announces (a 200-byte file claiming 65 535 x 2 700 pixels must not reserve half a gigabyte)
    if ((int64_t)hd.width * hd.height / 64 / 8 > size) { hd.why = "file too short for its dimensions"; return false; }
    hd.mcux = (hd.width + 8 * hd.hmax - 1) / (8 * hd.hmax);
    hd.mcuy = (hd.height + 8 * hd.vmax - 1) / (8 * hd.vmax);
    hd.supported = true;
    hd.why = "";
    return true;
}

// bit reader over the entropy-coded segment: 0xFF00 -> 0xFF, any other marker ends the data (zeros follow)
struct BitReader {
    const uint8_t *d;
    int64_t p, size;
    uint64_t acc = 0;
    int nbits = 0;
    int pad = 0;            // zero bytes appended after the data ran out (a marker or the end of the file)
    bool marker = false;

    // did the decoder consume bits that were not in the file?  (truncated or corrupt data: such files go to the host decoder)
    inline bool overran() const { return nbits < 8 * pad; }

    inline void fill()
    {
        while (nbits <= 56) {
            if (!marker && p + 8 <= size) {
                // eight bytes at once when none of them is 0xFF (no stuffing, no marker): the bits below the ones counted
                // in are the true next bits of the stream, and a later fill ORs the same bits over them
                uint64_t w;
                memcpy(&w, d + p, 8);
                w = __builtin_bswap64(w);
                const uint64_t x = ~w;
                if (((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) == 0) {
                    const int take = (64 - nbits) >> 3;
                    acc |= w >> nbits;
                    p += take;
                    nbits += 8 * take;
                    return;
                }
            }
            int byte = 0;
            if (!marker && p < size) {
                byte = d[p];
                if (byte == 0xFF) {
                    if (p + 1 < size && d[p + 1] == 0) p += 2;
                    else { marker = true; byte = 0; ++pad; }
                } else {
                    ++p;
                }
            } else {
                ++pad;
            }
            acc |= (uint64_t)byte << (56 - nbits);
            nbits += 8;
        }
    }
    inline int peek(int n) { return (int)(acc >> (64 - n)); }
    inline void skip(int n) { acc <<= n; nbits -= n; }
    inline int get(int n)
    {
        if (n == 0) return 0;
        const int v = peek(n);
        skip(n);
        return v;
    }
    // RSTn: drop the rest of the byte, step over the marker
    bool restart()
    {
        if (overran()) return false;
        acc = 0;
        nbits = 0;
        pad = 0;
        if (marker) {
            if (p + 1 < size && d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7) { p += 2; marker = false; return true; }
            return false;
        }
        // padding bits were consumed; look for the marker
        while (p + 1 < size && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) ++p;
        if (p + 1 >= size) return false;
        p += 2;
        return true;
    }
};

static inline int huff_decode(BitReader &br, const HuffTable &t)
{
    br.fill();
    const int look = br.peek(9);
    const int e = t.fast[look];
    if (e) {
        br.skip(e >> 8);
        return e & 255;
    }
    int code = br.peek(10), l = 10;
    while (l <= 16 && code > t.maxcode[l]) { ++l; code = br.peek(l); }
    if (l > 16) return -1;
    br.skip(l);
    return t.vals[t.valptr[l] + code - t.mincode[l]];
}

static inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// The DC predictor is a running sum over every block of a scan: a crafted file can push it past 2^31, so it is added and
// scaled in unsigned arithmetic (wraps, never undefined -- libjpeg-turbo does the same); what a sound file produces is
// unchanged, what a damaged one produces is caught by the coefficient range check of mdx_jpeg_coefficients.
static inline int wrap_add(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
static inline int wrap_shl(int a, int n) { return (int)((uint32_t)a << n); }

// one block of a sequential scan (all 64 coefficients)
static inline bool block_sequential(BitReader &br, const HuffTable &dct, const HuffTable &act, int &pred, int16_t *blk)
{
    int s = huff_decode(br, dct);
    if (s < 0 || s > 15) return false;
    if (s) {
        br.fill();
        pred = wrap_add(pred, extend(br.get(s), s));
    }
    blk[0] = (int16_t)pred;
    for (int k = 1; k < 64;) {
        br.fill();
        const int f = act.fast_ac[br.peek(9)];
        if (f) {                            // run, size and value in one look
            k += (f >> 4) & 15;
            if (k > 63) return false;
            br.skip(f & 15);
            blk[ZIGZAG[k++]] = (int16_t)(f >> 8);
            continue;
        }
        const int rs = huff_decode(br, act);
        if (rs < 0) return false;
        const int r = rs >> 4;
        s = rs & 15;
        if (s == 0) {
            if (r != 15) break;
            k += 16;
            continue;
        }
        k += r;
        if (k > 63) return false;
        br.fill();
        blk[ZIGZAG[k]] = (int16_t)extend(br.get(s), s);
        ++k;
    }
    return true;
}

// one block of a progressive AC scan (jdphuff.c decode_mcu_AC_first / decode_mcu_AC_refine)
static inline bool block_ac_first(BitReader &br, const HuffTable &act, int ss, int se, int al, int &eobrun, int16_t *blk)
{
    if (eobrun > 0) { --eobrun; return true; }
    for (int k = ss; k <= se; ++k) {
        const int rs = huff_decode(br, act);
        if (rs < 0) return false;
        const int r = rs >> 4, s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return false;
            br.fill();
            blk[ZIGZAG[k]] = (int16_t)wrap_shl(extend(br.get(s), s), al);
        } else if (r == 15) {
            k += 15;
        } else {
            eobrun = 1 << r;
            if (r) { br.fill(); eobrun += br.get(r); }
            --eobrun;
            break;
        }
    }
    return true;
}

static inline bool block_ac_refine(BitReader &br, const HuffTable &act, int ss, int se, int al, int &eobrun, int16_t *blk)
{
    const int p1 = 1 << al, m1 = -(1 << al);
    int k = ss;
    if (eobrun == 0) {
        for (; k <= se; ++k) {
            const int rs = huff_decode(br, act);
            if (rs < 0) return false;
            int r = rs >> 4, s = rs & 15;
            if (s) {
                br.fill();
                s = br.get(1) ? p1 : m1;
            } else if (r != 15) {
                eobrun = 1 << r;
                if (r) { br.fill(); eobrun += br.get(r); }
                break;
            }
            do {            // past the coefficients that are already non-zero (each takes a correction bit) and r zero ones
                int16_t *c = blk + ZIGZAG[k];
                if (*c != 0) {
                    br.fill();
                    if (br.get(1) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                } else if (--r < 0) {
                    break;
                }
                ++k;
            } while (k <= se);
            if (s) {
                if (k > 63) return false;
                blk[ZIGZAG[k]] = (int16_t)s;
            }
        }
    }
    if (eobrun > 0) {
        for (; k <= se; ++k) {
            int16_t *c = blk + ZIGZAG[k];
            if (*c != 0) {
                br.fill();
                if (br.get(1) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
            }
        }
        --eobrun;
    }
    return true;
}

// all scans of the file: sequential (one interleaved scan, or one scan per component) or progressive
static bool decode_scans(const uint8_t *d, int64_t size, JpegHeader &hd, const mdx_jpeg_info &info, int16_t *coef)
{
    memset(coef, 0, (size_t)info.nblocks * 64 * sizeof(int16_t));
    int64_t pos = hd.first_sos;                 // at the length field of an SOS segment
    for (int scans = 0; scans < 1000; ++scans) {
        if (pos + 2 > size) return false;
        const int len = be16(d + pos);
        if (len < 6 || pos + len > size) return false;
        const uint8_t *s = d + pos + 2;
        const int ns = s[0];
        if (ns < 1 || ns > hd.ncomp || len != 6 + 2 * ns) return false;
        int comp[3], td[3], ta[3];
        for (int i = 0; i < ns; ++i) {
            int c = 0;
            while (c < hd.ncomp && hd.cid[c] != s[1 + 2 * i]) ++c;
            if (c == hd.ncomp || (i && c <= comp[i - 1])) return false;
            comp[i] = c;
            td[i] = s[2 + 2 * i] >> 4;
            ta[i] = s[2 + 2 * i] & 15;
            if (td[i] > 3 || ta[i] > 3) return false;
        }
        const int ss = s[1 + 2 * ns], se = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
        if (!hd.progressive) {
            if (ss != 0 || se != 63 || ah != 0 || al != 0) return false;
        } else {
            if (ss > se || se > 63 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) return false;
        }
        for (int i = 0; i < ns; ++i) {
            if ((ss == 0 && ah == 0 && !hd.dc[td[i]].defined) || (se > 0 && !hd.ac[ta[i]].defined)) return false;
        }
        BitReader br{d, pos + len, size};
        int pred[3] = {0, 0, 0}, eobrun = 0, until_restart = hd.restart;
        // a scan of one component walks that component's own blocks (not padded to whole MCUs); several components are interleaved
        const int c0 = comp[0];
        const int cw = (hd.width * hd.h[c0] + hd.hmax - 1) / hd.hmax, ch = (hd.height * hd.v[c0] + hd.vmax - 1) / hd.vmax;
        const int64_t units = ns == 1 ? (int64_t)((cw + 7) / 8) * ((ch + 7) / 8) : (int64_t)hd.mcux * hd.mcuy;
        const int row_units = ns == 1 ? (cw + 7) / 8 : hd.mcux;
        for (int64_t u = 0; u < units; ++u) {
            if (hd.restart && until_restart == 0) {
                if (!br.restart()) return false;
                pred[0] = pred[1] = pred[2] = 0;
                eobrun = 0;
                until_restart = hd.restart;
            }
            const int uy = (int)(u / row_units), ux = (int)(u % row_units);
            for (int i = 0; i < ns; ++i) {
                const int c = comp[i];
                const int nv = ns == 1 ? 1 : hd.v[c], nh = ns == 1 ? 1 : hd.h[c];
                for (int by = 0; by < nv; ++by)
                    for (int bx = 0; bx < nh; ++bx) {
                        int16_t *blk = coef + (info.block_offset[c] + (int64_t)(uy * nv + by) * info.blocks_w[c] + (ux * nh + bx)) * 64;
                        bool ok = true;
                        if (!hd.progressive) {
                            ok = block_sequential(br, hd.dc[td[i]], hd.ac[ta[i]], pred[c], blk);
                        } else if (ss == 0) {
                            if (ah == 0) {                  // DC, first pass
                                const int t = huff_decode(br, hd.dc[td[i]]);
                                if (t < 0 || t > 15) return false;
                                if (t) { br.fill(); pred[c] = wrap_add(pred[c], extend(br.get(t), t)); }
                                blk[0] = (int16_t)wrap_shl(pred[c], al);
                            } else {                        // DC, one more bit
                                br.fill();
                                if (br.get(1)) blk[0] = (int16_t)(blk[0] | (1 << al));
                            }
                        } else if (ah == 0) {
                            ok = block_ac_first(br, hd.ac[ta[i]], ss, se, al, eobrun, blk);
                        } else {
                            ok = block_ac_refine(br, hd.ac[ta[i]], ss, se, al, eobrun, blk);
                        }
                        if (!ok) return false;
                    }
            }
            if (hd.restart) --until_restart;
        }
        if (br.overran()) return false;
        // on to the next marker that is not a restart marker; tables may be redefined between scans
        int64_t q = br.p;
        for (;;) {
            while (q + 1 < size && !(d[q] == 0xFF && d[q + 1] != 0x00 && d[q + 1] != 0xFF && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7))) ++q;
            // the data ends without an end-of-image marker: libjpeg would insert one with a warning, but Pillow -- what the
            // reference's loader calls -- reports such a file as truncated (a progressive file cut between two scans would
            // otherwise come out as a partially refined picture): declined, the host decoder decides
            if (q + 1 >= size) return false;
            const int m = d[q + 1];
            if (m == 0xD9) return true;
            if (q + 4 > size) return false;
            const int sl = be16(d + q + 2);
            if (sl < 2 || q + 2 + sl > size) return false;
            if (m == 0xDA) { pos = q + 2; break; }
            // a quantisation table redefined BETWEEN scans: libjpeg latches a component's table at that component's first
            // scan, the tables handed to the device were copied before the scans were walked -- rare enough to decline
            if (m == 0xDB) return false;
            if (!(m == 0xC4 || m == 0xDD || m == 0xFE || (m >= 0xE0 && m <= 0xEF))) return false;   // libjpeg: unsupported marker
            if (!table_segment(hd, m, d + q + 4, sl - 2)) return false;
            q += 2 + sl;
        }
    }
    return false;
}

static void fill_info(const JpegHeader &hd, mdx_jpeg_info *info)
{
    memset(info, 0, sizeof *info);
    info->width = hd.width;
    info->height = hd.height;
    info->ncomp = hd.ncomp;
    info->supported = hd.supported ? 1 : 0;
    if (!hd.supported) return;
    int64_t off = 0;
    for (int c = 0; c < hd.ncomp; ++c) {
        info->hsamp[c] = hd.h[c];
        info->vsamp[c] = hd.v[c];
        info->blocks_w[c] = hd.mcux * hd.h[c];
        info->blocks_h[c] = hd.mcuy * hd.v[c];
        info->block_offset[c] = off;
        off += (int64_t)info->blocks_w[c] * info->blocks_h[c];
    }
    info->nblocks = off;
}

// ---------------------------------------------------------------------------------------------------------- device
// jidctint.c (jpeg_idct_islow): 13-bit constants, 2 extra bits kept between the column pass and the row pass
#define JFIX_0_298631336 2446
#define JFIX_0_390180644 3196
#define JFIX_0_541196100 4433
#define JFIX_0_765366865 6270
#define JFIX_0_899976223 7373
#define JFIX_1_175875602 9633
#define JFIX_1_501321110 12299
#define JFIX_1_847759065 15137
#define JFIX_1_961570560 16069
#define JFIX_2_053119869 16819
#define JFIX_2_562915447 20995
#define JFIX_3_072711026 25172

__device__ __forceinline__ int32_t jdescale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

// eight inputs of one column / row -> eight outputs, descaled by `shift`
__device__ __forceinline__ void idct8(const int32_t (&in)[8], int32_t (&out)[8], int shift)
{
    int32_t z2 = in[2], z3 = in[6];
    int32_t z1 = (z2 + z3) * JFIX_0_541196100;
    int32_t tmp2 = z1 + z3 * (-JFIX_1_847759065);
    int32_t tmp3 = z1 + z2 * JFIX_0_765366865;
    z2 = in[0];
    z3 = in[4];
    int32_t tmp0 = (z2 + z3) << 13;
    int32_t tmp1 = (z2 - z3) << 13;
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * JFIX_1_175875602;
    tmp0 *= JFIX_0_298631336;
    tmp1 *= JFIX_2_053119869;
    tmp2 *= JFIX_3_072711026;
    tmp3 *= JFIX_1_501321110;
    z1 *= -JFIX_0_899976223;
    z2 *= -JFIX_2_562915447;
    z3 *= -JFIX_1_961570560;
    z4 *= -JFIX_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = jdescale(tmp10 + tmp3, shift);
    out[7] = jdescale(tmp10 - tmp3, shift);
    out[1] = jdescale(tmp11 + tmp2, shift);
    out[6] = jdescale(tmp11 - tmp2, shift);
    out[2] = jdescale(tmp12 + tmp1, shift);
    out[5] = jdescale(tmp12 - tmp1, shift);
    out[3] = jdescale(tmp13 + tmp0, shift);
    out[4] = jdescale(tmp13 - tmp0, shift);
}

// + 128, limited to 0..255: saturating, as libjpeg-turbo's SIMD IDCT packs its result (the C IDCT looks (x & 1023) up in a
// table that wraps beyond +-512; for every sample a sound file can produce the two agree, on damaged files Pillow's
// x86 build saturates)
__device__ __forceinline__ uint8_t idct_limit(int32_t x)
{
    x += 128;
    return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x));
}

struct JpegGeom {
    int32_t width, height, ncomp, hmax, vmax;
    int32_t bw[3], bh[3];
    int64_t boff[3];            // first block of the component
    int64_t poff[3];            // first byte of the component's plane ([bh*8][bw*8])
    int64_t nblocks;
};

// one thread per 8x8 block: dequantise, column pass, row pass, range limit -> the component's plane
__global__ __launch_bounds__(64) void jpeg_idct_kernel(const int16_t *__restrict__ coef, const uint16_t *__restrict__ quant, JpegGeom g,
                                                       uint8_t *__restrict__ planes)
{
    const int64_t blk = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (blk >= g.nblocks) return;
    int c = 0;
    if (g.ncomp == 3) c = blk >= g.boff[2] ? 2 : (blk >= g.boff[1] ? 1 : 0);
    const int64_t lb = blk - g.boff[c];
    const int by = (int)(lb / g.bw[c]), bx = (int)(lb % g.bw[c]);
    const int16_t *src = coef + blk * 64;
    const uint16_t *q = quant + c * 64;
    int32_t ws[64];
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int32_t in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int32_t)src[8 * r + col] * (int32_t)q[8 * r + col];
        idct8(in, out, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[8 * r + col] = out[r];
    }
    uint8_t *dst = planes + g.poff[c] + ((int64_t)by * 8) * ((int64_t)g.bw[c] * 8) + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int32_t in[8], out[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) in[k] = ws[8 * r + k];
        idct8(in, out, 13 + 2 + 3);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo |= (uint32_t)idct_limit(out[k]) << (8 * k);
            hi |= (uint32_t)idct_limit(out[4 + k]) << (8 * k);
        }
        uint32_t *row = (uint32_t *)(dst + (int64_t)r * g.bw[c] * 8);
        row[0] = lo;
        row[1] = hi;
    }
}

__device__ __forceinline__ uint8_t clamp8(int32_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// chroma sample at full resolution (jdsample.c): h2v2 / h2v1 "fancy" triangle filters, or the sample itself
__device__ __forceinline__ int32_t chroma_at(const uint8_t *__restrict__ pl, int64_t stride, int x, int y, int dw, int dh, int hs, int vs)
{
    if (hs == 1) return pl[(int64_t)y * stride + x];
    const int i = x >> 1;
    if (vs == 1) {          // h2v1: (3 * this + neighbour + 1 or 2) >> 2; the first and last column copy
        const int32_t cur = pl[(int64_t)y * stride + i];
        if (x & 1) return i == dw - 1 ? cur : (3 * cur + pl[(int64_t)y * stride + i + 1] + 2) >> 2;
        return i == 0 ? cur : (3 * cur + pl[(int64_t)y * stride + i - 1] + 1) >> 2;
    }
    // h2v2: column sums 3 * nearer row + farther row, then 3 * this column + neighbouring column
    const int r = y >> 1;
    int far = (y & 1) ? r + 1 : r - 1;
    far = far < 0 ? 0 : (far > dh - 1 ? dh - 1 : far);           // the rows beyond the image repeat its edge rows
    const uint8_t *n0 = pl + (int64_t)r * stride, *n1 = pl + (int64_t)far * stride;
    const int32_t cs = 3 * n0[i] + n1[i];
    if (x & 1) {
        if (i == dw - 1) return (4 * cs + 7) >> 4;
        return (3 * cs + 3 * n0[i + 1] + n1[i + 1] + 7) >> 4;
    }
    if (i == 0) return (4 * cs + 8) >> 4;
    return (3 * cs + 3 * n0[i - 1] + n1[i - 1] + 8) >> 4;
}

// one thread per output pixel: upsample the chroma, YCbCr -> RGB with jdcolor.c's 16-bit fixed point
__global__ __launch_bounds__(256) void jpeg_rgb_kernel(const uint8_t *__restrict__ planes, JpegGeom g, uint8_t *__restrict__ rgb)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)g.width * g.height) return;
    const int y = (int)(idx / g.width), x = (int)(idx % g.width);
    const int32_t lum = planes[g.poff[0] + (int64_t)y * g.bw[0] * 8 + x];
    uint8_t *o = rgb + idx * 3;
    if (g.ncomp == 1) {
        o[0] = o[1] = o[2] = (uint8_t)lum;
        return;
    }
    const int dw = (g.width + g.hmax - 1) / g.hmax, dh = (g.height + g.vmax - 1) / g.vmax;      // chroma size that counts
    const int32_t cb = chroma_at(planes + g.poff[1], (int64_t)g.bw[1] * 8, x, y, dw, dh, g.hmax, g.vmax) - 128;
    const int32_t cr = chroma_at(planes + g.poff[2], (int64_t)g.bw[2] * 8, x, y, dw, dh, g.hmax, g.vmax) - 128;
    // FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554; ONE_HALF = 32768
    o[0] = clamp8(lum + ((91881 * cr + 32768) >> 16));
    o[1] = clamp8(lum + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    o[2] = clamp8(lum + ((116130 * cb + 32768) >> 16));
}

// mdx_jpeg_pixels takes the geometry from its caller: only what mdx_jpeg_probe itself would have written is accepted (the
// kernels index the coefficient and plane buffers by these fields)
static bool info_consistent(const mdx_jpeg_info &in)
{
    if (!in.supported || (in.ncomp != 1 && in.ncomp != 3) || in.width <= 0 || in.height <= 0 || in.width > 65535 || in.height > 65535) return false;
    const int h0 = in.hsamp[0], v0 = in.vsamp[0];
    if (in.ncomp == 1) {
        if (h0 != 1 || v0 != 1) return false;
    } else {
        if (!((h0 == 1 && v0 == 1) || (h0 == 2 && v0 == 1) || (h0 == 2 && v0 == 2))) return false;
        if (in.hsamp[1] != 1 || in.vsamp[1] != 1 || in.hsamp[2] != 1 || in.vsamp[2] != 1) return false;
    }
    const int mcux = (in.width + 8 * h0 - 1) / (8 * h0), mcuy = (in.height + 8 * v0 - 1) / (8 * v0);
    int64_t off = 0;
    for (int c = 0; c < in.ncomp; ++c) {
        if (in.blocks_w[c] != mcux * in.hsamp[c] || in.blocks_h[c] != mcuy * in.vsamp[c] || in.block_offset[c] != off) return false;
        off += (int64_t)in.blocks_w[c] * in.blocks_h[c];
    }
    return in.nblocks == off;
}

static void geom_of(const mdx_jpeg_info &info, JpegGeom *g)
{
    g->width = info.width;
    g->height = info.height;
    g->ncomp = info.ncomp;
    g->hmax = info.hsamp[0];
    g->vmax = info.vsamp[0];
    g->nblocks = info.nblocks;
    int64_t p = 0;
    for (int c = 0; c < 3; ++c) {
        g->bw[c] = c < info.ncomp ? info.blocks_w[c] : 0;
        g->bh[c] = c < info.ncomp ? info.blocks_h[c] : 0;
        g->boff[c] = c < info.ncomp ? info.block_offset[c] : info.nblocks;
        g->poff[c] = p;
        p += (int64_t)g->bw[c] * g->bh[c] * 64;
    }
}

}  // namespace mdx

using namespace mdx;

extern "C" {

int mdx_jpeg_probe(const uint8_t *file, int64_t size, mdx_jpeg_info *info)
{
    MDX_CHECK_ARG(file && info && size > 0, "mdx_jpeg_probe: NULL pointer or empty file");
    JpegHeader hd;
    parse_header(file, size, hd);
    fill_info(hd, info);
    if (!hd.supported) set_error("mdx_jpeg_probe: left to the host decoder (%s)", hd.why);
    return MDX_OK;
}

int mdx_jpeg_coefficients(const uint8_t *file, int64_t size, int16_t *coef, int64_t coef_blocks, uint16_t *quant)
{
    MDX_CHECK_ARG(file && coef && quant && size > 0, "mdx_jpeg_coefficients: NULL pointer or empty file");
    JpegHeader hd;
    MDX_CHECK_ARG(parse_header(file, size, hd), "mdx_jpeg_coefficients: unsupported file (%s)", hd.why);
    mdx_jpeg_info info;
    fill_info(hd, &info);
    MDX_CHECK_ARG(coef_blocks >= info.nblocks, "mdx_jpeg_coefficients: room for %lld blocks, the image has %lld", (long long)coef_blocks,
                  (long long)info.nblocks);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 64; ++i) quant[c * 64 + i] = c < hd.ncomp ? hd.quant[hd.tq[c]][i] : 0;
    MDX_CHECK_ARG(decode_scans(file, size, hd, info, coef), "mdx_jpeg_coefficients: corrupt or unsupported entropy-coded data");
    // The DCT of 8-bit samples stays within +-1024 (x the quantiser's rounding); a dequantised coefficient beyond +-2048 is
    // damage.  libjpeg decodes such a file too, but what comes out then depends on its build (the SIMD IDCT multiplies and
    // packs in 16 bits with saturation, the C one does not), so those files are left to it.
    for (int c = 0; c < hd.ncomp; ++c) {
        const int16_t *blk = coef + info.block_offset[c] * 64;
        const int64_t nb = (int64_t)info.blocks_w[c] * info.blocks_h[c];
        int32_t lim[64];
        for (int i = 0; i < 64; ++i) lim[i] = 2048 / (quant[c * 64 + i] ? quant[c * 64 + i] : 1);
        int bad = 0;
        for (int64_t b = 0; b < nb; ++b)
            for (int i = 0; i < 64; ++i) {
                const int32_t v = blk[b * 64 + i];
                bad |= (v > lim[i]) | (v < -lim[i]);
            }
        MDX_CHECK_ARG(!bad, "mdx_jpeg_coefficients: coefficients out of the range of 8-bit samples (damaged file)");
    }
    return MDX_OK;
}

int mdx_jpeg_pixels(const int16_t *coef, const uint16_t *quant, const mdx_jpeg_info *info, uint8_t *planes, uint8_t *rgb, void *stream)
{
    MDX_CHECK_ARG(coef && quant && info && planes && rgb, "mdx_jpeg_pixels: NULL pointer");
    MDX_CHECK_ARG(info->supported && info->nblocks > 0 && info->width > 0 && info->height > 0, "mdx_jpeg_pixels: unsupported image");
    MDX_CHECK_ARG(info_consistent(*info), "mdx_jpeg_pixels: the geometry is not one mdx_jpeg_probe reports (sampling factors, block counts and "
                                          "offsets must follow from width, height and the component count)");
    JpegGeom g;
    geom_of(*info, &g);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)ceil_div(g.nblocks, 64)), dim3(64), 0, s, coef, quant, g, planes);
    hipLaunchKernelGGL(jpeg_rgb_kernel, dim3((unsigned)ceil_div((int64_t)g.width * g.height, 256)), dim3(256), 0, s,
                       (const uint8_t *)planes, g, rgb);
    MDX_LAUNCH_CHECK();
    return MDX_OK;
}

}  // extern "C"
