// Host half of the JPEG front end (scope row N4: what cv2.imread does at pipeline/run.py:250 before the hot path starts).
// A baseline JPEG is a serial Huffman bit stream -- that part stays on the CPU, as the corner search does -- followed by
// per-block arithmetic (dequantise, inverse DCT, chroma interpolation, colour conversion) that is embarrassingly parallel
// and runs on the GPU (k5_jpeg.hip).  This file: marker parsing (SOF0/SOF1, DQT, DHT, DRI, SOS, JFIF/Adobe/EXIF) and entropy
// decoding into dense int16 coefficient blocks, natural (de-zigzagged) order, laid out per component over the MCU-padded
// block grid -- the layout the reconstruction kernels index directly.
//
// Speed: a 64-bit bit buffer refilled a byte at a time only around 0xFF, a 10-bit look-ahead table per Huffman table that
// resolves (code length, symbol) in one probe, restart intervals located up front by a memchr scan and decoded in parallel
// (each interval is an independent bit stream with its own DC predictors), images of a batch spread over the same pool.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "sv_internal.h"
#include "host_pool.h"

namespace {

constexpr int kLook = 10;

const uint8_t kNatural[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    uint16_t look[1 << kLook];          // (length << 8) | symbol for codes of <= kLook bits, 0 otherwise
    int32_t maxcode[18];                // per length, left-aligned to 16 bits (exclusive upper bound); sentinel at [17]
    int32_t delta[17];                  // vals index = delta[len] + code
    uint8_t vals[256];
    int32_t fast_ac[1 << kLook];        // AC tables: (value << 16) | (run << 8) | total bits when code + magnitude bits fit kLook, else 0
    bool defined = false;

    bool define(const uint8_t *counts /*16*/, const uint8_t *symbols, int total)
    {
        memcpy(vals, symbols, (size_t)total);
        memset(look, 0, sizeof look);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            delta[len] = k - code;
            if (code + counts[len - 1] > (1 << len)) return false;  // over-subscribed table: reject before look[] is indexed with its codes
            for (int i = 0; i < counts[len - 1]; i++, k++, code++) {
                if (len <= kLook) {
                    const int first = code << (kLook - len), span = 1 << (kLook - len);
                    for (int j = 0; j < span; j++) look[first + j] = (uint16_t)((len << 8) | vals[k]);
                }
            }
            maxcode[len] = code << (16 - len);
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        // short code + short magnitude in one probe: the bits after the code ARE the magnitude field
        for (int i = 0; i < (1 << kLook); i++) {
            fast_ac[i] = 0;
            const int e = look[i];
            if (!e) continue;
            const int len = e >> 8, run = (e >> 4) & 15, size = e & 15;
            if (size == 0 || len + size > kLook) continue;
            int v = (i >> (kLook - len - size)) & ((1 << size) - 1);
            if (v < (1 << (size - 1))) v += 1 - (1 << size);
            fast_ac[i] = (int32_t)((unsigned)v << 16 | (unsigned)run << 8 | (unsigned)(len + size));
        }
        defined = true;
        return true;
    }
};

struct Component { int id, h, v, tq, bw, bh; long offset; };

struct Header {
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcu_cols = 0, mcu_rows = 0;
    int restart_interval = 0, orientation = 1;
    bool jfif = false, adobe = false;
    int adobe_transform = 0;
    Component comp[3];
    uint16_t quant[4][64];
    bool quant_defined[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    long coef_count = 0;
    size_t first_scan = 0;                                         // offset of the first SOS marker
};

inline unsigned be16(const uint8_t *p) { return (unsigned)p[0] << 8 | p[1]; }

// EXIF orientation tag (0x0112) of IFD0; cv2.imread rotates/flips accordingly
int exif_orientation(const uint8_t *seg, int n)
{
    if (n < 14 || memcmp(seg, "Exif\0\0", 6) != 0) return 1;
    const uint8_t *t = seg + 6;
    n -= 6;
    const bool little = t[0] == 'I' && t[1] == 'I';
    if (!little && !(t[0] == 'M' && t[1] == 'M')) return 1;
    auto u16 = [&](unsigned o) -> unsigned { return little ? (t[o] | t[o + 1] << 8) : (t[o] << 8 | t[o + 1]); };
    auto u32 = [&](unsigned o) -> unsigned { return little ? (u16(o) | u16(o + 2) << 16) : (u16(o) << 16 | u16(o + 2)); };
    if (u16(2) != 42) return 1;
    const unsigned ifd = u32(4);
    if (ifd > (unsigned)n || ifd + 2 > (unsigned)n) return 1;
    const unsigned entries = u16(ifd);
    for (unsigned i = 0; i < entries; i++) {
        const unsigned e = ifd + 2 + 12 * i;
        if (e + 12 > (unsigned)n) break;
        if (u16(e) == 0x0112) {
            const unsigned v = u16(e + 8);
            return v >= 1 && v <= 8 ? (int)v : 1;
        }
    }
    return 1;
}

// Table-definition segments can appear before any scan, so both the header pass and the scan loop use these
int read_dht(Header &h, const uint8_t *s, int n)
{
    while (n >= 17) {
        const int cls = s[0] >> 4, id = s[0] & 15;
        int total = 0;
        for (int i = 0; i < 16; i++) total += s[1 + i];
        if (cls > 1 || id > 3 || total > 256 || n < 17 + total) return sv_fail(SV_ERR_BAD_ARG, "jpeg: bad DHT segment");
        if (!(cls ? h.ac[id] : h.dc[id]).define(s + 1, s + 17, total)) return sv_fail(SV_ERR_BAD_ARG, "jpeg: over-subscribed Huffman table");
        s += 17 + total;
        n -= 17 + total;
    }
    return SV_OK;
}

int read_dqt(Header &h, const uint8_t *s, int n)
{
    while (n >= 65) {
        const int wide = s[0] >> 4, id = s[0] & 15;
        if (id > 3 || (wide && n < 129)) return sv_fail(SV_ERR_BAD_ARG, "jpeg: bad DQT segment");
        for (int i = 0; i < 64; i++) h.quant[id][kNatural[i]] = wide ? (uint16_t)be16(s + 1 + 2 * i) : s[1 + i];
        h.quant_defined[id] = true;
        s += wide ? 129 : 65;
        n -= wide ? 129 : 65;
    }
    return SV_OK;
}

// Walks marker segments from `pos`; returns the marker code and sets seg/len to its payload, or -1 at the end of data
int next_segment(const uint8_t *d, size_t size, size_t &pos, const uint8_t *&seg, int &len)
{
    for (;;) {
        const uint8_t *ff = pos < size ? (const uint8_t *)memchr(d + pos, 0xFF, size - pos) : nullptr;
        if (!ff) return -1;
        pos = (size_t)(ff - d);
        while (pos < size && d[pos] == 0xFF) pos++;
        if (pos >= size) return -1;
        const int m = d[pos++];
        if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD8)) continue;       // stuffing, TEM, RSTn, SOI: no payload
        if (m == 0xD9) { seg = nullptr; len = 0; return m; }
        if (pos + 2 > size) return -1;
        const int L = (int)be16(d + pos);
        if (L < 2 || pos + (size_t)L > size) return -1;
        seg = d + pos + 2;
        len = L - 2;
        pos += (size_t)L;
        return m;
    }
}

int parse_header(const uint8_t *d, size_t size, Header &h)
{
    if (!d || size < 4 || d[0] != 0xFF || d[1] != 0xD8) return sv_fail(SV_ERR_BAD_ARG, "jpeg: no SOI marker (not a JPEG file)");
    size_t pos = 2;
    bool have_frame = false;
    for (;;) {
        const uint8_t *s;
        int n;
        const size_t at = pos;
        const int m = next_segment(d, size, pos, s, n);
        if (m < 0 || m == 0xD9) return sv_fail(SV_ERR_BAD_ARG, "jpeg: no scan found");
        int rc = SV_OK;
        switch (m) {
        case 0xC0: case 0xC1: {
            if (n < 6) return sv_fail(SV_ERR_BAD_ARG, "jpeg: short SOF segment");
            if (s[0] != 8) return sv_fail(SV_ERR_UNSUPPORTED, "jpeg: %d-bit samples (only 8-bit is supported)", s[0]);
            h.height = (int)be16(s + 1); h.width = (int)be16(s + 3); h.ncomp = s[5];
            if (h.height == 0 || h.width == 0) return sv_fail(SV_ERR_UNSUPPORTED, "jpeg: image size deferred to a DNL marker");
            if (h.ncomp != 1 && h.ncomp != 3) return sv_fail(SV_ERR_UNSUPPORTED, "jpeg: %d components (gray or YCbCr only)", h.ncomp);
            if (n < 6 + 3 * h.ncomp) return sv_fail(SV_ERR_BAD_ARG, "jpeg: short SOF segment");
            for (int c = 0; c < h.ncomp; c++) h.comp[c] = {s[6 + 3 * c], s[7 + 3 * c] >> 4, s[7 + 3 * c] & 15, s[8 + 3 * c] & 3, 0, 0, 0};
            have_frame = true;
            break;
        }
        case 0xC2: return sv_fail(SV_ERR_UNSUPPORTED, "jpeg: progressive files are not supported (baseline / extended sequential Huffman only)");
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            return sv_fail(SV_ERR_UNSUPPORTED, "jpeg: SOF%d (lossless / hierarchical / arithmetic) is not supported", m - 0xC0);
        case 0xC4: rc = read_dht(h, s, n); break;
        case 0xDB: rc = read_dqt(h, s, n); break;
        case 0xDD: if (n < 2) return sv_fail(SV_ERR_BAD_ARG, "jpeg: short DRI segment"); h.restart_interval = (int)be16(s); break;
        case 0xE0: if (n >= 5 && memcmp(s, "JFIF", 5) == 0) h.jfif = true; break;
        case 0xE1: { const int o = exif_orientation(s, n); if (o != 1) h.orientation = o; break; }
        case 0xEE: if (n >= 12 && memcmp(s, "Adobe", 5) == 0) { h.adobe = true; h.adobe_transform = s[11]; } break;
        case 0xDA: h.first_scan = at; goto scanned;
        default: break;
        }
        if (rc) return rc;
    }
scanned:
    if (!have_frame) return sv_fail(SV_ERR_BAD_ARG, "jpeg: scan before frame header");
    if (h.ncomp == 1) h.comp[0].h = h.comp[0].v = 1;
    else {
        const Component *c = h.comp;
        const bool chroma_full = c[1].h == 1 && c[1].v == 1 && c[2].h == 1 && c[2].v == 1;
        const bool luma_ok = (c[0].h == 1 && c[0].v == 1) || (c[0].h == 2 && (c[0].v == 1 || c[0].v == 2));
        if (!chroma_full || !luma_ok)
            return sv_fail(SV_ERR_UNSUPPORTED, "jpeg: sampling %dx%d,%dx%d,%dx%d (4:4:4, 4:2:2 and 4:2:0 are supported)", c[0].h, c[0].v, c[1].h, c[1].v, c[2].h, c[2].v);
        if ((h.adobe && h.adobe_transform != 1) || (!h.jfif && !h.adobe && c[0].id == 'R' && c[1].id == 'G' && c[2].id == 'B'))
            return sv_fail(SV_ERR_UNSUPPORTED, "jpeg: not a YCbCr file");
    }
    h.hmax = h.comp[0].h;
    h.vmax = h.comp[0].v;
    h.mcu_cols = (h.width + 8 * h.hmax - 1) / (8 * h.hmax);
    h.mcu_rows = (h.height + 8 * h.vmax - 1) / (8 * h.vmax);
    long off = 0;
    for (int c = 0; c < h.ncomp; c++) {
        h.comp[c].bw = h.mcu_cols * h.comp[c].h;
        h.comp[c].bh = h.mcu_rows * h.comp[c].v;
        h.comp[c].offset = off;
        off += (long)h.comp[c].bw * h.comp[c].bh * 64;
    }
    h.coef_count = off;
    return SV_OK;
}

long sparse_capacity(const Header &h, size_t size);

void export_info(const Header &h, size_t size, sv_jpeg_info *o)
{
    o->sparse_capacity = sparse_capacity(h, size);
    const bool swap = h.orientation >= 5;
    o->width = h.width; o->height = h.height;
    o->out_width = swap ? h.height : h.width; o->out_height = swap ? h.width : h.height;
    o->components = h.ncomp; o->h_samp = h.hmax; o->v_samp = h.vmax;
    o->orientation = h.orientation; o->restart_interval = h.restart_interval;
    o->coef_count = h.coef_count;
}

// ---- the bit stream of one restart interval ----------------------------------------------------------
struct BitStream {
    const uint8_t *p, *end;
    uint64_t acc = 0;                   // bits left-aligned
    int have = 0;
    long fake = 0;                      // zero bits appended after the interval's last byte (they sit at the tail of the buffer)

    BitStream(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}

    inline void refill()                // afterwards at least 32 bits are buffered (code <= 16 + magnitude <= 16)
    {
        if (have > 32) return;
        if (p + 4 <= end) {             // four bytes at once when none of them is 0xFF (the common case)
            const uint32_t w = (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
            if (((~w - 0x01010101u) & w & 0x80808080u) == 0) {
                acc |= (uint64_t)w << (32 - have);
                have += 32;
                p += 4;
                return;
            }
        }
        while (have <= 56) {
            unsigned c = 0;
            if (p < end) {
                c = *p++;
                if (c == 0xFF && p < end && *p == 0x00) p++;        // byte stuffing; the interval ends before any real marker
            }
            else fake += 8;
            acc |= (uint64_t)c << (56 - have);
            have += 8;
        }
    }
    // true once decoding has consumed bits that are not in the file (a truncated or corrupt interval)
    inline bool starved() const { return (long)have < fake; }
    inline unsigned peek16() const { return (unsigned)(acc >> 48); }
    inline void drop(int n) { acc <<= n; have -= n; }
    inline int take_signed(int s)       // RECEIVE + EXTEND (T.81 F.2.2.1)
    {
        const int v = (int)(acc >> (64 - s));
        drop(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    inline int symbol(const HuffTable &t)
    {
        const unsigned top = peek16();
        const unsigned e = t.look[top >> (16 - kLook)];
        if (e) { drop((int)(e >> 8)); return (int)(e & 255); }
        int len = kLook + 1;
        while ((int)top >= t.maxcode[len]) len++;
        if (len > 16) { drop(16); return 0; }                      // invalid code: libjpeg substitutes 0 and goes on
        drop(len);
        return t.vals[(t.delta[len] + (int)(top >> (16 - len))) & 255];
    }
};

struct ScanComponent { int comp, dc, ac, nh, nv; };

struct Scan {
    int ns = 0;
    ScanComponent sc[3];
    int mcu_cols = 0, mcu_rows = 0;
};

// Where decoded blocks go.  Dense: 64 int16 per block, natural order.  Sparse (the PCIe transport): per block a 64-bit
// mask over ZIGZAG positions and the index of its first value; the non-zero values themselves are appended, in zigzag order,
// to a value stream -- a q90 1080p frame is ~0.6 M values + 12 B per block instead of 3.1 M int16.
struct DenseOut { int16_t *coef; };
struct SparseOut { uint64_t *masks; uint32_t *offs; int16_t *vals; };

// Decodes MCUs [m0, m1) of a scan from one restart interval's bytes.  Sparse: values are appended from index `vpos` on, never
// at or beyond `vlimit`; returns the index after the last value written (dense: 0).
// Insufficient data, as libjpeg handles it (jdhuff.c: insufficient_data, reset at every restart): once decoding needs bits the
// interval does not hold, the block being decoded and every later block of the interval stay zero (they reconstruct as mid-grey),
// which is what cv2.imread / Pillow(LOAD_TRUNCATED_IMAGES) show for the tail of a cut file.
template <bool SPARSE, class Out>
long decode_interval(const Header &h, const Scan &scan, const uint8_t *b, const uint8_t *e, long m0, long m1, const Out &out, long vpos, long vlimit)
{
    BitStream bs(b, e);
    int pred[3] = {0, 0, 0};
    bool dead = false;
    for (long mi = m0; mi < m1; mi++) {
        const int mx = (int)(mi % scan.mcu_cols), my = (int)(mi / scan.mcu_cols);
        for (int i = 0; i < scan.ns; i++) {
            const ScanComponent &s = scan.sc[i];
            const Component &c = h.comp[s.comp];
            const HuffTable &dct = h.dc[s.dc], &act = h.ac[s.ac];
            for (int v = 0; v < s.nv; v++)
                for (int u = 0; u < s.nh; u++) {
                    const long bi = c.offset / 64 + (long)(my * s.nv + v) * c.bw + (mx * s.nh + u);
                    int16_t *blk = nullptr;
                    uint64_t mask = 0;
                    const long vstart = vpos;
                    if constexpr (!SPARSE) {
                        blk = out.coef + bi * 64;
                        memset(blk, 0, 64 * sizeof(int16_t));       // cleared while the lines are about to be written anyway
                    }
                    if (dead) {
                        if constexpr (SPARSE) { out.masks[bi] = 0; out.offs[bi] = (uint32_t)vstart; }
                        continue;
                    }
                    bool full = false;                              // sparse: the interval's stretch of the value stream is used up
                    bs.refill();
                    const int t = bs.symbol(dct) & 15;
                    if (t) pred[i] += bs.take_signed(t);
                    if constexpr (SPARSE) {
                        if ((int16_t)pred[i]) {
                            if (vpos < vlimit) { mask = 1; out.vals[vpos++] = (int16_t)pred[i]; } else full = true;
                        }
                    } else blk[0] = (int16_t)pred[i];
                    for (int k = 1; k < 64 && !full;) {
                        bs.refill();
                        const int32_t f = act.fast_ac[bs.peek16() >> (16 - kLook)];
                        int val;
                        if (f) {
                            k += (f >> 8) & 15;
                            if (k > 63) break;
                            bs.drop(f & 255);
                            val = f >> 16;
                        } else {
                            const int rs = bs.symbol(act), run = rs >> 4, size = rs & 15;
                            if (size == 0) {
                                if (run != 15) break;               // EOB
                                k += 16;
                                continue;
                            }
                            k += run;
                            if (k > 63) break;
                            val = bs.take_signed(size);
                        }
                        if constexpr (SPARSE) {
                            if ((int16_t)val) {
                                if (vpos < vlimit) { mask |= 1ull << k; out.vals[vpos++] = (int16_t)val; } else full = true;
                            }
                        } else blk[kNatural[k]] = (int16_t)val;
                        k++;
                    }
                    if (bs.starved() || full) {                     // this block was (partly) made of bits that are not there: drop it and the rest
                        dead = true;
                        mask = 0;
                        vpos = vstart;
                        if constexpr (!SPARSE) memset(blk, 0, 64 * sizeof(int16_t));
                    }
                    if constexpr (SPARSE) { out.masks[bi] = mask; out.offs[bi] = (uint32_t)vstart; }
                }
        }
    }
    return vpos;
}

// every non-zero value costs at least 2 bits (code >= 1, magnitude >= 1): an interval of n bytes holds at most 4n values
inline long interval_value_bound(size_t bytes) { return 4 * (long)bytes + 2; }

// Splits the entropy-coded bytes that start at `pos` into restart intervals; returns the offset of the marker that ends the scan
size_t split_intervals(const uint8_t *d, size_t size, size_t pos, std::vector<std::pair<size_t, size_t>> &out)
{
    size_t start = pos;
    for (;;) {
        const uint8_t *ff = pos < size ? (const uint8_t *)memchr(d + pos, 0xFF, size - pos) : nullptr;
        if (!ff || (size_t)(ff - d) + 1 >= size) { out.emplace_back(start, size); return size; }
        const size_t at = (size_t)(ff - d);
        const int m = d[at + 1];
        if (m == 0x00) { pos = at + 2; continue; }
        if (m == 0xFF) { pos = at + 1; continue; }
        if (m >= 0xD0 && m <= 0xD7) { out.emplace_back(start, at); start = pos = at + 2; continue; }
        out.emplace_back(start, at);
        return at;
    }
}

template <bool SPARSE, class Out>
int entropy_decode(const uint8_t *d, size_t size, Header &h, const Out &out, int threads, long vals_cap, long *vals_used)
{
    bool first = true;
    size_t pos = h.first_scan;
    unsigned covered = 0;                                          // bit c: component c has been decoded
    const unsigned all = (1u << h.ncomp) - 1;
    long vbase = 0;                                                // sparse: where the next scan's value regions start
    while (covered != all) {
        const uint8_t *s;
        int n;
        const int m = next_segment(d, size, pos, s, n);
        if (m < 0 || m == 0xD9) return sv_fail(SV_ERR_BAD_ARG, "jpeg: data ends before all components were decoded");
        if (m == 0xC4) { const int rc = read_dht(h, s, n); if (rc) return rc; continue; }
        if (m == 0xDD) { if (n < 2) return sv_fail(SV_ERR_BAD_ARG, "jpeg: short DRI segment"); h.restart_interval = (int)be16(s); continue; }
        if (m != 0xDA) continue;
        Scan scan;
        scan.ns = n >= 1 ? s[0] : 0;
        if (scan.ns < 1 || scan.ns > h.ncomp || n < 4 + 2 * scan.ns) return sv_fail(SV_ERR_BAD_ARG, "jpeg: bad SOS segment");
        for (int i = 0; i < scan.ns; i++) {
            int c = 0;
            while (c < h.ncomp && h.comp[c].id != s[1 + 2 * i]) c++;
            const int td = s[2 + 2 * i] >> 4, ta = s[2 + 2 * i] & 15;
            if (c == h.ncomp || td > 3 || ta > 3 || !h.dc[td].defined || !h.ac[ta].defined) return sv_fail(SV_ERR_BAD_ARG, "jpeg: scan refers to an undefined component or table");
            if (covered >> c & 1) return sv_fail(SV_ERR_BAD_ARG, "jpeg: component %d appears in more than one scan (or twice in one)", c);
            covered |= 1u << c;
            scan.sc[i] = {c, td, ta, scan.ns == 1 ? 1 : h.comp[c].h, scan.ns == 1 ? 1 : h.comp[c].v};
        }
        const uint8_t *tail = s + 1 + 2 * scan.ns;
        if (tail[0] != 0 || tail[1] != 63 || tail[2] != 0) return sv_fail(SV_ERR_UNSUPPORTED, "jpeg: spectral selection / successive approximation in a sequential file");
        if (first && scan.ns < h.ncomp) {                          // only a full interleave reaches (and clears) every block: a
            if constexpr (SPARSE) {                                // component's own block grid leaves the MCU padding blocks out
                memset(out.masks, 0, (size_t)(h.coef_count / 64) * sizeof(uint64_t));
                memset(out.offs, 0, (size_t)(h.coef_count / 64) * sizeof(uint32_t));
            } else memset(out.coef, 0, (size_t)h.coef_count * sizeof(int16_t));
        }
        first = false;
        if (scan.ns == 1) {                                        // non-interleaved: the component's own block grid
            const Component &c = h.comp[scan.sc[0].comp];
            scan.mcu_cols = ((h.width * c.h + h.hmax - 1) / h.hmax + 7) / 8;
            scan.mcu_rows = ((h.height * c.v + h.vmax - 1) / h.vmax + 7) / 8;
        } else { scan.mcu_cols = h.mcu_cols; scan.mcu_rows = h.mcu_rows; }
        const long nmcu = (long)scan.mcu_cols * scan.mcu_rows;
        std::vector<std::pair<size_t, size_t>> iv;
        pos = split_intervals(d, size, pos, iv);
        const long per = h.restart_interval > 0 ? h.restart_interval : nmcu;
        const long need = (nmcu + per - 1) / per;
        if ((long)iv.size() < need) return sv_fail(SV_ERR_BAD_ARG, "jpeg: %ld restart intervals found, %ld needed (truncated file?)", (long)iv.size(), need);
        std::vector<long> vstart((size_t)need + 1, 0), vend((size_t)need, 0);
        if constexpr (SPARSE) {                                    // each interval appends to its own stretch of the value stream
            vstart[0] = vbase;
            for (long k = 0; k < need; k++) {
                long blocks = 0;
                for (int i = 0; i < scan.ns; i++) blocks += (long)scan.sc[i].nh * scan.sc[i].nv;
                const long m0 = k * per, m1 = std::min(nmcu, m0 + per);
                vstart[k + 1] = vstart[k] + std::min(interval_value_bound(iv[k].second - iv[k].first), (m1 - m0) * blocks * 64);
            }
            if (vstart[need] > vals_cap) return sv_fail(SV_ERR_BUFFER, "jpeg: value buffer holds %ld, this file may need %ld", vals_cap, vstart[need]);
        }
        const Header &hc = h;
        WorkerPool::instance().parallel_for((int)need, threads, [&](int k) {
            const long m0 = (long)k * per, m1 = std::min(nmcu, m0 + per);
            vend[k] = decode_interval<SPARSE>(hc, scan, d + iv[k].first, d + iv[k].second, m0, m1, out, vstart[k], SPARSE ? vstart[k + 1] : 0);
        });
        if constexpr (SPARSE) {
            vbase = vstart[need];
            if (vals_used) *vals_used = vend[need - 1];            // everything the kernels can reference lies below this index
        }
    }
    return SV_OK;
}

int fill_quant(const Header &h, uint16_t *quant)
{
    memset(quant, 0, 3 * 64 * sizeof(uint16_t));
    for (int c = 0; c < h.ncomp; c++) {
        if (!h.quant_defined[h.comp[c].tq]) return sv_fail(SV_ERR_BAD_ARG, "jpeg: component %d uses an undefined quantisation table", c);
        memcpy(quant + 64 * c, h.quant[h.comp[c].tq], 64 * sizeof(uint16_t));
    }
    return SV_OK;
}

long sparse_capacity(const Header &h, size_t size)
{
    const long nmcu = (long)h.mcu_cols * h.mcu_rows;
    const long intervals = h.restart_interval > 0 ? (nmcu + h.restart_interval - 1) / h.restart_interval : 1;
    return std::min(h.coef_count, 4 * (long)size + 2 * 3 * intervals + 64);
}

}  // namespace

// ---- C ABI -------------------------------------------------------------------------------------------
extern "C" int sv_jpeg_parse(const uint8_t *data, size_t size, sv_jpeg_info *info)
{
    if (!data || !info) return sv_fail(SV_ERR_BAD_ARG, "sv_jpeg_parse: NULL argument");
    std::unique_ptr<Header> h(new Header);
    const int rc = parse_header(data, size, *h);
    if (rc) return rc;
    export_info(*h, size, info);
    return SV_OK;
}

extern "C" int sv_jpeg_entropy_decode(const uint8_t *data, size_t size, int16_t *coef, uint16_t *quant, int threads)
{
    if (!data || !coef || !quant) return sv_fail(SV_ERR_BAD_ARG, "sv_jpeg_entropy_decode: NULL argument");
    std::unique_ptr<Header> h(new Header);
    int rc = parse_header(data, size, *h);
    if (rc || (rc = fill_quant(*h, quant))) return rc;
    return entropy_decode<false>(data, size, *h, DenseOut{coef}, threads < 1 ? 1 : threads, 0, nullptr);
}

extern "C" int sv_jpeg_entropy_decode_sparse(const uint8_t *data, size_t size, uint64_t *masks, uint32_t *offsets, int16_t *values, long values_cap,
                                             long *values_used, uint16_t *quant, int threads)
{
    if (!data || !masks || !offsets || !values || !values_used || !quant) return sv_fail(SV_ERR_BAD_ARG, "sv_jpeg_entropy_decode_sparse: NULL argument");
    std::unique_ptr<Header> h(new Header);
    int rc = parse_header(data, size, *h);
    if (rc || (rc = fill_quant(*h, quant))) return rc;
    *values_used = 0;
    return entropy_decode<true>(data, size, *h, SparseOut{masks, offsets, values}, threads < 1 ? 1 : threads, values_cap, values_used);
}

// Images of a batch over the pool's threads; every image decodes its restart intervals serially (a worker's nested
// parallel_for runs inline).  Dense when `coefs` is given, sparse when masks/offsets/values are.
extern "C" int sv_jpeg_entropy_decode_batch(const uint8_t *const *datas, const size_t *sizes, int n, int16_t *const *coefs, uint64_t *const *masks,
                                            uint32_t *const *offsets, int16_t *const *values, const long *values_cap, long *values_used,
                                            uint16_t *quants, int threads, int *status)
{
    const bool sparse = coefs == nullptr;
    if (!datas || !sizes || !quants || !status || n <= 0 || (sparse && (!masks || !offsets || !values || !values_cap || !values_used)))
        return sv_fail(SV_ERR_BAD_ARG, "sv_jpeg_entropy_decode_batch: bad argument");
    if (threads < 1) threads = 1;
    std::vector<std::unique_ptr<Header>> hs((size_t)n);
    auto one = [&](int i, int inner_threads) {
        hs[i].reset(new Header);
        const bool ok = datas[i] && (sparse ? (masks[i] && offsets[i] && values[i]) : coefs[i] != nullptr);
        int rc = ok ? parse_header(datas[i], sizes[i], *hs[i]) : SV_ERR_BAD_ARG;
        if (!rc) rc = fill_quant(*hs[i], quants + 192 * (size_t)i);
        if (!rc) {
            if (sparse) {
                values_used[i] = 0;
                rc = entropy_decode<true>(datas[i], sizes[i], *hs[i], SparseOut{masks[i], offsets[i], values[i]}, inner_threads, values_cap[i], values_used + i);
            } else rc = entropy_decode<false>(datas[i], sizes[i], *hs[i], DenseOut{coefs[i]}, inner_threads, 0, nullptr);
        }
        status[i] = rc;
    };
    if (threads == 1 || n == 1) for (int i = 0; i < n; i++) one(i, n == 1 ? threads : 1);
    else WorkerPool::instance().parallel_for(n, threads, [&](int i) { one(i, 1); });
    for (int i = 0; i < n; i++)
        if (status[i] != SV_OK) return sv_fail(status[i], "sv_jpeg_entropy_decode_batch: image %d failed (see status[])", i);
    return SV_OK;
}
