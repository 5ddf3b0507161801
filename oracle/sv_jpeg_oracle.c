/* oracle/sv_jpeg_oracle.c -- TEST INFRASTRUCTURE ONLY (checker for scope row N4, "JPEG decode for real-image feeds").
 *
 * CPU restatement of what `cv2.imread(path)` (pipeline/run.py:250, pipeline/run_v2.py:267, tests/test_integration.py:126)
 * does to a baseline JPEG.  The arithmetic lives in a third-party dependency that is not under /root/reference:
 * OpenCV's imread -> libjpeg(-turbo), requirement `opencv-python>=4.8` (ml/requirements.txt:3, unpinned).  Restated here
 * from libjpeg's published algorithm with its default decompression parameters (the ones cv2 leaves untouched):
 *   - Huffman entropy decoding of sequential DCT scans (ITU T.81 F.2), restart intervals;
 *   - dct_method = JDCT_ISLOW: the 13-bit fixed-point Loeffler-Ligtenberg-Moschytz inverse DCT (jidctint.c);
 *   - do_fancy_upsampling = TRUE: "triangle" chroma interpolation h2v1 / h2v2 (jdsample.c), plain replication when the
 *     down-sampled width is <= 2;
 *   - YCbCr -> RGB with the 16-bit fixed-point tables of jdcolor.c;
 *   - EXIF orientation applied (OpenCV >= 3.1 imread without IMREAD_IGNORE_ORIENTATION).
 * Pinned: tests/test_jpeg.py checks this file bit-for-bit against Pillow's decoder (libjpeg-turbo, same defaults) on the
 * reference's five photos (data/test_images/sample_1..5.jpg) and on synthetic 4:4:4 / 4:2:2 / 4:2:0 / gray / restart-interval
 * files, and the orientation mapping against PIL.ImageOps.exif_transpose.  cv2 itself is absent from the image, so
 * "imread == Pillow" is the working assumption (both sit on libjpeg-turbo) -- recorded in DESIGN.md.
 * Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may call this file.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int width, height, out_width, out_height, components, h_samp, v_samp, orientation, restart_interval;
    long coef_count;
} svo_jpeg_info;

typedef struct {
    uint8_t bits[17], vals[256];
    int mincode[17], maxcode[17], valptr[17], set;
} htab;

typedef struct {
    int W, H, nc, id[3], hs[3], vs[3], tq[3], hmax, vmax;
    uint16_t q[4][64];
    int qset[4];
    htab dc[4], ac[4];
    int ri, orientation, jfif, adobe, adobe_tf;
    int mcux, mcuy, bw[3], bh[3];
    long off[3], ncoef;
    int have_sof;
} jpg;

static const uint8_t ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static int rd16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

static void exif_orientation(const uint8_t *p, int n, int *orient)
{
    if (n < 14 || memcmp(p, "Exif\0\0", 6)) return;
    const uint8_t *t = p + 6;
    n -= 6;
    int le;
    if (t[0] == 'I' && t[1] == 'I') le = 1; else if (t[0] == 'M' && t[1] == 'M') le = 0; else return;
#define R16(o) (le ? (t[o] | (t[(o) + 1] << 8)) : ((t[o] << 8) | t[(o) + 1]))
#define R32(o) (le ? ((unsigned)t[o] | ((unsigned)t[(o) + 1] << 8) | ((unsigned)t[(o) + 2] << 16) | ((unsigned)t[(o) + 3] << 24)) \
                   : (((unsigned)t[o] << 24) | ((unsigned)t[(o) + 1] << 16) | ((unsigned)t[(o) + 2] << 8) | (unsigned)t[(o) + 3]))
    if (R16(2) != 42) return;
    unsigned ifd = R32(4);
    if (ifd + 2 > (unsigned)n) return;
    int cnt = R16(ifd);
    for (int i = 0; i < cnt; i++) {
        unsigned e = ifd + 2 + 12 * i;
        if (e + 12 > (unsigned)n) return;
        if (R16(e) == 0x0112) {
            int v = R16(e + 8);
            if (v >= 1 && v <= 8) *orient = v;
            return;
        }
    }
#undef R16
#undef R32
}

static void build(htab *h)
{
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        code += h->bits[l];
        k += h->bits[l];
        h->maxcode[l] = h->bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h->set = 1;
}

/* -1 malformed, -4 unsupported; on success *sos = offset of the first SOS marker segment */
static int parse_headers(const uint8_t *d, size_t len, jpg *j, size_t *sos)
{
    memset(j, 0, sizeof *j);
    j->orientation = 1;
    if (len < 4 || d[0] != 0xFF || d[1] != 0xD8) return -1;
    size_t p = 2;
    for (;;) {
        while (p < len && d[p] != 0xFF) p++;
        while (p < len && d[p] == 0xFF) p++;
        if (p >= len) return -1;
        int m = d[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return -1;
        if (p + 2 > len) return -1;
        int L = rd16(d + p);
        if (L < 2 || p + L > len) return -1;
        const uint8_t *s = d + p + 2;
        int n = L - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (n < 6 || s[0] != 8) return s[0] != 8 ? -4 : -1;
            j->H = rd16(s + 1); j->W = rd16(s + 3); j->nc = s[5];
            if (j->H == 0 || j->W == 0) return -4;
            if (j->nc != 1 && j->nc != 3) return -4;
            if (n < 6 + 3 * j->nc) return -1;
            for (int c = 0; c < j->nc; c++) {
                j->id[c] = s[6 + 3 * c]; j->hs[c] = s[7 + 3 * c] >> 4; j->vs[c] = s[7 + 3 * c] & 15; j->tq[c] = s[8 + 3 * c] & 3;
            }
            j->have_sof = 1;
        } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return -4;                                          /* progressive, lossless, arithmetic, hierarchical */
        } else if (m == 0xC4) {
            while (n >= 17) {
                int tc = s[0] >> 4, th = s[0] & 15, tot = 0;
                if (tc > 1 || th > 3) return -1;
                htab *h = tc ? &j->ac[th] : &j->dc[th];
                h->bits[0] = 0;
                for (int i = 1; i <= 16; i++) { h->bits[i] = s[i]; tot += s[i]; }
                if (tot > 256 || n < 17 + tot) return -1;
                memcpy(h->vals, s + 17, tot);
                build(h);
                s += 17 + tot; n -= 17 + tot;
            }
        } else if (m == 0xDB) {
            while (n >= 65) {
                int pq = s[0] >> 4, tq = s[0] & 15;
                if (tq > 3) return -1;
                if (pq) { if (n < 129) return -1; for (int i = 0; i < 64; i++) j->q[tq][ZZ[i]] = (uint16_t)rd16(s + 1 + 2 * i); s += 129; n -= 129; }
                else { for (int i = 0; i < 64; i++) j->q[tq][ZZ[i]] = s[1 + i]; s += 65; n -= 65; }
                j->qset[tq] = 1;
            }
        } else if (m == 0xDD) {
            if (n < 2) return -1;
            j->ri = rd16(s);
        } else if (m == 0xE0) {
            if (n >= 5 && !memcmp(s, "JFIF", 5)) j->jfif = 1;
        } else if (m == 0xE1) {
            exif_orientation(s, n, &j->orientation);
        } else if (m == 0xEE) {
            if (n >= 12 && !memcmp(s, "Adobe", 5)) { j->adobe = 1; j->adobe_tf = s[11]; }
        } else if (m == 0xDA) {
            *sos = p - 2;
            break;
        }
        p += L;
    }
    if (!j->have_sof) return -1;
    if (j->nc == 1) { j->hs[0] = j->vs[0] = 1; }
    else {
        if (j->hs[1] != 1 || j->vs[1] != 1 || j->hs[2] != 1 || j->vs[2] != 1) return -4;
        if (!((j->hs[0] == 1 || j->hs[0] == 2) && (j->vs[0] == 1 || j->vs[0] == 2)) || (j->hs[0] == 1 && j->vs[0] == 2)) return -4;
        if (j->adobe && j->adobe_tf != 1) return -4;            /* RGB / CMYK-family files */
        if (!j->jfif && !j->adobe && j->id[0] == 'R' && j->id[1] == 'G' && j->id[2] == 'B') return -4;
    }
    j->hmax = j->hs[0]; j->vmax = j->vs[0];
    j->mcux = (j->W + 8 * j->hmax - 1) / (8 * j->hmax);
    j->mcuy = (j->H + 8 * j->vmax - 1) / (8 * j->vmax);
    long off = 0;
    for (int c = 0; c < j->nc; c++) {
        j->bw[c] = j->mcux * j->hs[c]; j->bh[c] = j->mcuy * j->vs[c];
        j->off[c] = off;
        off += (long)j->bw[c] * j->bh[c] * 64;
    }
    j->ncoef = off;
    return 0;
}

static void fill_info(const jpg *j, svo_jpeg_info *o)
{
    o->width = j->W; o->height = j->H; o->components = j->nc; o->h_samp = j->hmax; o->v_samp = j->vmax;
    o->orientation = j->orientation; o->restart_interval = j->ri; o->coef_count = j->ncoef;
    const int swap = j->orientation >= 5;
    o->out_width = swap ? j->H : j->W; o->out_height = swap ? j->W : j->H;
}

int svo_jpeg_info_parse(const uint8_t *d, size_t len, svo_jpeg_info *o)
{
    jpg j;
    size_t sos;
    int rc = parse_headers(d, len, &j, &sos);
    if (rc) return rc;
    fill_info(&j, o);
    return 0;
}

/* ---- entropy decoding, one bit at a time (T.81 F.2.2.3 DECODE / RECEIVE / EXTEND) ---- */
typedef struct { const uint8_t *p, *end; int cur, n, hit_marker, starved; } bitrd;   /* starved: bits that are not in the file were consumed */

static int getbit(bitrd *b)
{
    if (b->n == 0) {
        int c = 0;
        if (!b->hit_marker && b->p < b->end) {
            c = *b->p++;
            if (c == 0xFF) {
                int c2 = b->p < b->end ? *b->p : 0xD9;
                if (c2 == 0) b->p++;
                else { b->p--; b->hit_marker = 1; c = 0; b->starved = 1; }   /* a marker inside the scan: libjpeg feeds zero bits */
            }
        } else b->starved = 1;                                 /* ... as it does past the end of the data (jdhuff.c: insufficient_data) */
        b->cur = c; b->n = 8;
    }
    b->n--;
    return (b->cur >> b->n) & 1;
}

static int decode_sym(bitrd *b, const htab *h)
{
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | getbit(b);
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) return h->vals[h->valptr[l] + code - h->mincode[l]];
    }
    return 0;                                                   /* bad code: libjpeg warns and returns 0 */
}

static int receive_extend(bitrd *b, int s)
{
    int v = 0;
    for (int i = 0; i < s; i++) v = (v << 1) | getbit(b);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

static int decode_scans(const uint8_t *d, size_t len, const jpg *j0, size_t p, int16_t *coef)
{
    jpg jj = *j0, *j = &jj;                                     /* tables may be redefined between scans */
    memset(coef, 0, (size_t)j->ncoef * sizeof(int16_t));
    int seen = 0;
    for (;;) {
        while (p < len && d[p] != 0xFF) p++;
        while (p < len && d[p] == 0xFF) p++;
        if (p >= len) break;
        int m = d[p++];
        if (m == 0xD9) break;
        if ((m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (p + 2 > len) return -1;
        int L = rd16(d + p);
        if (L < 2 || p + L > len) return -1;
        const uint8_t *s = d + p + 2;
        int n = L - 2;
        if (m == 0xC4) {
            while (n >= 17) {
                int tc = s[0] >> 4, th = s[0] & 15, tot = 0;
                if (tc > 1 || th > 3) return -1;
                htab *h = tc ? &j->ac[th] : &j->dc[th];
                for (int i = 1; i <= 16; i++) { h->bits[i] = s[i]; tot += s[i]; }
                if (tot > 256 || n < 17 + tot) return -1;
                memcpy(h->vals, s + 17, tot);
                build(h);
                s += 17 + tot; n -= 17 + tot;
            }
            p += L;
            continue;
        }
        if (m == 0xDD) { if (n < 2) return -1; j->ri = rd16(s); p += L; continue; }
        if (m != 0xDA) { p += L; continue; }
        /* SOS */
        int ns = s[0];
        if (ns < 1 || ns > j->nc || n < 1 + 2 * ns + 3) return -1;
        int ci[3], td[3], ta[3];
        for (int i = 0; i < ns; i++) {
            int c;
            for (c = 0; c < j->nc; c++) if (j->id[c] == s[1 + 2 * i]) break;
            if (c == j->nc) return -1;
            ci[i] = c; td[i] = s[2 + 2 * i] >> 4; ta[i] = s[2 + 2 * i] & 15;
            if (td[i] > 3 || ta[i] > 3 || !j->dc[td[i]].set || !j->ac[ta[i]].set) return -1;
        }
        if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return -4;
        p += L;
        bitrd b = {d + p, d + len, 0, 0, 0, 0};
        int dead = 0;                                           /* insufficient data: the rest of the restart interval stays zero */
        int pred[3] = {0, 0, 0};
        long nmcu;
        int mw, mh;
        if (ns == 1) {                                          /* non-interleaved: the component's own block grid */
            const int c = ci[0];
            mw = ((j->W * j->hs[c] + j->hmax - 1) / j->hmax + 7) / 8;
            mh = ((j->H * j->vs[c] + j->vmax - 1) / j->vmax + 7) / 8;
        } else { mw = j->mcux; mh = j->mcuy; }
        nmcu = (long)mw * mh;
        int rst = 0;
        for (long mi = 0; mi < nmcu; mi++) {
            if (j->ri && mi && mi % j->ri == 0) {
                b.n = 0;
                const uint8_t *q = b.p;
                while (q < b.end && *q != 0xFF) q++;            /* libjpeg also skips to the next marker */
                while (q + 1 < b.end && q[1] == 0xFF) q++;
                if (q + 1 < b.end && q[1] >= 0xD0 && q[1] <= 0xD7) { b.p = q + 2; b.hit_marker = 0; b.starved = 0; dead = 0; }
                else return -1;
                rst++;
                pred[0] = pred[1] = pred[2] = 0;
            }
            const int mx = (int)(mi % mw), my = (int)(mi / mw);
            for (int i = 0; i < ns; i++) {
                const int c = ci[i];
                const int nh = ns == 1 ? 1 : j->hs[c], nv = ns == 1 ? 1 : j->vs[c];
                for (int v = 0; v < nv; v++)
                    for (int h = 0; h < nh; h++) {
                        const int bx = mx * nh + h, by = my * nv + v;
                        int16_t *blk = coef + j->off[c] + ((long)by * j->bw[c] + bx) * 64;
                        if (dead) continue;                     /* coef was cleared up front */
                        const int t = decode_sym(&b, &j->dc[td[i]]);
                        if (t > 15) return -1;
                        pred[i] += t ? receive_extend(&b, t) : 0;
                        blk[0] = (int16_t)pred[i];
                        for (int k = 1; k < 64;) {
                            const int rs = decode_sym(&b, &j->ac[ta[i]]), r = rs >> 4, sz = rs & 15;
                            if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                            k += r;
                            if (k > 63) break;                  /* corrupt: libjpeg would overrun its own checks the same way */
                            blk[ZZ[k]] = (int16_t)receive_extend(&b, sz);
                            k++;
                        }
                        if (b.starved) { memset(blk, 0, 64 * sizeof(int16_t)); dead = 1; }   /* a block made of bits that are not there is dropped */
                    }
            }
        }
        seen += ns;
        /* continue at the next marker */
        p = (size_t)(b.p - d);
        if (seen >= j->nc) break;
    }
    return seen >= j->nc ? 0 : -1;
}

int svo_jpeg_coefficients(const uint8_t *d, size_t len, int16_t *coef, uint16_t *quant)
{
    jpg j;
    size_t sos;
    int rc = parse_headers(d, len, &j, &sos);
    if (rc) return rc;
    for (int c = 0; c < j.nc; c++) {
        if (!j.qset[j.tq[c]]) return -1;
        memcpy(quant + 64 * c, j.q[j.tq[c]], 128);
    }
    return decode_scans(d, len, &j, sos, coef);
}

/* ---- jidctint.c, jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2 ---- */
#define DESCALE(x, n) (((x) + (1L << ((n) - 1))) >> (n))
static uint8_t range_limit(long x)                              /* the post-IDCT table: index (x & 1023), centred on 128 */
{
    const int v = (int)(x & 1023);
    return (uint8_t)(v < 128 ? v + 128 : v < 512 ? 255 : v < 896 ? 0 : v - 896);
}

static void idct_islow(const int16_t *in, const uint16_t *q, uint8_t *out, long pitch)
{
    long ws[64];
    for (int c = 0; c < 8; c++) {
        long z2 = (long)in[16 + c] * q[16 + c], z3 = (long)in[48 + c] * q[48 + c];
        long z1 = (z2 + z3) * 4433;
        long tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
        z2 = (long)in[c] * q[c]; z3 = (long)in[32 + c] * q[32 + c];
        long tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = (long)in[56 + c] * q[56 + c]; tmp1 = (long)in[40 + c] * q[40 + c];
        tmp2 = (long)in[24 + c] * q[24 + c]; tmp3 = (long)in[8 + c] * q[8 + c];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3, z5 = (z3 + z4) * 9633;
        tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
        z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[c] = DESCALE(tmp10 + tmp3, 11); ws[56 + c] = DESCALE(tmp10 - tmp3, 11);
        ws[8 + c] = DESCALE(tmp11 + tmp2, 11); ws[48 + c] = DESCALE(tmp11 - tmp2, 11);
        ws[16 + c] = DESCALE(tmp12 + tmp1, 11); ws[40 + c] = DESCALE(tmp12 - tmp1, 11);
        ws[24 + c] = DESCALE(tmp13 + tmp0, 11); ws[32 + c] = DESCALE(tmp13 - tmp0, 11);
    }
    for (int r = 0; r < 8; r++) {
        const long *w = ws + 8 * r;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * 4433;
        long tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
        long tmp0 = (w[0] + w[4]) * 8192, tmp1 = (w[0] - w[4]) * 8192;
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3, z5 = (z3 + z4) * 9633;
        tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
        z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        uint8_t *o = out + r * pitch;
        o[0] = range_limit(DESCALE(tmp10 + tmp3, 18)); o[7] = range_limit(DESCALE(tmp10 - tmp3, 18));
        o[1] = range_limit(DESCALE(tmp11 + tmp2, 18)); o[6] = range_limit(DESCALE(tmp11 - tmp2, 18));
        o[2] = range_limit(DESCALE(tmp12 + tmp1, 18)); o[5] = range_limit(DESCALE(tmp12 - tmp1, 18));
        o[3] = range_limit(DESCALE(tmp13 + tmp0, 18)); o[4] = range_limit(DESCALE(tmp13 - tmp0, 18));
    }
}

/* ---- jdsample.c: one full-resolution row of a chroma plane ---- */
static void h2v1_fancy_row(const uint8_t *in, int dw, uint8_t *out)
{
    int inv = in[0];
    *out++ = (uint8_t)inv;
    *out++ = (uint8_t)((inv * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < dw - 1; i++) {
        inv = in[i] * 3;
        *out++ = (uint8_t)((inv + in[i - 1] + 1) >> 2);
        *out++ = (uint8_t)((inv + in[i + 1] + 2) >> 2);
    }
    inv = in[dw - 1];
    *out++ = (uint8_t)((inv * 3 + in[dw - 2] + 1) >> 2);
    *out++ = (uint8_t)inv;
}

static void h2v2_fancy_row(const uint8_t *near, const uint8_t *far, int dw, uint8_t *out)
{
    int thiscol = near[0] * 3 + far[0], nextcol = near[1] * 3 + far[1], lastcol;
    *out++ = (uint8_t)((thiscol * 4 + 8) >> 4);
    *out++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol; thiscol = nextcol;
    for (int i = 2; i < dw; i++) {
        nextcol = near[i] * 3 + far[i];
        *out++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        *out++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol; thiscol = nextcol;
    }
    *out++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
    *out++ = (uint8_t)((thiscol * 4 + 7) >> 4);
}

static uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

int svo_jpeg_decode_bgr(const uint8_t *d, size_t len, uint8_t *bgr)
{
    jpg j;
    size_t sos;
    int rc = parse_headers(d, len, &j, &sos);
    if (rc) return rc;
    int16_t *coef = (int16_t *)malloc((size_t)j.ncoef * sizeof(int16_t));
    if (!coef) return -1;
    rc = decode_scans(d, len, &j, sos, coef);
    if (rc) { free(coef); return rc; }
    for (int c = 0; c < j.nc; c++) if (!j.qset[j.tq[c]]) { free(coef); return -1; }
    uint8_t *plane[3] = {0, 0, 0};
    for (int c = 0; c < j.nc; c++) {
        const long pw = (long)j.bw[c] * 8, ph = (long)j.bh[c] * 8;
        plane[c] = (uint8_t *)malloc((size_t)(pw * ph));
        for (int by = 0; by < j.bh[c]; by++)
            for (int bx = 0; bx < j.bw[c]; bx++)
                idct_islow(coef + j.off[c] + ((long)by * j.bw[c] + bx) * 64, j.q[j.tq[c]], plane[c] + (long)by * 8 * pw + bx * 8, pw);
    }
    free(coef);
    const int W = j.W, H = j.H;
    const int dw = (W + j.hmax - 1) / j.hmax, dh = (H + j.vmax - 1) / j.vmax;      /* real size of the chroma planes */
    uint8_t *rows = (uint8_t *)malloc((size_t)(2 * (2 * (long)dw + 4)));
    uint8_t *urow[2] = {rows, rows + 2 * (long)dw + 4};
    const int o = j.orientation, OW = o >= 5 ? H : W;
    for (int y = 0; y < H; y++) {
        const uint8_t *yr = plane[0] + (long)y * j.bw[0] * 8;
        for (int c = 1; c < j.nc; c++) {
            const long pw = (long)j.bw[c] * 8;
            uint8_t *u = urow[c - 1];
            if (j.hmax == 1) memcpy(u, plane[c] + (long)y * pw, (size_t)W);
            else if (j.vmax == 1) {
                const uint8_t *in = plane[c] + (long)y * pw;
                if (dw > 2) h2v1_fancy_row(in, dw, u);
                else for (int x = 0; x < 2 * dw; x++) u[x] = in[x >> 1];
            } else {
                const int cy = y >> 1;
                const uint8_t *near = plane[c] + (long)cy * pw;
                if (dw > 2) {
                    int fy = (y & 1) ? cy + 1 : cy - 1;                           /* context row; the edge rows are duplicated */
                    if (fy < 0) fy = 0;
                    if (fy > dh - 1) fy = dh - 1;
                    h2v2_fancy_row(near, plane[c] + (long)fy * pw, dw, u);
                } else for (int x = 0; x < 2 * dw; x++) u[x] = near[x >> 1];
            }
        }
        for (int x = 0; x < W; x++) {
            int r, g, b;
            if (j.nc == 1) r = g = b = yr[x];
            else {
                const int Y = yr[x], cb = urow[0][x] - 128, cr = urow[1][x] - 128;
                r = clamp255(Y + (int)((91881L * cr + 32768) >> 16));
                b = clamp255(Y + (int)((116130L * cb + 32768) >> 16));
                g = clamp255(Y + (int)((-22554L * cb + 32768 - 46802L * cr) >> 16));
            }
            int ox, oy;
            switch (o) {
            case 2: ox = W - 1 - x; oy = y; break;
            case 3: ox = W - 1 - x; oy = H - 1 - y; break;
            case 4: ox = x; oy = H - 1 - y; break;
            case 5: ox = y; oy = x; break;
            case 6: ox = H - 1 - y; oy = x; break;
            case 7: ox = H - 1 - y; oy = W - 1 - x; break;
            case 8: ox = y; oy = W - 1 - x; break;
            default: ox = x; oy = y;
            }
            uint8_t *px = bgr + ((long)oy * OW + ox) * 3;
            px[0] = (uint8_t)b; px[1] = (uint8_t)g; px[2] = (uint8_t)r;
        }
    }
    free(rows);
    for (int c = 0; c < j.nc; c++) free(plane[c]);
    return 0;
}
