// Host-side grid corner search -- the part of the hot path that stays on the CPU (north_star;
// reference cv/grid.py:16-71).  The reference gets this from cv2 (findContours RETR_EXTERNAL /
// CHAIN_APPROX_SIMPLE, contourArea, arcLength, approxPolyDP); OpenCV is not a dependency here, so this
// file implements the same published algorithms: Suzuki-Abe outer-border following on a zero-padded
// 0/1 image, the shoelace area in double over float vertices, float-sqrt perimeter, and OpenCV's
// Douglas-Peucker variant for closed curves (farthest-point start search, explicit stack, collinear
// clean-up).
//
// Built for throughput: the raster scan skips runs 8 bytes at a time, small contours are rejected by
// bounding box before any area/approximation work, and a batch entry point spreads frames over threads.
#include <algorithm>
#include <immintrin.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "sv_internal.h"
#include "host_pool.h"

namespace {

struct Pt { int x, y; };
typedef uint64_t __attribute__((may_alias)) u64a;   // 64-bit view of words stored as uint32_t

const int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

// Labels written into the work image while following borders (OpenCV's convention).
constexpr int8_t kTraced = 2;                  // border pixel
constexpr int8_t kTracedRight = (int8_t)0x82;  // border pixel the border leaves to the right (negative)

struct Contour { size_t first, count; int x0, y0, x1, y1; };  // slice of `points` + bounding box

class BorderScanner {
  public:
    BorderScanner(const u8 *bin, int H, int W, ptrdiff_t pitch) : h_(H), w_(W), step_(W + 2), own_((size_t)(W + 2) * (H + 2), 0)
    {
        base_ = own_.data();
        for (int y = 0; y < H; y++) {
            const u8 *s = bin + (ptrdiff_t)y * pitch;
            int8_t *d = base_ + (size_t)(y + 1) * step_ + 1;
            for (int x = 0; x < W; x++) d[x] = s[x] != 0;
        }
        for (int k = 0; k < 8; k++) delta_[k] = delta_[k + 8] = kDy[k] * step_ + kDx[k];
    }

    // from a bit-packed image: 1 bit per pixel, LSB = leftmost, W/32 words per row (sv_despeckle_u8's packed output).
    // The label image lives in a per-thread buffer that is reused from frame to frame (a fresh 2 MB allocation costs more
    // in page faults than the whole search); only the byte spans the previous frame wrote are cleared.
    BorderScanner(const uint32_t *bits, int H, int W) : h_(H), w_(W), step_(W + 2), bits_(bits), wpr_(W >> 5)
    {
        ScanBuffer &tl = scan_buffer();
        if (tl.h != H || tl.w != W) {
            tl.img.assign((size_t)(W + 2) * (H + 2), 0);
            tl.span.assign((size_t)H + 2, {0, 0});
            tl.h = H; tl.w = W;
        } else {
            for (int y = 1; y <= H; y++) {
                auto &sp = tl.span[y];
                if (sp.second > sp.first) memset(tl.img.data() + (size_t)y * step_ + sp.first, 0, (size_t)(sp.second - sp.first));
                sp = {0, 0};
            }
        }
        base_ = tl.img.data();
        span_ = tl.span.data();
        for (int y = 0; y < H; y++) {
            const uint32_t *s = bits + (size_t)y * wpr_;
            int8_t *d = base_ + (size_t)(y + 1) * step_ + 1;
            int lo = -1, hi = -1;
            for (int k = 0; k < wpr_; k++) {
                uint32_t v = s[k];
                if (!v) continue;                              // rows are mostly empty after despeckling
                if (lo < 0) lo = k;
                hi = k;
                for (int b = 0; b < 4; b++, v >>= 8) {
                    // 8 bits -> 8 bytes of 0/1
                    uint64_t x = (uint64_t)(v & 0xFF) * 0x0101010101010101ull & 0x8040201008040201ull;
                    x = ((x + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
                    memcpy(d + 32 * k + 8 * b, &x, 8);
                }
            }
            if (lo >= 0) span_[y + 1] = {1 + 32 * lo, 1 + 32 * (hi + 1)};
        }
        for (int k = 0; k < 8; k++) delta_[k] = delta_[k + 8] = kDy[k] * step_ + kDx[k];
    }

    // Raster scan; every outer border that is not enclosed by an already-followed one is traced.
    void run()
    {
        for (int y = 1; y <= h_; y++) {
            if (bits_ && span_[y].second == 0) continue;       // nothing on this row
            int8_t *row = base_ + (size_t)y * step_;
            int last_marked = 0;  // x of the last labelled pixel seen on this row (0 = padding column)
            int prev = 0;
            int x = 1;
            while (x <= w_) {
                // background is never labelled, so from a background pixel the next non-zero label is the next set bit
                x = (bits_ && prev == 0) ? next_set(bits_ + (size_t)(y - 1) * wpr_, x) : skip_run(row, x, prev);
                if (x > w_) break;
                int p = row[x];
                if (prev == 0 && p == 1 && row[last_marked] <= 0) {
                    last_marked = x;
                    follow(row + x, x - 1, y - 1);
                    p = row[x];
                }
                prev = p;
                if (prev & ~1) {
                    last_marked = x;
                    if (prev > 0) {
                        // Inside a followed outer border (its mark here is positive): until the scan passes a right-exit mark
                        // (negative) `last_marked` keeps pointing at a positive label and no border can start, whatever lies in
                        // between -- the lines and digits inside the grid.  Go straight to the next negative label of the row.
                        x = next_negative(row, x + 1);
                        if (x > w_) break;
                        prev = row[x];
                        last_marked = x;
                    }
                }
                x++;
            }
        }
    }

    std::vector<Pt> points;
    std::vector<Contour> contours;  // in discovery order (cv2 reports them reversed)

  private:
    // first x' >= x (1-based) whose pixel is set in the bit row, or w_+1
    int next_set(const uint32_t *brow, int x) const
    {
        const int px = x - 1;
        int k = px >> 5;
        uint32_t v = brow[k] >> (px & 31);
        if (v) return x + __builtin_ctz(v);
        for (k++; k < wpr_; k++)
            if (brow[k]) return 32 * k + __builtin_ctz(brow[k]) + 1;
        return w_ + 1;
    }

    // first x' >= x whose label is negative (a right-exit mark), or w_+1
    int next_negative(const int8_t *row, int x) const
    {
        while (x + 8 <= w_ + 1) {
            uint64_t v;
            memcpy(&v, row + x, 8);
            v &= 0x8080808080808080ull;
            if (v) return x + (__builtin_ctzll(v) >> 3);
            x += 8;
        }
        while (x <= w_ && row[x] >= 0) x++;
        return x;
    }

    // first x' >= x with row[x'] != prev (or w_+1)
    int skip_run(const int8_t *row, int x, int prev) const
    {
        const uint64_t pat = 0x0101010101010101ull * (uint8_t)prev;
        while (x + 8 <= w_ + 1) {
            uint64_t v;
            memcpy(&v, row + x, 8);
            if (v != pat) break;
            x += 8;
        }
        while (x <= w_ && row[x] == prev) x++;
        return x;
    }

    // Follows one outer border starting at its top-left pixel; keeps the vertices where the direction changes.
    void follow(int8_t *start, int px, int py)
    {
        Contour c{points.size(), 0, px, py, px, py};
        int s = 4;
        const int s_stop = 4;
        int8_t *second;
        do {
            s = (s - 1) & 7;
            second = start + delta_[s];
        } while (*second == 0 && s != s_stop);
        if (s == s_stop) {  // isolated pixel
            *start = kTracedRight;
            points.push_back({px, py});
        } else {
            int8_t *cur = start, *next = nullptr;
            int prev_dir = s ^ 4;
            for (;;) {
                const int s_from = s;
                while (s < 15) {
                    next = cur + delta_[++s];
                    if (*next != 0) break;
                }
                s &= 7;
                if ((unsigned)(s - 1) < (unsigned)s_from) *cur = kTracedRight;
                else if (*cur == 1) *cur = kTraced;
                if (s != prev_dir) {
                    points.push_back({px, py});
                    c.x0 = std::min(c.x0, px); c.x1 = std::max(c.x1, px);
                    c.y0 = std::min(c.y0, py); c.y1 = std::max(c.y1, py);
                    prev_dir = s;
                }
                px += kDx[s];
                py += kDy[s];
                if (next == start && cur == second) break;
                cur = next;
                s = (s + 4) & 7;
            }
        }
        c.count = points.size() - c.first;
        contours.push_back(c);
    }

    struct ScanBuffer { std::vector<int8_t> img; std::vector<std::pair<int, int>> span; int h = 0, w = 0; };
    static ScanBuffer &scan_buffer() { static thread_local ScanBuffer b; return b; }

    int h_, w_, step_;
    std::vector<int8_t> own_;
    int8_t *base_ = nullptr;
    const uint32_t *bits_ = nullptr;
    int wpr_ = 0;
    std::pair<int, int> *span_ = nullptr;
    int delta_[16];
};

// The same scan on a bit-packed image (1 bit per pixel, LSB = leftmost, W/32 words per row: sv_despeckle_u8's packed output), without
// ever expanding it to bytes -- building and clearing a byte-per-pixel label image was half of the search's time.  A label is two
// bits in two side planes: T = "this pixel is on a followed border", R = "... and the border leaves it to the right" (the byte
// scanner's 2 and 0x82); an unlabelled set pixel is the byte scanner's 1.  The raster scan only ever has to look at the pixels
// where its state can change: labelled pixels, and unlabelled pixels that start a run (set, left neighbour clear).  A border may
// start at such a run start iff the last label passed on the row was a right-exit (or there was none); once a positive label has been
// passed nothing can start before the next right-exit, so the scan jumps there -- the lines and digits inside a followed grid
// border cost nothing.  Same borders, same points, same order as BorderScanner (tests/test_host_contours.py compares them).
//
// Layout (round 3): image and label planes live in per-thread buffers with one zero word left and right of every row and one zero
// row above and below, so neither the raster scan nor the border following ever tests a coordinate against the frame.  Following a
// border is table-driven: the 3x3 neighbourhood of the current pixel is three unaligned loads, and (neighbourhood, direction we came
// from) -> (next direction, "the right neighbour was examined and is clear") is one look-up in a 4-KB table built from the very loop
// BorderScanner::follow runs.  Per row, 64-bit masks say which words hold set pixels (from the sparse record, or built while the
// dense image is copied in) and which hold right-exit labels, so both the run-start search and the jump across a followed interior
// step from one interesting word to the next.
// Growable array without value-initialisation: the border follower appends with "store, then advance the count by 0 or 1", which needs
// room behind the count that costs nothing to provide.
template <class T>
struct RawBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { free(p); }
    size_t size() const { return n; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    const T *begin() const { return p; }
    const T *end() const { return p + n; }
    void clear() { n = 0; }
    void room(size_t extra)                     // p[n .. n + extra) may be written afterwards
    {
        if (n + extra <= cap) return;
        size_t c = cap ? cap : 1024;
        while (c < n + extra) c *= 2;
        T *q = static_cast<T *>(realloc(p, c * sizeof(T)));
        if (!q) throw std::bad_alloc();
        p = q;
        cap = c;
    }
    void push_back(const T &v) { room(1); p[n++] = v; }
};

struct BitScratch {
    static constexpr int PADY = 2;              // zero rows above and below the image (the follower keeps rows y-2 .. y+2 in registers)
    std::vector<uint32_t> img;                  // [(H + 2 PADY) * (wpr + 2)] (+ slack for the vector expansion's full-group stores)
    // Labels, one block of lab_row words per padded row (H + 2 rows): first the row's right-exit masks (gpr 64-bit words: which label
    // words of the row hold right-exit labels), then the two label planes interleaved (pair k: T word, R word of padded word k), so that
    // the follower addresses everything it writes for a pixel from one row pointer and a mark touches one cache line (two at most).
    std::vector<uint32_t> lab;
    std::vector<uint64_t> bmask;                // [H * gpr]: words of the row with set pixels
    struct Touched { uint32_t pair, rmask; };   // offsets into lab of a label pair / a right-exit mask written by the current frame's borders
    RawBuf<Touched> touched;
    RawBuf<Pt> points;
    std::vector<Contour> contours;
    int h = 0, wpr = 0, gpr = 0, stride = 0, lab_row = 0;
    bool dirty_img = false;

    static BitScratch &get() { static thread_local BitScratch s; return s; }

    // ready for a frame of this shape: image all zero, labels all zero, no points
    // Invariant between frames: img is non-zero exactly in the words bmask names.  keep_img: the caller replaces the old frame by the new
    // one chunk by chunk itself (the vector expansion, which needs the old masks for that); otherwise the image is zeroed here
    void begin(int H, int W, bool keep_img = false)
    {
        const int w = W >> 5;
        if (h != H || wpr != w) {               // (H, words per row) -- not their product: 480x640 and 640x480 share a word count
            h = H; wpr = w; gpr = (w + 63) / 64; stride = w + 2;
            img.assign((size_t)(H + 2 * PADY) * stride + 64, 0u);
            lab_row = 2 * gpr + 2 * stride;
            lab.assign((size_t)(H + 2) * lab_row, 0u);
            bmask.assign((size_t)H * gpr, 0ull);
            touched.clear();
            dirty_img = false;
        }
        if (dirty_img && !keep_img) memset(img.data(), 0, img.size() * 4);
        for (const Touched &x : touched) { lab[x.pair] = 0; lab[x.pair + 1] = 0; lab[x.rmask] = 0; lab[x.rmask + 1] = 0; }
        touched.clear();
        points.clear();
        contours.clear();
        dirty_img = true;
    }
    uint32_t *row(int y) { return img.data() + (size_t)(y + PADY) * stride + 1; }   // word 0 of image row y
    uint32_t *label_pairs(int y) { return lab.data() + (size_t)(y + 1) * lab_row + 2 * gpr; }   // pair of padded word 0 of row y (image word k = pair k + 1)

    // dense bit image (H rows of wpr words) -> padded image + row masks
    void load_dense(const uint32_t *bits)
    {
        for (int y = 0; y < h; y++) {
            const uint32_t *s = bits + (size_t)y * wpr;
            uint32_t *d = row(y);
            memcpy(d, s, (size_t)wpr * 4);
            for (int g = 0; g < gpr; g++) {
                uint64_t m = 0;
                const int k1 = std::min(wpr, 64 * g + 64);
                for (int k = 64 * g; k < k1; k++) m |= (uint64_t)(s[k] != 0) << (k & 63);
                bmask[(size_t)y * gpr + g] = m;
            }
        }
    }
};

// (3x3 neighbourhood, direction of the previous border pixel) -> the step to the next border pixel, packed so that nothing on the
// follower's latency chain has to be derived from the direction: bits 0-2 the direction back from the next pixel (the index of its own
// look-up), bit 3 "the right neighbour was examined and is clear", bit 4 "the border turns here" (the pixel is a vertex of the compressed
// chain), bits 8-9 dx + 1, bits 10-11 dy + 1; and the neighbourhood as an 8-bit mask over the directions of kDx/kDy.  code9 = row above | row << 3 | row below << 6, each 3 bits (x-1, x, x+1).
struct FollowTable {
    uint16_t next[512][8];
    uint8_t nb8[512];
    FollowTable()
    {
        for (int c = 0; c < 512; c++) {
            const int up = c & 7, mid = (c >> 3) & 7, dn = c >> 6;
            const int m = ((mid >> 2) & 1) | ((up >> 2) & 1) << 1 | ((up >> 1) & 1) << 2 | (up & 1) << 3 | (mid & 1) << 4 | (dn & 1) << 5 | ((dn >> 1) & 1) << 6 |
                          ((dn >> 2) & 1) << 7;
            nb8[c] = (uint8_t)m;
            for (int s_from = 0; s_from < 8; s_from++) {
                int s = s_from;
                while (s < 15) {                                   // BorderScanner::follow's search, counter-clockwise from the previous pixel
                    ++s;
                    if ((m >> (s & 7)) & 1) break;
                }
                s &= 7;
                const bool right = (unsigned)(s - 1) < (unsigned)s_from;
                next[c][s_from] = (uint16_t)((s ^ 4) | (right ? 8 : 0) | (s != (s_from ^ 4) ? 16 : 0) | (kDx[s] + 1) << 8 | (kDy[s] + 1) << 10);
            }
        }
    }
};
const FollowTable kFollow;

class BitScanner {
  public:
    // the scratch must hold the frame already (BitScratch::begin + load_dense, or sparse_expand)
    BitScanner(BitScratch &sc, int H, int W) : sc_(sc), h_(H), wpr_(W >> 5), gpr_(sc.gpr), stride_(sc.stride), img_(sc.img.data()), lab_(sc.lab.data()),
                                               points(sc.points), contours(sc.contours) {}

    void run()
    {
        for (int y = 0; y < h_; y++) {
            const uint64_t *bm = sc_.bmask.data() + (size_t)y * gpr_;
            // first word >= k with a set pixel (labels only ever sit on set pixels), or wpr_
            auto next_word = [&](int k) -> int {
                while (k < wpr_) {
                    const uint64_t m = bm[k >> 6] >> (k & 63);
                    if (m) return k + __builtin_ctzll(m);
                    k = (k | 63) + 1;
                }
                return wpr_;
            };
            int k = next_word(0);
            if (k >= wpr_) continue;                               // empty row
            const uint32_t *brow = img_ + (size_t)(y + BitScratch::PADY) * stride_ + 1;
            const uint32_t *lrow = sc_.label_pairs(y) + 2;                       // lrow[2k] = T word k, lrow[2k + 1] = R word k
            int x = 32 * k;
            for (;;) {
                // next pixel >= x that is labelled or starts a run (brow[-1] is the zero pad word)
                int pos = -1;
                k = x >> 5;
                if (k < wpr_) {
                    const uint32_t bw = brow[k];
                    const uint32_t e = (lrow[2 * k] | (bw & ~((bw << 1) | (brow[k - 1] >> 31)))) & (~0u << (x & 31));
                    if (e) pos = 32 * k + __builtin_ctz(e);
                    else
                        for (k = next_word(k + 1); k < wpr_; k = next_word(k + 1)) {
                            const uint32_t b2 = brow[k];
                            const uint32_t e2 = lrow[2 * k] | (b2 & ~((b2 << 1) | (brow[k - 1] >> 31)));
                            if (e2) { pos = 32 * k + __builtin_ctz(e2); break; }
                        }
                }
                if (pos < 0) break;
                const uint32_t m = 1u << (pos & 31);
                if (!(lrow[2 * (pos >> 5)] & m)) follow(pos, y);         // an unlabelled run start with no positive label pending: a new outer border
                x = pos + 1;
                if (!(lrow[2 * (pos >> 5) + 1] & m)) {               // positive label: nothing can start before the next right-exit label of the row
                    const u64a *rm = reinterpret_cast<const u64a *>(sc_.label_pairs(y)) - gpr_;
                    int nx = -1;
                    for (int kk = x >> 5; kk < wpr_;) {
                        uint64_t mm = rm[kk >> 6] >> (kk & 63);
                        if (!mm) { kk = (kk | 63) + 1; continue; }
                        kk += __builtin_ctzll(mm);
                        uint32_t e = lrow[2 * kk + 1];
                        if (kk == (x >> 5)) e &= ~0u << (x & 31);
                        if (e) { nx = 32 * kk + __builtin_ctz(e); break; }
                        kk++;
                    }
                    if (nx < 0) break;
                    x = nx + 1;
                }
            }
        }
    }

  private:
    // BorderScanner::follow on the bit planes (x, y are image coordinates; the byte scanner's are the same minus its padding).
    // The loop's latency chain is neighbourhood -> table -> direction -> position -> neighbourhood, so the neighbourhood comes out of
    // registers: five 64-bit windows hold bits [8 byte0, 8 byte0 + 64) of padded rows y-2 .. y+2.  A vertical step shifts the windows by
    // one row and loads the new outer one (needed two steps later at the earliest); a horizontal step moves the bit position, and only
    // when it leaves [1, 62] are the windows reloaded around it.  Everything else in the step is branch-free: the label planes, the
    // touched list and the vertex list are written every time and change / advance only when they should.
    void follow(int sx, int sy)
    {
        static_assert(sizeof(Pt) == 8, "a point is stored as one 64-bit word: x low, y high");
        Contour c{points.size(), 0, sx, sy, sx, sy};
        const ptrdiff_t rb = (ptrdiff_t)stride_ * 4, lrw = sc_.lab_row;
        const uint8_t *img8 = reinterpret_cast<const uint8_t *>(img_);
        uint32_t *lr = sc_.label_pairs(sy);
        const uint8_t *wp;
        uint64_t w0, w1, w2, w3, w4;
        int pb;
        auto load_windows = [&](uint32_t x, uint32_t y) {
            const unsigned bit = x + 32u;                              // bit index of pixel x in the padded row (>= 32)
            const unsigned byte0 = (bit >> 3) - 4u;                    // window = bytes [byte0, byte0 + 8): inside the row for every x in [0, W)
            pb = (int)(bit - 8u * byte0);                              // 32 .. 39
            wp = img8 + (ptrdiff_t)(y + BitScratch::PADY) * rb + byte0;
            memcpy(&w0, wp - 2 * rb, 8);
            memcpy(&w1, wp - rb, 8);
            memcpy(&w2, wp, 8);
            memcpy(&w3, wp + rb, 8);
            memcpy(&w4, wp + 2 * rb, 8);
        };
        auto code9 = [&]() -> unsigned {
            const unsigned sh = (unsigned)pb - 1u;
            return (unsigned)((w1 >> sh) & 7) | (unsigned)(((w2 >> sh) & 7) << 3) | (unsigned)(((w3 >> sh) & 7) << 6);
        };
        // labels pixel x of the row whose label pairs start at lr: T always, R and the row's right-exit mask if `right`
        auto &touched = sc_.touched;
        BitScratch::Touched *tp = touched.p;
        size_t tn = touched.n;
        auto mark = [&](uint32_t x, uint32_t right) {
            if (tn == touched.cap) { touched.n = tn; touched.room(1); tp = touched.p; }
            uint32_t *w = lr + 2 * ((x >> 5) + 1);
            u64a *g = reinterpret_cast<u64a *>(lr) - gpr_ + (x >> 11);
            const uint32_t m = 1u << (x & 31);
            tp[tn] = {(uint32_t)(w - lab_), (uint32_t)(reinterpret_cast<uint32_t *>(g) - lab_)};
            tn += w[0] == 0;
            w[0] |= m;
            w[1] |= (0u - right) & m;
            *g |= (uint64_t)right << ((x >> 5) & 63);
        };
        load_windows((uint32_t)sx, (uint32_t)sy);
        const unsigned nb0 = kFollow.nb8[code9()];
        int s = 4;
        do {
            s = (s - 1) & 7;
        } while (!((nb0 >> s) & 1) && s != 4);
        if (s == 4) {  // isolated pixel (the start pixel's left neighbour is clear: it starts a run)
            mark((uint32_t)sx, 1u);
            points.push_back({sx, sy});
        } else {
            auto pack = [](int x, int y) { return (uint64_t)(uint32_t)y << 32 | (uint32_t)x; };
            const uint64_t start = pack(sx, sy), second = pack(sx + kDx[s], sy + kDy[s]);
            uint64_t pos = start;                                      // y << 32 | x: one add per step, one store per vertex
            unsigned from = (unsigned)s;                               // direction (from the current pixel) of the pixel we came from
            Pt *pp = points.p;
            size_t np = points.n;
            for (;;) {
                if (np == points.cap) { points.n = np; points.room(1); pp = points.p; }
                const unsigned e = kFollow.next[code9()][from];
                from = e & 7;
                mark((uint32_t)pos, (e >> 3) & 1u);
                memcpy(pp + np, &pos, 8);                              // kept iff the border turns here
                np += (e >> 4) & 1u;
                const int dx = (int)((e >> 8) & 3u) - 1, dy = (int)(e >> 10) - 1;
                const uint64_t npos = pos + (uint64_t)(int64_t)dx + ((uint64_t)(int64_t)dy << 32);
                if (npos == start && pos == second) break;
                pos = npos;
                pb += dx;
                if (dy > 0) {
                    w0 = w1; w1 = w2; w2 = w3; w3 = w4;
                    wp += rb; lr += lrw;
                    memcpy(&w4, wp + 2 * rb, 8);
                } else if (dy < 0) {
                    w4 = w3; w3 = w2; w2 = w1; w1 = w0;
                    wp -= rb; lr -= lrw;
                    memcpy(&w0, wp - 2 * rb, 8);
                }
                if ((unsigned)(pb - 1) > 61u) load_windows((uint32_t)pos, (uint32_t)(pos >> 32));
            }
            points.n = np;
            for (size_t i = c.first; i < np; i++) {
                const Pt q = pp[i];
                c.x0 = std::min(c.x0, q.x); c.x1 = std::max(c.x1, q.x);
                c.y0 = std::min(c.y0, q.y); c.y1 = std::max(c.y1, q.y);
            }
        }
        touched.n = tn;
        c.count = points.size() - c.first;
        contours.push_back(c);
    }

    BitScratch &sc_;
    int h_, wpr_, gpr_, stride_;
    const uint32_t *img_;
    uint32_t *lab_;

  public:
    RawBuf<Pt> &points;
    std::vector<Contour> &contours;  // in discovery order (cv2 reports them reversed)
};

double contour_area(const Pt *p, size_t n)
{
    if (n == 0) return 0.0;
    double a = 0.0;
    float px = (float)p[n - 1].x, py = (float)p[n - 1].y;
    for (size_t i = 0; i < n; i++) {
        const float x = (float)p[i].x, y = (float)p[i].y;
        a += (double)px * y - (double)py * x;
        px = x;
        py = y;
    }
    return std::fabs(a * 0.5);
}

double arc_length(const Pt *p, size_t n, bool closed)
{
    if (n <= 1) return 0.0;
    double len = 0.0;
    const size_t last = closed ? n - 1 : 0;
    float px = (float)p[last].x, py = (float)p[last].y;
    for (size_t i = 0; i < n; i++) {
        const float x = (float)p[i].x, y = (float)p[i].y, dx = x - px, dy = y - py;
        len += std::sqrt(dx * dx + dy * dy);
        px = x;
        py = y;
    }
    return len;
}

// OpenCV's approxPolyDP for integer points.
std::vector<Pt> approx_poly(const Pt *src, int count, double eps, bool closed_in)
{
    std::vector<Pt> dst;
    if (count == 0) return dst;
    struct Range { int start, end; };
    std::vector<Range> stack;
    auto next = [&](int &pos) { Pt q = src[pos]; if (++pos >= count) pos = 0; return q; };
    bool closed = closed_in, le_eps = false;
    int init_iters = 3, pos = 0;
    Range slice{0, 0}, right{0, 0};
    Pt start{-1000000, -1000000}, end{0, 0}, pt{0, 0};
    eps *= eps;
    if (!closed) {
        right.start = count;
        end = src[0];
        start = src[count - 1];
        if (start.x != end.x || start.y != end.y) {
            stack.push_back({0, count - 1});
        } else {
            closed = true;
            init_iters = 1;
        }
    }
    if (closed) {
        right.start = 0;
        for (int it = 0; it < init_iters; it++) {  // approximately the two farthest points
            double max_dist = 0;
            pos = (pos + right.start) % count;
            start = next(pos);
            for (int j = 1; j < count; j++) {
                pt = next(pos);
                const double dx = pt.x - start.x, dy = pt.y - start.y, dist = dx * dx + dy * dy;
                if (dist > max_dist) { max_dist = dist; right.start = j; }
            }
            le_eps = max_dist <= eps;
        }
        if (!le_eps) {
            right.end = slice.start = pos % count;
            slice.end = right.start = (right.start + slice.start) % count;
            stack.push_back(right);
            stack.push_back(slice);
        } else {
            dst.push_back(start);
        }
    }
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        end = src[slice.end];
        pos = slice.start;
        start = next(pos);
        if (pos != slice.end) {
            const double dx = end.x - start.x, dy = end.y - start.y;
            double max_dist = 0;
            while (pos != slice.end) {
                pt = next(pos);
                const double dist = std::fabs((pt.y - start.y) * dx - (pt.x - start.x) * dy);
                if (dist > max_dist) { max_dist = dist; right.start = (pos + count - 1) % count; }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
            start = src[slice.start];
        }
        if (le_eps) {
            dst.push_back(start);
        } else {
            right.end = slice.end;
            slice.end = right.start;
            stack.push_back(right);
            stack.push_back(slice);
        }
    }
    if (!closed) dst.push_back(src[count - 1]);

    // remove vertices on (almost) straight runs
    const int n0 = (int)dst.size();
    int n = n0, wpos;
    auto nextd = [&](int &q) { Pt r = dst[q]; if (++q >= n0) q = 0; return r; };
    pos = closed_in ? n0 - 1 : 0;
    start = nextd(pos);
    wpos = pos;
    pt = nextd(pos);
    const int open = closed_in ? 0 : 1;
    for (int i = open; i < n0 - open && n > 2; i++) {
        end = nextd(pos);
        const double dx = end.x - start.x, dy = end.y - start.y;
        const double dist = std::fabs((pt.x - start.x) * dy - (pt.y - start.y) * dx);
        const double inner = (double)(pt.x - start.x) * (end.x - pt.x) + (double)(pt.y - start.y) * (end.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && inner >= 0) {
            n--;
            dst[wpos] = start = end;
            if (++wpos >= n0) wpos = 0;
            pt = nextd(pos);
            i++;
            continue;
        }
        dst[wpos] = start = pt;
        if (++wpos >= n0) wpos = 0;
        pt = end;
    }
    if (!closed_in) dst[wpos] = pt;
    dst.resize(n);
    return dst;
}

template <class Scanner>
bool grid_corners_from(Scanner &sc, int H, int W, double min_area_ratio, double eps_ratio, int *out8);

// find_grid_contour, cv/grid.py:37-71
bool grid_corners(const u8 *bin, int H, int W, ptrdiff_t pitch, double min_area_ratio, double eps_ratio, int *out8)
{
    BorderScanner sc(bin, H, W, pitch);
    return grid_corners_from(sc, H, W, min_area_ratio, eps_ratio, out8);
}

template <class Scanner>
bool grid_corners_from(Scanner &sc, int H, int W, double min_area_ratio, double eps_ratio, int *out8)
{
    sc.run();
    const double min_area = min_area_ratio * ((double)H * (double)W);
    struct Cand { double area; size_t idx; };
    std::vector<Cand> cand;
    // cv2 lists contours last-found-first; the reference's stable descending sort keeps that order for ties
    for (size_t k = sc.contours.size(); k-- > 0;) {
        const Contour &c = sc.contours[k];
        if ((double)(c.x1 - c.x0) * (double)(c.y1 - c.y0) < min_area) continue;  // area <= bounding box
        const double a = contour_area(sc.points.data() + c.first, c.count);
        if (a >= min_area) cand.push_back({a, k});
    }
    std::stable_sort(cand.begin(), cand.end(), [](const Cand &a, const Cand &b) { return a.area > b.area; });
    for (const Cand &cd : cand) {
        const Contour &c = sc.contours[cd.idx];
        const Pt *p = sc.points.data() + c.first;
        const double eps = eps_ratio * arc_length(p, c.count, true);
        const std::vector<Pt> poly = approx_poly(p, (int)c.count, eps, true);
        if (poly.size() == 4) {
            for (int i = 0; i < 4; i++) { out8[2 * i] = poly[i].x; out8[2 * i + 1] = poly[i].y; }
            return true;
        }
    }
    return false;
}

}  // namespace

// ---- C ABI -------------------------------------------------------------------------------------------
extern "C" int sv_find_grid_corners_u8(const uint8_t *binary, int H, int W, ptrdiff_t pitch, double min_area_ratio,
                                       double epsilon_ratio, int *corners)
{
    if (!binary || !corners || H <= 0 || W <= 0 || pitch < W) return sv_fail(SV_ERR_BAD_ARG, "sv_find_grid_corners_u8: bad argument");
    return grid_corners(binary, H, W, pitch, min_area_ratio, epsilon_ratio, corners) ? 1 : 0;
}

extern "C" int sv_find_grid_corners_batch_u8(const uint8_t *binary, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride,
                                             double min_area_ratio, double epsilon_ratio, int *corners, uint8_t *found, int threads)
{
    if (!binary || !corners || !found || n <= 0 || H <= 0 || W <= 0 || pitch < W) return sv_fail(SV_ERR_BAD_ARG, "sv_find_grid_corners_batch_u8: bad argument");
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    WorkerPool::instance().parallel_for(n, threads, [&](int i) {
        found[i] = grid_corners(binary + (ptrdiff_t)i * img_stride, H, W, pitch, min_area_ratio, epsilon_ratio, corners + 8 * i) ? 1 : 0;
    });
    return SV_OK;
}

extern "C" int sv_find_grid_corners_bits_batch(const uint32_t *bits, int n, int H, int W, double min_area_ratio, double epsilon_ratio, int *corners,
                                               uint8_t *found, int threads)
{
    if (!bits || !corners || !found || n <= 0 || H <= 0 || W <= 0 || (W & 31)) return sv_fail(SV_ERR_BAD_ARG, "sv_find_grid_corners_bits_batch: bad argument");
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    WorkerPool::instance().parallel_for(n, threads, [&](int i) {
        BitScratch &scratch = BitScratch::get();
        scratch.begin(H, W);
        scratch.load_dense(bits + (size_t)i * H * (W >> 5));
        BitScanner sc(scratch, H, W);
        found[i] = grid_corners_from(sc, H, W, min_area_ratio, epsilon_ratio, corners + 8 * i) ? 1 : 0;
    });
    return SV_OK;
}

// Expansion of a validated record with AVX-512 expand-loads: 16 words per instruction, zero fill included, so the image needs no
// memset: the scratch image still holds the thread's previous frame exactly where `prev` (its row masks) has bits, and a 16-word chunk
// is rewritten iff the old or the new frame has a word in it.  A chunk of a row's last group may reach past the row: it lands, as
// zeros (mask bits beyond the row are validated to be clear), in pad words and in the first words of the next row(s), which are
// processed after it, and the scratch image has 64 words of slack behind its last row.  prev is replaced by the new masks.
__attribute__((target("avx512f,popcnt"))) static void sparse_expand_avx512(const uint64_t *masks, uint64_t *prev, const uint32_t *val, int H, int gpr, uint32_t *row0,
                                                                   int stride)
{
    for (int y = 0; y < H; y++) {
        uint32_t *row = row0 + (size_t)y * stride;
        for (int g = 0; g < gpr; g++) {
            const uint64_t m = masks[(size_t)y * gpr + g], either = m | prev[(size_t)y * gpr + g];
            prev[(size_t)y * gpr + g] = m;
            if (!either) continue;
            // where each chunk's values start comes from prefix counts of the mask, not from the chunk before it: the four expand-loads
            // of a group are independent, and the chain from group to group is one popcount
            const uint32_t *v0 = val, *v1 = v0 + __builtin_popcountll(m & 0xFFFFull), *v2 = v0 + __builtin_popcountll(m & 0xFFFFFFFFull),
                           *v3 = v0 + __builtin_popcountll(m & 0xFFFFFFFFFFFFull);
            val += __builtin_popcountll(m);
            uint32_t *d = row + 64 * g;
            if (either & 0xFFFFull) _mm512_storeu_si512(d, _mm512_maskz_expandloadu_epi32((__mmask16)m, v0));
            if (either & 0xFFFF0000ull) _mm512_storeu_si512(d + 16, _mm512_maskz_expandloadu_epi32((__mmask16)(m >> 16), v1));
            if (either & 0xFFFF00000000ull) _mm512_storeu_si512(d + 32, _mm512_maskz_expandloadu_epi32((__mmask16)(m >> 32), v2));
            if (either & 0xFFFF000000000000ull) _mm512_storeu_si512(d + 48, _mm512_maskz_expandloadu_epi32((__mmask16)(m >> 48), v3));
        }
    }
}

// number of values a record's row masks announce, or -1 if a mask has a bit beyond the last word of its row.  Compiled twice: built
// without -mpopcnt the popcount is fifteen instructions, and this loop runs over every mask of every frame.
__attribute__((always_inline)) static inline long count_record_values(const uint64_t *masks, int H, int wpr)
{
    const int gpr = (wpr + 63) / 64, tail = wpr & 63;            // tail: valid bits of a row's last mask word (0 = all 64)
    const uint64_t tail_bad = tail ? ~0ull << tail : 0ull;
    long total = 0;
    uint64_t bad = 0;
    for (int y = 0; y < H; y++) {
        for (int g = 0; g < gpr; g++) total += __builtin_popcountll(masks[(size_t)y * gpr + g]);
        bad |= masks[(size_t)y * gpr + gpr - 1] & tail_bad;
    }
    return bad ? -1 : total;
}
__attribute__((target("popcnt"))) static long count_record_values_popcnt(const uint64_t *masks, int H, int wpr) { return count_record_values(masks, H, wpr); }
static long count_record_values_generic(const uint64_t *masks, int H, int wpr) { return count_record_values(masks, H, wpr); }

// sparse record (include/sudoku_vision_hip.h, sv_pack_sparse_bits) -> dense bit image.  The record is validated before anything is
// written from it (it crosses PCIe and may be stale, torn or built for another shape): 1 = expanded, 0 = the record overflowed its
// capacity (n_values > cap_values: use the dense image), -1 = malformed (a mask bit beyond the row, more mask bits than values, or
// a capacity that does not fit `avail` bytes; avail < 0 = unknown, trust cap_values).
static int sparse_expand(const uint8_t *record, long avail, int H, int W, uint32_t *bits, BitScratch *scratch)
{
    const int wpr = W >> 5, gpr = (wpr + 63) / 64;
    const size_t head_bytes = 8 + 8 * (size_t)H * gpr;
    if (avail >= 0 && (size_t)avail < head_bytes) return -1;
    uint32_t head[2];
    memcpy(head, record, 8);
    if (avail >= 0 && head_bytes + 4 * (size_t)head[1] > (size_t)avail) return -1;
    if (head[0] > head[1]) return 0;
    const uint64_t *masks = reinterpret_cast<const uint64_t *>(record + 8);
    const uint32_t *val = reinterpret_cast<const uint32_t *>(record + head_bytes);
    static const bool has_popcnt = __builtin_cpu_supports("popcnt");
    const long total = has_popcnt ? count_record_values_popcnt(masks, H, wpr) : count_record_values_generic(masks, H, wpr);
    if (total < 0 || (size_t)total != head[0]) return -1;
    if (scratch) {                                               // the search's padded per-thread image, row masks taken over as they are
        static const bool vec = __builtin_cpu_supports("avx512f");
        scratch->begin(H, W, vec);
        if (vec) { sparse_expand_avx512(masks, scratch->bmask.data(), val, H, gpr, scratch->row(0), scratch->stride); return 1; }
        memcpy(scratch->bmask.data(), masks, 8 * (size_t)H * gpr);
    } else {
        memset(bits, 0, (size_t)H * wpr * 4);
    }
    for (int y = 0; y < H; y++)
        for (int g = 0; g < gpr; g++) {
            uint64_t m = masks[(size_t)y * gpr + g];
            uint32_t *row = (scratch ? scratch->row(y) : bits + (size_t)y * wpr) + 64 * g;
            while (m) { row[__builtin_ctzll(m)] = *val++; m &= m - 1; }
        }
    return 1;
}

extern "C" int sv_sparse_bits_expand(const uint8_t *record, int H, int W, uint32_t *bits)
{
    if (!record || !bits || H <= 0 || W <= 0 || (W & 31) || ((uintptr_t)record & 7)) return sv_fail(SV_ERR_BAD_ARG, "sv_sparse_bits_expand: bad argument");
    const int r = sparse_expand(record, -1, H, W, bits, nullptr);
    if (r == 0) return sv_fail(SV_ERR_BUFFER, "sv_sparse_bits_expand: the record overflowed its capacity");
    if (r < 0) return sv_fail(SV_ERR_BAD_ARG, "sv_sparse_bits_expand: malformed record (mask bits outside the row, or mask bits != n_values)");
    return SV_OK;
}

extern "C" int sv_find_grid_corners_sparse_batch(const uint8_t *records, long record_stride, int n, int H, int W, double min_area_ratio, double epsilon_ratio,
                                                 int *corners, uint8_t *found, int threads)
{
    if (!records || !corners || !found || n <= 0 || H <= 0 || W <= 0 || (W & 31) || (record_stride & 7) || ((uintptr_t)records & 7) ||
        record_stride < 8 + 8 * (long)H * ((((long)W >> 5) + 63) / 64))
        return sv_fail(SV_ERR_BAD_ARG, "sv_find_grid_corners_sparse_batch: bad argument");
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    WorkerPool::instance().parallel_for(n, threads, [&](int i) {
        BitScratch &scratch = BitScratch::get();
        // overflowed or malformed (stale / torn / foreign) record: the caller searches the dense image of that frame instead
        if (sparse_expand(records + (size_t)i * record_stride, record_stride, H, W, nullptr, &scratch) != 1) { found[i] = 2; return; }
        BitScanner sc(scratch, H, W);
        found[i] = grid_corners_from(sc, H, W, min_area_ratio, epsilon_ratio, corners + 8 * i) ? 1 : 0;
    });
    return SV_OK;
}

extern "C" int sv_host_pool_set_affinity(const int *cpus, int n)
{
    if (n < 0 || (n > 0 && !cpus)) return sv_fail(SV_ERR_BAD_ARG, "sv_host_pool_set_affinity: bad argument");
    WorkerPool::instance().set_affinity(cpus, n);
    return SV_OK;
}

extern "C" int sv_find_contours_u8(const uint8_t *binary, int H, int W, ptrdiff_t pitch, int *points, long cap_points, int *sizes,
                                   int cap_contours, long *n_points, int *n_contours)
{
    if (!binary || !n_points || !n_contours || H <= 0 || W <= 0 || pitch < W) return sv_fail(SV_ERR_BAD_ARG, "sv_find_contours_u8: bad argument");
    BorderScanner sc(binary, H, W, pitch);
    sc.run();
    *n_points = (long)sc.points.size();
    *n_contours = (int)sc.contours.size();
    if ((long)sc.points.size() > cap_points || (int)sc.contours.size() > cap_contours || !points || !sizes)
        return sv_fail(SV_ERR_BUFFER, "sv_find_contours_u8: need room for %ld points in %d contours", *n_points, *n_contours);
    long w = 0;
    int ci = 0;
    for (size_t k = sc.contours.size(); k-- > 0; ci++) {  // cv2 order: last found first
        const Contour &c = sc.contours[k];
        sizes[ci] = (int)c.count;
        for (size_t i = 0; i < c.count; i++) { points[2 * w] = sc.points[c.first + i].x; points[2 * w + 1] = sc.points[c.first + i].y; w++; }
    }
    return SV_OK;
}

// the same from a bit-packed image (BitScanner): what sv_find_grid_corners_bits_batch runs on, exposed so that tests can compare
// the two scanners contour by contour
extern "C" int sv_find_contours_bits(const uint32_t *bits, int H, int W, int *points, long cap_points, int *sizes, int cap_contours,
                                     long *n_points, int *n_contours)
{
    if (!bits || !n_points || !n_contours || H <= 0 || W <= 0 || (W & 31)) return sv_fail(SV_ERR_BAD_ARG, "sv_find_contours_bits: bad argument");
    BitScratch &scratch = BitScratch::get();
    scratch.begin(H, W);
    scratch.load_dense(bits);
    BitScanner sc(scratch, H, W);
    sc.run();
    *n_points = (long)sc.points.size();
    *n_contours = (int)sc.contours.size();
    if ((long)sc.points.size() > cap_points || (int)sc.contours.size() > cap_contours || !points || !sizes)
        return sv_fail(SV_ERR_BUFFER, "sv_find_contours_bits: need room for %ld points in %d contours", *n_points, *n_contours);
    long w = 0;
    int ci = 0;
    for (size_t k = sc.contours.size(); k-- > 0; ci++) {  // cv2 order: last found first
        const Contour &c = sc.contours[k];
        sizes[ci] = (int)c.count;
        for (size_t i = 0; i < c.count; i++) { points[2 * w] = sc.points[c.first + i].x; points[2 * w + 1] = sc.points[c.first + i].y; w++; }
    }
    return SV_OK;
}

extern "C" int sv_contour_area_i32(const int *xy, int n, double *area)
{
    if ((!xy && n > 0) || n < 0 || !area) return sv_fail(SV_ERR_BAD_ARG, "sv_contour_area_i32: bad argument");
    *area = contour_area(reinterpret_cast<const Pt *>(xy), (size_t)n);
    return SV_OK;
}

extern "C" int sv_arc_length_i32(const int *xy, int n, int closed, double *length)
{
    if ((!xy && n > 0) || n < 0 || !length) return sv_fail(SV_ERR_BAD_ARG, "sv_arc_length_i32: bad argument");
    *length = arc_length(reinterpret_cast<const Pt *>(xy), (size_t)n, closed != 0);
    return SV_OK;
}

extern "C" int sv_approx_poly_dp_i32(const int *xy, int n, double epsilon, int closed, int *out, int *n_out)
{
    if ((!xy && n > 0) || n < 0 || !out || !n_out || epsilon < 0) return sv_fail(SV_ERR_BAD_ARG, "sv_approx_poly_dp_i32: bad argument");
    const std::vector<Pt> r = approx_poly(reinterpret_cast<const Pt *>(xy), n, epsilon, closed != 0);
    *n_out = (int)r.size();
    for (size_t i = 0; i < r.size(); i++) { out[2 * i] = r[i].x; out[2 * i + 1] = r[i].y; }
    return SV_OK;
}
