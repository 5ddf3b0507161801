// Host-side fuzz of the corner search under AddressSanitizer + UBSan (host code only; the GPU pool has no GPU sanitizers):
//   hipcc -O1 -g -std=c++17 -ffp-contract=off -x hip --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
//         -I sudoku-vision_amd/csrc tools/dev/host_fuzz_asan.cpp -o /tmp/host_fuzz_asan -lpthread && /tmp/host_fuzz_asan
// Random images of many shapes and densities, as sequences on one thread: byte scanner vs bit scanner through the dense and the sparse
// entry (records packed here the way k_pack_sparse packs them), plus malformed records.
#include "../../sudoku-vision_amd/csrc/host_contours.cpp"
#include <random>
int sv_fail(int code, const char *, ...) { return code; }

static std::vector<uint8_t> pack(const std::vector<uint32_t> &bits, int H, int W, long cap, long &stride)
{
    const int wpr = W >> 5, gpr = (wpr + 63) / 64;
    stride = (8 + 8L * gpr * H + 4 * cap + 15) / 16 * 16;
    cap = (stride - 8 - 8L * gpr * H) / 4;
    std::vector<uint8_t> rec(stride + 64, 0xEE);
    uint8_t *r = rec.data();
    while ((uintptr_t)r & 7) r++;
    std::vector<uint8_t> out(r, r + stride);                      // (vector storage is 16-byte aligned)
    uint32_t n = 0;
    uint64_t *masks = (uint64_t *)(out.data() + 8);
    uint32_t *val = (uint32_t *)(out.data() + 8 + 8L * gpr * H);
    for (int y = 0; y < H; y++)
        for (int g = 0; g < gpr; g++) {
            uint64_t m = 0;
            for (int k = 64 * g; k < std::min(wpr, 64 * g + 64); k++)
                if (bits[(size_t)y * wpr + k]) { m |= 1ull << (k & 63); if (n < cap) val[n] = bits[(size_t)y * wpr + k]; n++; }
            masks[(size_t)y * gpr + g] = m;
        }
    ((uint32_t *)out.data())[0] = n;
    ((uint32_t *)out.data())[1] = (uint32_t)cap;
    return out;
}

int main()
{
    std::mt19937 rng(7);
    const int shapes[][2] = {{1, 32}, {3, 64}, {64, 64}, {65, 96}, {33, 2048}, {40, 2080}, {17, 4128}, {200, 320}, {480, 640}, {640, 480}, {1080, 1920}};
    long checked = 0;
    for (int rep = 0; rep < 3; rep++)
        for (auto &sh : shapes) {
            const int H = sh[0], W = sh[1], wpr = W >> 5;
            for (double density : {0.0, 0.001, 0.03, 0.3, 0.6, 1.0}) {
                std::vector<uint8_t> img((size_t)H * W);
                std::vector<uint32_t> bits((size_t)H * wpr, 0);
                std::bernoulli_distribution d(density);
                for (int y = 0; y < H; y++)
                    for (int x = 0; x < W; x++) {
                        bool v = d(rng);
                        if (density > 0.0 && density < 1.0 && H > 20 && W > 40 && (y == 4 || y == H - 5 || x == 6 || x == W - 7) && y >= 4 && y <= H - 5 && x >= 6 && x <= W - 7) v = true;
                        img[(size_t)y * W + x] = v ? 255 : 0;
                        if (v) bits[(size_t)y * wpr + (x >> 5)] |= 1u << (x & 31);
                    }
                int want[8] = {0}, got[8] = {0};
                uint8_t f = 9;
                const bool w = grid_corners(img.data(), H, W, W, 0.05, 0.02, want);
                if (sv_find_grid_corners_bits_batch(bits.data(), 1, H, W, 0.05, 0.02, got, &f, 1) != 0 || (f != 0) != w || (w && memcmp(want, got, sizeof want))) { printf("dense mismatch %dx%d %.3f\n", H, W, density); return 1; }
                long stride;
                for (long cap : {(long)H * wpr, (long)H * wpr / 3, 4L}) {
                    std::vector<uint8_t> rec = pack(bits, H, W, cap, stride);
                    f = 9;
                    memset(got, 0, sizeof got);
                    if (sv_find_grid_corners_sparse_batch(rec.data(), stride, 1, H, W, 0.05, 0.02, got, &f, 1) != 0) { printf("sparse rc %dx%d\n", H, W); return 1; }
                    const uint32_t n = ((uint32_t *)rec.data())[0], c = ((uint32_t *)rec.data())[1];
                    if (n > c) { if (f != 2) { printf("overflow not reported %dx%d\n", H, W); return 1; } continue; }
                    if ((f != 0) != w || f == 2 || (w && memcmp(want, got, sizeof want))) { printf("sparse mismatch %dx%d %.3f cap %ld f %d\n", H, W, density, cap, (int)f); return 1; }
                    // malformed variants of a good record must be refused (found = 2), never expanded
                    std::vector<uint8_t> bad = rec;
                    ((uint32_t *)bad.data())[0] = n + 1 <= c ? n + 1 : (n ? n - 1 : 0);
                    if (((uint32_t *)bad.data())[0] != n) {
                        f = 9;
                        sv_find_grid_corners_sparse_batch(bad.data(), stride, 1, H, W, 0.05, 0.02, got, &f, 1);
                        if (f != 2) { printf("bad count accepted %dx%d\n", H, W); return 1; }
                    }
                    if (wpr & 63) {
                        bad = rec;
                        ((uint64_t *)(bad.data() + 8))[((wpr + 63) / 64) - 1] |= 1ull << 63;
                        f = 9;
                        sv_find_grid_corners_sparse_batch(bad.data(), stride, 1, H, W, 0.05, 0.02, got, &f, 1);
                        if (f != 2) { printf("mask bit beyond the row accepted %dx%d\n", H, W); return 1; }
                    }
                    checked++;
                }
            }
        }
    printf("host fuzz ok: %ld sparse + dense comparisons against the byte scanner, malformed records refused\n", checked);
    return 0;
}
