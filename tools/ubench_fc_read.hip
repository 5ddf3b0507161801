// Micro-benchmark (round 3): what does reading the 260-MB feature tensor cost, by access pattern?  k_fc_head_h2p's launch shape (one 768-thread
// workgroup per CU, 81 rows of 12,544 B each, 49 stages of 256 B per row, two stages of loads in flight) with nothing but the loads:
//   P0  the MFMA-fragment pattern of the fc kernels: lane (r, q) reads 64 B of row r at 64 q -- adjacent lanes are different rows, every lane its own
//       16-byte access; waves w and w + 6 read the same rows (the two N halves)
//   P1  the same bytes per stage, lane-contiguous: 16 adjacent lanes read 256 contiguous bytes of a row; every byte read once per workgroup
//   P2  the same amount of data as one sequential stream per wave (what a tile-major layout would allow)
// and a writer pass (P3) that produces the tensor first, like k_conv_features does, so that the reads see the same cache state.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32;

constexpr int FEAT_B = 12544, NST = 49;

template <int P>
__global__ __launch_bounds__(768) void k_read(const unsigned char *__restrict__ feat, long rows, long per, u32 *__restrict__ sink)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r0 = (long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    uint4 acc = {0, 0, 0, 0};
    auto add = [&](const uint4 &v) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; };
    if (P == 0) {
        const int r = lane & 15, q = lane >> 4, mt = wave % 6;
        long row = r0 + 16 * mt + r;
        if (row >= r1) row = r1 - 1;
        const unsigned char *p = feat + row * FEAT_B + 64 * q;
        uint4 a[3][4];
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) a[s][j] = *(const uint4 *)(p + 256 * s + 16 * j);
#pragma unroll 3
        for (int st = 0; st < NST; st++) {
            const int nx = st + 2 < NST ? st + 2 : NST - 1;
#pragma unroll
            for (int j = 0; j < 4; j++) a[(st + 2) % 3][j] = *(const uint4 *)(p + 256 * nx + 16 * j);
#pragma unroll
            for (int j = 0; j < 4; j++) add(a[st % 3][j]);
            __builtin_amdgcn_s_barrier();
        }
    } else if (P == 1) {
        uint4 a[3][2];
        auto addr = [&](int st, int i) {
            const int id = tid + 768 * i;                          // 1536 16-byte pieces of a stage: 96 rows x 16
            long row = r0 + (id >> 4);
            if (row >= r1) row = r1 - 1;
            return feat + row * FEAT_B + 256 * st + 16 * (id & 15);
        };
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int i = 0; i < 2; i++) a[s][i] = *(const uint4 *)addr(s, i);
#pragma unroll 3
        for (int st = 0; st < NST; st++) {
            const int nx = st + 2 < NST ? st + 2 : NST - 1;
#pragma unroll
            for (int i = 0; i < 2; i++) a[(st + 2) % 3][i] = *(const uint4 *)addr(nx, i);
#pragma unroll
            for (int i = 0; i < 2; i++) add(a[st % 3][i]);
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // one sequential stream per wave: the workgroup's per * 12,544 bytes split into 12 contiguous parts
        const long bytes = (r1 - r0) * FEAT_B, part = (bytes / 12) & ~1023L;
        const unsigned char *p = feat + r0 * FEAT_B + wave * part + 16 * lane;
        const long n = part / 1024;
        for (long i = 0; i + 4 <= n; i += 4) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = *(const uint4 *)(p + 1024 * (i + j));
#pragma unroll
            for (int j = 0; j < 4; j++) add(v[j]);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[tid] = acc.x;
}

// P4 / P5: the weight stream of k_fc_head_h2p alone -- every workgroup reads the same 1.6 MB in 49 stages of 32 KB (waves 0-7: 4 KB each), two stages in
// flight, a barrier per stage.  P4: all workgroups walk the stages in the same order (what the kernel does); P5: workgroup b starts at stage 7 b mod 49.
template <bool ROTATE>
__global__ __launch_bounds__(768) void k_wstream(const unsigned char *__restrict__ w, u32 *__restrict__ sink)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int off = ROTATE ? (blockIdx.x * 7) % NST : 0;
    uint4 acc = {0, 0, 0, 0};
    auto add = [&](const uint4 &v) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; };
    const unsigned char *p = w + (wave & 7) * 4096 + lane * 16;
    uint4 a[3][4];
    auto ld = [&](int st, uint4 (&d)[4]) {
        const int s2 = (st + off) % NST;
#pragma unroll
        for (int j = 0; j < 4; j++) d[j] = *(const uint4 *)(p + (size_t)s2 * 32768 + 1024 * j);
    };
    if (wave < 8) { ld(0, a[0]); ld(1, a[1]); }
#pragma unroll 3
    for (int st = 0; st < NST; st++) {
        if (wave < 8) {
            ld(st + 2 < NST ? st + 2 : NST - 1, a[(st + 2) % 3]);
#pragma unroll
            for (int j = 0; j < 4; j++) add(a[st % 3][j]);
        }
        __builtin_amdgcn_s_barrier();
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[tid] = acc.x;
}

// P6: both streams of a per-CU fc kernel as LDS-DMA (global_load_lds_dwordx4), no compute: per stage 24 feature pieces (4 rows x 256 B each, lane-contiguous,
// two per wave, ring of three 24-KB buffers: two stages in flight) and 32 weight pieces (1 KB each, waves 0-7, ring of two 32-KB buffers: one stage in flight),
// one counted wait (vmcnt(2)) and one barrier per stage.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(768) void k_both(const unsigned char *__restrict__ feat, const unsigned char *__restrict__ w, long rows, long per, u32 *__restrict__ sink)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 32768 + 3 * 24576];
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long r0 = (long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    const unsigned char *ap[2];
    for (int i = 0; i < 2; i++) {
        long row = r0 + 8 * wave + 4 * i + (lane >> 4);
        if (row >= r1) row = r1 - 1;
        ap[i] = feat + row * FEAT_B + 16 * (lane & 15);
    }
    const unsigned char *wp = w + (wave & 7) * 4096 + lane * 16;
    auto issue_w = [&](int st) {
        if (wave < 8)
            for (int j = 0; j < 4; j++) glds16(wp + (size_t)st * 32768 + 1024 * j, base + (st & 1) * 32768 + (wave & 7) * 4096 + 1024 * j);
    };
    auto issue_a = [&](int st) {
        for (int i = 0; i < 2; i++) glds16(ap[i] + 256 * st, base + 65536 + (st % 3) * 24576 + (2 * wave + i) * 1024);
    };
    issue_w(0); issue_a(0); issue_a(1);
    u32 acc = 0;
    for (int st = 0; st < NST; st++) {
        if (st + 1 < NST) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 1 < NST) issue_w(st + 1);
        if (st + 2 < NST) issue_a(st + 2);
        acc ^= *(const u32 *)(lds + (st & 1) * 32768 + tid * 4) ^ *(const u32 *)(lds + 65536 + (st % 3) * 24576 + tid * 4);
    }
    if (acc == 0x12345678u) sink[tid] = acc;
}

__global__ __launch_bounds__(512) void k_write(uint4 *__restrict__ feat, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) feat[i] = uint4{(u32)i, 1, 2, 3};
}

int main()
{
    const long rows = 20736, per = 81;
    unsigned char *feat; u32 *sink;
    hipMalloc(&feat, (size_t)rows * FEAT_B); hipMalloc(&sink, 4096);
    auto run = [&](auto kern, const char *name, bool rewrite) {
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 12; i++) {
            if (rewrite) k_write<<<256, 512>>>((uint4 *)feat, (size_t)rows * FEAT_B / 16);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            kern<<<256, 768>>>(feat, rows, per, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
            hipEventDestroy(a); hipEventDestroy(b);
        }
        printf("%-44s %s: min %.4f ms, mean %.4f ms  (%.2f TB/s at the mean)\n", name, rewrite ? "freshly written" : "read again      ", best, sum / 10,
               rows * FEAT_B / (sum / 10) * 1e-9);
    };
    {
        unsigned char *w; hipMalloc(&w, NST * 32768); hipMemset(w, 3, NST * 32768);
        for (int v = 0; v < 2; v++) {
            float best = 1e9f;
            for (int i = 0; i < 12; i++) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipEventRecord(a);
                if (v) k_wstream<true><<<256, 768>>>(w, sink); else k_wstream<false><<<256, 768>>>(w, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (i >= 2 && ms < best) best = ms;
            }
            printf("%s: min %.4f ms (%.2f TB/s out of the L2s)\n", v ? "P5 weight stream, stage order rotated per workgroup" : "P4 weight stream, all workgroups in step          ", best,
                   256.0 * NST * 32768 / best * 1e-9);
        }
    }
    {
        unsigned char *w; hipMalloc(&w, NST * 32768); hipMemset(w, 3, NST * 32768);
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 12; i++) {
            k_write<<<256, 512>>>((uint4 *)feat, (size_t)rows * FEAT_B / 16);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            k_both<<<256, 768>>>(feat, w, rows, per, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("P6 features + weights as LDS-DMA, rings of 3 and 2, no compute: min %.4f ms, mean %.4f ms\n", best, sum / 10);
    }
    for (int rw = 1; rw >= 0; rw--) {
        run(k_read<0>, "P0 fragment pattern (as k_fc_head_h2p)", rw);
        run(k_read<1>, "P1 lane-contiguous 256-B row pieces", rw);
        run(k_read<2>, "P2 one sequential stream per wave", rw);
    }
    return 0;
}
