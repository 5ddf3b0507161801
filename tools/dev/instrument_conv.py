#!/usr/bin/env python3
"""Development aid: patches cycle-counter stamps into k_conv_features_wstream (a scratch copy of csrc/k3_cnn.hip is
written in place -- restore with `git checkout` afterwards) so that tools/dev/trace_conv.py can print a per-step timeline
of one workgroup: consumer wave 0 (GEMM / output transform / help / barrier wait) and producer wave 4
(transform / conv1 + input store / barrier wait)."""
import os
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sudoku-vision_amd", "csrc", "k3_cnn.hip")
s = open(p).read()
a = s.index('template <bool U8IN>\n__global__ __launch_bounds__(512, 2) void k_conv_features_wstream')
s = s[:a] + '''__device__ unsigned long long g_trace[64 * 8 * 6];
extern "C" int sv_debug_conv_trace(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(g_trace)); }
#define TR(slot) do { if (blockIdx.x == 7 && m >= 40 && m < 104 && lane == 0) g_trace[((m - 40) * 8 + wave) * 6 + (slot)] = __builtin_readcyclecounter(); } while (0)
''' + s[a:]
a = s.index('__global__ __launch_bounds__(512, 2) void k_conv_features_wstream')
b = s.index('// 64 cells per workgroup, 16 per wave; K = 3136 in 196 chunks of 16.')
k = s[a:b]


def rep(old, new):
    global k
    assert old in k, old
    k = k.replace(old, new, 1)


rep('''        for (int m = 0; m < NM; m++) {
            const long sc = wstream_need(m + 1, ncell);''', '''        for (int m = 0; m < NM; m++) {
            TR(0);
            const long sc = wstream_need(m + 1, ncell);''')
rep('''            if (m + 1 < NM) transform(m + 1, 0, 256);
            const long cc = wstream_need(m, ncell);''', '''            if (m + 1 < NM) transform(m + 1, 0, 256);
            TR(1);
            const long cc = wstream_need(m, ncell);''')
rep('''            if (sc > staged) { stage_store(sc); staged = sc; }
            __syncthreads();''', '''            if (sc > staged) { stage_store(sc); staged = sc; }
            TR(2); TR(3);
            __syncthreads();
            TR(4);''')
rep('''    for (int m = 0; m < NM; m++) {
        const float *ap = v_base + (m & 1) * VSLOT + q * 16 + r16;''', '''    for (int m = 0; m < NM; m++) {
        TR(0);
        const float *ap = v_base + (m & 1) * VSLOT + q * 16 + r16;''')
rep('''#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int T = 16 * m + 4 * q + reg;''', '''        TR(1);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int T = 16 * m + 4 * q + reg;''')
rep('''        if (m + 1 < NM) transform(m + 1, 256, 512);
        __syncthreads();''', '''        TR(2);
        if (m + 1 < NM) transform(m + 1, 256, 512);
        TR(3);
        __syncthreads();
        TR(4);''')
s = s[:a] + k + s[b:]
open(p, 'w').write(s)
print("instrumented", p)
