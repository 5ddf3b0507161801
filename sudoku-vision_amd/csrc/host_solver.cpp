// N4 -- the sudoku solver as an in-process call (the reference shells out to solver/sudoku_solver through temp files,
// pipeline/run.py:163-202; the solver itself is solver/src/sudoku.c: solve_sudoku, sudoku.c:72-81).
// Written fresh around bitmasks and a flat peer table, but following the reference's decision order exactly -- validation;
// candidate initialisation; propagation sweeps (naked singles row-major, then hidden singles by row, by column, by box,
// digits ascending, repeated while anything was placed); branching on the first cell with the fewest candidates, digits
// ascending -- so that it returns the SAME grid as the reference also for puzzles with several solutions.
#include <cstdint>
#include <cstring>

#include "sv_internal.h"

namespace {

struct Peers {
    uint8_t p[81][20];
    Peers()
    {
        for (int i = 0; i < 81; i++) {
            const int r = i / 9, c = i % 9, br = r / 3 * 3, bc = c / 3 * 3;
            int n = 0;
            bool seen[81] = {false};
            auto add = [&](int j) { if (j != i && !seen[j]) { seen[j] = true; p[i][n++] = (uint8_t)j; } };
            for (int k = 0; k < 9; k++) add(r * 9 + k);
            for (int k = 0; k < 9; k++) add(k * 9 + c);
            for (int y = br; y < br + 3; y++)
                for (int x = bc; x < bc + 3; x++) add(y * 9 + x);
        }
    }
};
const Peers kPeers;

struct State { uint8_t g[81]; uint16_t cand[81]; };   // cand bit d (1..9) set = digit d still possible; 0 for filled cells

inline void place(State &s, int i, int d)
{
    s.g[i] = (uint8_t)d;
    s.cand[i] = 0;
    const uint16_t m = (uint16_t)~(1u << d);
    for (int k = 0; k < 20; k++) s.cand[kPeers.p[i][k]] &= m;
}

bool valid(const uint8_t *g)
{
    for (int i = 0; i < 81; i++)
        if (g[i] > 9) return false;
    for (int u = 0; u < 27; u++) {          // 9 rows, 9 columns, 9 boxes
        unsigned seen = 0;
        for (int k = 0; k < 9; k++) {
            const int i = u < 9 ? u * 9 + k : u < 18 ? k * 9 + (u - 9) : ((u - 18) / 3 * 3 + k / 3) * 9 + (u - 18) % 3 * 3 + k % 3;
            if (g[i]) { if (seen >> g[i] & 1) return false; seen |= 1u << g[i]; }
        }
    }
    return true;
}

// hidden single of digit d among the 9 cells idx[]: -1 contradiction, 1 placed, 0 nothing
inline int hidden_single(State &s, const int *idx, int d)
{
    int count = 0, last = -1;
    for (int k = 0; k < 9; k++) {
        const int i = idx[k];
        if (s.g[i] == d) return 0;                       // already placed in this unit
        if (s.g[i] == 0 && (s.cand[i] >> d & 1)) { count++; last = i; }
    }
    if (count == 0) return -1;
    if (count == 1) { place(s, last, d); return 1; }
    return 0;
}

bool propagate(State &s)
{
    static int unit[27][9];
    static bool init = false;
    if (!init) {
        for (int u = 0; u < 9; u++)
            for (int k = 0; k < 9; k++) {
                unit[u][k] = u * 9 + k;
                unit[9 + u][k] = k * 9 + u;
                unit[18 + u][k] = (u / 3 * 3 + k / 3) * 9 + u % 3 * 3 + k % 3;
            }
        init = true;
    }
    for (bool progress = true; progress;) {
        progress = false;
        for (int i = 0; i < 81; i++)
            if (s.g[i] == 0) {
                const unsigned c = s.cand[i];
                if (c == 0) return false;
                if ((c & (c - 1)) == 0) { place(s, i, __builtin_ctz(c)); progress = true; }
            }
        for (int u = 0; u < 27; u++)                     // rows, then columns, then boxes (boxes row-major)
            for (int d = 1; d <= 9; d++) {
                const int r = hidden_single(s, unit[u], d);
                if (r < 0) return false;
                progress |= r > 0;
            }
    }
    return true;
}

bool search(State &s)
{
    if (!propagate(s)) return false;
    int best = -1, best_n = 10;
    for (int i = 0; i < 81; i++)
        if (s.g[i] == 0) {
            const int n = __builtin_popcount(s.cand[i]);
            if (n < best_n) { best_n = n; best = i; }
        }
    if (best < 0) return true;                           // no empty cell left: solved
    const unsigned c = s.cand[best];
    for (int d = 1; d <= 9; d++)
        if (c >> d & 1) {
            State t = s;
            place(t, best, d);
            if (search(t)) { s = t; return true; }
        }
    return false;
}

}  // namespace

// result: 1 solved (solution filled), 0 valid input without solution, -1 invalid input (the reference's SOLVE_* codes,
// solver/include/sudoku.h:13-16).  grid: 81 digits row-major, 0 = empty.
extern "C" int sv_solve_sudoku(const uint8_t *grid, uint8_t *solution, int *result)
{
    if (!grid || !solution || !result) return sv_fail(SV_ERR_BAD_ARG, "sv_solve_sudoku: NULL argument");
    memcpy(solution, grid, 81);
    if (!valid(grid)) { *result = -1; return SV_OK; }
    State s;
    memcpy(s.g, grid, 81);
    for (int i = 0; i < 81; i++) s.cand[i] = grid[i] ? 0 : 0x3FE;
    for (int i = 0; i < 81; i++)
        if (grid[i]) {
            const uint16_t m = (uint16_t)~(1u << grid[i]);
            for (int k = 0; k < 20; k++) s.cand[kPeers.p[i][k]] &= m;
        }
    if (search(s)) { memcpy(solution, s.g, 81); *result = 1; }
    else *result = 0;
    return SV_OK;
}
