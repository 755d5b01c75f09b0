// Host-side check of the layout in blocks (csrc/rt_lay.h, rt_ctx.h): the
// block plan of rt_reserve, the address of a ray, and the segments host-side
// loops walk.  Compiled with hipcc, runs without a GPU.  Test infrastructure.
#include "../../rayopt_amd/csrc/rt_ctx.h"
#include <cstdio>
#include <cstdlib>

static long bad = 0;
#define CHECK(c)                                                            \
    do {                                                                    \
        if (!(c)) {                                                         \
            ++bad;                                                          \
            if (bad < 20)                                                   \
                printf("line %d: %s\n", __LINE__, #c);                      \
        }                                                                   \
    } while (0)

int main()
{
    // the plan: expected block counts for 13 elements (1040 B per ray)
    struct { int64_t n; int blocks; } want[] = {
        {1, 1}, {64, 1}, {8000000, 1}, {8173076, 1}, {8173077, 2},
        {10000000, 2}, {12500000, 2}, {13461538, 2}, {13461539, 3},
        {20000000, 3}, {30000000, 5}, {100000000, 15}, {1000000000, 149}};
    for (auto w : want) {
        int64_t bs;
        int nb;
        rt_block_plan(13, 0, 64, w.n, &bs, &nb);
        CHECK(nb == w.blocks);
        CHECK(bs * nb >= w.n);
        CHECK(nb == 1 ? bs % 64 == 0 && bs - w.n < 64
                      : bs % 256 == 0 && bs * (nb - 1) < w.n);
        CHECK(nb == 1 || 1040. * bs <= 7.0e9 + 1040. * 256);
    }
    // forced block sizes, every n: whole workgroups, no empty block
    srand(7);
    for (int t = 0; t < 200000; ++t) {
        const int64_t n = 1 + rand() % 300000, b = 1 + rand() % 70000;
        int64_t bs;
        int nb;
        rt_block_plan(2 + rand() % 255, b, 64, n, &bs, &nb);
        CHECK(bs * nb >= n && bs * (nb - 1) < n);
        CHECK(nb == 1 ? bs % 64 == 0 : bs % 256 == 0);
        CHECK(nb == 1 || nb <= (n + b - 1) / b);
        // a laboratory layout keeps one block
        rt_block_plan(13, b, 4096, n, &bs, &nb);
        CHECK(nb == 1 && bs % 4096 == 0 && bs >= n);
    }
    // addresses and segments
    rt_ctx *c = (rt_ctx *)calloc(1, sizeof(rt_ctx));
    for (int t = 0; t < 20000; ++t) {
        const int L = 2 + rand() % 20;
        const int64_t n = 1 + rand() % 200000;
        int64_t bs;
        int nb;
        rt_block_plan(L, rand() % 3 ? 1 + rand() % 50000 : 0, 64, n, &bs, &nb);
        c->n = n;
        c->bs = bs;
        c->nblk = nb;
        c->ld = bs * nb;
        c->bts = nb > 1 ? (int64_t)10 * L * bs : 0;
        int64_t lo = rand() % n, hi = lo + rand() % (n - lo + 1);
        int64_t next = lo, count = 0;
        RT_FOR_SEGMENTS(c, g, lo, hi) {
            CHECK(g.ray == next && g.cnt > 0);
            CHECK(g.off == rt_block_col(c->bs, c->bts, g.ray));
            // contiguous inside the segment, and inside one block
            CHECK(rt_block_col(c->bs, c->bts, g.ray + g.cnt - 1) ==
                  g.off + g.cnt - 1);
            CHECK(g.ray / bs == (g.ray + g.cnt - 1) / bs);
            next = g.ray + g.cnt;
            ++count;
        }
        CHECK(next == (lo < hi ? hi : lo));
        CHECK(count <= nb);
        // no two rays share an address; rows of a block do not overlap the next block
        const int64_t j = rand() % n, k = rand() % n;
        CHECK((rt_block_col(bs, c->bts, j) == rt_block_col(bs, c->bts, k)) == (j == k));
        if (nb > 1)
            CHECK(rt_block_col(bs, c->bts, j) % c->bts < bs);
    }
    // the block of a workgroup without a division (rt_lay_set_window,
    // rt_wg_block): every workgroup of every window, against the division
    for (int t = 0; t < 3000; ++t) {
        rt_lay a = {};
        const int64_t wgs = t < 40 ? 1 + t : 1 + rand() % 40000;
        const int nb = 2 + rand() % 9;
        a.bs = wgs * 256;
        a.ts = 10 * (2 + rand() % 20) * a.bs;
        const int64_t ld = a.bs * nb;
        const int64_t lo = (int64_t)(rand() % (int)(ld / 256)) * 256;
        rt_lay_set_window(a, lo, ld);
        CHECK(a.wgs == wgs && a.j0 == lo && a.w0 == lo / 256);
        const int64_t nwg = (ld - lo) / 256;
        for (int64_t w = 0; w < nwg; w += (t < 200 ? 1 : 1 + rand() % 7)) {
            const int64_t j = w * 256 + rand() % 256; // a ray of the window
            CHECK((int64_t)rt_wg_block(a, (uint32_t)w) == (j + lo) / a.bs);
            CHECK(j + a.j0 + (int64_t)rt_wg_block(a, (uint32_t)w) * (a.ts - a.bs) ==
                  rt_block_col(a.bs, a.ts, j + lo));
        }
        // the last workgroup 2^24 - 2 of the largest batch the form covers
        rt_lay_set_window(a, 0, ((int64_t)1 << 32) - 512);
        CHECK(a.wgs == wgs);
        for (uint32_t w : {0u, (uint32_t)wgs - 1, (uint32_t)wgs,
                           (1u << 24) - 3, (1u << 24) - 2})
            CHECK(rt_wg_block(a, w) == w / (uint64_t)wgs);
        // windows off a workgroup border, blocks that are not whole
        // workgroups and one block: the dividing form stays in charge
        rt_lay_set_window(a, lo + 64, ld);
        CHECK(a.wgs == 0 && a.j0 == lo + 64);
        rt_lay_set_window(a, 0, (int64_t)1 << 32);
        CHECK(a.wgs == 0);
        a.bs += 64;
        rt_lay_set_window(a, 0, ld);
        CHECK(a.wgs == 0);
        a.ts = 0;
        rt_lay_set_window(a, 0, ld);
        CHECK(a.wgs == 0);
    }
    // all divisors up to 5000 and a few large ones, every w at the edges
    for (int64_t d = 1; d <= (1 << 24); d = d < 5000 ? d + 1 : d * 3 + 1) {
        rt_lay a = {};
        a.bs = d * 256;
        a.ts = 100 * a.bs;
        rt_lay_set_window(a, 0, ((int64_t)1 << 32) - 512);
        CHECK(a.wgs == d);
        for (uint64_t q = 0; q * d < (1u << 24) - 1; q += 1 + q / 64) {
            for (int64_t e = -1; e <= 1; ++e) {
                const int64_t w = (int64_t)(q * d) + e;
                if (w < 0 || w >= (1 << 24) - 1)
                    continue;
                CHECK(rt_wg_block(a, (uint32_t)w) == (uint64_t)w / (uint64_t)d);
            }
        }
    }
    // the turns of a generated batch (rt_gen_wg): a permutation of the
    // workgroups of the bundles, padding left alone, runs of `turn`
    // consecutive workgroups per bundle, a turn of every bundle before the
    // next turn of any
    for (int t = 0; t < 400; ++t) {
        rt_gen_order o;
        o.per = 1 + rand() % 300;
        o.nf = 1 + rand() % 7;
        o.turn = t % 5 == 0 ? 0 : 1 + rand() % (o.per + 20);
        const uint32_t all = o.per * o.nf, grid = all + rand() % 9;
        unsigned char *seen = (unsigned char *)calloc(grid, 1);
        uint32_t before_q = 0;
        for (uint32_t w = 0; w < grid; ++w) {
            const uint32_t v = rt_gen_wg(o, w);
            CHECK(v < grid && !seen[v]);
            if (v < grid)
                seen[v] = 1;
            if (w >= all || !o.turn) {
                CHECK(v == w);
                continue;
            }
            const uint32_t f = v / o.per, k = v % o.per, q = k / o.turn;
            CHECK(q >= before_q); // turns never go back
            before_q = q;
            if (w + 1 < all) { // the next one: same run, or the next run
                const uint32_t v1 = rt_gen_wg(o, w + 1);
                const uint32_t f1 = v1 / o.per, k1 = v1 % o.per;
                const bool same_run = f1 == f && k1 == k + 1 &&
                                      k1 / o.turn == q;
                const bool next_bundle = f1 == f + 1 && k1 == q * o.turn;
                const bool next_turn = f == o.nf - 1 && f1 == 0 &&
                                       k1 == (q + 1) * o.turn;
                CHECK(same_run || next_bundle || next_turn);
            }
        }
        free(seen);
    }
    free(c);
    printf("blocks_host: %ld failures\n", bad);
    return bad != 0;
}
