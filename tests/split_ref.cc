// split_ref.cc — the work split of the Schur tile kernel (ptam_cg_amd/csrc/ba_split.h) compiled for the HOST: test infrastructure
// (tests/test_gpu_prepare.py builds it with g++ and compares the device's lists against it).  The budget search is the device's,
// one budget after another instead of one lane each.
#define __host__
#define __device__
#include "../ptam_cg_amd/csrc/ba_split.h"

static long long smallest_budget(const SplitRuns& R, const SplitCfg& c, long long cost_x, int n_wg) {
    SplitBracket br = split_bracket_begin(cost_x, n_wg, R.np, c);
    auto none = [](int, int, int, int) {};
    for (int round = 0; round < 64 && !br.done; round++) {
        int jmin = SPLIT_NC;
        for (int j = 0; j < SPLIT_NC; j++)
            if (split_fill(split_candidate(br, j), R, c, none, n_wg) <= n_wg) {
                jmin = j;
                break;
            }
        split_bracket_step(br, jmin);
    }
    return br.hi;
}

extern "C" {
// cfg8: seg_cost, second_lag, min_room, min_seg, n_first, cost_model, slots, greedy.  pl_e / pl_h: the pairs' entry and H prefixes
// (ba_split.h: the even split).  Returns the number of segments (cuts) written — each (k, begin, end, wg) — or -1 if `cap` is too
// small; *n_wgs_out, *t_cut_out as the device publishes them.
int split_ref_list(const int* run_cnt, const int* run_cost, const int* pl_run0, const int* pl_n, int np, long long cost_x,
                   long long ent_x, const int* cfg8, int* cuts4, int cap, int* n_wgs_out, long long* t_cut_out, const int* pl_e,
                   const long long* pl_h) {
    SplitCfg c;
    c.seg_cost = cfg8[0], c.second_lag = cfg8[1], c.min_room = cfg8[2], c.min_seg = cfg8[3], c.n_first = cfg8[4], c.cost_model = cfg8[5],
    c.slots = cfg8[6], c.greedy = cfg8[7];
    SplitRuns R{run_cnt, run_cost, pl_run0, pl_n, np, pl_e, pl_h};
    if (!c.greedy) {
        const SplitEven ev = split_even_budget(pl_h[np], ent_x, c);
        int n = 0;
        bool over = false;
        const int n_wgs = split_even_cut(ev, R, c, [&](int k, int b, int e, int wg) {
            if (n >= cap) {
                over = true;
                return;
            }
            cuts4[4 * n] = k, cuts4[4 * n + 1] = b, cuts4[4 * n + 2] = e, cuts4[4 * n + 3] = wg;
            n++;
        });
        *n_wgs_out = n_wgs;
        *t_cut_out = ev.T;
        return over ? -1 : n;
    }
    long long n_wg_max = ent_x / (4 * c.min_seg);
    if (n_wg_max > c.slots) n_wg_max = c.slots;
    if (n_wg_max < 1) n_wg_max = 1;
    const bool two = n_wg_max > c.n_first;
    long long t_cut = smallest_budget(R, c, cost_x, (int)n_wg_max);
    if (two && cost_x / c.n_first + c.seg_cost < t_cut) {
        const long long tb = smallest_budget(R, c, cost_x, c.n_first);
        if (tb < t_cut) t_cut = tb;
    }
    int n = 0;
    bool over = false;
    const int n_wgs = split_fill(t_cut, R, c, [&](int k, int b, int e, int wg) {
        if (n >= cap) {
            over = true;
            return;
        }
        cuts4[4 * n] = k, cuts4[4 * n + 1] = b, cuts4[4 * n + 2] = e, cuts4[4 * n + 3] = wg;
        n++;
    });
    *n_wgs_out = n_wgs;
    *t_cut_out = t_cut;
    return over ? -1 : n;
}
int split_ref_entry_cost(int F, int a, int b, int pat, int cost_model) { return split_entry_cost(F, a, b, pat, cost_model); }
int split_ref_full_products(int F, int a, int b) { return split_full_products(F, a, b); }
}
