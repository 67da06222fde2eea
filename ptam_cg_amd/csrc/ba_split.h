// ba_split.h — the work split of the Schur tile kernel (ba_schur.inc) as integer arithmetic on RUN LISTS, shared by the device
// kernels that build the lists (ba_prepare.inc) and by the host-side checker of the instrumented build (bundle.hip, PTAM_AB_SWITCHES).
//
// One XCD's work list is the tile pairs' entries one pair after another; inside a pair the entries are sorted by fragment pattern,
// and an entry's model cost depends on (pair, pattern) only — so a pair's list is at most 16 RUNS of equal-cost entries, and
// everything the split needs ("how many whole entries fit this budget", "what do entries [pos, pos + take) cost") is a walk over
// one to three runs instead of a binary search over per-entry prefix sums.  The split then needs the (XCD, pair, pattern) counts
// only — not the entries — and runs before they are placed.  Budgets and costs are integers (the cost model's units are ~ns):
// host and device take the same decisions bit for bit.
#pragma once

struct SplitCfg {
    int seg_cost;     // fixed cost of a segment (pipeline fill, cross-wave reduction, partial tile out)
    int second_lag;   // what a CU's second workgroup is given less than its first
    int min_room;     // a workgroup begins another segment only for at least this much work
    int min_seg;      // entries of the shortest segment
    int n_first;      // list positions >= n_first are second workgroups of their CUs
    int cost_model;   // load term of an entry's cost (0: every entry costs 1)
    int slots;        // workgroups one XCD holds at once
    int greedy;       // 1: the greedy fill + budget search of round 6's first form (measurement build only); 0: the even split
};

// one XCD's list: pairs k = 0..np-1 (the non-empty ones, in list order), pair k's runs are [pl_run0[k], pl_run0[k + 1]).
// P: the pointer type of the arrays (the device kernel keeps them in LDS and says so in the type: a fill is a serial chain of
// dependent reads, and through generic pointers each was a FLAT load of several hundred cycles).
template <class P, class P64 = const long long*>
struct SplitRunsT {
    P run_cnt;
    P run_cost;
    P pl_run0;   // [np + 1]
    P pl_n;      // entries of the pair
    int np;
    // the even split's prefixes over the pairs: entries before pair k (pl_e[np]: all of them), and H at the pair's start — the model
    // cost of the pairs before it + one segment cost per pair start before it
    P pl_e;      // [np + 1]
    P64 pl_h;    // [np + 1]
};
typedef SplitRunsT<const int*> SplitRuns;

// 16x16 fragments per tile (row mappings of ba_schur.inc: 1-2 cameras in a tile = 1 fragment, 3-5 = 2, 6-8 = 3)
__host__ __device__ inline int split_frags(int F, int t) {
    int n = F - t * 8;
    if (n > 8) n = 8;
    return n <= 2 ? 1 : (n <= 5 ? 2 : 3);
}
// Model cost of one entry of tile pair (a, b), a >= b, with fragment pattern `pat` (bit 0: a camera of the point in slots 0..5 of
// tile a, bit 1: in slots 5..7, bits 2, 3: the same for tile b): 10 units per 16x16 fragment product + cost_model for its loads
// (fitted to per-workgroup stamps of the tile kernel, tools/dev/schur_fit.py, docs/LOG_r05.md).
__host__ __device__ inline int split_entry_cost(int F, int a, int b, int pat, int cost_model) {
    if (!cost_model) return 1;
    const int ma = split_frags(F, a), mb = split_frags(F, b);
    const int a01 = (pat & 1) ? (ma < 2 ? ma : 2) : 0, a2 = (ma == 3 && (pat & 2)) ? 1 : 0;
    int f;
    if (a == b)
        f = a01 * a01 + a2 * a01 + a2;
    else {
        const int b01 = (pat & 4) ? (mb < 2 ? mb : 2) : 0, b2 = (mb == 3 && (pat & 8)) ? 1 : 0;
        f = (a01 + a2) * (b01 + b2);
    }
    return 10 * f + cost_model;
}
// fragment products of a pair whose points have cameras everywhere (the order of the pairs in a list: fewest first)
__host__ __device__ inline int split_full_products(int F, int a, int b) {
    const int ma = split_frags(F, a), mb = split_frags(F, b);
    return a == b ? (ma == 3 ? 7 : (ma == 2 ? 4 : 1)) : ma * mb;
}

// n / (cost of run i), n >= 0: a list type may bring its own (the device's lists in LDS keep the runs' reciprocals: an integer division
// is ~45 dependent instructions of the fill's serial chain)
template <class RUNS, class I>
__host__ __device__ inline I split_quot(const RUNS& R, int i, I n) {
    return n / (I)R.run_cost[i];
}

// The greedy fill: the pairs' entries, one pair after another, into workgroups of whole 4-entry groups such that no workgroup's
// cost — entries + seg_cost per segment (+ second_lag for list positions >= n_first) — exceeds T.  Returns the workgroups made;
// emit(k, begin, end, wg) is called for every segment (entries [begin, end) of pair k, relative to the pair's first entry in
// this list, in workgroup wg).  I: the integer type budgets and costs are held in (int where the list's total cost fits 31 bits:
// the device's 64-bit divisions are software; the decisions are the same).  The fill gives up — returns a count > limit — as
// soon as it has made more than `limit` workgroups: a budget near the mean cost makes hundreds of them (every one the
// minimum segment), and the budget search's wave runs as long as its slowest lane.
template <class I, class RUNS, class Emit>
__host__ __device__ inline int split_fill(I T, const RUNS& R, const SplitCfg& c, Emit&& emit, int limit = 0x7fffffff) {
    int n_out = 0, cur_n = 0;
    I cur_cost = 0;
    for (int k = 0; k < R.np; k++) {
        const int n = R.pl_n[k], run1 = R.pl_run0[k + 1];
        int pos = 0, ri = R.pl_run0[k], ro = 0;
        int rc = R.run_cnt[ri], rs = R.run_cost[ri];   // the run `pos` is in
        if (run1 - ri == 1) {
            // ONE pattern in this pair's part of the list (every pair of a dense problem): the same decisions with the walks over
            // the runs folded away — the loop below, executed by one lane per budget, is a serial chain, and its length is the kernel
            while (pos < n) {
                const int left = n - pos;
                const I room = T - (I)(n_out >= c.n_first ? c.second_lag : 0) - cur_cost - (I)c.seg_cost;
                int take = 0;
                if (room >= 0) take = (I)left * rs <= room ? left : (int)split_quot(R, ri, room);
                take = take >= left ? left : take / 4 * 4;
                if (cur_n > 0 && left > take && (take < c.min_seg || room < (I)c.min_room)) {
                    n_out++;
                    if (n_out > limit) return n_out;
                    cur_n = 0;
                    cur_cost = 0;
                    continue;
                }
                if (take < c.min_seg) take = c.min_seg;
                if (left - take < c.min_seg) take = left;
                if (take > left) take = left;
                emit(k, pos, pos + take, n_out);
                cur_n++;
                cur_cost += (I)take * rs + (I)c.seg_cost;
                pos += take;
            }
            continue;
        }
        while (pos < n) {
            const int left = n - pos;
            const I room = T - (I)(n_out >= c.n_first ? c.second_lag : 0) - cur_cost - (I)c.seg_cost;
            // as many whole entries as the workgroup's remaining budget pays for
            int take = 0;
            if (room >= 0) {
                const int avail0 = rc - ro;
                if ((I)avail0 * rs <= room) {
                    take = avail0;
                    I rem = room - (I)avail0 * rs;
                    for (int i = ri + 1; i < run1; i++) {
                        const int avail = R.run_cnt[i], cst = R.run_cost[i];
                        if ((I)avail * cst <= rem) {
                            take += avail;
                            rem -= (I)avail * cst;
                        } else {
                            take += (int)split_quot(R, i, rem);
                            break;
                        }
                    }
                } else
                    take = (int)split_quot(R, ri, room);
            }
            take = take >= left ? left : take / 4 * 4;
            // a sliver at the end of a full workgroup — less work than the segment itself would cost: start the next one
            if (cur_n > 0 && left > take && (take < c.min_seg || room < (I)c.min_room)) {
                n_out++;
                if (n_out > limit) return n_out;
                cur_n = 0;
                cur_cost = 0;
                continue;
            }
            if (take < c.min_seg) take = c.min_seg;
            if (left - take < c.min_seg) take = left;   // ... or at the end of the pair's chunk: take it along
            if (take > left) take = left;
            I cst = 0;
            for (int t = take; t > 0;) {
                const int avail = rc - ro, s = avail < t ? avail : t;
                cst += (I)s * rs;
                t -= s;
                ro += s;
                if (ro == rc) {
                    ri++;
                    ro = 0;
                    if (ri < run1) rc = R.run_cnt[ri], rs = R.run_cost[ri];
                }
            }
            emit(k, pos, pos + take, n_out);
            cur_n++;
            cur_cost += cst + (I)c.seg_cost;
            pos += take;
        }
    }
    if (cur_n > 0) n_out++;
    return n_out;
}

// The budget search looks at SPLIT_NC budgets per round, evenly spaced in (lo, hi], and keeps the bracket around the smallest
// one whose fill needs no more than n_wg workgroups (the device: one lane per budget; the host checker: one after another).
#define SPLIT_NC 256
struct SplitBracket {
    long long lo, hi, t_all;
    int done;
};
__host__ __device__ inline SplitBracket split_bracket_begin(long long cost_x, int n_wg, int chunks_x, const SplitCfg& c) {
    SplitBracket b;
    b.lo = cost_x / n_wg + c.seg_cost;                      // a lower bound: the mean cost + the one segment every workgroup has
    b.hi = b.lo + b.lo / 4 + c.second_lag;                  // the answer is 1.1 - 1.4 times that
    b.t_all = cost_x + (long long)c.seg_cost * (chunks_x + n_wg) + c.second_lag + 1;   // one workgroup takes it all
    if (b.hi > b.t_all) b.hi = b.t_all;
    if (b.lo >= b.hi) b.lo = b.hi - 1;
    b.done = 0;
    return b;
}
__host__ __device__ inline long long split_candidate(const SplitBracket& b, int j) { return b.lo + (b.hi - b.lo) * (j + 1) / SPLIT_NC; }
// jmin: the smallest j whose candidate was feasible (SPLIT_NC: none)
__host__ __device__ inline void split_bracket_step(SplitBracket& b, int jmin) {
    if (jmin >= SPLIT_NC) {
        if (b.hi >= b.t_all) {   // (cannot happen: one workgroup always takes it all)
            b.done = 1;
            return;
        }
        b.lo = b.hi;
        b.hi = b.hi + b.hi / 4 + 1;
        if (b.hi > b.t_all) b.hi = b.t_all;
        return;
    }
    const long long nhi = split_candidate(b, jmin), nlo = jmin > 0 ? split_candidate(b, jmin - 1) : b.lo;
    b.lo = nlo;
    b.hi = nhi;
    if (b.hi - b.lo <= b.hi / 200 + 1) b.done = 1;   // 0.5 %: one round of 256 budgets over a bracket of 0.7 means
}


// ---- the even split ---------------------------------------------------------------------------------------------------------------
// No walk, no search: along the list, H(e) = model cost of the entries before e + one segment cost per pair start before e grows
// monotonically; a workgroup that starts at e and ends at e' costs H(e') - H(e) + one more segment (its own first one — charged
// twice where it starts exactly at a pair's start: such a workgroup is given a little less than it could take).  With n workgroups
// of which those at list positions >= n_first are given second_lag less, the budget that makes the parts add up is
//   T = ceil((H_total + n seg + lag (n - n_first)+) / n),
// workgroup w starts where H reaches  w (T - seg) - lag (w - n_first)+ , rounded down to a whole 4-entry group of its pair and
// snapped to the pair's start / the next pair's start where less than min_seg entries would be left on either side.  Every start is
// a closed form of w: the device computes them one per lane, the host one after the other — the same integers.
struct SplitEven {
    long long T;
    int n_wg;
    int lag;   // second_lag, or 0 where the list is too light for it (a second workgroup's budget must stay positive)
};
__host__ __device__ inline SplitEven split_even_budget(long long h_total, long long ent_x, const SplitCfg& c) {
    long long n_max = ent_x / (4 * c.min_seg);
    if (n_max > c.slots) n_max = c.slots;
    if (n_max < 1) n_max = 1;
    SplitEven best;
    best.T = -1;
    best.n_wg = 0;
    best.lag = 0;
    for (int pass = 0; pass < 2; pass++) {
        const long long n = pass == 0 ? n_max : c.n_first;
        if (pass == 1 && n_max <= c.n_first) break;   // (one workgroup per CU is a different cut only where both slots would be used)
        const long long n2 = n > c.n_first ? n - c.n_first : 0;
        long long lag = c.second_lag;
        long long T = (h_total + (long long)c.seg_cost * n + lag * n2 + n - 1) / n;
        if (T - c.seg_cost - lag < (long long)c.min_seg) {   // too light a list for the lag
            lag = 0;
            T = (h_total + (long long)c.seg_cost * n + n - 1) / n;
        }
        if (best.T < 0 || T < best.T) {
            best.T = T;
            best.n_wg = (int)n;
            best.lag = (int)lag;
        }
    }
    return best;
}
// where workgroup w starts: entry position in the list (0 .. all entries); *k_out / *o_out: its pair and the offset inside it
template <class RUNS>
__host__ __device__ inline int split_even_start(int w, const SplitEven& e, const RUNS& R, const SplitCfg& c, int* k_out, int* o_out) {
    if (w == 0) {
        *k_out = 0;
        *o_out = 0;
        return 0;
    }
    const long long cw = (long long)w * (e.T - c.seg_cost) - (long long)e.lag * (w > c.n_first ? w - c.n_first : 0);
    int lo = 0, hi = R.np - 1;   // the last pair whose start H has reached
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (R.pl_h[mid] <= cw)
            lo = mid;
        else
            hi = mid - 1;
    }
    int k = lo, o = 0;
    long long r = cw - R.pl_h[k] - c.seg_cost;   // what is left for the pair's entries behind its own segment cost
    if (r > 0)
        for (int i = R.pl_run0[k]; i < R.pl_run0[k + 1]; i++) {
            const int cnt = R.run_cnt[i], cst = R.run_cost[i];
            if ((long long)cnt * cst <= r) {
                o += cnt;
                r -= (long long)cnt * cst;
            } else {
                o += (int)(r / cst);
                break;
            }
        }
    o = o / 4 * 4;
    if (o < c.min_seg) o = 0;
    if (R.pl_n[k] - o < c.min_seg) {
        k++;
        o = 0;
    }
    *k_out = k;
    *o_out = o;
    return R.pl_e[k] + o;   // (k == np: the list's end — an empty workgroup)
}
// The host's form of the cut (the device builds the same segments from ranks, ba_prepare.inc: prep_split_even): emit(k, begin, end,
// wg) per segment in list order; returns the workgroups made.
template <class RUNS, class Emit>
inline int split_even_cut(const SplitEven& e, const RUNS& R, const SplitCfg& c, Emit&& emit) {
    const int ent = R.pl_e[R.np];
    int wg = -1, next_w = 0, n_wgs = 0;
    int next_start = 0;   // where the next workgroup that starts something begins
    auto advance = [&](int from_w) {   // the first w >= from_w whose start lies behind the current one
        int w = from_w, k, o;
        int cur = next_start;
        for (; w < e.n_wg; w++) {
            const int g = split_even_start(w, e, R, c, &k, &o);
            if (w == 0 || g > cur) {
                next_start = g;
                return w;
            }
        }
        next_start = ent;
        return e.n_wg;
    };
    next_w = advance(0);
    for (int k = 0; k < R.np; k++) {
        int pos = R.pl_e[k];
        const int end_k = R.pl_e[k + 1];
        while (pos < end_k) {
            if (next_w < e.n_wg && next_start == pos) {   // a workgroup starts here
                wg++;
                n_wgs++;
                next_w = advance(next_w + 1);
            }
            const int stop = next_w < e.n_wg && next_start < end_k ? next_start : end_k;
            emit(k, pos - R.pl_e[k], stop - R.pl_e[k], wg);
            pos = stop;
        }
    }
    return n_wgs;
}
