/* Zero-intelligence limit-order-book generator that emits LOBSTER-format message + orderbook rows.
 *
 * Host-side data tooling (plain C, no GPU): it produces the "synthetic 10-level SPY-shaped message
 * stream" BASELINE.json's configs name, in the *file format the reference ingests*
 * (reference rl4mm/database/populate_database.py:38-95 reads `time,type,id,size,price,direction`
 * rows plus one `ask_p,ask_s,bid_p,bid_s,...` book row per message), so the same rows can be fed to
 * the reference (through its sqlite test path), to oracle/ and to the HIP path.
 *
 * Model (assumptions, stated in DESIGN.md): tick 100, Poisson event arrivals, event mix
 * ~48% submit / 2% partial cancel / 44% delete / 6% execute with a mean-reverting live-order count,
 * sizes 100 x Geometric, submissions placed 1..10 ticks behind the opposite best (touch-weighted, with
 * a geometric tail; so the spread is mostly one tick and ~10 levels per side carry most of the depth),
 * executions sweep FIFO heads and emit one type-4 row per resting order touched, all with the same
 * timestamp.  Like LOBSTER's N-level files, only events that touch
 * one of the N visible levels of their side are emitted; deeper events still change the true book.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WIN 8192 /* price window in ticks around the initial mid */
#define NIL (-1)

typedef struct {
    int64_t id;
    int32_t size;
    int32_t pidx; /* price index in window */
    int32_t side; /* 0 buy 1 sell */
    int32_t prev, next;
    int32_t live_pos;
} Ord;

typedef struct {
    int32_t head, tail;
    int32_t n;
    int64_t vol;
} Lvl;

typedef struct {
    uint64_t s[4];
    Ord* pool;
    int32_t pool_cap, pool_n, free_head;
    int32_t* live;
    int32_t live_n;
    Lvl lv[2][WIN];
    int32_t best[2]; /* best price idx per side; buy: highest non-empty, sell: lowest non-empty; NIL if empty */
    int64_t base_price; /* price of index 0 */
    int64_t next_id;
    int32_t tick;
} Sim;

static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t rnd(Sim* S) {
    uint64_t* s = S->s;
    uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
}
static double unif(Sim* S) { return (double)(rnd(S) >> 11) * (1.0 / 9007199254740992.0); }
static int geom(Sim* S, double p) { /* >= 1 */
    int k = 1;
    while (unif(S) > p && k < 64) k++;
    return k;
}
static void seed_rng(Sim* S, uint64_t seed) {
    for (int i = 0; i < 4; i++) {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        S->s[i] = z ^ (z >> 31);
    }
}

static int32_t alloc_ord(Sim* S) {
    if (S->free_head != NIL) {
        int32_t i = S->free_head;
        S->free_head = S->pool[i].next;
        return i;
    }
    if (S->pool_n == S->pool_cap) {
        S->pool_cap *= 2;
        S->pool = (Ord*)realloc(S->pool, sizeof(Ord) * (size_t)S->pool_cap);
        S->live = (int32_t*)realloc(S->live, sizeof(int32_t) * (size_t)S->pool_cap);
    }
    return S->pool_n++;
}

static void fix_best_after_remove(Sim* S, int side) {
    int32_t b = S->best[side];
    if (b == NIL || S->lv[side][b].n > 0) return;
    if (side == 0) {
        while (b >= 0 && S->lv[0][b].n == 0) b--;
        S->best[0] = b >= 0 ? b : NIL;
    } else {
        while (b < WIN && S->lv[1][b].n == 0) b++;
        S->best[1] = b < WIN ? b : NIL;
    }
}

static int32_t add_order(Sim* S, int side, int32_t pidx, int32_t size) {
    int32_t i = alloc_ord(S);
    Ord* o = &S->pool[i];
    o->id = S->next_id;
    S->next_id += 1 + (int64_t)(rnd(S) % 3);
    o->size = size; o->pidx = pidx; o->side = side; o->next = NIL;
    Lvl* L = &S->lv[side][pidx];
    o->prev = L->tail;
    if (L->tail != NIL) S->pool[L->tail].next = i; else L->head = i;
    L->tail = i; L->n++; L->vol += size;
    o->live_pos = S->live_n; S->live[S->live_n++] = i;
    if (S->best[side] == NIL || (side == 0 ? pidx > S->best[0] : pidx < S->best[1])) S->best[side] = pidx;
    return i;
}

static void remove_order(Sim* S, int32_t i) {
    Ord* o = &S->pool[i];
    Lvl* L = &S->lv[o->side][o->pidx];
    if (o->prev != NIL) S->pool[o->prev].next = o->next; else L->head = o->next;
    if (o->next != NIL) S->pool[o->next].prev = o->prev; else L->tail = o->prev;
    L->n--; L->vol -= o->size;
    int32_t last = S->live[--S->live_n];
    S->live[o->live_pos] = last; S->pool[last].live_pos = o->live_pos;
    int side = o->side;
    o->next = S->free_head; S->free_head = i;
    fix_best_after_remove(S, side);
}

/* 1-based rank of price index among the non-empty levels of its side (counting from the best); large if
 * beyond `cap`. `pidx` itself need not be populated (rank it would have). */
static int level_rank(const Sim* S, int side, int32_t pidx, int cap) {
    int32_t b = S->best[side];
    if (b == NIL) return 1;
    int r = 1;
    if (side == 0) {
        if (pidx >= b) return 1;
        for (int32_t p = b; p > pidx; p--) if (S->lv[0][p].n > 0) { if (++r > cap) return r; }
    } else {
        if (pidx <= b) return 1;
        for (int32_t p = b; p < pidx; p++) if (S->lv[1][p].n > 0) { if (++r > cap) return r; }
    }
    return r;
}

static void write_book(const Sim* S, int n_levels, int64_t* row) {
    int32_t pa = S->best[1], pb = S->best[0];
    for (int l = 0; l < n_levels; l++) {
        while (pa != NIL && pa < WIN && S->lv[1][pa].n == 0) pa++;
        if (pa == NIL || pa >= WIN) { row[4 * l] = 9999999999ll; row[4 * l + 1] = 0; pa = NIL; }
        else { row[4 * l] = S->base_price + (int64_t)pa * S->tick; row[4 * l + 1] = S->lv[1][pa].vol; pa++; }
        while (pb != NIL && pb >= 0 && S->lv[0][pb].n == 0) pb--;
        if (pb == NIL || pb < 0) { row[4 * l + 2] = -9999999999ll; row[4 * l + 3] = 0; pb = NIL; }
        else { row[4 * l + 2] = S->base_price + (int64_t)pb * S->tick; row[4 * l + 3] = S->lv[0][pb].vol; pb--; }
    }
}

typedef struct {
    int64_t *time_ns, *type, *id, *size, *price, *dir, *books;
    int64_t n, max;
    int n_levels;
} Out;

static int emit(Sim* S, Out* O, int64_t t, int type, int64_t id, int64_t size, int64_t price, int dir) {
    if (O->n >= O->max) return 0;
    int64_t k = O->n++;
    O->time_ns[k] = t; O->type[k] = type; O->id[k] = id; O->size[k] = size; O->price[k] = price; O->dir[k] = dir;
    write_book(S, O->n_levels, O->books + k * 4 * (int64_t)O->n_levels);
    return 1;
}

/* Returns the number of message rows written (<= max_msgs). */
int64_t synth_lob_generate(uint64_t seed, int n_levels, double lambda, int64_t mid_price, int64_t start_ns,
                           int64_t end_ns, int64_t max_msgs, int64_t* time_ns, int64_t* type, int64_t* id,
                           int64_t* size, int64_t* price, int64_t* dir, int64_t* books) {
    Sim* S = (Sim*)calloc(1, sizeof(Sim));
    seed_rng(S, seed);
    S->tick = 100;
    S->pool_cap = 1 << 14;
    S->pool = (Ord*)malloc(sizeof(Ord) * (size_t)S->pool_cap);
    S->live = (int32_t*)malloc(sizeof(int32_t) * (size_t)S->pool_cap);
    S->free_head = NIL;
    S->best[0] = S->best[1] = NIL;
    for (int s = 0; s < 2; s++) for (int p = 0; p < WIN; p++) S->lv[s][p].head = S->lv[s][p].tail = NIL;
    S->base_price = (mid_price / 100) * 100 - (int64_t)(WIN / 2) * 100;
    S->next_id = 10000000 + (int64_t)(seed % 1000) * 7;
    const int bid0 = WIN / 2, ask0 = WIN / 2 + 1;
    const int init_levels = 40;
    for (int l = 0; l < init_levels; l++) {
        int nb = 18 + (int)(rnd(S) % 16), na = 18 + (int)(rnd(S) % 16);
        if (l > 15) { nb = 4 + (int)(rnd(S) % 8); na = 4 + (int)(rnd(S) % 8); }
        for (int k = 0; k < nb; k++) add_order(S, 0, bid0 - l, 100 * geom(S, 0.6));
        for (int k = 0; k < na; k++) add_order(S, 1, ask0 + l, 100 * geom(S, 0.6));
    }
    const int target_live = S->live_n;
    Out O = {time_ns, type, id, size, price, dir, books, 0, max_msgs, n_levels};
    int64_t t = start_ns;
    while (O.n < max_msgs) {
        double u = unif(S);
        if (u < 1e-300) u = 1e-300;
        int64_t dt = (int64_t)llround(-log(u) / lambda * 1e9);
        if (dt < 1) dt = 1;
        t += dt;
        if (t >= end_ns) break;
        if (S->best[0] == NIL || S->best[1] == NIL || S->best[0] < 200 || S->best[1] > WIN - 200) break;
        double p_submit = 0.48 + 0.25 * (double)(target_live - S->live_n) / (double)target_live;
        if (p_submit < 0.05) p_submit = 0.05;
        if (p_submit > 0.90) p_submit = 0.90;
        double rest = 1.0 - p_submit;
        double p_exec = rest * (3.5 / 52.0), p_canc = rest * (5.0 / 52.0);
        double r = unif(S);
        if (r < p_submit) {
            int side = (int)(rnd(S) & 1);
            /* touch replenishment thins out as the touch fills up (keeps ~5k shares there and lets it deplete) */
            int32_t p1 = side == 0 ? S->best[1] - 1 : S->best[0] + 1, p2 = side == 0 ? p1 - 1 : p1 + 1;
            double v1 = (double)S->lv[side][p1].vol, v2 = (double)S->lv[side][p2].vol;
            double q1 = 0.04 + 0.26 * (v1 < 6000 ? 1.0 - v1 / 6000 : 0.0), q2 = 0.05 + 0.10 * (v2 < 6000 ? 1.0 - v2 / 6000 : 0.0);
            double ud = unif(S);
            int d = ud < q1 ? 1 : ud < q1 + q2 ? 2 : ud < 0.88 ? 3 + (int)(rnd(S) % 8) : 10 + geom(S, 0.3);
            if (d > 40) d = 40;
            int32_t pidx = side == 0 ? S->best[1] - d : S->best[0] + d;
            int32_t sz = 100 * geom(S, 0.6);
            int32_t i = add_order(S, side, pidx, sz);
            if (level_rank(S, side, pidx, n_levels) <= n_levels)
                if (!emit(S, &O, t, 1, S->pool[i].id, sz, S->base_price + (int64_t)pidx * 100, side == 0 ? 1 : -1)) break;
        } else if (r < p_submit + p_exec) {
            int aggr = (int)(rnd(S) & 1); /* 0: buyer initiated -> consumes asks */
            int rs = aggr == 0 ? 1 : 0;  /* resting side */
            int64_t rem = 100 * (int64_t)geom(S, 0.5);
            if (unif(S) < 0.04) rem *= 25; /* occasional sweep: this is what moves the price */
            int touched = 0;
            while (rem > 0 && S->best[rs] != NIL && touched < 40) {
                Lvl* L = &S->lv[rs][S->best[rs]];
                int32_t h = L->head;
                Ord* o = &S->pool[h];
                int64_t ex = rem < o->size ? rem : o->size;
                int64_t oid = o->id, pr = S->base_price + (int64_t)o->pidx * 100;
                if (ex == o->size) remove_order(S, h); else { o->size -= (int32_t)ex; L->vol -= ex; }
                rem -= ex; touched++;
                /* the head of the best level is always visible */
                if (!emit(S, &O, t, 4, oid, ex, pr, rs == 0 ? 1 : -1)) { rem = 0; break; }
            }
        } else {
            if (S->live_n == 0) continue;
            int32_t i = S->live[rnd(S) % (uint64_t)S->live_n];
            Ord* o = &S->pool[i];
            int vis = level_rank(S, o->side, o->pidx, n_levels) <= n_levels;
            int64_t oid = o->id, pr = S->base_price + (int64_t)o->pidx * 100;
            int d = o->side == 0 ? 1 : -1;
            if (r < p_submit + p_exec + p_canc && o->size > 100) {
                int32_t cut = 100 * (1 + (int32_t)(rnd(S) % (uint64_t)(o->size / 100 - 1 > 0 ? o->size / 100 - 1 : 1)));
                if (cut >= o->size) cut = o->size - 100;
                o->size -= cut; S->lv[o->side][o->pidx].vol -= cut;
                if (vis) if (!emit(S, &O, t, 2, oid, cut, pr, d)) break;
            } else {
                int64_t sz = o->size;
                remove_order(S, i);
                if (vis) if (!emit(S, &O, t, 3, oid, sz, pr, d)) break;
            }
        }
    }
    int64_t n = O.n;
    free(S->pool); free(S->live); free(S);
    return n;
}
