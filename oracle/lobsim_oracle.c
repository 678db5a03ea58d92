/* lobsim_oracle.c - CPU restatement of rl4mm's LOB simulation step.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product path
 * (rl4mm_amd/, include/lobsim.h's lobsim_* entry points) never does.
 *
 * It follows the reference's *algorithm and data structures* function by function (sorted price map of FIFO
 * deques, a second "internal" book for the agent, an external->internal id map, Python-order float arithmetic),
 * deliberately NOT the GPU data layout, so that the two implementations fail independently.  Each function
 * cites the reference file:line it restates (paths relative to /root/reference).
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against golden vectors emitted by the reference
 * itself (oracle/gen_golden.py, run in the build container) - engine scenarios of
 * rl4mm/orderbook/tests/testExchange.py, the LOBSTER-agreement replay of
 * rl4mm/simulation/tests/testOrderbookSimulator.py, full env trajectories on the MSFT test slice and on
 * synthetic streams, and a scipy Beta-volume table.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp (see oracle/Makefile); contraction is off so every double
 * operation rounds exactly like CPython's.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/lobsim.h"

#define DUMMY_PRICE 9999999999ll
#define IID_NONE 0
#define VOL_NONE (-1)

/* ------------------------------------------------------------------------------------------------------------
 * Order records (rl4mm/orderbook/models.py:17-61)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t iid;    /* internal_id; -1 for snapshot orders */
    int64_t ext_id; /* external_id (unused for agent / snapshot orders) */
    int64_t vol;
    int is_ext;
    int64_t tag; /* agent orders: the submission tag - the id a caller of oracle_forward gave the order (row id > 0: rl4mm_amd/simulator.py names its
                    orders by it), else the next value of the env's own counter; lobsim_order_view.reserved */
} Resting;

typedef struct {
    int type; /* lobsim_order_type: LIMIT, CANCELLATION, DELETION, MARKET */
    int side;
    int64_t price;
    int64_t vol; /* VOL_NONE = None */
    int64_t iid; /* IID_NONE = None */
    int64_t ext_id;
    int is_ext;
    int64_t tag; /* agent limit orders: the caller's id for it (0: none) */
} Order;

typedef struct {
    int64_t price;
    Resting* q; /* deque, index 0 = head */
    int n, cap;
} Level;

typedef struct {
    Level* lv; /* SortedDict: ascending price */
    int n, cap;
} Side;

typedef struct {
    Side s[2];
} Book; /* models.py:64-69 */

typedef struct {
    int side;
    int64_t price, vol;
} Fill;

typedef struct {
    Fill* f;
    int n, cap;
} FillList;

/* OrderIdConvertor (rl4mm/orderbook/OrderIDConvertor.py:7-37): dict external_id -> internal_id + counter */
typedef struct {
    int64_t* key;
    int64_t* val;
    uint8_t* st; /* 0 empty 1 used 2 deleted */
    int64_t cap, n_used, n_filled;
    int64_t counter;
} IdMap;

typedef struct {
    Book central, internal; /* Exchange.py:41-46 */
    IdMap ids;
} Exchange;

/* ---- feature state (rl4mm/features/Features.py) ---- */
typedef struct {
    lobsim_feature d;
    double current_value;
    int64_t first_usage_us;
    double* prices; /* deque contents, oldest first for Volatility; newest first for PriceMove/PriceRange */
    int n_prices;
    int64_t *buys, *sells; /* trade imbalance deques */
    int n_trades;
    int64_t total, diff;
    /* AmihudLambda (Features.py:245-324) */
    int64_t* dvols; /* dollar_volumes deque */
    int n_dvols;
    double partial_price_sum;
    int64_t partial_dollar_volume;
    int steps_until_update;
    /* normalisation history, deque(maxlen=max_norm_len) (Features.py:60-61): oldest first */
    double* hist;
    int64_t n_hist;
} Feature;

typedef struct {
    const int64_t* msgs;
    int64_t n_msgs;
    const int64_t* snaps;
    int64_t n_snaps;
    int64_t end_sec; /* first whole second (since midnight) that lies 2 s or more beyond the last message / snapshot */
} Stream;

typedef struct {
    Exchange ex;
    int stream_id;
    int64_t now_us, terminal_us;
    /* OrderbookSimulator.py:49-53 */
    double min_buy_price, max_sell_price; /* ints or inf in the reference: hold as double (exact below 2^53) */
    double initial_buy_range, initial_sell_range;
    /* State / Portfolio (Features.py:24-36) */
    int64_t inventory, cash;
    double price;
    FillList fin, fex; /* state.filled_orders.internal / .external of the last step */
    int hw_queue, hw_agent, hw_levels[2]; /* high-water marks since creation: orders at one price, live agent orders, levels per side
                                             (the reference's containers are unbounded; the HIP layout's capacities are checked against these) */
    Feature feat[LOBSIM_MAX_FEATURES];
    int status;
    int n_resyncs;
    int64_t agent_seq; /* next submission tag of an agent order that names none (the HIP layout's AgentOrd.seq) */
    double prev_action[LOBSIM_MAX_ACTION_DIM];
    int64_t cash_delta_dbg;
    /* RollingSharpe state (RewardFunctions.py:50-59), [0] per-step instance, [1] terminal instance; never reset */
    double rs_aum[2][LOBSIM_SHARPE_MAX_WINDOW];
    int rs_filled[2];
} Env;

struct oracle {
    lobsim_cfg cfg;
    Env* env;
    Stream* streams;
    int n_streams;
    char err[256];
};
typedef struct oracle oracle_t;

/* ------------------------------------------------------------------------------------------------------------
 * containers
 * ---------------------------------------------------------------------------------------------------------- */
static void fills_push(FillList* L, int side, int64_t price, int64_t vol) {
    if (L->n == L->cap) {
        L->cap = L->cap ? 2 * L->cap : 16;
        L->f = (Fill*)realloc(L->f, sizeof(Fill) * (size_t)L->cap);
    }
    L->f[L->n].side = side;
    L->f[L->n].price = price;
    L->f[L->n].vol = vol;
    L->n++;
}

static uint64_t hash64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
static void idmap_init(IdMap* m) {
    m->cap = 1024;
    m->key = (int64_t*)calloc((size_t)m->cap, 8);
    m->val = (int64_t*)calloc((size_t)m->cap, 8);
    m->st = (uint8_t*)calloc((size_t)m->cap, 1);
    m->n_used = m->n_filled = 0;
    m->counter = 0;
}
static void idmap_free(IdMap* m) { free(m->key); free(m->val); free(m->st); }
static void idmap_put(IdMap* m, int64_t k, int64_t v);
static void idmap_grow(IdMap* m) {
    IdMap o = *m;
    m->cap = o.cap * 2;
    m->key = (int64_t*)calloc((size_t)m->cap, 8);
    m->val = (int64_t*)calloc((size_t)m->cap, 8);
    m->st = (uint8_t*)calloc((size_t)m->cap, 1);
    m->n_used = m->n_filled = 0;
    for (int64_t i = 0; i < o.cap; i++) if (o.st[i] == 1) idmap_put(m, o.key[i], o.val[i]);
    free(o.key); free(o.val); free(o.st);
}
static void idmap_put(IdMap* m, int64_t k, int64_t v) {
    if ((m->n_filled + 1) * 2 > m->cap) idmap_grow(m);
    int64_t i = (int64_t)(hash64((uint64_t)k) & (uint64_t)(m->cap - 1)), first_del = -1;
    while (m->st[i] != 0) {
        if (m->st[i] == 1 && m->key[i] == k) { m->val[i] = v; return; }
        if (m->st[i] == 2 && first_del < 0) first_del = i;
        i = (i + 1) & (m->cap - 1);
    }
    if (first_del >= 0) i = first_del; else m->n_filled++;
    m->st[i] = 1; m->key[i] = k; m->val[i] = v; m->n_used++;
}
static int idmap_get(const IdMap* m, int64_t k, int64_t* v) {
    int64_t i = (int64_t)(hash64((uint64_t)k) & (uint64_t)(m->cap - 1));
    while (m->st[i] != 0) {
        if (m->st[i] == 1 && m->key[i] == k) { *v = m->val[i]; return 1; }
        i = (i + 1) & (m->cap - 1);
    }
    return 0;
}
static void idmap_del(IdMap* m, int64_t k) {
    int64_t i = (int64_t)(hash64((uint64_t)k) & (uint64_t)(m->cap - 1));
    while (m->st[i] != 0) {
        if (m->st[i] == 1 && m->key[i] == k) { m->st[i] = 2; m->n_used--; return; }
        i = (i + 1) & (m->cap - 1);
    }
}

static void side_clear(Side* s) {
    for (int i = 0; i < s->n; i++) free(s->lv[i].q);
    s->n = 0;
}
static void book_clear(Book* b) { side_clear(&b->s[0]); side_clear(&b->s[1]); }
static void book_free(Book* b) { book_clear(b); free(b->s[0].lv); free(b->s[1].lv); b->s[0].lv = b->s[1].lv = NULL; b->s[0].cap = b->s[1].cap = 0; }

/* position of first level with price >= p */
static int side_lower(const Side* s, int64_t p) {
    int lo = 0, hi = s->n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (s->lv[mid].price < p) lo = mid + 1; else hi = mid; }
    return lo;
}
static Level* side_find(Side* s, int64_t p) {
    int i = side_lower(s, p);
    return (i < s->n && s->lv[i].price == p) ? &s->lv[i] : NULL;
}
static Level* side_insert(Side* s, int64_t p) { /* new empty level at price p (must not exist) */
    int i = side_lower(s, p);
    if (s->n == s->cap) { s->cap = s->cap ? 2 * s->cap : 64; s->lv = (Level*)realloc(s->lv, sizeof(Level) * (size_t)s->cap); }
    memmove(&s->lv[i + 1], &s->lv[i], sizeof(Level) * (size_t)(s->n - i));
    s->n++;
    s->lv[i].price = p; s->lv[i].q = NULL; s->lv[i].n = 0; s->lv[i].cap = 0;
    return &s->lv[i];
}
static void side_pop(Side* s, int64_t p) {
    int i = side_lower(s, p);
    if (i >= s->n || s->lv[i].price != p) return;
    free(s->lv[i].q);
    memmove(&s->lv[i], &s->lv[i + 1], sizeof(Level) * (size_t)(s->n - i - 1));
    s->n--;
}
static void level_append(Level* l, Resting r) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 8; l->q = (Resting*)realloc(l->q, sizeof(Resting) * (size_t)l->cap); }
    l->q[l->n++] = r;
}
static void level_erase(Level* l, int pos) {
    memmove(&l->q[pos], &l->q[pos + 1], sizeof(Resting) * (size_t)(l->n - pos - 1));
    l->n--;
}
static int64_t level_volume(const Level* l) { int64_t v = 0; for (int i = 0; i < l->n; i++) v += l->q[i].vol; return v; }

/* ------------------------------------------------------------------------------------------------------------
 * Orderbook properties (models.py:71-101) - prices as double because the reference mixes int, 0 and np.infty
 * ---------------------------------------------------------------------------------------------------------- */
static double book_best_buy(const Book* b) { return b->s[0].n ? (double)b->s[0].lv[b->s[0].n - 1].price : 0.0; }       /* :72-73 */
static double book_best_sell(const Book* b) { return b->s[1].n ? (double)b->s[1].lv[0].price : INFINITY; }            /* :76-77 */
static int book_microprice(const Book* b, double* out) {                                                               /* :80-97 */
    if (!b->s[0].n || !b->s[1].n) return 0; /* KeyError on self.buy[0] / self.sell[inf] */
    int64_t bv = level_volume(&b->s[0].lv[b->s[0].n - 1]), av = level_volume(&b->s[1].lv[0]);
    if (bv + av == 0) return 0; /* ZeroDivisionError */
    double imb = (double)(bv - av) / (double)(bv + av);
    double ask = book_best_sell(b), bid = book_best_buy(b);
    *out = (1 + imb) / 2 * ask + (1 - imb) / 2 * bid;
    return 1;
}
static int book_imbalance(const Book* b, double* out) {
    if (!b->s[0].n || !b->s[1].n) return 0;
    int64_t bv = level_volume(&b->s[0].lv[b->s[0].n - 1]), av = level_volume(&b->s[1].lv[0]);
    if (bv + av == 0) return 0;
    *out = (double)(bv - av) / (double)(bv + av);
    return 1;
}

/* ------------------------------------------------------------------------------------------------------------
 * Exchange (rl4mm/orderbook/Exchange.py)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    Env* env;
    FillList *fin, *fex;
} Ctx;

static void process_order(Ctx* c, Order* o);
static double np_pairwise_sum(const double* a, int n);
static void submit_order(Ctx* c, Order* o);

static int crosses(const Exchange* ex, const Order* o) { /* _does_order_cross_spread, Exchange.py:188-194 */
    if (o->type == LOBSIM_MARKET) return 1;
    if (o->side == LOBSIM_BUY) return (double)o->price >= book_best_sell(&ex->central);
    return (double)o->price <= book_best_buy(&ex->central);
}

/* _clear_empty_orders_and_prices, Exchange.py:238-247 */
static void clear_empty(Exchange* ex, Book* b, int side, int64_t price, int qpos) {
    Level* l = side_find(&b->s[side], price);
    if (l->q[qpos].vol == 0) {
        if (l->q[qpos].is_ext && l->q[qpos].iid != -1) idmap_del(&ex->ids, l->q[qpos].ext_id); /* snapshot orders carry external_id None */
        level_erase(l, qpos);
    }
    if (l->n == 0) side_pop(&b->s[side], price);
}

/* _reduce_order_with_queue_position, Exchange.py:219-236.  Returns 0 on CancellationVolumeExceededError. */
static int reduce_at(Exchange* ex, Book* b, int side, int64_t price, int qpos, int64_t vol, Resting* removed) {
    Level* l = side_find(&b->s[side], price);
    if (vol > l->q[qpos].vol) return 0;
    if (removed) { *removed = l->q[qpos]; removed->vol = vol; }
    l->q[qpos].vol -= vol;
    clear_empty(ex, b, side, price, qpos);
    return 1;
}

/* _find_queue_position, Exchange.py:196-217.  -1 = None. */
static int find_queue_position(Exchange* ex, const Order* o, Book* b) {
    int64_t iid = o->iid;
    int have = iid != IID_NONE;
    if (!have) { /* `order.internal_id or convertor.get_internal_order_id(order)` */
        if (!o->is_ext) have = 0; /* returns order.internal_id (None) */
        else have = idmap_get(&ex->ids, o->ext_id, &iid);
    }
    if (!have && o->is_ext) return -1;
    Level* l = side_find(&b->s[o->side], o->price);
    if (!l) return -1;
    int left = 0, right = l->n - 1;
    while (left <= right) {
        int mid = (left + right) / 2;
        int64_t mid_id = l->q[mid].iid;
        if (mid_id == iid) return mid;
        if (mid_id < iid) left = mid + 1; else right = mid - 1;
    }
    return -1;
}

/* remove_order, Exchange.py:122-147 */
static void remove_order(Ctx* c, Order* o) {
    Exchange* ex = &c->env->ex;
    Book* books[2] = {&ex->central, &ex->internal};
    int nb = o->is_ext ? 1 : 2;
    for (int k = 0; k < nb; k++) {
        Book* b = books[k];
        int qpos = find_queue_position(ex, o, b);
        if (qpos < 0) {
            Level* l = side_find(&b->s[o->side], o->price);
            if (!l) continue; /* KeyError */
            if (l->q[0].iid == -1) { /* initial orders remain in book */
                if (o->vol == VOL_NONE) { c->env->status |= LOBSIM_ST_ASSERT; return; }
                o->iid = -1;
                qpos = 0;
            } else continue;
        } else if (o->vol == VOL_NONE) { /* isinstance(order, Deletion) and order.volume is None */
            o->vol = side_find(&b->s[o->side], o->price)->q[qpos].vol;
        }
        if (!reduce_at(ex, b, o->side, o->price, qpos, o->vol, NULL)) {
            int64_t all = side_find(&b->s[o->side], o->price)->q[qpos].vol;
            reduce_at(ex, b, o->side, o->price, qpos, all, NULL);
        }
    }
}

/* execute_order, Exchange.py:85-120 */
static void execute_order(Ctx* c, Order* o) {
    Env* e = c->env;
    Exchange* ex = &e->ex;
    int64_t rem = o->vol;
    while (rem > 0 && crosses(ex, o)) {
        int opp = o->side == LOBSIM_BUY ? LOBSIM_SELL : LOBSIM_BUY;
        Side* os = &ex->central.s[opp];
        if (os->n == 0) { e->status |= LOBSIM_ST_EMPTY_BOOK; return; } /* EmptyOrderbookError, :180-186 */
        Level* bl = opp == LOBSIM_SELL ? &os->lv[0] : &os->lv[os->n - 1];
        Resting best = bl->q[0];
        int64_t bprice = bl->price;
        if (!o->is_ext && !best.is_ext) { /* cannot fill our own order: delete it (:91-94) */
            Order d = {LOBSIM_DELETION, opp, bprice, best.vol, best.iid, best.ext_id, 0, 0};
            process_order(c, &d);
            if (e->status) return;
            continue;
        }
        int64_t v = rem < best.vol ? rem : best.vol;
        Resting ex_order;
        if (!best.is_ext) reduce_at(ex, &ex->internal, opp, bprice, 0, v, &ex_order);
        reduce_at(ex, &ex->central, opp, bprice, 0, v, &ex_order);
        if (ex_order.is_ext) fills_push(c->fex, opp, bprice, v);
        else fills_push(c->fin, opp, bprice, v);
        rem -= v;
        if (!o->is_ext) fills_push(c->fin, o->side, bprice, v); /* the agent's own aggressing fill (:110-115) */
    }
    if (rem > 0 && o->type == LOBSIM_LIMIT) {
        Order r = *o;
        r.vol = rem;
        submit_order(c, &r);
    }
}

/* submit_order, Exchange.py:71-83 */
static void submit_order(Ctx* c, Order* o) {
    Exchange* ex = &c->env->ex;
    if (crosses(ex, o)) { execute_order(c, o); return; }
    ex->ids.counter += 1; /* OrderIDConvertor.py:12-19 */
    Resting r = {ex->ids.counter, o->ext_id, o->vol, o->is_ext, 0};
    if (!o->is_ext) { /* (the tag is bookkeeping of the test interface, not of the reference: it orders nothing) */
        Env* en = c->env;
        const int64_t sq = o->tag > 0 ? o->tag : en->agent_seq;
        r.tag = sq & 0x7fffffff;
        en->agent_seq = (sq > en->agent_seq ? sq : en->agent_seq) + 1;
    }
    if (o->is_ext) idmap_put(&ex->ids, o->ext_id, r.iid);
    Book* books[2] = {&ex->central, &ex->internal};
    int nb = o->is_ext ? 1 : 2;
    for (int k = 0; k < nb; k++) {
        Level* l = side_find(&books[k]->s[o->side], o->price);
        if (!l) l = side_insert(&books[k]->s[o->side], o->price);
        level_append(l, r);
        if (k == 0 && l->n > c->env->hw_queue) c->env->hw_queue = l->n;
    }
    {
        Env* e = c->env;
        if (ex->central.s[o->side].n > e->hw_levels[o->side]) e->hw_levels[o->side] = ex->central.s[o->side].n;
        if (!o->is_ext) {
            int na = 0;
            for (int sd = 0; sd < 2; sd++) for (int i = 0; i < ex->internal.s[sd].n; i++) na += ex->internal.s[sd].lv[i].n;
            if (na > e->hw_agent) e->hw_agent = na;
        }
    }
}

/* process_order, Exchange.py:58-69 */
static void process_order(Ctx* c, Order* o) {
    if (o->vol != VOL_NONE && !(o->vol > 0)) { c->env->status |= LOBSIM_ST_ASSERT; return; }
    if (o->type == LOBSIM_LIMIT) submit_order(c, o);
    else if (o->type == LOBSIM_MARKET) execute_order(c, o);
    else if (o->type == LOBSIM_CANCELLATION || o->type == LOBSIM_DELETION) remove_order(c, o);
    else c->env->status |= LOBSIM_ST_BAD_MESSAGE;
}

/* orderbook_price_range, Exchange.py:158-169.  0 on StopIteration. */
static int price_range(const Exchange* ex, int64_t* worst_buy, int64_t* worst_sell) {
    const Side* ss = &ex->central.s[1];
    int i = ss->n - 1;
    while (i >= 0 && ss->lv[i].price >= DUMMY_PRICE) i--;
    if (i < 0) return 0;
    *worst_sell = ss->lv[i].price;
    const Side* bs = &ex->central.s[0];
    int j = 0;
    while (j < bs->n && bs->lv[j].price <= 0) j++;
    if (j >= bs->n) return 0;
    *worst_buy = bs->lv[j].price;
    return 1;
}

/* ------------------------------------------------------------------------------------------------------------
 * OrderbookSimulator (rl4mm/simulation/OrderbookSimulator.py)
 * ---------------------------------------------------------------------------------------------------------- */
static int64_t last_snapshot(const Stream* st, int L, int64_t t_us) { /* HistoricalDatabase.get_last_snapshot:46-62 */
    int64_t lo = 0, hi = st->n_snaps, w = 2 + 4 * (int64_t)L;
    while (lo < hi) { int64_t mid = (lo + hi) / 2; if (st->snaps[mid * w] <= t_us) lo = mid + 1; else hi = mid; }
    return lo - 1;
}
static int64_t first_msg_after(const Stream* st, int64_t t_us) { /* get_messages: timestamp > start, :109 */
    int64_t lo = 0, hi = st->n_msgs;
    while (lo < hi) { int64_t mid = (lo + hi) / 2; if (st->msgs[mid * 4] <= t_us) lo = mid + 1; else hi = mid; }
    return lo;
}

/* snapshot row -> one aggregated LimitOrder(internal_id=-1) per level; buy levels first, then sell
 * (_get_initial_orders_from_snapshot, :156-175).  kind: 0 = all, 1 = outside [min_buy, max_sell] filter (:99-103). */
static int snapshot_orders(const Env* e, const int64_t* row, int L, int filter, int* sides, int64_t* prices, int64_t* vols) {
    int n = 0;
    for (int side = 0; side < 2; side++)
        for (int l = 0; l < L; l++) {
            int64_t p = side == LOBSIM_BUY ? row[4 * l + 2] : row[4 * l];
            int64_t v = side == LOBSIM_BUY ? row[4 * l + 3] : row[4 * l + 1];
            if (filter) {
                int pass = (side == LOBSIM_BUY && (double)p < e->min_buy_price) || (side == LOBSIM_SELL && (double)p > e->max_sell_price);
                if (!pass) continue;
            }
            sides[n] = side; prices[n] = p; vols[n] = v; n++;
        }
    return n;
}

static void set_level_to_snapshot(Book* b, int side, int64_t price, int64_t vol) { /* book[direction][price] = deque([order]) */
    Level* l = side_find(&b->s[side], price);
    if (!l) l = side_insert(&b->s[side], price);
    l->n = 0;
    Resting r = {-1, 0, vol, 1, 0};
    level_append(l, r);
}

static void reset_price_ranges(Env* e) { /* _reset_initial_price_ranges, :185-188 */
    int64_t wb, ws;
    if (!price_range(&e->ex, &wb, &ws)) { e->status |= LOBSIM_ST_EMPTY_BOOK; return; }
    e->min_buy_price = (double)wb; e->max_sell_price = (double)ws;
    e->initial_buy_range = book_best_buy(&e->ex.central) - e->min_buy_price;
    e->initial_sell_range = e->max_sell_price - book_best_sell(&e->ex.central);
}

static int near_exit(const oracle_t* h, const Env* e) { /* _near_exiting_initial_price_range, :177-183 */
    double prop = (double)h->cfg.outer_levels / (double)h->cfg.n_levels;
    return book_best_buy(&e->ex.central) < e->min_buy_price + prop * e->initial_buy_range ||
           book_best_sell(&e->ex.central) > e->max_sell_price - prop * e->initial_sell_range;
}

/* update_outer_levels, :105-135 */
static void update_outer_levels(oracle_t* h, Env* e, Ctx* c) {
    const Stream* st = &h->streams[e->stream_id];
    int L = h->cfg.n_levels;
    e->n_resyncs++;
    int64_t si = last_snapshot(st, L, e->now_us);
    int n = 0;
    int* sides = (int*)malloc(sizeof(int) * 2 * (size_t)L);
    int64_t* prices = (int64_t*)malloc(8 * 2 * (size_t)L);
    int64_t* vols = (int64_t*)malloc(8 * 2 * (size_t)L);
    if (si >= 0) n = snapshot_orders(e, st->snaps + si * (2 + 4 * (int64_t)L) + 2, L, 1, sides, prices, vols);
    Order* replace = NULL;
    int n_rep = 0, cap_rep = 0;
    for (int i = 0; i < n && !e->status; i++) {
        Level* il = side_find(&e->ex.internal.s[sides[i]], prices[i]);
        if (il) {
            int m = il->n;
            Order* canc = (Order*)malloc(sizeof(Order) * (size_t)m);
            for (int k = 0; k < m; k++) {
                Order o = {LOBSIM_CANCELLATION, sides[i], prices[i], il->q[k].vol, il->q[k].iid, il->q[k].ext_id, 0, 0};  /* (re-added under a fresh tag) */
                canc[k] = o;
                if (n_rep == cap_rep) { cap_rep = cap_rep ? 2 * cap_rep : 16; replace = (Order*)realloc(replace, sizeof(Order) * (size_t)cap_rep); }
                o.type = LOBSIM_LIMIT;
                replace[n_rep++] = o;
            }
            for (int k = 0; k < m && !e->status; k++) process_order(c, &canc[k]);
            free(canc);
        }
        set_level_to_snapshot(&e->ex.central, sides[i], prices[i], vols[i]);
    }
    for (int k = 0; k < n_rep && !e->status; k++) process_order(c, &replace[k]);
    free(replace); free(sides); free(prices); free(vols);
    if (e->status) return;
    int64_t wb, ws;
    if (!price_range(&e->ex, &wb, &ws)) { e->status |= LOBSIM_ST_EMPTY_BOOK; return; }
    if ((double)wb < e->min_buy_price) e->min_buy_price = (double)wb;
    if ((double)ws > e->max_sell_price) e->max_sell_price = (double)ws;
}

/* get_order_from_external_message + create_order (HistoricalOrderGenerator.py:77-90, create_order.py:4-42) */
static Order order_from_message(const int64_t* m) {
    Order o;
    o.type = (int)((m[3] >> LOBSIM_TYPE_SHIFT) & 0xF);
    o.side = (int)((m[3] >> LOBSIM_SIDE_SHIFT) & 1);
    o.price = m[2];
    o.vol = m[3] & LOBSIM_VOL_MASK;
    o.iid = IID_NONE;
    o.ext_id = m[1];
    o.is_ext = 1;
    return o;
}

/* forward_step, :70-88: agent orders first, then the window (now, until] of historical messages */
static void forward_step_impl(oracle_t* h, Env* e, int64_t until, Order* internal, int n_internal, int keep_fills) {
    Ctx c = {e, &e->fin, &e->fex};
    if (!keep_fills) e->fin.n = e->fex.n = 0;
    for (int i = 0; i < n_internal && !e->status; i++) process_order(&c, &internal[i]);
    if (e->stream_id >= 0) {
        const Stream* st = &h->streams[e->stream_id];
        for (int64_t k = first_msg_after(st, e->now_us); k < st->n_msgs && st->msgs[k * 4] <= until && !e->status; k++) {
            Order o = order_from_message(st->msgs + k * 4);
            if (o.type == LOBSIM_CROSS_TRADE || o.type == LOBSIM_TRADING_HALT) { e->status |= LOBSIM_ST_BAD_MESSAGE; break; }
            process_order(&c, &o);
        }
    }
    e->now_us = until;
    if (e->status) return;
    /* not a reference behaviour but the boundary's (lobsim.h, LOBSIM_ST_STREAM_END): a step that ends two seconds or more
     * beyond the loaded data is a usage error, flagged instead of silently replaying nothing */
    if (e->stream_id >= 0 && until / 1000000 >= h->streams[e->stream_id].end_sec) { e->status |= LOBSIM_ST_STREAM_END; return; }
    if (e->stream_id >= 0 && near_exit(h, e) && e->now_us % 1000000 == 0) update_outer_levels(h, e, &c);
}

static void forward_step(oracle_t* h, Env* e, int64_t until, Order* internal, int n_internal) { forward_step_impl(h, e, until, internal, n_internal, 0); }
/* (oracle_forward: the internal orders of the call - whose cancellations name their target by FIFO rank at that moment - were
 * processed by the caller, their fills already in the lists) */
static void forward_step_keep_fills(oracle_t* h, Env* e, int64_t until) { forward_step_impl(h, e, until, NULL, 0, 1); }

/* reset_episode + get_historical_start_book, :55-68, 90-97 */
static void reset_episode(oracle_t* h, Env* e, int64_t start_us, const int64_t* explicit_row) {
    int L = h->cfg.n_levels;
    book_clear(&e->ex.central);
    book_clear(&e->ex.internal);
    e->agent_seq = 0;
    const int64_t* row = explicit_row;
    if (!row && e->stream_id >= 0) {
        const Stream* st = &h->streams[e->stream_id];
        int64_t si = last_snapshot(st, L, start_us);
        if (si < 0) { e->status |= LOBSIM_ST_STREAM_END; return; }
        row = st->snaps + si * (2 + 4 * (int64_t)L) + 2;
    }
    if (row) {
        int* sides = (int*)malloc(sizeof(int) * 2 * (size_t)L);
        int64_t* prices = (int64_t*)malloc(8 * 2 * (size_t)L);
        int64_t* vols = (int64_t*)malloc(8 * 2 * (size_t)L);
        int n = snapshot_orders(e, row, L, 0, sides, prices, vols);
        for (int i = 0; i < n; i++) {
            if (explicit_row && prices[i] == 0) continue; /* debug_set_book: price 0 marks an absent level */
            set_level_to_snapshot(&e->ex.central, sides[i], prices[i], vols[i]);
        }
        free(sides); free(prices); free(vols);
        reset_price_ranges(e);
    }
    e->now_us = start_us;
}

/* ------------------------------------------------------------------------------------------------------------
 * Features (rl4mm/features/Features.py)
 * ---------------------------------------------------------------------------------------------------------- */
/* Feature.clamp, :99-100: max(min(number, max_value), min_value) with PYTHON's min / max - the first argument unless the second is
 * strictly smaller / larger - so a NaN (the z-score of a constant history) comes back as NaN, not as a bound: the features whose
 * update reads current_value (Volatility's running sum, AmihudLambda between slow updates) carry it on */
static double clampd(double x, double lo, double hi) { double m = hi < x ? hi : x; return lo > m ? lo : m; }
static double (*volatile libm_pow)(double, double) = pow;

static void feature_alloc(Feature* f) {
    int lb = f->d.lookback;
    f->prices = (double*)calloc((size_t)lb + 2, sizeof(double));
    f->buys = (int64_t*)calloc((size_t)lb + 1, 8);
    f->sells = (int64_t*)calloc((size_t)lb + 1, 8);
    f->dvols = (int64_t*)calloc((size_t)lb + 1, 8);
    f->hist = f->d.norm_len > 0 ? (double*)calloc((size_t)f->d.norm_len, sizeof(double)) : NULL;
}
static void feature_free(Feature* f) { free(f->prices); free(f->buys); free(f->sells); free(f->dvols); free(f->hist); }

static void trades_append(Feature* f, int64_t b, int64_t s) { /* deque(maxlen=lookback).append */
    int lb = f->d.lookback;
    if (lb == 0) return;
    if (f->n_trades == lb) {
        memmove(f->buys, f->buys + 1, 8 * (size_t)(lb - 1));
        memmove(f->sells, f->sells + 1, 8 * (size_t)(lb - 1));
        f->n_trades--;
    }
    f->buys[f->n_trades] = b; f->sells[f->n_trades] = s; f->n_trades++;
}

/* Feature._update for each kind.  Returns 0 when the reference would raise. */
static int feature_raw_update(const oracle_t* h, Env* e, Feature* f) {
    (void)h;
    const Book* cb = &e->ex.central;
    switch (f->d.kind) {
    case LOBSIM_F_SPREAD: /* :128-129 */
        f->current_value = book_best_sell(cb) - book_best_buy(cb);
        return 1;
    case LOBSIM_F_BOOK_IMBALANCE: /* :142-143 */
        return book_imbalance(cb, &f->current_value);
    case LOBSIM_F_PRICE: /* :342-343 */
        f->current_value = e->price;
        return 1;
    case LOBSIM_F_PRICE_MOVE: /* :173-175 */
    case LOBSIM_F_PRICE_RANGE: { /* :198-200 */
        int maxlen = f->d.lookback + 1;
        int n = f->n_prices < maxlen ? f->n_prices : maxlen - 1; /* appendleft drops the oldest when full */
        memmove(f->prices + 1, f->prices, sizeof(double) * (size_t)n);
        f->prices[0] = e->price;
        f->n_prices = n + 1;
        if (f->d.kind == LOBSIM_F_PRICE_MOVE) f->current_value = f->prices[0] - f->prices[f->n_prices - 1];
        else {
            double mx = f->prices[0], mn = f->prices[0];
            for (int i = 1; i < f->n_prices; i++) { if (f->prices[i] > mx) mx = f->prices[i]; if (f->prices[i] < mn) mn = f->prices[i]; }
            f->current_value = mx - mn;
        }
        return 1;
    }
    case LOBSIM_F_VOLATILITY: { /* :226-242 */
        int lb = f->d.lookback;
        if (f->n_prices < lb) {
            f->prices[f->n_prices++] = e->price;
            f->current_value = 0.0;
        } else if (f->n_prices == lb) {
            f->prices[f->n_prices++] = e->price;
            double p0 = f->prices[0], s = 0.0; /* python sum() starts from int 0; 0 + x is exact */
            for (int i = 0; i < lb; i++) { double r = (f->prices[i + 1] - f->prices[i]) / p0; s = s + r * r; }
            f->current_value = s / lb;
        } else {
            double oldest = f->prices[0];
            memmove(f->prices, f->prices + 1, sizeof(double) * (size_t)lb);
            double old_ret = (f->prices[0] - oldest) / oldest;
            double last = f->prices[lb - 1];
            double new_ret = (e->price - last) / last;
            f->prices[lb] = e->price;
            /* float ** 2 in CPython is libm's pow(x, 2.0), called at run time - through a volatile pointer here: the compiler folds a
             * literal pow(x, 2.0) into x * x, which is the correctly rounded square and libm's pow, good to 1 ulp, is not always
             * (found by oracle/fuzz_reference.py: one last bit in 90 000 steps) */
            double ss = f->current_value * lb - libm_pow(old_ret, 2.0) + libm_pow(new_ret, 2.0);
            f->current_value = ss / lb;
        }
        return 1;
    }
    case LOBSIM_F_TRADE_DIRECTION_IMBALANCE: /* :376-403 */
    case LOBSIM_F_TRADE_VOLUME_IMBALANCE: {  /* :435-462 */
        int64_t nb = 0, ns = 0;
        for (int i = 0; i < e->fex.n; i++) {
            int64_t w = f->d.kind == LOBSIM_F_TRADE_DIRECTION_IMBALANCE ? 1 : e->fex.f[i].vol;
            if (e->fex.f[i].side == LOBSIM_BUY) nb += w; else ns += w;
        }
        if (f->d.iparam) /* track_internal (:379-381, :438-440): the agent's own fills count as well */
            for (int i = 0; i < e->fin.n; i++) {
                int64_t w = f->d.kind == LOBSIM_F_TRADE_DIRECTION_IMBALANCE ? 1 : e->fin.f[i].vol;
                if (e->fin.f[i].side == LOBSIM_BUY) nb += w; else ns += w;
            }
        int lb = f->d.lookback;
        if (f->n_trades < lb) {
            trades_append(f, nb, ns);
            f->current_value = 0.0;
            return 1;
        } else if (f->total == 0) {
            trades_append(f, nb, ns);
            int64_t sb = 0, ss = 0;
            for (int i = 0; i < f->n_trades; i++) { sb += f->buys[i]; ss += f->sells[i]; }
            f->total = sb + ss; f->diff = sb - ss;
        } else {
            int64_t ob = f->buys[0], os = f->sells[0];
            memmove(f->buys, f->buys + 1, 8 * (size_t)(lb - 1));
            memmove(f->sells, f->sells + 1, 8 * (size_t)(lb - 1));
            f->n_trades--;
            f->total -= ob + os; f->total += nb + ns;
            f->diff -= ob - os; f->diff += nb - ns;
            trades_append(f, nb, ns);
        }
        f->current_value = f->total != 0 ? (double)f->diff / (double)f->total : 0.5;
        return 1;
    }
    case LOBSIM_F_AMIHUD_LAMBDA: { /* :285-324 */
        int slowing = (int)f->d.iparam, tlb = f->d.lookback / slowing - 1; /* true_lookback_periods */
        if (f->steps_until_update > 0) {
            f->steps_until_update -= 1;
            f->partial_price_sum += e->price;
            for (int i = 0; i < e->fex.n; i++) f->partial_dollar_volume += e->fex.f[i].vol;
            return 1;
        }
        f->steps_until_update = slowing - 1;
        if (f->n_prices < tlb) { /* _update_prices_and_dollar_volumes; value 0 */
            f->prices[f->n_prices++] = f->partial_price_sum / slowing;
            if (f->n_dvols == tlb) { memmove(f->dvols, f->dvols + 1, 8 * (size_t)(tlb - 1)); f->n_dvols--; }
            f->dvols[f->n_dvols++] = f->partial_dollar_volume;
            f->current_value = 0.0;
        } else if (f->n_prices == tlb) {
            f->prices[f->n_prices++] = f->partial_price_sum / slowing;
            if (f->n_dvols == tlb) { memmove(f->dvols, f->dvols + 1, 8 * (size_t)(tlb - 1)); f->n_dvols--; }
            f->dvols[f->n_dvols++] = f->partial_dollar_volume;
            double p0 = f->prices[0], sum = 0.0; /* python sum() of a float array starting from int 0 */
            for (int i = 0; i < tlb; i++) {
                double r = (f->prices[i + 1] - f->prices[i]) / p0;
                if (f->dvols[i] > 0) sum = sum + fabs(r) / (double)f->dvols[i];
            }
            f->current_value = sum / tlb;
        } else {
            double oldest = f->prices[0];
            /* Python floats: a division by 0.0 raises ZeroDivisionError (:306, :309).  The averaged price is 0.0 exactly when
             * slowing_factor is 1 - the partial sum is only added to between slow updates (:287-290), of which there are none - so
             * such a feature raises at its (lookback + 2)-th update: the env stops (found by oracle/fuzz_reference.py) */
            if (oldest == 0.0 || f->prices[tlb] == 0.0) return 0;
            memmove(f->prices, f->prices + 1, sizeof(double) * (size_t)tlb);
            double old_ret = (f->prices[0] - oldest) / oldest;
            int64_t old_dv = f->dvols[0];
            memmove(f->dvols, f->dvols + 1, 8 * (size_t)(tlb - 1));
            f->n_dvols--;
            double new_price = f->partial_price_sum / slowing;
            double new_ret = (new_price - f->prices[tlb - 1]) / f->prices[tlb - 1];
            int64_t new_dv = f->partial_dollar_volume;
            f->prices[tlb] = new_price;
            f->dvols[f->n_dvols++] = new_dv;
            double new_ratio = new_dv > 0 ? fabs(new_ret) / (double)new_dv : 0.0;
            double old_ratio = old_dv > 0 ? fabs(old_ret) / (double)old_dv : 0.0;
            double sr = f->current_value * tlb + new_ratio - old_ratio;
            f->current_value = sr / tlb;
        }
        f->partial_price_sum = 0; f->partial_dollar_volume = 0; /* _reset_partial_values */
        return 1;
    }
    case LOBSIM_F_INVENTORY: /* :489-490 */
        f->current_value = (double)e->inventory;
        return 1;
    case LOBSIM_F_EPISODE_PROPORTION: /* :511-512 */
        f->current_value += f->d.dparam;
        return 1;
    case LOBSIM_F_TIME_OF_DAY: { /* :535-536; min_time = 10:00 of the day (:16,:531) */
        int64_t num = e->now_us - 10ll * 3600 * 1000000, den = f->d.iparam;
        int64_t q = num / den;
        if (num % den != 0 && num < 0) q -= 1; /* floor division */
        f->current_value = (double)(q > 0 ? q : 0);
        return 1;
    }
    }
    return 0;
}

/* numpy's add.reduce over a contiguous f64 vector of any length, optionally of (x - mean)^2: pairwise summation
 * (8 accumulators up to 128 elements, halves rounded down to a multiple of 8 above) inside chunks of the ufunc
 * buffer size (np.getbufsize() = 8192 elements), the chunks added left to right.  Checked against numpy for lengths
 * up to 1e5 (oracle/gen_golden.py pins it through scipy.stats.zscore). */
static double np_pw(const double* a, int64_t n, int centered, double mean) {
#define NPT(i) (centered ? (a[i] - mean) * (a[i] - mean) : a[i])
    if (n < 8) { double r = 0.; for (int64_t i = 0; i < n; i++) r += NPT(i); return r; }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = NPT(j);
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += NPT(i + j);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += NPT(i);
        return res;
    }
#undef NPT
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pw(a, n2, centered, mean) + np_pw(a + n2, n - n2, centered, mean);
}
static double np_sum(const double* a, int64_t n, int centered, double mean) {
    double total = 0.;
    for (int64_t off = 0; off < n; off += 8192) total = total + np_pw(a + off, n - off < 8192 ? n - off : 8192, centered, mean);
    return total;
}

/* Feature.normalise (Features.py:67-74): history.append(value) [preceded by value + 1e-06 when the history is empty],
 * then scipy.stats.zscore(history)[-1] = (value - mean) / std with numpy's mean and std(ddof=0). */
static double feature_normalise(Feature* f, double value) {
    const int64_t cap = f->d.norm_len;
    for (int k = f->n_hist == 0 ? 0 : 1; k < 2; k++) {
        if (f->n_hist == cap) { memmove(f->hist, f->hist + 1, sizeof(double) * (size_t)(cap - 1)); f->n_hist--; }
        f->hist[f->n_hist++] = k == 0 ? value + 1e-06 : value;
    }
    const int64_t n = f->n_hist;
    const double mean = np_sum(f->hist, n, 0, 0.) / (double)n;
    const double var = np_sum(f->hist, n, 1, mean) / (double)n;
    return (value - mean) / sqrt(var);
}

static void feature_reset(const oracle_t* h, Env* e, Feature* f, int64_t first_usage_us) { /* reset/_reset, :92-96 */
    f->n_hist = 0; /* history.clear() */
    f->n_prices = 0; f->n_trades = 0; f->total = 0; f->diff = 0;
    f->n_dvols = 0; f->partial_price_sum = 0; f->partial_dollar_volume = 0;
    f->steps_until_update = f->d.kind == LOBSIM_F_AMIHUD_LAMBDA ? (int)f->d.iparam - 1 : 0; /* :278-283 */
    f->first_usage_us = first_usage_us;
    if (!feature_raw_update(h, e, f)) e->status |= LOBSIM_ST_EMPTY_BOOK;
    if (f->d.kind == LOBSIM_F_EPISODE_PROPORTION) f->current_value = 0.0; /* :508-509 */
}

static void feature_update(const oracle_t* h, Env* e, Feature* f) { /* update, :80-86 and :102-105 */
    if (e->now_us < f->first_usage_us) return;
    if ((e->now_us % 60000000ll) % f->d.update_us != 0) return;
    if (!feature_raw_update(h, e, f)) { e->status |= LOBSIM_ST_EMPTY_BOOK; return; }
    f->current_value = clampd(f->current_value, f->d.min_value, f->d.max_value);
    if (f->d.norm_len > 0) f->current_value = feature_normalise(f, f->current_value); /* :86 */
}

/* ------------------------------------------------------------------------------------------------------------
 * action -> orders (rl4mm/gym/action_interpretation/OrderDistributors.py, HistoricalOrderbookEnvironment.py)
 * ---------------------------------------------------------------------------------------------------------- */
static double np_pairwise_sum(const double* a, int n) { /* numpy add.reduce on a contiguous f64 vector, n <= 128 */
    if (n < 8) { double r = 0.; for (int i = 0; i < n; i++) r += a[i]; return r; }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* BetaOrderDistributor.convert_action/_convert_action, OrderDistributors.py:23-25,39-56.  scipy's beta.pdf is
 * x^(a-1) (1-x)^(b-1) / B(a,b); the 1/B cancels in the normalisation. */
/* Returns nonzero where the reference's distribution is NaN: scipy's beta.pdf of a shape parameter <= 0 (concentration mode with an
 * action at or above the concentration, :43-44; in the four-parameter mode that needs an action below the Box), or a pdf that sums to
 * nothing / overflows (0 / 0, inf / inf at :52-53).  np.round(nan * 100).astype(int) is then the most negative integer at every level
 * and _volume_diff_to_orders looks up orders to cancel at a price that holds none: KeyError, the step raises
 * (HistoricalOrderbookEnvironment.py:238-240) - unless the step clears the inventory, which replaces the volumes by zeros first
 * (:208-209).  Recognised the way the kernels do: a side whose volumes are all zero (no distribution gives that: the largest of Q <= 32
 * shares of 100 rounds to 3 at least).  The pdf here drops the 1 / B(a, b) that cancels in the normalisation, so shape parameters in
 * the hundreds (no Box of the reference's configurations holds them) underflow where scipy's would not: out of range, same marker. */
static int beta_volumes(const lobsim_cfg* cfg, const double* action, int64_t* buy, int64_t* sell) {
    const double EPS = 0.000001;
    int Q = cfg->max_quote_level - cfg->min_quote_level;
    double a[LOBSIM_MAX_ACTION_DIM];
    for (int i = 0; i < cfg->action_dim; i++) a[i] = action[i] + EPS;
    double ab, bb, as, bs;
    if (cfg->use_concentration) { ab = a[0]; bb = cfg->concentration - a[0] + EPS; as = a[1]; bs = cfg->concentration - a[1] + EPS; }
    else { ab = a[0]; bb = a[1]; as = a[2]; bs = a[3]; }
    double pb[128] = {0}, ps[128] = {0};
    for (int i = 0; i < Q; i++) {
        double x = 1.0 / Q * (i + 0.5);
        pb[i] = pow(x, ab - 1.0) * pow(1.0 - x, bb - 1.0);
        ps[i] = pow(x, as - 1.0) * pow(1.0 - x, bs - 1.0);
    }
    double sb = np_pairwise_sum(pb, Q), ss = np_pairwise_sum(ps, Q);
    const int nan_b = (cfg->use_concentration && !(bb > 0.0)) || !(sb > 0.0 && sb < INFINITY);
    const int nan_s = (cfg->use_concentration && !(bs > 0.0)) || !(ss > 0.0 && ss < INFINITY);
    int64_t tb = 0, ts = 0;
    for (int i = 0; i < Q; i++) {
        double qb = pb[i] / sb * 100.0, qs = ps[i] / ss * 100.0;
        buy[i] = nan_b || qb != qb ? 0 : (int64_t)rint(qb); /* np.round: half to even */
        sell[i] = nan_s || qs != qs ? 0 : (int64_t)rint(qs);
        tb += buy[i]; ts += sell[i];
    }
    return tb == 0 || ts == 0;
}

static int64_t internal_volume_at(const Book* ib, int side, int64_t price) { /* _get_volumes_at_prices, :324-331 */
    const Level* l = side_find((Side*)&ib->s[side], price);
    return l ? level_volume(l) : 0;
}

typedef struct {
    Order* o;
    int n, cap;
} OrderList;
static void olist_push(OrderList* L, Order o) {
    if (L->n == L->cap) { L->cap = L->cap ? 2 * L->cap : 64; L->o = (Order*)realloc(L->o, sizeof(Order) * (size_t)L->cap); }
    L->o[L->n++] = o;
}

/* convert_action_to_orders + _volume_diff_to_orders + _get_best_prices, HistoricalOrderbookEnvironment.py:206-331 */
static void convert_action_to_orders(const oracle_t* h, Env* e, const double* action, OrderList* out) {
    const lobsim_cfg* cfg = &h->cfg;
    int Q = cfg->max_quote_level - cfg->min_quote_level;
    int64_t tick = cfg->tick_size;
    int64_t desired[2][128];
    const int nan_volumes = beta_volumes(cfg, action, desired[0], desired[1]);
    int64_t abs_inv = e->inventory < 0 ? -e->inventory : e->inventory;
    int clearing = cfg->market_order_clearing && (double)abs_inv > action[cfg->action_dim - 1];
    if (clearing) for (int i = 0; i < Q; i++) desired[0][i] = desired[1][i] = 0;
    const Book* cb = &e->ex.central;
    double bb = book_best_buy(cb), bs = book_best_sell(cb);
    if (!(bs < INFINITY)) { e->status |= LOBSIM_ST_EMPTY_BOOK; return; } /* np.arange(inf) raises */
    if (nan_volumes && !clearing) { e->status |= LOBSIM_ST_ASSERT; return; } /* (beta_volumes) the KeyError of :238-240 */
    int64_t best_buy, best_sell;
    if (cfg->enter_spread) { /* :300-309 */
        double mid = (bs + bb) / 2;
        best_buy = (int64_t)(floor(mid / (double)tick) * (double)tick);
        best_sell = (int64_t)(ceil(mid / (double)tick) * (double)tick);
    } else { best_buy = (int64_t)bb; best_sell = (int64_t)bs; }
    int64_t prices[2][128];
    for (int k = 0; k < Q; k++) {
        prices[0][k] = best_buy - (cfg->min_quote_level + k) * tick;
        prices[1][k] = best_sell + (cfg->min_quote_level + k) * tick;
    }
    Book* ib = &e->ex.internal;
    for (int side = 0; side < 2; side++) {
        for (int k = 0; k < Q; k++) {
            int64_t price = prices[side][k];
            int64_t diff = desired[side][k] - internal_volume_at(ib, side, price);
            if (diff > 0) {
                Order o = {LOBSIM_LIMIT, side, price, diff, IID_NONE, 0, 0, 0};
                olist_push(out, o);
            }
            if (diff < 0) { /* cancel from the newest order backwards, :240-249 */
                Level* l = side_find(&ib->s[side], price);
                int j = l->n - 1;
                while (diff < 0) {
                    int64_t v = l->q[j].vol < -diff ? l->q[j].vol : -diff;
                    Order o = {LOBSIM_CANCELLATION, side, price, v, l->q[j].iid, 0, 0, 0};
                    olist_push(out, o);
                    diff += v;
                    j--;
                }
            }
        }
        /* orders at prices we no longer quote: cancel fully (:250-257); price order is unobservable (Q12) */
        for (int i = 0; i < ib->s[side].n; i++) {
            Level* l = &ib->s[side].lv[i];
            int quoted = 0;
            for (int k = 0; k < Q; k++) if (prices[side][k] == l->price) quoted = 1;
            if (quoted) continue;
            for (int j = 0; j < l->n; j++) {
                Order o = {LOBSIM_CANCELLATION, side, l->price, l->q[j].vol, l->q[j].iid, 0, 0, 0};
                olist_push(out, o);
            }
        }
    }
    if (clearing) { /* _get_inventory_clearing_market_order, :260-266 */
        double v = rint((double)abs_inv * cfg->market_order_fraction_of_inventory);
        Order o = {LOBSIM_MARKET, e->inventory < 0 ? LOBSIM_BUY : LOBSIM_SELL, 0, (int64_t)v, IID_NONE, 0, 0, 0};
        olist_push(out, o);
    }
}

/* _update_portfolio, :280-289 */
static void update_portfolio(Env* e) {
    for (int i = 0; i < e->fin.n; i++) {
        Fill* f = &e->fin.f[i];
        if (f->side == LOBSIM_SELL) { e->inventory -= f->vol; e->cash += f->vol * f->price; }
        else { e->inventory += f->vol; e->cash -= f->vol * f->price; }
    }
}

/* _forward + update_internal_state, :180-187, 199-204 */
static void env_forward(oracle_t* h, Env* e, Order* internal, int n_internal) {
    forward_step(h, e, e->now_us + h->cfg.step_us, internal, n_internal);
    if (e->status) return;
    update_portfolio(e);
    if (!book_microprice(&e->ex.central, &e->price)) e->status |= LOBSIM_ST_EMPTY_BOOK;
}

/* get_sharpe (RewardFunctions.py:10-22) with numpy's summation order.  Returns 0 and sets *bad when min(aum) <= 0. */
static double get_sharpe(const double* aum, int n, int* bad) {
    double s[LOBSIM_SHARPE_MAX_WINDOW] = {0}, x[LOBSIM_SHARPE_MAX_WINDOW] = {0};
    for (int i = 0; i < n; i++) if (aum[i] <= 0) { *bad = 1; return 0.0; }
    int m = n - 1;
    for (int i = 0; i < m; i++) s[i] = exp(log(aum[i + 1]) - log(aum[i])) - 1;
    double mean = np_pairwise_sum(s, m) / m;                    /* np.mean */
    for (int i = 0; i < m; i++) { double d = s[i] - mean; x[i] = d * d; } /* np.std: mean again with the same sum */
    double var = np_pairwise_sum(x, m) / (m - 1);
    return mean / (sqrt(var) + 2.2250738585072014e-308);        /* + sys.float_info.min */
}

/* RollingSharpe.calculate (RewardFunctions.py:81-94) */
static double rolling_sharpe(const lobsim_reward* r, Env* e, int role) {
    int W = r->max_window;
    double new_aum = (double)e->cash + e->price * (double)e->inventory; /* calculate_aum of next_state */
    memmove(e->rs_aum[role], e->rs_aum[role] + 1, sizeof(double) * (size_t)(W - 1)); /* aum[0] = new; roll(-1) */
    e->rs_aum[role][W - 1] = new_aum;
    if (e->rs_filled[role] < W) e->rs_filled[role]++;
    int n = e->rs_filled[role];
    if (n < r->min_window) return 0.0;
    int bad = 0;
    double v = get_sharpe(e->rs_aum[role] + (W - n), n, &bad); /* the non-nan tail, oldest first */
    if (bad) e->status |= LOBSIM_ST_ASSERT; /* raise Exception("AUM has gone non_positive") */
    return v;
}

static double reward_calc(const lobsim_reward* r, int64_t inv0, int64_t cash0, double p0, int64_t inv1, int64_t cash1, double p1) {
    /* PnL.calculate, RewardFunctions.py:98-101 */
    double cur = (double)cash0 + (double)inv0 * p0;
    double nxt = (double)cash1 + (double)inv1 * p1;
    double pnl = nxt - cur;
    if (r->kind == LOBSIM_R_PNL) return pnl;
    /* InventoryAdjustedPnL.calculate, :113-118 */
    double delta = p1 - p0;
    double term = r->inventory_aversion * (double)inv1 * delta;
    if (r->asymmetrically_dampened) term = term > 0 ? term : 0; /* max(0, term) */
    return pnl - term;
}

static void write_obs(const oracle_t* h, const Env* e, double* obs, int with_action) {
    int F = h->cfg.n_features;
    for (int i = 0; i < F; i++) obs[i] = e->feat[i].current_value;
    if (h->cfg.inc_prev_action_in_obs)
        for (int i = 0; i < h->cfg.action_dim; i++) obs[F + i] = with_action ? e->prev_action[i] : 0.0;
}

static int64_t feature_window_us(const lobsim_feature* f) { return (int64_t)f->lookback * f->update_us; } /* :63-65 */

/* ------------------------------------------------------------------------------------------------------------
 * public API (mirrors include/lobsim.h with an oracle_ prefix)
 * ---------------------------------------------------------------------------------------------------------- */
/* test helper: feeds a sequence of (inventory=1, cash=0, price=aum[i]) states through RollingSharpe.calculate, the shape
   of rl4mm/rewards/tests/testRewards.py:45-78.  raised[i] = 1 where the reference raises. */
void oracle_rolling_sharpe(int max_window, int min_window, const double* aum, int n, double* out, int32_t* raised) {
    Env* e = (Env*)calloc(1, sizeof(Env));
    lobsim_reward r = {LOBSIM_R_ROLLING_SHARPE, 0, 0.0, max_window, min_window};
    e->inventory = 1;
    for (int i = 0; i < n; i++) {
        e->price = aum[i];
        e->status = 0;
        out[i] = rolling_sharpe(&r, e, 0);
        raised[i] = e->status != 0;
    }
    free(e);
}

int oracle_create(const lobsim_cfg* cfg, oracle_t** out) {
    if (!cfg || !out || cfg->n_envs <= 0 || cfg->n_features > LOBSIM_MAX_FEATURES) return LOBSIM_ERR_ARG;
    if (cfg->max_quote_level - cfg->min_quote_level > 128 || cfg->max_quote_level <= cfg->min_quote_level) return LOBSIM_ERR_ARG;
    /* reset_episode asserts that the replay starts on the second (OrderbookSimulator.py:61); it starts `warm-up` before the episode */
    if (cfg->warmup_us < 0 || cfg->warmup_us % 1000000 != 0 || cfg->step_us <= 0) return LOBSIM_ERR_ARG;
    const lobsim_reward* rw[2] = {&cfg->per_step_reward, &cfg->terminal_reward};
    for (int i = 0; i < 2; i++) {
        if (rw[i]->kind < LOBSIM_R_PNL || rw[i]->kind > LOBSIM_R_ROLLING_SHARPE) return LOBSIM_ERR_ARG;
        if (rw[i]->kind == LOBSIM_R_ROLLING_SHARPE && (rw[i]->min_window < 2 || rw[i]->max_window < rw[i]->min_window || rw[i]->max_window > LOBSIM_SHARPE_MAX_WINDOW))
            return LOBSIM_ERR_ARG;
    }
    oracle_t* h = (oracle_t*)calloc(1, sizeof(oracle_t));
    h->cfg = *cfg;
    h->env = (Env*)calloc((size_t)cfg->n_envs, sizeof(Env));
    for (int i = 0; i < cfg->n_envs; i++) {
        Env* e = &h->env[i];
        idmap_init(&e->ex.ids);
        e->stream_id = -1;
        e->status = LOBSIM_ST_NOT_RESET;
        e->cash = 1000; /* Portfolio(inventory=0, cash=1000), HistoricalOrderbookEnvironment.py:104 */
        for (int f = 0; f < cfg->n_features; f++) { e->feat[f].d = cfg->features[f]; feature_alloc(&e->feat[f]); }
    }
    *out = h;
    return LOBSIM_OK;
}

void oracle_destroy(oracle_t* h) {
    if (!h) return;
    for (int i = 0; i < h->cfg.n_envs; i++) {
        Env* e = &h->env[i];
        book_free(&e->ex.central); book_free(&e->ex.internal); idmap_free(&e->ex.ids);
        free(e->fin.f); free(e->fex.f);
        for (int f = 0; f < h->cfg.n_features; f++) feature_free(&e->feat[f]);
    }
    for (int s = 0; s < h->n_streams; s++) { free((void*)h->streams[s].msgs); free((void*)h->streams[s].snaps); }
    free(h->streams); free(h->env); free(h);
}

const char* oracle_last_error(const oracle_t* h) { return h ? h->err : "null handle"; }

int oracle_add_stream(oracle_t* h, const int64_t* msgs, int64_t n_msgs, const int64_t* snaps, int64_t n_snaps) {
    h->streams = (Stream*)realloc(h->streams, sizeof(Stream) * (size_t)(h->n_streams + 1));
    Stream* s = &h->streams[h->n_streams];
    size_t mb = (size_t)n_msgs * 4 * 8, sb = (size_t)n_snaps * (2 + 4 * (size_t)h->cfg.n_levels) * 8;
    int64_t* m = (int64_t*)malloc(mb ? mb : 8); memcpy(m, msgs, mb);
    int64_t* sn = (int64_t*)malloc(sb ? sb : 8); memcpy(sn, snaps, sb);
    s->msgs = m; s->n_msgs = n_msgs; s->snaps = sn; s->n_snaps = n_snaps;
    {
        int64_t sw = 2 + 4 * (int64_t)h->cfg.n_levels, tmax = INT64_MIN;
        if (n_msgs) tmax = m[4 * (n_msgs - 1)];
        if (n_snaps && sn[sw * (n_snaps - 1)] > tmax) tmax = sn[sw * (n_snaps - 1)];
        if (tmax == INT64_MIN) tmax = 0;
        s->end_sec = (tmax >= 0 ? tmax / 1000000 : -((-tmax + 999999) / 1000000)) + 3;
    }
    return h->n_streams++;
}

int oracle_set_portfolio(oracle_t* h, const int32_t* env_ids, int32_t n, const int64_t* inventory, const int64_t* cash) {
    for (int i = 0; i < n; i++) {
        int id = env_ids ? env_ids[i] : i;
        if (id < 0 || id >= h->cfg.n_envs) return LOBSIM_ERR_ARG;
        h->env[id].inventory = inventory[i];
        h->env[id].cash = cash[i];
    }
    return LOBSIM_OK;
}

int oracle_obs_dim(const oracle_t* h) { return h->cfg.n_features + (h->cfg.inc_prev_action_in_obs ? h->cfg.action_dim : 0); }

/* HistoricalOrderbookEnvironment.reset, :147-161 (start time chosen by the caller) */
static void env_reset(oracle_t* h, Env* e, int stream_id, int64_t episode_start_us) {
    const lobsim_cfg* cfg = &h->cfg;
    e->status = 0;
    e->n_resyncs = 0;
    e->stream_id = stream_id;
    e->terminal_us = episode_start_us + cfg->episode_us;
    int64_t now = episode_start_us - cfg->warmup_us;
    reset_episode(h, e, now, NULL);
    if (e->status) return;
    if (!book_microprice(&e->ex.central, &e->price)) { e->status |= LOBSIM_ST_EMPTY_BOOK; return; }
    e->fin.n = e->fex.n = 0;
    for (int f = 0; f < cfg->n_features; f++) feature_reset(h, e, &e->feat[f], episode_start_us - feature_window_us(&cfg->features[f]));
    int64_t n_warm = cfg->warmup_us / cfg->step_us;
    for (int64_t s = 0; s < n_warm && !e->status; s++) {
        env_forward(h, e, NULL, 0);
        if (e->status) break;
        for (int f = 0; f < cfg->n_features; f++) feature_update(h, e, &e->feat[f]);
    }
}

int oracle_reset(oracle_t* h, const int32_t* env_ids, int32_t n, const int32_t* stream_ids, const int64_t* episode_start_us, double* obs) {
    int D = oracle_obs_dim(h);
    for (int i = 0; i < n; i++) {
        int id = env_ids ? env_ids[i] : i;
        if (id < 0 || id >= h->cfg.n_envs || stream_ids[i] < 0 || stream_ids[i] >= h->n_streams) return LOBSIM_ERR_ARG;
        if (episode_start_us[i] % 1000000 != 0) return LOBSIM_ERR_ARG; /* "Episodes must be started on the second" */
    }
    /* static schedule here and in oracle_rollout: a thread keeps its envs (heap locality on multi-socket hosts) */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        int id = env_ids ? env_ids[i] : i;
        env_reset(h, &h->env[id], stream_ids[i], episode_start_us[i]);
        if (obs) write_obs(h, &h->env[id], obs + (size_t)i * D, 0);
    }
    return LOBSIM_OK;
}

/* HistoricalOrderbookEnvironment.step, :163-178 */
static void env_step(oracle_t* h, Env* e, const double* action, double* obs, double* reward, uint8_t* done) {
    const lobsim_cfg* cfg = &h->cfg;
    if (e->status) { *done = 1; *reward = 0.0; if (obs) write_obs(h, e, obs, 1); return; }
    OrderList ol = {0};
    convert_action_to_orders(h, e, action, &ol);
    int64_t inv0 = e->inventory, cash0 = e->cash;
    double p0 = e->price;
    if (!e->status) env_forward(h, e, ol.o, ol.n);
    free(ol.o);
    for (int i = 0; i < cfg->action_dim; i++) e->prev_action[i] = action[i];
    if (e->status) { *done = 1; *reward = 0.0; if (obs) write_obs(h, e, obs, 1); return; }
    for (int f = 0; f < cfg->n_features; f++) feature_update(h, e, &e->feat[f]);
    double r = cfg->per_step_reward.kind == LOBSIM_R_ROLLING_SHARPE ? rolling_sharpe(&cfg->per_step_reward, e, 0)
                                                                  : reward_calc(&cfg->per_step_reward, inv0, cash0, p0, e->inventory, e->cash, e->price);
    int d = 0;
    if (2 * (e->terminal_us - e->now_us) < cfg->step_us) { /* terminal_time - now < step_size / 2 */
        r = cfg->terminal_reward.kind == LOBSIM_R_ROLLING_SHARPE ? rolling_sharpe(&cfg->terminal_reward, e, 1)
                                                                 : reward_calc(&cfg->terminal_reward, inv0, cash0, p0, e->inventory, e->cash, e->price);
        d = 1;
    }
    if (e->status) d = 1;
    *reward = r; *done = (uint8_t)d;
    if (obs) write_obs(h, e, obs, 1);
}

int oracle_step(oracle_t* h, const double* actions, const uint8_t* active, double* obs, double* reward, uint8_t* done) {
    int D = oracle_obs_dim(h), A = h->cfg.action_dim;
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < h->cfg.n_envs; i++) {
        if (active && !active[i]) continue;
        env_step(h, &h->env[i], actions + (size_t)i * A, obs ? obs + (size_t)i * D : NULL, &reward[i], &done[i]);
    }
    return LOBSIM_OK;
}

/* Teradactyl.get_action (rl4mm/agents/baseline_agents.py:58-87) */
static void teradactyl_action(const double* p, double inventory, double* a) {
    const double max_inv = p[0], def_kappa = p[1], w = p[2], max_kappa = p[3], e = p[4], denom = p[7], eps = 0.00001;
    double x = inventory / denom;
    double u = x < 1 - eps ? x : 1 - eps;
    u = u > -1 + eps ? u : -1 + eps;
    double omega_bid, omega_ask;
    if (inventory >= 0) {
        omega_bid = w * (1 + (1 / w - 1) * pow(u, e));
        omega_ask = w * (1 - pow(u, e));
    } else {
        omega_bid = w * (1 - pow(fabs(u), e));
        omega_ask = w * (1 + (1 / w - 1) * pow(fabs(u), e));
    }
    double kappa = (max_kappa - def_kappa) * pow(fabs(inventory / max_inv), e) + def_kappa;
    a[0] = omega_bid * (kappa - 2) + 1; a[1] = (1 - omega_bid) * (kappa - 2) + 1;
    a[2] = omega_ask * (kappa - 2) + 1; a[3] = (1 - omega_ask) * (kappa - 2) + 1;
    a[4] = p[6] != 0.0 ? max_inv * 2 : 0.0;
}

/* generate_trajectory (rl4mm/gym/utils.py:100-117) with the action of every step produced here: host mirror of
 * lobsim_rollout_device.  actions_out (may be NULL) [T][n_envs][A] receives the actions taken. */
int oracle_rollout(oracle_t* h, int32_t policy, const double* params8, int32_t T, const double* actions, const uint8_t* active,
                   double* obs, double* reward, uint8_t* done, double* return_acc, int32_t* steps_taken, double* actions_out) {
    int D = oracle_obs_dim(h), A = h->cfg.action_dim, N = h->cfg.n_envs;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        if (active && !active[i]) continue;
        Env* e = &h->env[i];
        double ob[LOBSIM_MAX_FEATURES + LOBSIM_MAX_ACTION_DIM], r = 0.0, ret = 0.0;
        uint8_t d = e->status ? 1 : 0;
        int steps = 0;
        for (int t = 0; t < T && !d; t++) {
            double a[LOBSIM_MAX_ACTION_DIM] = {0, 0, 0, 0, 0};
            if (policy == LOBSIM_POLICY_GIVEN) for (int k = 0; k < A; k++) a[k] = actions[((size_t)t * N + i) * A + k];
            else if (policy == LOBSIM_POLICY_FIXED) for (int k = 0; k < A; k++) a[k] = params8[k];
            else teradactyl_action(params8, e->feat[(int)params8[5]].current_value, a);
            if (actions_out) for (int k = 0; k < A; k++) actions_out[((size_t)t * N + i) * A + k] = a[k];
            env_step(h, e, a, ob, &r, &d);
            ret += r;
            steps++;
        }
        write_obs(h, e, ob, 1);
        if (obs) memcpy(obs + (size_t)i * D, ob, sizeof(double) * (size_t)D);
        reward[i] = r; done[i] = d;
        if (return_acc) return_acc[i] += ret;
        if (steps_taken) steps_taken[i] = steps;
    }
    return LOBSIM_OK;
}

int oracle_get_env_state(oracle_t* h, int32_t env, lobsim_env_state* out) {
    if (env < 0 || env >= h->cfg.n_envs) return LOBSIM_ERR_ARG;
    Env* e = &h->env[env];
    memset(out, 0, sizeof(*out));
    out->now_us = e->now_us; out->terminal_us = e->terminal_us;
    out->inventory = e->inventory; out->cash = e->cash; out->price = e->price;
    double bb = book_best_buy(&e->ex.central), bs = book_best_sell(&e->ex.central);
    out->best_buy = (int64_t)bb; out->best_sell = bs < INFINITY ? (int64_t)bs : INT64_MAX;
    out->min_buy_price = (int64_t)e->min_buy_price; out->max_sell_price = (int64_t)e->max_sell_price;
    out->initial_buy_range = (int64_t)e->initial_buy_range; out->initial_sell_range = (int64_t)e->initial_sell_range;
    out->n_buy_levels = e->ex.central.s[0].n; out->n_sell_levels = e->ex.central.s[1].n;
    int na = 0;
    for (int s = 0; s < 2; s++) for (int i = 0; i < e->ex.internal.s[s].n; i++) na += e->ex.internal.s[s].lv[i].n;
    out->n_agent_orders = na;
    out->status = e->status; out->stream_id = e->stream_id; out->n_resyncs = e->n_resyncs;
    for (int i = 0; i < e->fin.n; i++) {
        out->agent_fill_count++;
        if (e->fin.f[i].side == LOBSIM_BUY) out->agent_buy_volume += e->fin.f[i].vol; else out->agent_sell_volume += e->fin.f[i].vol;
    }
    for (int i = 0; i < e->fex.n; i++) {
        if (e->fex.f[i].side == LOBSIM_BUY) { out->ext_buy_count++; out->ext_buy_volume += e->fex.f[i].vol; }
        else { out->ext_sell_count++; out->ext_sell_volume += e->fex.f[i].vol; }
    }
    return LOBSIM_OK;
}

int oracle_get_portfolios(oracle_t* h, int64_t* inventory, int64_t* cash, double* price) {
    for (int i = 0; i < h->cfg.n_envs; i++) {
        if (inventory) inventory[i] = h->env[i].inventory;
        if (cash) cash[i] = h->env[i].cash;
        if (price) price[i] = h->env[i].price;
    }
    return LOBSIM_OK;
}

int oracle_get_best_prices(oracle_t* h, int64_t* best_buy, int64_t* best_sell) {
    for (int i = 0; i < h->cfg.n_envs; i++) {
        double bb = book_best_buy(&h->env[i].ex.central), bs = book_best_sell(&h->env[i].ex.central);
        best_buy[i] = (int64_t)bb;
        best_sell[i] = bs < INFINITY ? (int64_t)bs : INT64_MAX;
    }
    return LOBSIM_OK;
}

int oracle_get_book(oracle_t* h, int32_t env, int32_t side, int64_t* prices, int64_t* volumes, int32_t* n_orders, int32_t max_levels, int32_t* n_out) {
    if (env < 0 || env >= h->cfg.n_envs || side < 0 || side > 1) return LOBSIM_ERR_ARG;
    Side* s = &h->env[env].ex.central.s[side];
    int n = 0;
    for (int k = 0; k < s->n && n < max_levels; k++) {
        Level* l = side == LOBSIM_BUY ? &s->lv[s->n - 1 - k] : &s->lv[k];
        prices[n] = l->price; volumes[n] = level_volume(l);
        if (n_orders) n_orders[n] = l->n;
        n++;
    }
    *n_out = n;
    return LOBSIM_OK;
}

int oracle_get_level_orders(oracle_t* h, int32_t env, int32_t side, int64_t price, lobsim_order_view* out, int32_t max_orders, int32_t* n_out) {
    if (env < 0 || env >= h->cfg.n_envs || side < 0 || side > 1) return LOBSIM_ERR_ARG;
    Exchange* ex = &h->env[env].ex;
    Level* l = side_find(&ex->central.s[side], price);
    int n = 0;
    if (l) {
        /* agent sequence number = rank of the order among all agent orders ever submitted is not observable from
         * outside; expose the position of the agent order within the internal book's FIFO at this price instead */
        int agent_rank = 0;
        for (int i = 0; i < l->n && n < max_orders; i++) {
            out[n].volume = l->q[i].vol; out[n].reserved = 0;
            if (l->q[i].iid == -1) { out[n].kind = 2; out[n].id = -1; }
            else if (l->q[i].is_ext) { out[n].kind = 0; out[n].id = l->q[i].ext_id; }
            else { out[n].kind = 1; out[n].id = agent_rank++; out[n].reserved = (int32_t)l->q[i].tag; }
            n++;
        }
    }
    *n_out = n;
    return LOBSIM_OK;
}

int oracle_debug_set_book(oracle_t* h, int32_t env, const int64_t* snapshot_row, int64_t now_us) {
    if (env < 0 || env >= h->cfg.n_envs) return LOBSIM_ERR_ARG;
    Env* e = &h->env[env];
    if (now_us < 0) { /* -1: replace the book of a simulator that has been reset (reset_episode(start_date, start_book), :55-68); -2: only the
                         initial price ranges again, from the book as it is (include/lobsim.h) */
        e->n_resyncs = 0;
        e->min_buy_price = INFINITY; e->max_sell_price = 0; e->initial_buy_range = INFINITY; e->initial_sell_range = INFINITY;
        if (now_us == -1) {
            const int sid = e->stream_id;
            const int64_t now = e->now_us;
            e->stream_id = -1;  /* (reset_episode: the explicit row, not the stream's snapshot) */
            if (snapshot_row) reset_episode(h, e, now, snapshot_row);
            else { book_clear(&e->ex.central); book_clear(&e->ex.internal); e->agent_seq = 0; reset_price_ranges(e); }
            e->stream_id = sid;
        } else reset_price_ranges(e);
        e->price = 0.0;
        if (!e->status && !book_microprice(&e->ex.central, &e->price)) e->price = 0.0;
        e->fin.n = e->fex.n = 0;
        return LOBSIM_OK;
    }
    e->status = 0; e->stream_id = -1; e->n_resyncs = 0;
    e->min_buy_price = INFINITY; e->max_sell_price = 0; e->initial_buy_range = INFINITY; e->initial_sell_range = INFINITY;
    reset_episode(h, e, now_us, snapshot_row);
    e->status = 0; /* a one-sided debug book is legal; price-range bookkeeping is only used with a stream */
    e->terminal_us = INT64_MAX / 4;
    e->fin.n = e->fex.n = 0;
    return LOBSIM_OK;
}

/* One order row {type, side, price, volume (-1 none), id, who} of oracle_debug_process_orders / oracle_forward (include/lobsim.h).
 * who = 1: an external order, id = its external id.  who = 0: the agent's; a limit order's id > 0 is the tag the caller names it by, a
 * cancellation / deletion addresses its target by FIFO rank among the agent's orders at that price (see get_level_orders).  who = 2: the
 * agent's cancellation / deletion naming its target by TAG - the internal id of rl4mm_amd/simulator.py's orders: what the reference's
 * Cancellation carries (Exchange.py:196-217 looks the internal id up in the level's deque). */
static Order row_order(Env* e, const int64_t* r) {
    const int who = (int)r[5];
    Order o = {(int)r[0], (int)r[1], r[2], r[3], IID_NONE, r[4], who == 1, 0};
    if (who == 1) return o;
    o.ext_id = 0;
    if (o.type == LOBSIM_LIMIT) { o.tag = r[4] > 0 ? r[4] : 0; return o; }
    if (o.type == LOBSIM_CANCELLATION || o.type == LOBSIM_DELETION) {
        Level* l = side_find(&e->ex.internal.s[o.side], o.price);
        o.iid = INT64_MAX / 2; /* (addresses no order) */
        if (l && who == 0 && r[4] >= 0 && r[4] < l->n) o.iid = l->q[r[4]].iid;
        if (l && who == 2) for (int k = 0; k < l->n; k++) if (l->q[k].tag == r[4]) { o.iid = l->q[k].iid; break; }
    }
    return o;
}

int oracle_debug_process_orders(oracle_t* h, int32_t env, const int64_t* orders, int32_t n, int64_t* fills_out) {
    if (env < 0 || env >= h->cfg.n_envs) return LOBSIM_ERR_ARG;
    Env* e = &h->env[env];
    Ctx c = {e, &e->fin, &e->fex};
    e->fin.n = e->fex.n = 0;
    int64_t cash0 = e->cash;
    for (int i = 0; i < n && !e->status; i++) {
        const int64_t* r = orders + 6 * (size_t)i;
        Order o = row_order(e, r);
        process_order(&c, &o);
    }
    update_portfolio(e);
    if (fills_out) {
        lobsim_env_state st;
        oracle_get_env_state(h, env, &st);
        fills_out[0] = st.agent_fill_count; fills_out[1] = st.agent_buy_volume; fills_out[2] = st.agent_sell_volume;
        fills_out[3] = st.ext_buy_count; fills_out[4] = st.ext_sell_count; fills_out[5] = st.ext_buy_volume; fills_out[6] = st.ext_sell_volume;
        fills_out[7] = e->cash - cash0;
    }
    return LOBSIM_OK;
}

/* Number of OpenMP threads of the parallel regions above (bench.py's cpu_baseline: 1-thread and all-core runs). */
void oracle_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

/* test-only helpers --------------------------------------------------------------------------------------- */
/* Beta action -> integer volumes (pinned against scipy in tests/golden/beta_table.npz). */
void oracle_beta_volumes(const lobsim_cfg* cfg, const double* action, int64_t* buy, int64_t* sell) { (void)beta_volumes(cfg, action, buy, sell); }

/* FilledOrders of the last step / forward: out [max_rows][4] = {list (0 internal, 1 external), side, price, volume} (lobsim.h). */
int oracle_get_fills(oracle_t* h, int32_t env, int64_t* out, int32_t max_rows, int32_t* n_out) {
    if (!h || env < 0 || env >= h->cfg.n_envs || !n_out || max_rows < 0 || (max_rows && !out)) return LOBSIM_ERR_ARG;
    Env* e = &h->env[env];
    int n = 0;
    for (int i = 0; i < e->fin.n && n < max_rows; i++, n++) { out[4 * n] = 0; out[4 * n + 1] = e->fin.f[i].side; out[4 * n + 2] = e->fin.f[i].price; out[4 * n + 3] = e->fin.f[i].vol; }
    for (int i = 0; i < e->fex.n && n < max_rows; i++, n++) { out[4 * n] = 1; out[4 * n + 1] = e->fex.f[i].side; out[4 * n + 2] = e->fex.f[i].price; out[4 * n + 3] = e->fex.f[i].vol; }
    *n_out = e->fin.n + e->fex.n;
    return LOBSIM_OK;
}

/* Agent order list of the *next* step for `action` without processing it (convert_action_to_orders):
 * out [max][5] = {type, side, price, volume, fifo_rank_of_cancelled_order or -1}. */
int oracle_peek_agent_orders(oracle_t* h, int32_t env, const double* action, int64_t* out, int32_t max) {
    Env* e = &h->env[env];
    OrderList ol = {0};
    convert_action_to_orders(h, e, action, &ol);
    int n = ol.n < max ? ol.n : max;
    for (int i = 0; i < n; i++) {
        out[5 * i] = ol.o[i].type; out[5 * i + 1] = ol.o[i].side; out[5 * i + 2] = ol.o[i].price; out[5 * i + 3] = ol.o[i].vol;
        out[5 * i + 4] = -1;
    }
    free(ol.o);
    return n;
}

/* --- simulator-level entry (lobsim.h): OrderbookSimulator.reset_episode / forward_step on their own ------------------------ */
int oracle_abi_version(void) { return LOBSIM_ABI_VERSION; }

int oracle_get_limits(const oracle_t* h, lobsim_limits* out) { /* the restatement's containers are unbounded, like the reference's */
    if (!h || !out) return LOBSIM_ERR_ARG;
    memset(out, 0, sizeof(*out));
    out->level_slabs = out->far_levels = out->queue_slots = out->agent_orders = out->fill_log_rows = INT32_MAX;
    return LOBSIM_OK;
}

int oracle_get_status(oracle_t* h, int32_t* status) {
    if (!h || !status) return LOBSIM_ERR_ARG;
    for (int i = 0; i < h->cfg.n_envs; i++) status[i] = h->env[i].status;
    return LOBSIM_OK;
}

int oracle_reset_episode(oracle_t* h, const int32_t* env_ids, int32_t n, const int32_t* stream_ids, const int64_t* start_us) {
    if (!h || n <= 0 || n > h->cfg.n_envs || !stream_ids || !start_us) return LOBSIM_ERR_ARG;
    for (int i = 0; i < n; i++) {
        int id = env_ids ? env_ids[i] : i;
        if (id < 0 || id >= h->cfg.n_envs || stream_ids[i] < 0 || stream_ids[i] >= h->n_streams || start_us[i] % 1000000 != 0) return LOBSIM_ERR_ARG;
    }
    for (int i = 0; i < n; i++) {
        Env* e = &h->env[env_ids ? env_ids[i] : i];
        e->status = 0; e->stream_id = stream_ids[i]; e->n_resyncs = 0;
        e->min_buy_price = INFINITY; e->max_sell_price = 0; e->initial_buy_range = INFINITY; e->initial_sell_range = INFINITY;
        reset_episode(h, e, start_us[i], NULL);
        e->terminal_us = INT64_MAX / 4;
        e->fin.n = e->fex.n = 0;
        if (!book_microprice(&e->ex.central, &e->price)) e->price = 0.0;
    }
    return LOBSIM_OK;
}

int oracle_forward(oracle_t* h, const int32_t* env_ids, int32_t n, const int64_t* until_us, const int64_t* orders, const int32_t* n_orders,
                   int32_t max_orders, int32_t flags, double* obs) {
    if (!h || n <= 0 || n > h->cfg.n_envs || !until_us || max_orders < 0 || ((orders == NULL) != (n_orders == NULL))) return LOBSIM_ERR_ARG;
    const int D = oracle_obs_dim(h);
    for (int i = 0; i < n; i++) {
        int id = env_ids ? env_ids[i] : i;
        if (id < 0 || id >= h->cfg.n_envs || (n_orders && (n_orders[i] < 0 || n_orders[i] > max_orders))) return LOBSIM_ERR_ARG;
    }
    for (int i = 0; i < n; i++) {
        Env* e = &h->env[env_ids ? env_ids[i] : i];
        if (obs) for (int f = 0; f < D; f++) obs[(size_t)i * D + f] = 0.0;
        if (e->status) { e->fin.n = e->fex.n = 0; continue; }
        if (!(until_us[i] > e->now_us)) { e->status |= LOBSIM_ST_ASSERT; e->fin.n = e->fex.n = 0; continue; } /* :71-73 */
        int ni = n_orders ? n_orders[i] : 0;
        /* agent cancellations name their target by FIFO rank at the price: resolved as each order is reached, so rows are
         * converted one at a time (forward_step processes the list in order; ranks refer to the book at that moment) */
        Ctx c = {e, &e->fin, &e->fex};
        e->fin.n = e->fex.n = 0;
        for (int k = 0; k < ni && !e->status; k++) {
            const int64_t* r = orders + ((size_t)i * max_orders + k) * 6;
            Order o = row_order(e, r);
            process_order(&c, &o);
        }
        if (!e->status) forward_step_keep_fills(h, e, until_us[i]);
        if (e->status) continue;
        update_portfolio(e);
        if (flags & LOBSIM_FWD_UPDATE_FEATURES) {
            if (!book_microprice(&e->ex.central, &e->price)) { e->status |= LOBSIM_ST_EMPTY_BOOK; continue; }
            for (int f = 0; f < h->cfg.n_features; f++) feature_update(h, e, &e->feat[f]);
            if (obs) for (int f = 0; f < h->cfg.n_features; f++) obs[(size_t)i * D + f] = e->feat[f].current_value;
        }
    }
    return LOBSIM_OK;
}

int oracle_features_reset(oracle_t* h, const int32_t* env_ids, int32_t n) {
    if (!h || n <= 0 || n > h->cfg.n_envs) return LOBSIM_ERR_ARG;
    for (int i = 0; i < n; i++) {
        int id = env_ids ? env_ids[i] : i;
        if (id < 0 || id >= h->cfg.n_envs) return LOBSIM_ERR_ARG;
        Env* e = &h->env[id];
        if (e->status) continue;
        if (!book_microprice(&e->ex.central, &e->price)) e->price = 0.0;
        e->fin.n = e->fex.n = 0;
        for (int f = 0; f < h->cfg.n_features; f++) feature_reset(h, e, &e->feat[f], e->now_us);
    }
    return LOBSIM_OK;
}

/* High-water marks of an env since the handle was created: out = {most orders at one price, most live agent orders, most buy
 * levels, most sell levels} (test infrastructure: the capacities of the HIP layout are held against these). */
int oracle_get_high_water(oracle_t* h, int32_t env, int32_t* out) {
    if (!h || env < 0 || env >= h->cfg.n_envs || !out) return LOBSIM_ERR_ARG;
    const Env* e = &h->env[env];
    int lv[2] = {e->hw_levels[0], e->hw_levels[1]};
    for (int sd = 0; sd < 2; sd++) if (e->ex.central.s[sd].n > lv[sd]) lv[sd] = e->ex.central.s[sd].n; /* (levels the snapshot / resync put there) */
    out[0] = e->hw_queue; out[1] = e->hw_agent; out[2] = lv[0]; out[3] = lv[1];
    return LOBSIM_OK;
}
