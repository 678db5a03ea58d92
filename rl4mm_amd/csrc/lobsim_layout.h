// lobsim_layout.h - HBM / LDS data layout of the MI355X LOB simulator (shared by kernels and host glue).
//
// One wavefront (64 lanes) owns one env.  Everything an env owns is private to it, so no atomics, no
// inter-workgroup communication: the only shared data is the read-only packed message stream.
//
// Per env in HBM (sized for 288 GB: we spend capacity to make every access O(1) and coalesced):
//   EnvHdr                      scalars (clock, portfolio, resync bookkeeping, fill aggregates, free masks)
//   Ladder                      per side LV price levels, sorted best-first (SoA: price, packed slab/head/tail/live, volume)
//   FarLevel far[2][FAR_CAP]    price levels beyond the best LVT of a side (the reference's SortedDict is unbounded and keeps
//                               every level that left LOBSTER's visible range): a ring per side, best-first from its front
//   hi_free[..]                 free masks of the slabs whose mask is not in the header (lobsim_layout.h: HiWords)
//   Slot pages[n_pages][64]     the slab pool in 1 KiB pages.  One slab per price level = that level's FIFO queue, CONTIGUOUS, in one
//                               of four sizes - 1, 4, 16 or 64 pages = 64 / 256 / 1024 / 4096 slots; a level starts in the smallest and moves
//                               to a larger one when it outgrows its slab (rare: make_room).  Slot index = time priority; removed
//                               orders leave tombstones, squeezed out once LOBSIM_TOMB of them have piled up (or the slab is
//                               full), so a scan touches ceil((live+tombstones)/64) chunks
//   AgentOrd agent[na]          the agent's own live orders (the reference's "internal orderbook"); the first 128 are staged on chip
//   feature scalars + rings     rolling windows of the observation features
// During a launch the ladder, the agent orders and the op list live in LDS; slabs stay in HBM/L2 and are read
// 1 KiB per wave-instruction (16 B per lane), so "find order by id", "volume ahead" and "sweep the level" are
// one coalesced load + ballot / prefix-sum.
#pragma once
#include <stdint.h>

#include "../../include/lobsim.h"

namespace lobk {

constexpr int WAVE = 64;
constexpr int LV = 128;          // price levels per side of the NEAR ladder (the part of a side an env keeps in LDS; 2 per lane)
// The slab of a level: a contiguous run of 1 KiB pages (64 slots of 16 B: one wave-load) in the env's pool, in one of four size
// classes (round 6 added the 64-page class: all four values of the 2-bit field are in use).  Slab id (the 16-bit field of a level's meta word, of a position-cache entry) = class << 14 | first page: a slot's address
// is a function of the meta word alone - no table in between - and the class says how many slots the level may use.  Until round 5
// every level owned a fixed 16 KiB slab (16.5 MB of device memory per env at the default level budget, ~400 x what a day's live orders
// take); now a level takes 1 KiB while it is thin - the hundreds of stale levels a drifting day leaves behind hold an order or two -
// and only the levels that really hold hundreds of orders take 16 KiB.
constexpr int PAGE_SLOTS = 64;
constexpr int SLAB_CLASS_SHIFT = 14, SLAB_PAGE_MASK = (1 << SLAB_CLASS_SHIFT) - 1;
constexpr int SLAB_CLASS_MAX = 3;  // 64 << 2 * class slots: 64 / 256 / 1024 / 4096
constexpr int SLOTS = PAGE_SLOTS << (2 * SLAB_CLASS_MAX);  // order slots of the largest class = the most live orders + tombstones one price can hold (4096; 1024 until round 6)
constexpr int FLAT_SLOTS = PAGE_SLOTS << 4;  // (-DLOBSIM_FLAT_SLABS diagnostic builds: round 4's fixed 16 KiB slab per level)
constexpr int NSLAB = 2 * LV;    // class-0 slabs whose free mask lives in the header / in LDS (the first ones handed out): slab id = page < NSLAB
// Levels beyond the near ladder ("far" levels: the worse end of a side) live in HBM, one ring per side that is ordered
// best-first from its front: the level next to the near ladder's worst is at `front`, the k-th best far level at
// (front - k) mod FAR_CAP.  Both ends are O(1): the near ladder evicts to / refills from the front, the once-a-second
// outer-level resync appends at the back.  The ring's capacity is fixed; how many levels an env can really hold is set
// by its slab pool (cfg.max_levels -> DevPtrs.n_slab).
constexpr int FAR_CAP = 2048;
constexpr int NSLAB_MAX = NSLAB + 64 * 64;  // the free mask of the class-0 slabs beyond NSLAB is one 64-bit word per lane
// free masks in HBM (u64 words of the env's aux block) of the buddy pool behind the first NSLAB pages: blocks of 1 page (class 0), 4
// pages (class 1), 16 pages (class 2), 64 pages (class 3), bit b of a size's mask = block b of that size (block b of class c = pages
// NSLAB + b * 4^c ...); and the env's counts (level budget | 16-page blocks << 32: the cold routines read them here instead of carrying
// them in registers)
constexpr int HI_W0 = 0, HI_W1 = 64, HI_W2 = 80, HI_W3 = 84, HI_COUNTS = 85;
constexpr int NMID_MAX = 16 * 64, NBIG_MAX = 4 * 64, NHUGE_MAX = 64;
struct alignas(16) FarLevel { int64_t price; uint64_t meta; int64_t vol; int64_t pad; };  // meta as in Ladder
static_assert(sizeof(FarLevel) == 32, "");
constexpr size_t FAR_BYTES = 2 * (size_t)FAR_CAP * sizeof(FarLevel);
constexpr size_t HI_BYTES = 1024;                  // hi_free[128] u64 (86 used; pages stay 1 KiB-aligned)
constexpr size_t AGHI_BYTES = 3072;                // the agent-order entries beyond the 128 that are staged on chip: AgentOrd[128] (2048 B), then their slot indices u16[128]
constexpr size_t AUX_BYTES = FAR_BYTES + AGHI_BYTES + HI_BYTES;  // per env, in front of its pages: far rings, upper agent entries, hi_free
// Agent orders per env (live ones + those freed in the current step): a property of the HANDLE since round 5 (cfg.agent_orders), the
// engine is a template on it (Eng<.., NAT>: NAT / 64 entries per lane).  What bounds the reference's internal book is VOLUME - a side
// quotes sum(round(100 pdf)) <= 100 + Q/2 shares, so at most ~105 one-share orders per side - and 2 x 105 live entries + the <= 2Q a
// step adds fit into 256: with 256 entries LOBSIM_ST_AGENT_OVERFLOW cannot happen under the default configuration, which is why 256 is
// what lobsim_create picks unless told otherwise.  128 is the faster table (7.2 instead of 10.2 KB of LDS per env: 20 instead of 16
// envs per CU) that random and rule-based policies fill to 53 - 67; an env that outgrows it says so.
constexpr int NA_MAX = 256;      // the header's free mask and the op code's index field are sized for this
#ifndef LOBSIM_IDC_N
#define LOBSIM_IDC_N 32768
#endif
constexpr int IDC_N = LOBSIM_IDC_N;    // entries of the per-env external-order position cache (256 KiB; a cache: every hit is
                                 // verified against the slot it points to, misses take the queue scan)
constexpr int QMAX = 32;         // quote levels per side (2 * QMAX <= 64 lanes)
// agent op list of a step: limits + cancellations + the clearing market order (resync re-adds reuse it; its last 64 entries double as
// the Beta-pdf scratch).  With 256 agent entries the list is sized by the LDS budget of 16 envs per CU (10 240 B): 286 ops hold the
// <= 2 x (100 + Q/2) cancellations of one-share orders + 2Q limits + the clearing order for up to 28 quote levels (the default has 10; the
// table itself is provably large enough for up to 18).  Beyond: LOBSIM_ST_AGENT_OVERFLOW
constexpr int na_ops_max(int na) { return na == 128 ? 2 * QMAX + 128 + 8 : 286; }

constexpr int LPT_CLASSES = 16;  // sorter waves at the head of a step launch = residue classes of the envs (lobsim_kernels.hip: lpt_sort_body)

constexpr int64_t DUMMY_PRICE = 9999999999ll;

enum SlotKind : int32_t { K_EXT = 0, K_AGENT = 1, K_SNAP = 2, K_DEAD = 3 };

struct alignas(16) Slot {
    int64_t id;    // K_EXT: external id; K_AGENT: index into agent[]; K_SNAP: -1
    int32_t vol;
    int32_t kind;  // SlotKind
};
static_assert(sizeof(Slot) == 16, "slot must be one dwordx4");

struct alignas(16) AgentOrd {
    int64_t price;
    int32_t vol;        // 0 = free entry
    uint32_t seq_side;  // seq (31 bits, submission order within the episode) | side << 31
};
static_assert(sizeof(AgentOrd) == 16, "");

struct Ladder {  // HBM image of the LDS ladder
    int64_t price[2][LV];
    uint64_t meta[2][LV];  // slab | head << 16 | tail << 32 | live << 48
    int32_t vol[2][LV];    // total resting volume of the level
};

struct alignas(16) EnvHdr {
    int64_t now_us, terminal_us, episode_start_us;
    int64_t inventory, cash;
    double price;
    double min_buy, max_sell, init_buy_range, init_sell_range;  // OrderbookSimulator.py:49-53 (ints or inf)
    int64_t msg_cursor;  // next unreplayed message of the stream
    int32_t stream_id;
    int32_t status;
    int32_t n_lv[2];
    int32_t n_resyncs;
    uint32_t agent_seq;
    uint64_t slab_free[NSLAB / 64];  // bit set = slab free
    uint64_t agent_free[NA_MAX / 64];  // bit set = agent entry free (the words beyond the handle's table: all set)
    // fills of the last step
    int64_t agent_fill_count, agent_vol[2], ext_count[2], ext_vol[2];
    int64_t cash_delta;
    uint16_t far_n[2], far_front[2];  // far levels per side and the ring position of the best one (one 8-byte word)
    int64_t info_moc_count;    // SimpleInfoCalculator.market_order_count / market_order_total_volume (InfoCalculators.py:26-27, 85-87):
    double info_moc_volume;    // live as long as the env object, never reset
    double prev_action[LOBSIM_MAX_ACTION_DIM];
    // the env's stream as the step kernel needs it at the very top (set at reset: a copy of streams[stream_id].msgs / .n_msgs, whose
    // arrays live as long as the handle): read with the rest of the header, it saves the dependent trip through the stream table that
    // stood between the header and every other load of the step's prologue
    const int64_t* msgs;
    int64_t n_msgs;
};

// Feature scalars, one per (env, feature).
struct alignas(16) FeatState {
    double current_value;
    int64_t first_usage_us;
    int32_t count;   // elements in the deque
    int32_t pos;     // ring write position
    double p_first;  // Volatility: first price of the fill-up window; also S accumulator partner
    double p_last;   // newest price in the deque
    double acc;      // Volatility: running sum of squares during fill-up
    int64_t total, diff;  // trade imbalance running sums
};

struct StreamDev {
    const int64_t* msgs;
    const int64_t* snaps;
    const int32_t* sec_to_snap;  // [n_secs]: last snapshot with t <= second, -1 if none
    const int32_t* sec_to_msg;   // [n_secs]: first message with t > second
    int64_t n_msgs, n_snaps;
    int64_t first_sec;
    int32_t n_secs;
    int32_t n_win;               // entries of wtab
    // weight of every window of one step of this stream: window g = (first_sec s + g step, first_sec s + (g + 1) step], messages in it +
    // 8 x runs of executions, capped at 255 (built by lobsim_add_stream; read by the step launch's sorter waves: longest job first)
    const uint8_t* wtab;
};

struct DevCfg {
    lobsim_cfg c;
    double outer_prop;          // outer_levels / n_levels
    int32_t Q;                  // quote levels
    int32_t obs_dim;
    int32_t ring_off[LOBSIM_MAX_FEATURES];  // offset of feature f's ring (in 8-byte words) inside an env's ring block
    int32_t ring_words;         // ring words per env
    uint32_t tick_magic;        // ceil(2^32 / tick_size) when (min_quote_level + Q) * tick_size^2 < 2^32, else 0
    int32_t norm_off[LOBSIM_MAX_FEATURES];  // offset (8-byte words) of feature f's normalisation history inside an env's block
    int32_t norm_words;                     // history words per env (0: no feature is normalised)
    // ln(x_k), ln(1 - x_k) at the Q quote-level midpoints x_k = (k + 0.5) / Q: Beta pdf = exp((a-1) ln x + (b-1) ln(1-x))
    double log_x[QMAX], log_1mx[QMAX];
};

struct DevPtrs {
    EnvHdr* hdr;
    Ladder* ladder;
    uint8_t* pool;     // [env]{FarLevel far[2][FAR_CAP]; AgentOrd agent_hi[128]; u16 apos_hi[128] (+pad); u64 hi_free[128]; Slot pages[NSLAB + 16 n_big][64]}
    int64_t pool_stride;  // bytes per env
    int32_t n_slab;    // the env's budget of price levels (NSLAB .. NSLAB_MAX, a multiple of 64)
    int32_t na;        // agent entries per env (128 or 256: the NAT of the kernels this handle launches)
    int32_t n_mid, n_big;  // n_big: 16-page blocks of the env's buddy pool (pages NSLAB .. NSLAB + 16 n_big; slabs of 1 / 4 / 16 pages are cut from it); n_mid = 4 n_big
    AgentOrd* agent;   // [env][128] the agent-order entries that are staged in LDS (entries 128 .. na - 1: in the env's pool block, see `pool`)
    uint16_t* apos;    // [env][128] slot index of each live agent order inside its level's slab
    uint64_t* idc;     // [env][IDC_N] position cache of external orders: tag(id) << 32 | slab << 16 | slot, direct-mapped by id
    FeatState* feat;   // [env][LOBSIM_MAX_FEATURES]
    int64_t* rings;    // [env][ring_words]
    const StreamDev* streams;
    int32_t n_streams;
    double* norm;      // [env][norm_words]: per normalised feature one header word (count | start << 32) + norm_len values
    double* sharpe;    // [env][2][1 + LOBSIM_SHARPE_MAX_WINDOW] RollingSharpe windows (per-step / terminal), or null when unused
    uint64_t* prof;    // [env][8] phase cycle sums; only a -DLOBSIM_PROFILE diagnostic build allocates and writes it
};

}  // namespace lobk
