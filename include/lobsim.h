/* lobsim.h - C-ABI of the MI355X-native limit-order-book simulation step.
 *
 * Drop-in boundary for the hot path of JJJerome/rl4mm (reference paths are relative to /root/reference):
 *   rl4mm/gym/HistoricalOrderbookEnvironment.py:147-178  reset() / step()          -> lobsim_reset / lobsim_step
 *   rl4mm/simulation/OrderbookSimulator.py:55-135        reset_episode / forward_step / update_outer_levels
 *   rl4mm/orderbook/Exchange.py:58-247                   process_order and the matching engine
 *   rl4mm/features/Features.py, rl4mm/rewards/RewardFunctions.py  observation + reward reductions
 * Everything is plain pointers and sizes: no torch / numpy / C++ types cross this boundary.  The Python host
 * side (rl4mm_amd/) binds it with ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Threading: one handle per GPU; calls on one handle must be serialised by the caller (the reference env is
 * single-threaded as well, rl4mm/gym/utils.py:242-246).  All functions return 0 on success, a negative
 * lobsim_error otherwise; lobsim_last_error() gives the message.
 *
 * oracle/lobsim_oracle.c exports the same entry points with an `oracle_` prefix (CPU restatement, test
 * infrastructure only).
 */
#ifndef LOBSIM_H
#define LOBSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOBSIM_MAX_FEATURES 16
#define LOBSIM_MAX_ACTION_DIM 5
#define LOBSIM_MSG_WORDS 4 /* packed message = 4 x int64, see rl4mm_amd/stream.py */

/* message word 3: volume | type << 48 | side << 52 */
#define LOBSIM_TYPE_SHIFT 48
#define LOBSIM_SIDE_SHIFT 52
#define LOBSIM_VOL_MASK ((1ll << 48) - 1)

enum lobsim_order_type { /* LOBSTER event types = rl4mm/database/database_population_helpers.py:151-160 */
    LOBSIM_LIMIT = 1,
    LOBSIM_CANCELLATION = 2,
    LOBSIM_DELETION = 3,
    LOBSIM_MARKET = 4,
    LOBSIM_HIDDEN = 5,
    LOBSIM_CROSS_TRADE = 6,
    LOBSIM_TRADING_HALT = 7
};
enum lobsim_side { LOBSIM_BUY = 0, LOBSIM_SELL = 1 };

enum lobsim_error {
    LOBSIM_OK = 0,
    LOBSIM_ERR_ARG = -1,
    LOBSIM_ERR_HIP = -2,
    LOBSIM_ERR_STATE = -3,
    LOBSIM_ERR_NOMEM = -4
};

/* Per-env status bits (sticky until the env is reset).  They stand in for the Python exceptions the reference
 * raises on the same conditions; an env with any bit set reports done=1, reward 0 and stops advancing.  The reference has no
 * observation for such a step (the exception leaves `step` / `reset`): the observation row of an env whose status is nonzero is
 * unspecified - compare statuses, not rows (the env facade raises / puts the condition into info["error"]). */
enum lobsim_status {
    LOBSIM_ST_EMPTY_BOOK = 1,      /* EmptyOrderbookError, rl4mm/orderbook/Exchange.py:180-186; np.arange(inf) of a book without asks
                                      (HistoricalOrderbookEnvironment.py:298-322); a feature's _update that raises: the
                                      ZeroDivisionError of BookImbalance / microprice on a side of volume 0, of AmihudLambda with
                                      slowing_factor 1 (Features.py:306) */
    LOBSIM_ST_BAD_MESSAGE = 2,     /* cross_trade assert / trading_halt crash, HistoricalOrderGenerator.py:54-56 */
    LOBSIM_ST_LEVEL_OVERFLOW = 4,  /* more price levels than cfg.max_levels provides slabs for */
    LOBSIM_ST_QUEUE_OVERFLOW = 8,  /* more live orders at one price than the build holds */
    LOBSIM_ST_AGENT_OVERFLOW = 16, /* more live agent orders than the build holds */
    LOBSIM_ST_STREAM_END = 32,     /* no snapshot at or before the episode's start, or a step that ends 2 s or more beyond the
                                      last message / snapshot of the loaded stream */
    LOBSIM_ST_ASSERT = 64,         /* `assert order.volume > 0` (Exchange.py:59-60: a clearing order that rounds to 0 shares, a
                                      zero-volume message); the KeyError of a step whose Beta volumes are NaN (an action at or above
                                      the concentration, shape parameters whose pdf underflows) unless it clears the inventory
                                      (HistoricalOrderbookEnvironment.py:208-209, 238-240); RollingSharpe's non-positive AUM; a
                                      forward_step that does not advance the clock (OrderbookSimulator.py:71-73) */
    LOBSIM_ST_NOT_RESET = 128,
    LOBSIM_ST_INTERNAL = 256       /* only ever set by the diagnostic build of the library (-DLOBSIM_BOUNDS, liblobsim_bounds.so): a computed index
                                      left its array; bits 16 - 30 hold the line of lobsim_kernels.hip whose check failed */
};

enum lobsim_feature_kind { /* rl4mm/features/Features.py */
    LOBSIM_F_SPREAD = 1,                    /* :113 */
    LOBSIM_F_PRICE_MOVE = 2,                /* :151 */
    LOBSIM_F_VOLATILITY = 3,                /* :203 */
    LOBSIM_F_INVENTORY = 4,                 /* :474 */
    LOBSIM_F_EPISODE_PROPORTION = 5,        /* :493 */
    LOBSIM_F_TIME_OF_DAY = 6,               /* :515 */
    LOBSIM_F_TRADE_DIRECTION_IMBALANCE = 7, /* :351 */
    LOBSIM_F_TRADE_VOLUME_IMBALANCE = 8,    /* :410 */
    LOBSIM_F_BOOK_IMBALANCE = 9,            /* :132 */
    LOBSIM_F_PRICE_RANGE = 10,              /* :178 */
    LOBSIM_F_PRICE = 11,                    /* :327 */
    LOBSIM_F_AMIHUD_LAMBDA = 12             /* :245; lookback = (lookback_periods + 1) * slowing_factor, iparam = slowing_factor */
};

typedef struct lobsim_feature {
    int32_t kind;      /* lobsim_feature_kind */
    int32_t lookback;  /* lookback_periods (0 for book/agent features) */
    int64_t update_us; /* update_frequency in microseconds */
    double min_value, max_value;
    double dparam;  /* EpisodeProportion: update_frequency / episode_length (Features.py:505) */
    int64_t iparam; /* TimeOfDay: bucket size in us (Features.py:528); AmihudLambda: slowing_factor;
                       TradeDirectionImbalance / TradeVolumeImbalance: track_internal (0 / 1, Features.py:379-381,438-440) */
    /* normalisation_on (Features.py:60-74): > 0 = capacity of the value history (deque(maxlen=max_norm_len)); every
     * update then returns scipy.stats.zscore(history)[-1], an O(len) pass in numpy's summation order.  0 = off (the
     * reference's default).  Memory: 8 * norm_len bytes per env and feature - pass the number of updates an episode
     * plus its warm-up can make when that is smaller than max_norm_len (the history is cleared at every reset). */
    int64_t norm_len;
} lobsim_feature;

enum lobsim_reward_kind {
    LOBSIM_R_PNL = 1,                /* RewardFunctions.py:97 */
    LOBSIM_R_INVENTORY_ADJUSTED = 2, /* :107 */
    LOBSIM_R_ROLLING_SHARPE = 3      /* :38; stateful: one AUM window per env and per role (per-step / terminal) */
};
#define LOBSIM_SHARPE_MAX_WINDOW 128

typedef struct lobsim_reward {
    int32_t kind;
    int32_t asymmetrically_dampened;
    double inventory_aversion;
    int32_t max_window; /* RollingSharpe(max_window_size=120, min_window_size=60); max_window <= 128 */
    int32_t min_window;
} lobsim_reward;

/* Mirrors the constructor arguments of HistoricalOrderbookEnvironment (HistoricalOrderbookEnvironment.py:54-81)
 * and OrderbookSimulator (OrderbookSimulator.py:24-35) that reach the step. */
typedef struct lobsim_cfg {
    int32_t n_envs;
    int32_t n_levels;        /* book levels per side in the snapshot rows (simulator n_levels) */
    int32_t tick_size;       /* Exchange.tick_size, 100 */
    int32_t outer_levels;    /* OrderbookSimulator.outer_levels, 20 */
    int32_t min_quote_level; /* quotes sit min..max-1 ticks behind the best price */
    int32_t max_quote_level;
    int32_t action_dim;            /* 4; 2 with concentration; +1 with market_order_clearing */
    int32_t use_concentration;     /* BetaOrderDistributor(concentration=...) */
    int32_t market_order_clearing;
    int32_t enter_spread;
    int32_t inc_prev_action_in_obs;
    int32_t n_features;
    int32_t device_id; /* HIP device ordinal (ignored by the oracle) */
    int32_t ladder_levels; /* price levels per side an env keeps ON CHIP (LDS): 0 = auto (64 for n_levels <= 56, else 128), 64 or
                              128.  A performance knob only: levels beyond it live in HBM (ignored by the oracle) */
    int32_t max_levels;    /* the env's budget of price levels: 2 x max_levels levels in all, shared by the two
                              sides (a side can take nearly all of it, up to ladder_levels + 2048).  The reference's book is an
                              unbounded SortedDict (rl4mm/orderbook/models.py:64-69) that keeps every level that left LOBSTER's visible
                              range, so a drifting price piles up stale levels on the side it leaves (10 + one per tick of net excursion).
                              0 = the default, 512 (1.75 MiB of order queues per env - a level takes 1 KiB, 4 KiB or 16 KiB by the
                              orders it holds -, whatever else is on the GPU); otherwise 64 .. 2048, raised to 128 at least and
                              rounded up to a multiple of 32.  If the pool cannot be allocated, lobsim_create
                              retries with half (down to 128): lobsim_get_limits reports what the handle really has.  The operation
                              that needs one level more sets LOBSIM_ST_LEVEL_OVERFLOW on that env (ignored by the oracle) */
    int32_t agent_orders;  /* the env's table of live agent orders (the reference's internal book, rl4mm/orderbook/models.py:64-69, is
                              unbounded): 0 = 256, the size that CANNOT overflow under the reference's action interpretation for up to 18
                              quote levels - a side quotes at most 100 + Q/2 shares (OrderDistributors.py:39-56), every share its own order
                              at worst, plus the <= 2Q orders of the step: 2 x (100 + Q/2) + 2Q <= 256.  128 = the smaller table (2.5 KB less
                              on-chip memory per env: 20 instead of 16 envs resident per CU, a few per cent faster); an env that outgrows
                              it stops with LOBSIM_ST_AGENT_OVERFLOW.  lobsim_get_limits reports it (ignored by the oracle) */
    int64_t step_us;    /* step_size */
    int64_t episode_us; /* episode_length */
    int64_t warmup_us;  /* max feature window (whole seconds) */
    double concentration;
    double market_order_fraction_of_inventory;
    lobsim_reward per_step_reward;
    lobsim_reward terminal_reward;
    lobsim_feature features[LOBSIM_MAX_FEATURES];
    int32_t queue_pages; /* 1 KiB pages (64 order slots) of queue memory per env: a level's queue takes 1, 4 or 16 of them by the orders it
                            holds.  0 = auto: two pages per level of the budget (2 x max_levels levels), at least 1280 - 2304 at the default
                            budget, which a 30-minute trending day on a thick book (300 levels of ~180 orders) fills to 1300.  Otherwise
                            rounded up to 256 + a multiple of 16, at most 4352.  What a handle got: lobsim_limits.queue_pages (ignored by the oracle) */
    int32_t reserved1;
} lobsim_cfg;

typedef struct lobsim_env_state {
    int64_t now_us;        /* simulator clock, microseconds since midnight */
    int64_t terminal_us;   /* episode start + episode length */
    int64_t inventory;
    int64_t cash;          /* integer-valued, kept exact */
    double price;          /* state.price (microprice) after the last step */
    int64_t best_buy, best_sell;
    int64_t min_buy_price, max_sell_price;       /* outer-level resync bookkeeping, OrderbookSimulator.py:185-188 */
    int64_t initial_buy_range, initial_sell_range;
    int32_t n_buy_levels, n_sell_levels;
    int32_t n_agent_orders;
    int32_t status;
    int32_t stream_id;
    int32_t n_resyncs; /* update_outer_levels calls since reset */
    /* fills of the last step (a19 / Q10): agent fills by the agent's side; external fills by the resting side */
    int64_t agent_fill_count, agent_buy_volume, agent_sell_volume;
    int64_t ext_buy_count, ext_sell_count, ext_buy_volume, ext_sell_volume;
} lobsim_env_state;

/* One order of the central book as seen from outside (FIFO position = array index). */
typedef struct lobsim_order_view {
    int64_t id;      /* tracked external order: external id; agent order: FIFO rank among the agent's own orders at
                        this price (0 = oldest); snapshot aggregate: -1 */
    int64_t volume;
    int32_t kind;    /* 0 tracked external, 1 agent, 2 snapshot aggregate (internal_id == -1) */
    int32_t reserved; /* agent order: its TAG - the id the caller of lobsim_forward / lobsim_debug_process_orders gave the limit order (row id > 0), else
                         the env's submission counter at that moment; unique among an env's live agent orders, ascending in submission order
                         (the role of the reference's internal_id, OrderIDConvertor.py:12-19).  0 for the other kinds */
} lobsim_order_view;

typedef struct lobsim lobsim_t;

/* Bumped whenever a struct of this header changes layout or an entry point changes meaning: a binding compares it with the
 * LOBSIM_ABI_VERSION it was written against before it passes a config struct: a stale prebuilt library would otherwise
 * read the struct at the wrong offsets and still produce plausible numbers. */
#define LOBSIM_ABI_VERSION 6
int lobsim_abi_version(void);

int lobsim_create(const lobsim_cfg* cfg, lobsim_t** out);
void lobsim_destroy(lobsim_t* h);
const char* lobsim_last_error(const lobsim_t* h);

/* Launch everything on this hipStream_t (0 = the handle's own stream).  Lets a torch caller time with
 * torch.cuda.Event on torch's current stream. */
int lobsim_set_stream(lobsim_t* h, void* hip_stream);

/* Upload one packed (ticker, day) stream (host pointers; copied to HBM).  Returns the stream id >= 0.
 * Replaces HistoricalDatabase.get_messages / get_last_snapshot (rl4mm/database/HistoricalDatabase.py:46,103). */
int lobsim_add_stream(lobsim_t* h, const int64_t* msgs, int64_t n_msgs, const int64_t* snaps, int64_t n_snaps);
/* Free the device copy of a stream (a (ticker, day) no env replays any more: the reference reads its days from a database on demand,
 * rl4mm/database/HistoricalDatabase.py:46-62,103-135, and holds none of them).  Its id is handed out again by a later
 * lobsim_add_stream.  LOBSIM_ERR_STATE while an env's last reset named this stream - also an env that has finished: reset it onto
 * another stream first. */
int lobsim_drop_stream(lobsim_t* h, int32_t stream_id);

/* Portfolio carried by each env (HistoricalOrderbookEnvironment.initial_portfolio, :104).  The reference does NOT
 * reset it between episodes; neither does lobsim_reset. */
int lobsim_set_portfolio(lobsim_t* h, const int32_t* env_ids, int32_t n, const int64_t* inventory, const int64_t* cash);

/* env.reset() for n envs: snapshot at (episode_start - warmup), warm-up replay without agent orders, first
 * observation.  episode_start_us must be whole seconds.  obs (host, [n][obs_dim] f64) may be NULL. */
int lobsim_reset(lobsim_t* h, const int32_t* env_ids, int32_t n, const int32_t* stream_ids,
                 const int64_t* episode_start_us, double* obs);

/* env.step(action) for every env with active[i] != 0 (active == NULL: all).  Host pointers:
 * actions [n_envs][action_dim] f64 -> obs [n_envs][obs_dim] f64, reward [n_envs] f64, done [n_envs] u8. */
int lobsim_step(lobsim_t* h, const double* actions, const uint8_t* active, double* obs, double* reward, uint8_t* done);

/* Same with DEVICE pointers, asynchronous on the handle's stream (no host sync).  obs_f32 (device, may be NULL)
 * additionally receives the observation as float32, the dtype of observation_space (:131-135); d_return_acc (device,
 * may be NULL) accumulates reward per env (episode returns for the rollout's gather) in the kernel epilogue. */
int lobsim_step_device(lobsim_t* h, const double* d_actions, const uint8_t* d_active, double* d_obs, float* d_obs_f32,
                       double* d_reward, uint8_t* d_done, double* d_return_acc);
int lobsim_sync(lobsim_t* h);

/* Open-loop / rule-based rollouts: up to T consecutive env.step()s per env in ONE launch, the action of every step
 * produced on the device (the reference's rule-based baselines, rl4mm/agents/baseline_agents.py, and its
 * generate_trajectory loop, rl4mm/gym/utils.py:100-117).  An env stops early when it reports done.
 *   policy LOBSIM_POLICY_GIVEN      d_actions [T][n_envs][action_dim] f64 (e.g. pre-drawn RandomAgent actions)
 *   policy LOBSIM_POLICY_FIXED      FixedActionAgent: params[0 .. action_dim) is the action
 *   policy LOBSIM_POLICY_TERADACTYL Teradactyl.get_action (baseline_agents.py:58-87) on the observation entry
 *                                   params = {max_inventory, default_kappa, default_omega, max_kappa, exponent,
 *                                             inventory_index, market_clearing (0/1), denom (max_inventory or 100)}
 * Outputs (device pointers, any of d_obs / d_obs_f32 / d_return_acc / d_steps_taken may be NULL): observation, reward
 * and done flag of the LAST step each env took, the sum of its rewards added to d_return_acc, the number of steps it
 * took.  Asynchronous on the handle's stream. */
enum lobsim_policy { LOBSIM_POLICY_GIVEN = 0, LOBSIM_POLICY_FIXED = 1, LOBSIM_POLICY_TERADACTYL = 2 };
int lobsim_rollout_device(lobsim_t* h, int32_t policy, const double* params8, int32_t T, const double* d_actions,
                          const uint8_t* d_active, double* d_obs, float* d_obs_f32, double* d_reward, uint8_t* d_done,
                          double* d_return_acc, int32_t* d_steps_taken);

/* --- info arrays (rl4mm/gym/order_tracking/InfoCalculators.py:17-89, SimpleInfoCalculator) ------------------------- */
/* One row of doubles per env: what SimpleInfoCalculator.calculate(state_after_step, action) returns, computed in the step
 * kernel's epilogue.  (Its `agent_weighted_spread` entry is the literal [1] in the reference; the value it computes and
 * then drops is kept here.)  market_order_count / market_order_total_volume accumulate for the lifetime of the env, like the
 * reference's calculator object. */
enum lobsim_info_col {
    LOBSIM_INFO_ASSET_PRICE = 0,
    LOBSIM_INFO_INVENTORY = 1,
    LOBSIM_INFO_CASH = 2,
    LOBSIM_INFO_AUM = 3,
    LOBSIM_INFO_MARKET_SPREAD = 4,
    LOBSIM_INFO_AGENT_SPREAD = 5,
    LOBSIM_INFO_AGENT_WEIGHTED_SPREAD = 6,
    LOBSIM_INFO_MIDPRICE_OFFSET = 7,
    LOBSIM_INFO_WEIGHTED_MIDPRICE_OFFSET = 8,
    LOBSIM_INFO_MARKET_ORDER_COUNT = 9,
    LOBSIM_INFO_MARKET_ORDER_TOTAL_VOLUME = 10,
    LOBSIM_INFO_DIM = 11
};
/* d_info: DEVICE buffer [n_envs][LOBSIM_INFO_DIM] f64 that every later lobsim_step / lobsim_step_device /
 * lobsim_rollout_device fills for the envs that stepped (NULL: off, the default - the reference's default is
 * info_calculator=None).  Rows of envs that did not step keep their previous contents. */
int lobsim_set_info_buffer(lobsim_t* h, double* d_info);
/* Per-step series of a multi-step launch, the columns get_episode_summary_dict reads (rl4mm/gym/utils.py:145-198):
 * d_trace: DEVICE buffer [t_capacity][n_envs][LOBSIM_TRACE_DIM] f64; lobsim_rollout_device (1 <= T <= t_capacity) writes row t of
 * every env that took step t.  A step that ends with a status bit (the reference would have raised there) counts in
 * d_steps_taken - the env did step, into its error state - but writes NO row: pre-fill the buffer with NaN and the failed
 * step shows as a NaN row.  NULL: off. */
enum lobsim_trace_col {
    LOBSIM_TRACE_REWARD = 0,
    LOBSIM_TRACE_AUM = 1,
    LOBSIM_TRACE_INVENTORY = 2,
    LOBSIM_TRACE_ASSET_PRICE = 3,
    LOBSIM_TRACE_MARKET_SPREAD = 4,
    LOBSIM_TRACE_WEIGHTED_MIDPRICE_OFFSET = 5,
    LOBSIM_TRACE_ACTION = 6, /* .. +LOBSIM_MAX_ACTION_DIM: the action taken (what an on-device policy chose) */
    LOBSIM_TRACE_DIM = 11
};
int lobsim_set_trace_buffer(lobsim_t* h, double* d_trace, int32_t t_capacity);

int lobsim_obs_dim(const lobsim_t* h);

/* What this handle can hold per env (the reference's containers are unbounded; each of these is a status bit when exceeded). */
typedef struct lobsim_limits {
    int32_t level_slabs;     /* price levels an env can hold in all, both sides together (= 2 x the effective max_levels) */
    int32_t ladder_levels;   /* levels per side kept on chip (64 / 128); a performance property, bounds nothing */
    int32_t far_levels;      /* levels per side beyond the on-chip ladder (2048) */
    int32_t queue_slots;     /* live orders + tombstones at one price (4096; 1024 until ABI 6) */
    int32_t agent_orders;    /* live agent orders (cfg.agent_orders: 256 or 128) */
    int32_t fill_log_rows;   /* rows per list of the fill log of lobsim_forward */
    int64_t bytes_per_env;   /* device memory per env */
    int32_t queue_pages;     /* 1 KiB pages (64 order slots) of queue memory per env.  A level's queue is 1, 4, 16 or 64 pages - it moves up when
                                it fills its pages before 128 tombstones have piled up - cut from this pool; with no larger piece left a
                                level stays where it is (and squeezes its tombstones out more often): LOBSIM_ST_QUEUE_OVERFLOW only when its
                                pages are full of LIVE orders, LOBSIM_ST_LEVEL_OVERFLOW when a new level finds no page */
    int32_t reserved;
} lobsim_limits;
int lobsim_get_limits(const lobsim_t* h, lobsim_limits* out);

/* Status bits (enum lobsim_status) of every env into a host array [n_envs]: which envs stopped, and why.  The batched env asks
 * when a step reported done, to tell an episode's end from an error (the reference raises there: Exchange.py:32-37,180-186). */
int lobsim_get_status(lobsim_t* h, int32_t* status);

/* --- host-mapped step I/O ------------------------------------------------------------------------------------------------- */
/* The host-pointer lobsim_step without its copies: pinned host buffers the GPU reads the actions from and writes observations,
 * rewards and done flags to directly (no hipMemcpy in the step).  The caller fills io.actions [n_envs][action_dim] (and
 * io.active [n_envs] when it passes use_active), calls lobsim_step_host_io, and reads io.obs [n_envs][obs_dim] / io.obs_f32 /
 * io.reward / io.done in place once it returns (it synchronises).  The buffers belong to the handle and live as long as it does.
 * want_obs: bit 0 = f64 observations, bit 1 = f32 (the dtype of observation_space, HistoricalOrderbookEnvironment.py:131-135). */
typedef struct lobsim_host_io {
    double* actions;
    uint8_t* active;
    double* obs;
    float* obs_f32;
    double* reward;
    uint8_t* done;
} lobsim_host_io;
int lobsim_host_io_buffers(lobsim_t* h, lobsim_host_io* out);
int lobsim_step_host_io(lobsim_t* h, int32_t use_active, int32_t want_obs);

/* lobsim_step_device where the mask says which envs NOT to step: d_skip [n_envs] (device, may be NULL), envs with d_skip[i] != 0
 * are left alone.  d_skip may be the very array the done flags are written to: an env that reported done then stays as it is -
 * outputs included - until the caller resets it and clears its flag (a closed loop on the device with no host round trip per step). */
int lobsim_step_device_skip(lobsim_t* h, const double* d_actions, const uint8_t* d_skip, double* d_obs, float* d_obs_f32,
                            double* d_reward, uint8_t* d_done, double* d_return_acc);
/* env.reset() with device outputs: like lobsim_reset, the first observation of env e = env_ids[i] written to ROW e of the caller's
 * d_obs [n_envs][obs_dim] f64 and / or d_obs_f32 (device pointers, either may be NULL: the buffers it also hands to
 * lobsim_step_device), and d_done[e] cleared when d_done is given.  Synchronises (the arguments are host arrays). */
int lobsim_reset_device(lobsim_t* h, const int32_t* env_ids, int32_t n, const int32_t* stream_ids, const int64_t* episode_start_us,
                        double* d_obs, float* d_obs_f32, uint8_t* d_done);

/* --- inspection (parity tests, info dicts) ---------------------------------------------------------------- */
int lobsim_get_env_state(lobsim_t* h, int32_t env, lobsim_env_state* out);
/* Device-side bulk read of (inventory, cash, price) for all envs into host arrays (any may be NULL). */
int lobsim_get_portfolios(lobsim_t* h, int64_t* inventory, int64_t* cash, double* price);
/* Best prices of the central book of every env (Orderbook.best_buy_price / best_sell_price, models.py:71-77): 0 when the
 * buy side is empty, INT64_MAX when the sell side is empty.  market spread = best_sell - best_buy (info calculators). */
int lobsim_get_best_prices(lobsim_t* h, int64_t* best_buy, int64_t* best_sell);
/* Central book, best level first: prices/volumes/order counts of up to max_levels levels of one side. */
int lobsim_get_book(lobsim_t* h, int32_t env, int32_t side, int64_t* prices, int64_t* volumes, int32_t* n_orders,
                    int32_t max_levels, int32_t* n_levels_out);
/* FIFO queue at one price. */
int lobsim_get_level_orders(lobsim_t* h, int32_t env, int32_t side, int64_t price, lobsim_order_view* out,
                            int32_t max_orders, int32_t* n_out);

/* --- engine-level entry (unit parity of Exchange.process_order, Exchange.py:58) ---------------------------- */
/* Start env from an explicit snapshot row [4 * n_levels] (sell_p, sell_v, buy_p, buy_v per level; pass NULL for an
 * empty book) at time now_us, without a stream. */
int lobsim_debug_set_book(lobsim_t* h, int32_t env, const int64_t* snapshot_row, int64_t now_us); /* price 0 = absent level */
/*   now_us == -1: the env is a simulator that HAS been reset (lobsim_reset_episode) - its book is replaced by the row, clock, stream and cursor
 *   stay, the initial price ranges are those of the new book: OrderbookSimulator.reset_episode(start_date, start_book), OrderbookSimulator.py:55-68.
 *   now_us == -2: nothing but the initial price ranges, worked out again from the book as it is (after lobsim_debug_process_orders added the
 *   orders a start book holds beyond one per level).  An empty side: LOBSIM_ST_EMPTY_BOOK (the reference's StopIteration, Exchange.py:158-169). */
/* Process orders [n][6] = {type, side, price, volume (-1 = None), id, who} in sequence.  who = 1: an external order, id = its external id.
 * who = 0: the agent's own - a limit order with id > 0 is filed under that TAG (lobsim_order_view.reserved; ids must ascend in submission
 * order, as the reference's internal ids do), id = 0 takes the env's counter; a cancellation / deletion names its target by FIFO RANK among
 * the agent's orders at that price (as in lobsim_order_view.id; a rank that does not exist addresses no order).  who = 2: the agent's
 * cancellation / deletion naming its target by TAG - what the reference's Cancellation carries (its internal_id: Exchange.py:196-217); a tag no
 * live agent order at that side and price holds addresses no order (the reference warns and, if the level's head is a snapshot order, reduces
 * that: Exchange.py:128-139 - the same here).  fills_out [8] (may be
 * NULL) receives the aggregate fills {agent_fill_count, agent_buy_vol, agent_sell_vol, ext_buy_count,
 * ext_sell_count, ext_buy_vol, ext_sell_vol, cash_delta}. */
int lobsim_debug_process_orders(lobsim_t* h, int32_t env, const int64_t* orders, int32_t n, int64_t* fills_out);
/* The memory-safety net of the diagnostic build (liblobsim_bounds.so = -DLOBSIM_BOUNDS -DLOBSIM_REDZONE; the pool this was built on has no GPU
 * AddressSanitizer): 4 KiB sentinel zones in front of and behind every per-env device array and, for the arrays whose per-env stride is a
 * run-time value (queue pages, feature rings, normalisation histories), behind every env's part.  *n_zones = zones looked at, *n_damaged = zones
 * that no longer hold the sentinel (lobsim_last_error names the first).  The product library has no zones: 0 / 0.  The reference is Python: an
 * index out of range raises there (IndexError / KeyError), it cannot write into a neighbouring object. */
int lobsim_debug_check_redzones(lobsim_t* h, int32_t* n_zones, int32_t* n_damaged);

/* --- simulator-level entry (rl4mm/simulation/OrderbookSimulator.py:55-88 on its own: no env, no features, no warm-up) -------- */
/* OrderbookSimulator.reset_episode(start) (:55-68) for n envs: central book <- the last snapshot at or before start_us
 * (get_historical_start_book, :90-97), internal book emptied, initial price ranges (:185-188), clock = start_us (whole seconds,
 * :61).  No snapshot there: LOBSIM_ST_STREAM_END on that env. */
int lobsim_reset_episode(lobsim_t* h, const int32_t* env_ids, int32_t n, const int32_t* stream_ids, const int64_t* start_us);
enum lobsim_forward_flags {
    LOBSIM_FWD_UPDATE_FEATURES = 1 /* after the step: state.price = microprice, Feature.update(state) for every feature of the
                                      cfg on the step's fills (Features.py:80-86), values to obs - the reference's feature tests */
};
/* OrderbookSimulator.forward_step(until, internal_orders) (:70-88) for n envs: env i first processes its n_orders[i] internal
 * orders - rows orders[i][0 .. n_orders[i]) of [max_orders][6], the row format of lobsim_debug_process_orders - then the
 * messages of (now, until_us[i]], sets its clock to until_us[i] and, on a whole second and near the edge of its initial price
 * range, runs update_outer_levels (:99-135).  until <= now is the reference's assert (:71-73), an exception that leaves the simulator as it
 * was: the clocks of all envs of the call are checked first and the call returns LOBSIM_ERR_ARG with NO env stepped (lobsim_last_error lists the
 * envs; until ABI 5 the other envs of the call had already moved).  Every fill of the call
 * goes to the env's fill log (lobsim_get_fills).  orders / n_orders may be NULL (no internal orders); obs (host, [n][obs_dim])
 * may be NULL. */
int lobsim_forward(lobsim_t* h, const int32_t* env_ids, int32_t n, const int64_t* until_us, const int64_t* orders,
                   const int32_t* n_orders, int32_t max_orders, int32_t flags, double* obs);
/* FilledOrders of the env's last lobsim_forward (rl4mm/orderbook/models.py:104-107): rows {list, side, price, volume} with
 * list 0 = .internal (the agent's resting orders that were hit, by their side; the agent's aggressing fills, one MarketOrder copy
 * at the fill price per resting order), then list 1 = .external (resting external orders, by their side), each in fill order.
 * *n_out = rows of the step; at most max_rows are written.  LOBSIM_ERR_NOMEM if the step made more fills per list than the log
 * holds (lobsim_limits.fill_log_rows; the rows that fit are still returned). */
int lobsim_get_fills(lobsim_t* h, int32_t env, int64_t* out, int32_t max_rows, int32_t* n_out);
/* Feature.reset(state) on the simulator's current state for every feature of the cfg (Features.py:88-96), first usage = now. */
int lobsim_features_reset(lobsim_t* h, const int32_t* env_ids, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* LOBSIM_H */
