// engine.hip -- host side of the self-play pool and the C ABI (include/corintho_hip.h).
//
// Mirrors the reference Trainer (corintho_ai/cpp/src/trainer.cpp) method by
// method; the per-game work of every method runs in the kernels of kernels.h.
// Compat mode keeps the reference protocol (the caller evaluates the network
// between doIteration calls, main.pyx:123-187); fused mode (nn_*.hip) keeps
// the whole loop on the device.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../../include/corintho_hip.h"
#include "kernels.h"
#include "nn.h"
#include "logfmt.h"
#include "rt.h"

#ifdef CO_EMU
thread_local int co_emu_block_idx = 0;
#ifdef CO_SB_STATS
unsigned long long co_sb_stats[32];
unsigned long long co_sb_ply[8][8];
extern "C" unsigned long long *co_emu_sb_stats(void) { return co_sb_stats; }
extern "C" unsigned long long *co_emu_sb_ply(void) { return &co_sb_ply[0][0]; }
#endif
#endif

static thread_local std::string g_last_error;
extern "C" const char *ca_last_error(void) { return g_last_error.c_str(); }

#define CA_TRY try {
#define CA_CATCH                          \
  }                                       \
  catch (const std::exception &e) {       \
    g_last_error = e.what();              \
    return CA_ERR_DEVICE;                 \
  }

struct EngineError : std::runtime_error {
  int code;
  EngineError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

extern "C" int ca_device_check(int device) {
#ifdef CO_EMU
  (void)device;
  return CA_OK;
#else
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= device) {
    g_last_error = "no HIP device " + std::to_string(device) + " visible";
    return CA_ERR_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    g_last_error = "hipGetDeviceProperties failed";
    return CA_ERR_DEVICE;
  }
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    g_last_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return CA_ERR_DEVICE;
  }
  return CA_OK;
#endif
}

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  void alloc(size_t count, rt_stream_t s) { /* zeroed, the clear queued on s (rt.h) */
    release();
    n = count;
    rt_malloc((void **)&p, count * sizeof(T), s);
  }
  void release() {
    if (p) rt_free(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
};

#define CO_MAX_POOLS 4
#ifndef CO_POOL_POLL
#define CO_POOL_POLL 16 /* fused training: iterations between polls of a pool's counter (round 5, three pools: 4 / 8 / 16 / 32 / 64 = 134.4 / 131.6 / 130.0 / 129.8 / 130.0 ms per mlp12x100 generation, 395.0 / 393.6 / 391.9 / 391.8 / 392.9 with rescnn4) */
#endif

/* one independent slice of the games in fused training (run_pools) */
struct Pool {
  /* (every handle starts null: a pool whose set-up failed half way is torn down like any other) */
  rt_stream_t st = {};
  int lo = 0, n = 0, row_base = 0;
  bool finished = false;
  int idle = 0;
  int running = 0; /* games of the pool still running at its last poll */
  rt_event_t ev[2][4] = {}; /* per window parity: start / after search / after cache probe / after network of the TIMED iteration */
  rt_event_t polled[2] = {};
  int launched[2] = {0, 0};  /* iterations queued in the window of that parity */
  int word_iter[2] = {0, 0}; /* Trainer::searches_done_ of the iteration whose counter word was copied */
  int first_start = 0;       /* iteration at which the stagger releases the pool's first game (trainer.cpp:184-186) */
  int timed[2] = {0, 0};     /* the window's last iteration carries the events */
  unsigned long long *word = nullptr; /* pinned: [parity] counter word copied at the end of window parity 0 / 1, [2 + parity] rows the
                             * network evaluated in that iteration (evaluation cache) */
  /* the pool's view of the evaluation cache (EvalCache).  The table (c_hdr, c_val, c_done) is ONE for the trainer and
   * belongs to pool 0 -- the other pools hold copies of the pointers, not buffers of their own */
  EvalCache cache = {};
  uint32_t *c_hdr = nullptr, *c_count = nullptr, *c_done = nullptr;
  float *c_val = nullptr;
  int32_t *c_in_idx = nullptr, *c_out_idx = nullptr;
  unsigned long long *c_totals = nullptr;
  size_t c_entries = 0;
  double c_inserted_est = 0; /* entries taken since the table was last emptied (estimate: timed iteration x window) */
  rt_event_t quiet = {};     /* emptying the shared table: the pool's stream has reached the iteration boundary */
};

struct ca_trainer {
  ca_config cfg;
  int G = 0, spe = 0; /* games of this trainer */
  int R = 0;          /* slots of the pool = games resident at a time; R < G: slots are recycled (EngineParams::results) */
  bool recycle = false;
  uint32_t cap_units = 0;
  rt_stream_t stream = {};
  EngineParams P;
  DevBuf<GameCtl> games;
  DevBuf<TreeCtl> trees;
  DevBuf<uint4> arena, pend_key;
  DevBuf<int32_t> pend_src;
  DevBuf<uint32_t> pend_leaf, pend_path, pend_n, noise_raw, rng;
  DevBuf<int32_t> pend_depth, req_offset, trace, all_done;
  DevBuf<int32_t> logbuf; /* per-game text logs: EngineParams::log */
  DevBuf<int32_t> log_index; /* tournament: EngineParams::log_index */
  std::vector<int> log_game;          /* record k belongs to game log_game[k] ... */
  std::vector<std::string> log_paths; /* ... and goes to this file */
  int num_logged = 0;
  bool logs_written = false;
  DevBuf<float> req, nn_in, nn_in70, nn_eval, nn_probs, samples;
  DevBuf<int32_t> row_idx;
  DevBuf<unsigned long long> row_counter, pack_counter, work_counter, prof, next_game;
  DevBuf<GameCtl> results;   /* [G] finished games by index (recycling pools) */
  DevBuf<uint32_t> seeds_dev; /* [G] per-game generator seeds (recycling pools) */
  DevBuf<int32_t> ctl;
  int32_t *h_ctl = nullptr; /* pinned: {batch rows, all done, any error, games not done} of the last scan */
  /* buffers of the host-facing calls, kept (and grown on demand) instead of allocated per call */
  DevBuf<int32_t> ws_off, fw_rows;
  DevBuf<float> ws_gs, ws_ev, ws_pr, fw_in70, fw_in, fw_ev, fw_pr;
  /* caller buffers page-locked for direct DMA (ca_trainer_pin_host) */
  std::vector<std::pair<void *, size_t>> host_regs;
  /* host state */
  int64_t iterations = 0;
  int32_t trainer_iteration = 0; /* Trainer::searches_done_ (train mode only) */
  int scan_valid_for = 0;      /* to_play / model id the current req_offset/nn_in describe (if scan_valid) */
  int32_t last_total = 0;
  bool finished = false;
  std::vector<GameCtl> host_games;
  bool host_games_valid = false;
  /* fused mode */
  std::unique_ptr<CoNet> nets[2];
  double mcts_ms = 0, nn_ms = 0, pack_ms = 0;
  int64_t mcts_launches = 0, nn_launches = 0, nn_rows = 0;
  /* fused training: the launches that carried timing events (one per pool and window) */
  double mcts_timed_ms = 0, nn_timed_ms = 0, pack_timed_ms = 0;
  int64_t timed_launches = 0, nn_timed_rows = 0;
  int64_t nn_rows_evaluated = 0; /* rows the network kernels worked on (= nn_rows without the evaluation cache) */
  int64_t steps_cut = 0, step_budget_last = 0; /* ca_config.step_budget: steps of the last run that stopped at their budget; the last budget */
  bool cache_clean = false;      /* the pools' tables hold nothing of an earlier generation */
  int64_t cache_clears = 0;
  /* the evaluation cache serves fused training (and fused analysis): one network, rows packed by the search kernel */
  /* Not for the analysis of caller-given positions (the table's header encoding relies on boards that doMove
   * produced).  Automatic (eval_cache = 0): only for a network whose rows cost more than resolving them -- the 9.65
   * MFLOP residual CNN, not the 0.25 MFLOP MLP (measured: the MLP's whole launch is 43 us for 12 k rows, the probe 15). */
  bool use_cache() const {
    if (cfg.eval_cache < 0 || tourney || cfg.analyse) return false;
    if (cfg.eval_cache > 0) return true;
    return nets[0] && nets[0]->flop_per_row() >= 1e6;
  }

  /* tournament mode (ca_tourney): per-match players, per-match seeds, the reference's read offsets */
  bool tourney = false;
  std::vector<PlayerCfg> host_pcfg; /* [2G] */
  std::vector<uint32_t> match_seeds; /* [G] */
  DevBuf<PlayerCfg> pcfg;
  DevBuf<int32_t> arena_state; /* fused arena: see EngineParams::arena_state */
  DevBuf<int32_t> read_offset;
  bool scan_valid = false;

  /* analysis mode: the positions (DockerMC constructor arguments) */
  std::vector<uint32_t> an_pos;   /* [G][3] board lo, board hi, meta */
  std::vector<uint32_t> an_seed;  /* [G] */
  std::vector<int32_t> an_pre;    /* [G] Node result of a position that is terminal as given, else 0 */

  std::vector<Pool> pools;
  void free_pools() {
    for (auto &q : pools) {
      for (int w = 0; w < 2; ++w) {
        for (auto &e : q.ev[w]) rt_event_destroy(e);
        rt_event_destroy(q.polled[w]);
      }
      rt_host_free(q.word);
      rt_event_destroy(q.quiet);
      const bool owner = &q == &pools[0]; /* the shared table is pool 0's */
      for (void *b : {owner ? (void *)q.c_hdr : nullptr, (void *)q.c_count, owner ? (void *)q.c_val : nullptr, (void *)q.c_in_idx,
                      (void *)q.c_out_idx, (void *)q.c_totals, owner ? (void *)q.c_done : nullptr})
        rt_free(b);
      rt_stream_destroy(q.st);
    }
    pools.clear();
  }
  ~ca_trainer() {
    free_pools();
    for (auto &r : host_regs) rt_host_unregister(r.first);
    rt_host_free(h_ctl);
    rt_stream_destroy(stream);
  }

  template <typename T>
  void ensure(DevBuf<T> &b, size_t count) {
    if (b.n < count) b.alloc(count + count / 4, stream);
  }

  /* Page-lock a caller buffer (the three arrays of main.pyx:132-134 live as long as the Trainer):
   * copies from / to it are then direct DMA at PCIe speed instead of staged through the runtime's
   * bounce buffers.  The buffer must stay allocated until it is unpinned or the trainer destroyed. */
  bool pin_host(void *p, size_t bytes) {
    for (auto &r : host_regs)
      if (r.first == p) {
        if (r.second >= bytes) return true;
        rt_host_unregister(p);
        r = host_regs.back();
        host_regs.pop_back();
        break;
      }
    if (!rt_host_register(p, bytes)) return false;
    host_regs.emplace_back(p, bytes);
    return true;
  }
  void unpin_host(void *p) {
    for (size_t i = 0; i < host_regs.size(); ++i)
      if (host_regs[i].first == p) {
        rt_host_unregister(p);
        host_regs[i] = host_regs.back();
        host_regs.pop_back();
        return;
      }
  }

  void init(const ca_config &c) {
    cfg = c;
    if (cfg.analyse) {
      cfg.testing = 1;    /* trainmc.cpp:43 */
      cfg.no_stagger = 1;
      cfg.game_base = cfg.total_games = 0;
    }
    if (cfg.max_searches <= 0) cfg.max_searches = 1600;
    if (cfg.searches_per_eval <= 0) cfg.searches_per_eval = 16;
    if (cfg.c_puct <= 0.0f) cfg.c_puct = 1.0f;
    G = cfg.num_games;
    spe = cfg.searches_per_eval;
    rt_set_device(cfg.device);
    rt_stream_create(&stream);
    uint32_t cap = cfg.arena_units;
    if (cap == 0) {
      /* a tree gains at most one node per simulation on its own turns; a node is
       * 2 + (legal moves) units, ~34 on average.  Sized for ~20 own turns of
       * typical growth; overflow is detected and reported, never silent. */
      uint64_t nodes = (uint64_t)cfg.max_searches * 14 + 64;
      cap = (uint32_t)std::min<uint64_t>(nodes * 40, 0x7FFFFFF0ull);
    }
    cap_units = cap;
    /* Resident slots.  The reference holds every game's trees at once and staggers the starts to bound them
     * (trainer.cpp:184-186); here `resident` slots hold the games in play and a slot whose game ends takes the
     * next one (training only: an arena game's trajectory depends on its batch, quirk 12).  0 = automatic: all
     * games resident if their trees fit in 5/8 of the free device memory, else as many slots as fit. */
    R = G;
    const bool can_recycle = !cfg.testing && !cfg.analyse && !tourney;
    const size_t per_slot = (size_t)2 * ((size_t)cap + CO_ARENA_PAD) * sizeof(uint4) +
                            (size_t)spe * (CO_PATH_MAX * 4 + CO_NUM_MOVES * 4 + 2 * CO_STATE_STRIDE * 4 + CO_GAME_STATE_SIZE * 4 +
                                           CO_NUM_MOVES * 4 + 64) + CO_MT_N * 4 + 512;
    if (can_recycle && cfg.resident > 0 && cfg.resident < G) R = cfg.resident;
    if (can_recycle && cfg.resident == 0) {
      const size_t budget = rt_mem_free() / 8 * 5;
      const size_t per_game = (size_t)CO_MAX_PLIES * CO_SAMPLE_FLOATS * 4 + sizeof(GameCtl) + 8;
      if ((size_t)G * (per_slot + per_game) > budget) {
        size_t fit = budget > (size_t)G * per_game ? (budget - (size_t)G * per_game) / per_slot : 0;
        if (fit < 1) throw EngineError(CA_ERR_DEVICE, "not enough device memory for a single resident game");
        R = (int)std::min<size_t>(fit, (size_t)G);
        if (R >= 512) R &= ~255; /* whole workgroups of the network kernels' row tiles */
      }
    }
    recycle = R < G;
    size_t T = (size_t)2 * R;
    games.alloc(R, stream);
    trees.alloc(T, stream);
    arena.alloc(T * ((size_t)cap + CO_ARENA_PAD), stream);
    pend_leaf.alloc((size_t)R * spe, stream);
    pend_depth.alloc((size_t)R * spe, stream);
    pend_n.alloc((size_t)R * spe * 4, stream);
    if (cfg.eval_cache >= 0 && !tourney && !cfg.analyse) {
      pend_key.alloc((size_t)R * spe, stream);
      pend_src.alloc((size_t)R * spe, stream);
    }
    noise_raw.alloc((size_t)R * spe * CO_NUM_MOVES, stream);
    pend_path.alloc((size_t)R * spe * CO_PATH_MAX, stream);
    rng.alloc((size_t)R * CO_MT_N, stream);
    req.alloc((size_t)R * spe * CO_STATE_STRIDE, stream);
    req_offset.alloc((size_t)R + 1, stream);
    nn_in.alloc((size_t)R * spe * CO_STATE_STRIDE, stream);
    row_idx.alloc((size_t)R * spe, stream);
    nn_in70.alloc((size_t)R * spe * CO_GAME_STATE_SIZE, stream);
    ctl.alloc(4, stream);
    rt_host_alloc((void **)&h_ctl, 16);
    nn_eval.alloc((size_t)R * spe, stream);
    nn_probs.alloc((size_t)R * spe * CO_NUM_MOVES, stream);
    if (!cfg.testing) samples.alloc((size_t)G * CO_MAX_PLIES * CO_SAMPLE_FLOATS, stream); /* by game */
    if (cfg.trace) trace.alloc((size_t)G * CO_TRACE_CAP, stream);                            /* by game */
    if (recycle) {
      results.alloc(G, stream);
      seeds_dev.alloc(G, stream);
    }
    next_game.alloc(1, stream);
    all_done.alloc(1, stream);
    row_counter.alloc(1, stream);
    pack_counter.alloc((size_t)CO_PACK_STRIDE * CO_MAX_POOLS, stream); /* a 128-byte line per pool: every wavefront of a launch adds to its pool's words */
    work_counter.alloc((size_t)CO_WC_WORDS * CO_MAX_POOLS, stream);
    arena_state.alloc(8, stream);
    if (tourney) {
      pcfg.alloc(host_pcfg.size(), stream);
      rt_h2d(pcfg.p, host_pcfg.data(), host_pcfg.size() * sizeof(PlayerCfg), stream);
      read_offset.alloc((size_t)R, stream);
    }

    /* (an analysis trainer is reset by set_positions, once the positions and their seeds are known) */
    if (!cfg.analyse) reset_games(cfg.seed);
    memset(&P, 0, sizeof P);
    fill_params(cap, cfg.total_games > 0 ? cfg.total_games : G);
  }

  /* Trainer::initialize (trainer.cpp:238-256): game i is seeded with the i-th
   * output of mt19937(seed), in global game order.  Also used to start a new
   * generation in the same pool (ca_trainer_reset). */
  void reset_games(int32_t seed) {
    cfg.seed = seed;
    size_t T = (size_t)2 * R;
    int total = cfg.total_games > 0 ? cfg.total_games : G;
    std::mt19937 gen((uint32_t)cfg.seed);
    std::vector<uint32_t> seeds(total);
    for (int i = 0; i < total; ++i) seeds[i] = (uint32_t)gen();
    std::vector<uint32_t> st((size_t)R * CO_MT_N);
    std::vector<GameCtl> hg(R);
    std::vector<TreeCtl> ht(T);
    for (int g = 0; g < R; ++g) { /* slot g starts with game g */
      uint32_t *x = &st[(size_t)g * CO_MT_N];
      x[0] = tourney ? match_seeds[g] : (cfg.analyse && !an_seed.empty()) ? an_seed[g] : seeds[cfg.game_base + g];
      for (int i = 1; i < CO_MT_N; ++i) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
      memset(&hg[g], 0, sizeof(GameCtl));
      hg[g].gid = g;
      hg[g].parity = (cfg.game_base + g) % 2;
      hg[g].rng_idx = CO_MT_N;
      hg[g].pos_meta = CO_META_START; /* Match::root_ = Node{} (match.h:91): the empty board */
      if (cfg.analyse && !an_pos.empty()) {
        hg[g].parity = 0;
        hg[g].pos_lo = an_pos[3 * g];
        hg[g].pos_hi = an_pos[3 * g + 1];
        hg[g].pos_meta = an_pos[3 * g + 2];
        if (an_pre[g]) hg[g].done = 1; /* choose_move.pyx:194-197: a terminal position is not searched */
      }
    }
    if (cfg.analyse && !an_pos.empty()) {
      /* result rows of the positions that were terminal as given */
      std::vector<uint32_t> rows((size_t)R * spe * CO_STATE_STRIDE, 0u);
      for (int g = 0; g < R; ++g)
        if (an_pre[g]) {
          uint32_t *o = &rows[(size_t)g * spe * CO_STATE_STRIDE];
          o[0] = 0xFFFFFFFFu;
          o[1] = (uint32_t)an_pre[g];
          o[2] = 1u;
          o[7] = 1u;
        }
      rt_h2d(req.p, rows.data(), rows.size() * 4, stream);
    }
    for (size_t t = 0; t < T; ++t) {
      ht[t].root = CO_NONE;
      ht[t].searches_done = 0;
      ht[t].units_used = 0;
      ht[t].peak_units = 0;
    }
    rt_h2d(rng.p, st.data(), st.size() * 4, stream);
    rt_h2d(games.p, hg.data(), hg.size() * sizeof(GameCtl), stream);
    rt_h2d(trees.p, ht.data(), ht.size() * sizeof(TreeCtl), stream);
    if (recycle) {
      /* the games behind the first R: their seeds (the Trainer stream in game order) and the counter they are taken from */
      rt_h2d(seeds_dev.p, seeds.data() + cfg.game_base, (size_t)G * 4, stream);
      rt_memset(results.p, 0, (size_t)G * sizeof(GameCtl), stream);
    }
    const unsigned long long first_unstarted = (unsigned long long)R;
    rt_h2d(next_game.p, &first_unstarted, 8, stream);
    rt_memset(row_counter.p, 0, 8, stream);
    rt_memset(pack_counter.p, 0, (size_t)8 * CO_PACK_STRIDE * CO_MAX_POOLS, stream);
    rt_memset(work_counter.p, 0, (size_t)8 * CO_WC_WORDS * CO_MAX_POOLS, stream);
    rt_memset(arena_state.p, 0, 32, stream);
    rt_sync(stream);
    iterations = 0;
    trainer_iteration = 0;
    scan_valid = false;
    last_total = 0;
    finished = false;
    host_games_valid = false;
    mcts_ms = nn_ms = pack_ms = 0;
    mcts_launches = nn_launches = nn_rows = 0;
    mcts_timed_ms = nn_timed_ms = pack_timed_ms = 0;
    timed_launches = nn_timed_rows = 0;
    nn_rows_evaluated = 0;
    cache_clean = false; /* run_pools empties the tables before the first iteration of the new generation */
    if (logbuf.p) {
      rt_memset(logbuf.p, 0, (size_t)num_logged * CO_LOG_CAP * 4, stream);
      rt_sync(stream);
    }
    logs_written = false;
  }

  /* Trainer::initialize, trainer.cpp:243-250: the first num_logged games write `<log_folder>/game_<i>.txt` (i = the game's
   * index in the generation).  Before the first iteration only. */
  void set_logging(const char *folder, int n) {
    if (tourney || cfg.analyse) throw EngineError(CA_ERR_STATE, "per-game logs belong to Trainer games (self-play or arena)");
    if (iterations != 0 || trainer_iteration != 0) throw EngineError(CA_ERR_STATE, "ca_trainer_set_logging: the games have started");
    if (n < 0) throw EngineError(CA_ERR_ARG, "ca_trainer_set_logging: num_logged < 0");
    n -= cfg.game_base; /* a shard logs the games of the generation's first num_logged that it owns */
    if (n < 0) n = 0;
    if (n > G) n = G;
    if (n > R) /* logged games start in their own slots (mcts.h): fewer slots than logged games would write fewer files than the reference */
      throw EngineError(CA_ERR_ARG, "ca_trainer_set_logging: " + std::to_string(n) + " logged games on " + std::to_string(R) +
                                        " resident slots -- a logged game must start in its own slot; raise ca_config.resident");
    std::vector<int> games;
    std::vector<std::string> paths;
    for (int g = 0; g < n; ++g) {
      games.push_back(g);
      paths.push_back(std::string(folder ? folder : "") + "/game_" + std::to_string(cfg.game_base + g) + ".txt");
    }
    set_log_records(games, paths, false);
  }

  /* record k = game games[k], printed to paths[k]; with_index: the device finds a game's record through
   * EngineParams::log_index (tournament matches added with logging = true) instead of "the first num_logged games" */
  void set_log_records(const std::vector<int> &games, const std::vector<std::string> &paths, bool with_index) {
    log_game = games;
    log_paths = paths;
    num_logged = (int)games.size();
    if (num_logged > 0) {
      logbuf.alloc((size_t)num_logged * CO_LOG_CAP, stream);
      if (with_index) {
        std::vector<int32_t> idx((size_t)G, -1);
        for (int k = 0; k < num_logged; ++k) idx[(size_t)games[k]] = k;
        log_index.alloc((size_t)G, stream);
        rt_h2d(log_index.p, idx.data(), idx.size() * 4, stream);
      }
      rt_sync(stream);
    } else {
      logbuf.release();
      log_index.release();
    }
    P.log = logbuf.p;
    P.num_logged = num_logged;
    P.log_index = with_index ? log_index.p : nullptr;
    logs_written = false;
  }

  /* the files, once every game is over (the reference writes them as the games go; a file that cannot be opened is
   * skipped without a word there too: an ofstream in its fail state) */
  void maybe_write_logs() {
    if (!logbuf.p || logs_written) return;
    logs_written = true;
    fetch_games();
    std::vector<int32_t> rec((size_t)num_logged * CO_LOG_CAP);
    rt_d2h(rec.data(), logbuf.p, rec.size() * 4, stream);
    rt_sync(stream);
    for (int k = 0; k < num_logged; ++k) {
      const int32_t *r = rec.data() + (size_t)k * CO_LOG_CAP;
      if (r[0] < 0 || r[0] > CO_LOG_CAP - 1)
        throw EngineError(CA_ERR_ENGINE, "text log " + log_paths[k] + " does not fit its record (" + std::to_string(r[0]) + " words)");
      FILE *f = fopen(log_paths[k].c_str(), "w");
      if (!f) continue;
      CoLogWriter wr(f);
      const bool ok = wr.write_game(r + 1, r[0], host_games[log_game[k]].result);
      fclose(f);
      if (!ok) throw EngineError(CA_ERR_ENGINE, "malformed text-log record for " + log_paths[k]);
    }
  }

  void fill_params(uint32_t cap, int total) {
    P.num_games = R;
    P.total_local = G;
    P.results = recycle ? results.p : nullptr;
    P.next_game = next_game.p;
    P.seeds = recycle ? seeds_dev.p : nullptr;
    P.max_searches = cfg.max_searches;
    P.searches_per_eval = spe;
    P.c_puct = cfg.c_puct;
    P.epsilon = cfg.epsilon;
    P.testing = cfg.testing;
    size_t div = (size_t)total / (size_t)cfg.max_searches;
    if (div < 1) div = 1;
    /* the staggered start bounds the reference's memory (trainer.cpp:184-186); a recycling pool is bounded by its slots */
    P.stagger_div = (cfg.no_stagger || recycle) ? 0 : (int32_t)div;
    P.iteration = 0;
    P.to_play = -1;
    P.game_base = cfg.game_base;
    P.cap_units = cap;
    P.trace_on = cfg.trace;
    P.analyse = cfg.analyse;
    P.pcfg = tourney ? pcfg.p : nullptr;
    P.read_offset = tourney ? read_offset.p : nullptr;
    P.arena_state = nullptr;
    P.scan_phase = 0;
    P.games = games.p;
    P.trees = trees.p;
    P.arena = arena.p;
    P.pend_leaf = pend_leaf.p;
    P.pend_depth = pend_depth.p;
    P.pend_n = pend_n.p;
    P.pend_key = pend_key.p;
    P.pend_src = pend_src.p;
    P.noise_raw = noise_raw.p;
    P.pend_path = pend_path.p;
    P.rng = rng.p;
    P.req = req.p;
    P.req_offset = req_offset.p;
    P.nn_eval = nn_eval.p;
    P.nn_probs = nn_probs.p;
    P.nn_in = nn_in.p;
    P.row_idx = row_idx.p;
    P.nn_in70 = nn_in70.p;
    P.ctl = ctl.p;
    P.samples = samples.p;
    P.trace = trace.p;
    P.log = logbuf.p;
    P.num_logged = num_logged;
    P.log_index = log_index.p;
    P.all_done = all_done.p;
    P.row_counter = nullptr; /* counted in fused mode only */
    P.fused_pack = 0;
    P.defer_handover = 0;
    /* ca_config.step_budget: n > 0 that many scans, 0 automatic (CO_STEP_BUDGET_K16 / 16 x the pool's mean), -1 none;
     * below -1 (diagnostic): automatic with the factor -n / 16 */
    P.step_budget = cfg.step_budget > 0 ? cfg.step_budget : 0;
    P.step_budget_k16 = cfg.step_budget == 0 ? CO_STEP_BUDGET_K16 : cfg.step_budget < -1 ? -cfg.step_budget : 0;
    P.work_counter = nullptr; /* a pool's own, set by run_pools */
    P.pool_lo = 0;
    P.pool_n = R;
    P.pool_row_base = 0;
    P.pack_counter = pack_counter.p;
#ifdef CO_PROF
    prof.alloc((size_t)R * CO_NPROF + 24 + CO_NPROF, stream);
    P.prof = prof.p;
#else
    P.prof = nullptr;
#endif
  }

  /* offsets + compact batch for model `to_play` (K4); ONE host synchronisation, on 16 bytes */
  bool any_error = false;
  void pack(int to_play) {
    if (scan_valid && scan_valid_for == to_play) return;
    P.to_play = to_play;
    RT_LAUNCH(co_k_scan, 1, CO_WAVE, stream, P);
    RT_LAUNCH(co_k_compact, R, CO_WAVE, stream, P);
    rt_d2h(h_ctl, ctl.p, 16, stream);
    rt_sync(stream);
    last_total = h_ctl[0];
    finished = h_ctl[1] != 0;
    any_error = h_ctl[2] != 0;
    scan_valid_for = to_play;
    scan_valid = true;
    if (finished && !any_error) maybe_write_logs();
  }

  /* host_games[i] = control block of GAME i of this trainer: the slot itself without recycling; else the filed
   * result of a finished game, the slot of a game in play, or an untouched block for a game not yet started */
  void fetch_games() {
    if (host_games_valid) return;
    host_games.resize(G);
    if (!recycle) {
      rt_d2h(host_games.data(), games.p, (size_t)G * sizeof(GameCtl), stream);
      rt_sync(stream);
    } else {
      std::vector<GameCtl> slots(R);
      rt_d2h(slots.data(), games.p, (size_t)R * sizeof(GameCtl), stream);
      rt_d2h(host_games.data(), results.p, (size_t)G * sizeof(GameCtl), stream);
      rt_sync(stream);
      for (int i = 0; i < G; ++i)
        if (!host_games[i].done) {
          memset(&host_games[i], 0, sizeof(GameCtl));
          host_games[i].gid = i;
        }
      for (int sl = 0; sl < R; ++sl) {
        const int i = slots[sl].gid;
        if (i >= 0 && i < G && !host_games[i].done) host_games[i] = slots[sl];
      }
    }
    host_games_valid = true;
  }

  void check_errors() {
    check_net_range();
    if (scan_valid && !any_error) return; /* the last scan saw no error bit in any game */
    fetch_games();
    for (int g = 0; g < G; ++g) {
      if (host_games[g].error) {
        char buf[256];
        int e = host_games[g].error;
        snprintf(buf, sizeof buf, "game %d reported engine error 0x%x (%s%s%s%s)", g, e,
                 (e & CO_ERR_ARENA_FULL) ? "search-tree arena full: raise ca_config.arena_units; " : "",
                 (e & CO_ERR_PATH_TOO_DEEP) ? "search path deeper than CO_PATH_MAX; " : "",
                 (e & CO_ERR_TOO_MANY_PLIES) ? "game longer than CO_MAX_PLIES; " : "",
                 (e & CO_ERR_INTERNAL) ? "internal inconsistency; " : "");
        throw EngineError(CA_ERR_ENGINE, buf);
      }
    }
  }

  /* Trainer::doIteration (trainer.cpp:164-236), compat protocol */
  void need_positions() const {
    if (cfg.analyse && an_pos.empty()) throw EngineError(CA_ERR_STATE, "analysis trainer: ca_trainer_set_positions first");
  }
  bool do_iteration(const float *evals, const float *probs, int to_play) {
    need_positions();
    if (to_play != 0 && to_play != 1) to_play = -1;
    if (iterations > 0) {
      pack(to_play); /* offsets the reference computes at entry */
      if (last_total > 0) {
        if (!evals || !probs) throw EngineError(CA_ERR_ARG, "doIteration: null evaluations/probabilities");
        rt_h2d(nn_eval.p, evals, (size_t)last_total * 4, stream);
        rt_h2d(nn_probs.p, probs, (size_t)last_total * CO_NUM_MOVES * 4, stream);
      }
    } else {
      rt_memset(req_offset.p, 0, ((size_t)R + 1) * 4, stream);
    }
    P.to_play = to_play;
    P.iteration = trainer_iteration;
    RT_LAUNCH(co_k_mcts_step, ((R) + CO_K3_WAVES - 1) / CO_K3_WAVES, CO_WAVE * CO_K3_WAVES, stream, P);
    if (to_play == -1) ++trainer_iteration;
    ++iterations;
    ++mcts_launches;
    scan_valid = false;
    host_games_valid = false;
    pack(to_play);
    check_errors();
    return finished;
  }

  int32_t num_requests(int to_play) {
    if (to_play != 0 && to_play != 1) to_play = -1;
    pack(to_play);
    return last_total;
  }

  void write_requests(float *out, int to_play) {
    if (to_play != 0 && to_play != 1) to_play = -1;
    pack(to_play);
    if (last_total == 0) return;
    /* K4 has laid the rows out as the caller's array holds them: one copy, straight into it */
    rt_d2h(out, nn_in70.p, (size_t)last_total * CO_GAME_STATE_SIZE * 4, stream);
    rt_sync(stream);
  }

  /* ---- Tourney (tourney.cpp) on the same pool: one match per game slot ---- */
  int32_t tourney_num_requests(int id) {
    pack(id);
    return last_total;
  }
  void tourney_write_requests(float *out, int id) {
    pack(id);
    if (last_total == 0) return;
    rt_d2h(out, nn_in70.p, (size_t)last_total * CO_GAME_STATE_SIZE * 4, stream);
    rt_sync(stream);
  }
  /* Tourney::doIteration (tourney.cpp:53-70).  `rows` = rows of the caller's two arrays: the
   * reference reads them at its own offset table (quirk 10), so the whole arrays travel. */
  void tourney_do_iteration(const float *evals, const float *probs, int32_t rows, int id) {
    size_t cap = (size_t)R * spe;
    if (rows < 0 || (size_t)rows > cap) rows = (int32_t)cap;
    P.to_play = id;
    RT_LAUNCH(co_k_scan, 1, CO_WAVE, stream, P); /* offsets at entry */
    if (rows > 0 && evals && probs) {
      rt_h2d(nn_eval.p, evals, (size_t)rows * 4, stream);
      rt_h2d(nn_probs.p, probs, (size_t)rows * CO_NUM_MOVES * 4, stream);
    }
    P.iteration = trainer_iteration;
    RT_LAUNCH(co_k_mcts_step, ((R) + CO_K3_WAVES - 1) / CO_K3_WAVES, CO_WAVE * CO_K3_WAVES, stream, P);
    ++iterations;
    ++mcts_launches;
    scan_valid = false;
    host_games_valid = false;
    pack(id);
    check_errors();
  }
  bool tourney_all_done() {
    fetch_games();
    for (int g = 0; g < G; ++g)
      if (!host_games[g].done) return false;
    return true;
  }

  int32_t num_samples() {
    fetch_games();
    int32_t n = 0;
    for (int g = 0; g < G; ++g) n += host_games[g].n_samples;
    return n;
  }

  /* SelfPlayer::score (selfplayer.cpp:57-64) + Trainer::score (trainer.cpp:59-68) */
  static float game_score(const GameCtl &gc) {
    if (gc.result == CO_RESULT_LOSS) return 0.0f;
    if (gc.result == CO_RESULT_WIN) return 1.0f;
    return 0.5f;
  }
  float score() {
    fetch_games();
    /* colour alternates with the GLOBAL game index (trainer.cpp:61-66) */
    float s = 0;
    for (int g = 0; g < G; ++g)
      if ((cfg.game_base + g) % 2 == 0) s += game_score(host_games[g]);
    for (int g = 0; g < G; ++g)
      if ((cfg.game_base + g) % 2 == 1) s = (float)((double)s + (1.0 - (double)game_score(host_games[g])));
    return s / (float)(size_t)G;
  }
  float avg_mate_length() {
    fetch_games();
    int32_t total = 0;
    for (int g = 0; g < G; ++g) {
      const GameCtl &gc = host_games[g];
      total += gc.mate_turn == 0 ? 0 : gc.n_samples - gc.mate_turn + 1; /* selfplayer.cpp:66-71 */
    }
    return (float)total / (float)(size_t)G;
  }

  /* ws_off = {sample offsets [G + 1], per game (plies | result << 8) [G]}: the sample kernels work by game, not by slot */
  void upload_sample_index(const std::vector<int32_t> &off) {
    std::vector<int32_t> idx(off);
    idx.resize((size_t)2 * G + 1);
    for (int g = 0; g < G; ++g) idx[(size_t)G + 1 + g] = host_games[g].n_samples | (host_games[g].result << 8);
    rt_h2d(ws_off.p, idx.data(), idx.size() * 4, stream);
    rt_sync(stream); /* idx is a local */
  }

  void write_samples(float *gs, float *ev, float *pr) {
    if (cfg.testing) throw EngineError(CA_ERR_STATE, "writeSamples in testing mode");
    fetch_games();
    std::vector<int32_t> off(G + 1, 0);
    for (int g = 0; g < G; ++g) off[g + 1] = off[g] + host_games[g].n_samples;
    size_t n = (size_t)off[G];
    if (n == 0) return;
    ensure(ws_off, (size_t)2 * G + 1);
    ensure(ws_gs, n * 8 * CO_GAME_STATE_SIZE);
    ensure(ws_ev, n * 8);
    ensure(ws_pr, n * 8 * CO_NUM_MOVES);
    upload_sample_index(off);
    RT_LAUNCH(co_k_write_samples, G, CO_WAVE, stream, P, G, (const int32_t *)ws_off.p, (const int32_t *)ws_off.p + G + 1, ws_gs.p,
              ws_ev.p, ws_pr.p);
    rt_d2h(gs, ws_gs.p, n * 8 * CO_GAME_STATE_SIZE * 4, stream);
    rt_d2h(ev, ws_ev.p, n * 8 * 4, stream);
    rt_d2h(pr, ws_pr.p, n * 8 * CO_NUM_MOVES * 4, stream);
    rt_sync(stream);
  }

  void export_samples(float *state_policy, float *outcome) {
    if (cfg.testing) throw EngineError(CA_ERR_STATE, "export_samples in testing mode");
    fetch_games();
    std::vector<float> all((size_t)G * CO_MAX_PLIES * CO_SAMPLE_FLOATS);
    rt_d2h(all.data(), samples.p, all.size() * 4, stream);
    rt_sync(stream);
    size_t row = 0;
    for (int g = 0; g < G; ++g) {
      int n = host_games[g].n_samples;
      for (int i = 0; i < n; ++i, ++row) {
        memcpy(state_policy + row * CO_SAMPLE_FLOATS, &all[((size_t)g * CO_MAX_PLIES + i) * CO_SAMPLE_FLOATS],
               CO_SAMPLE_FLOATS * 4);
        float e = host_games[g].result == CO_RESULT_DRAW ? 0.0f : 1.0f;
        if ((n - 1 - i) & 1) e = (float)((double)e * -1.0);
        outcome[row] = e;
      }
    }
  }

  /* un-augmented samples packed on the device into caller-owned device memory
   * (the multi-GPU gather hands these straight to RCCL) */
  int32_t pack_samples_device(float *d_state_policy, float *d_outcome, int32_t cap_rows) {
    if (cfg.testing) throw EngineError(CA_ERR_STATE, "pack_samples in testing mode");
    fetch_games();
    std::vector<int32_t> off(G + 1, 0);
    for (int g = 0; g < G; ++g) off[g + 1] = off[g] + host_games[g].n_samples;
    if (off[G] > cap_rows) throw EngineError(CA_ERR_ARG, "pack_samples: destination too small");
    if (off[G] == 0) return 0;
    ensure(ws_off, (size_t)2 * G + 1);
    upload_sample_index(off);
    RT_LAUNCH(co_k_pack_samples, G, CO_WAVE, stream, P, G, (const int32_t *)ws_off.p, (const int32_t *)ws_off.p + G + 1,
              d_state_policy, d_outcome);
    rt_sync(stream);
    return off[G];
  }

  /* Trainer::writeScores (trainer.cpp:115-162) */
  void write_scores(const char *file) {
    fetch_games();
    size_t n = (size_t)G;
    std::vector<float> scores(n);
    for (size_t i = 0; i < n; i += 2) scores[i] = game_score(host_games[i]);
    for (size_t i = 1; i < n; i += 2) scores[i] = (float)(1.0 - (double)game_score(host_games[i]));
    FILE *f = fopen(file, "w");
    if (!f) throw EngineError(CA_ERR_IO, std::string("cannot open ") + file);
    const char *who[2] = {"First", "Second"};
    for (int side = 0; side < 2; ++side) {
      int wins = 0, draws = 0;
      for (size_t i = side; i < n; i += 2) {
        if (scores[i] == 1.0f) ++wins;
        else if (scores[i] == 0.5f) ++draws;
      }
      size_t half = n / 2;
      auto ratio = [&](size_t k) { return (double)((float)k / (float)half); };
      fprintf(f, "%s player wins: %d / %zu = %g\n", who[side], wins, half, ratio(wins));
      fprintf(f, "%s player draws: %d / %zu = %g\n", who[side], draws, half, ratio(draws));
      fprintf(f, "%s player losses: %zu / %zu = %g\n", who[side], half - wins - draws, half, ratio(half - wins - draws));
    }
    fclose(f);
  }

  /* ------------------------------------------------------------ analysis mode (DockerMC) */
  void set_positions(const int32_t *boards, const int32_t *to_play, const int32_t *pieces, const int32_t *seeds) {
    if (!cfg.analyse) throw EngineError(CA_ERR_STATE, "set_positions: not an analysis trainer (ca_config.analyse)");
    if (iterations > 0) throw EngineError(CA_ERR_STATE, "set_positions after the first iteration");
    an_pos.assign((size_t)3 * G, 0u);
    an_seed.assign(G, 0u);
    an_pre.assign(G, 0);
    std::vector<uint64_t> hb(G);
    std::vector<uint32_t> hm(G);
    for (int g = 0; g < G; ++g) {
      uint64_t b = 0;
      for (int i = 0; i < 64; ++i) {
        int v = boards[(size_t)g * 64 + i];
        if (v != 0 && v != 1) throw EngineError(CA_ERR_ARG, "set_positions: board entries must be 0 or 1");
        if (v) b |= 1ull << i;
      }
      uint32_t meta = 0;
      for (int i = 0; i < 6; ++i) {
        int pc = pieces[(size_t)g * 6 + i];
        if (pc < 0 || pc > 4) throw EngineError(CA_ERR_ARG, "set_positions: piece counts must be 0..4");
        meta |= (uint32_t)pc << (3 * i);
      }
      if (to_play[g] != 0 && to_play[g] != 1) throw EngineError(CA_ERR_ARG, "set_positions: to_play must be 0 or 1");
      meta |= (uint32_t)to_play[g] << 18;
      an_pos[3 * g] = (uint32_t)b;
      an_pos[3 * g + 1] = (uint32_t)(b >> 32);
      an_pos[3 * g + 2] = meta;
      an_seed[g] = (uint32_t)seeds[g];
      hb[g] = b;
      hm[g] = meta;
    }
    /* Node result of every given position (node.cpp:256-271), by the rule kernel */
    DevBuf<uint64_t> db;
    DevBuf<uint32_t> dm, dk;
    DevBuf<int32_t> dl;
    db.alloc(G, stream); dm.alloc(G, stream); dk.alloc((size_t)G * 3, stream); dl.alloc(G, stream);
    rt_h2d(db.p, hb.data(), (size_t)G * 8, stream);
    rt_h2d(dm.p, hm.data(), (size_t)G * 4, stream);
    RT_LAUNCH(co_k_rules_batch, G, CO_WAVE, stream, (const uint64_t *)db.p, (const uint32_t *)dm.p, G, dk.p, dl.p);
    std::vector<uint32_t> mk((size_t)G * 3);
    std::vector<int32_t> ln(G);
    rt_d2h(mk.data(), dk.p, mk.size() * 4, stream);
    rt_d2h(ln.data(), dl.p, ln.size() * 4, stream);
    rt_sync(stream);
    for (int g = 0; g < G; ++g)
      if ((mk[3 * g] | mk[3 * g + 1] | mk[3 * g + 2]) == 0u) an_pre[g] = ln[g] ? CO_RESULT_LOSS : CO_RESULT_DRAW;
    reset_games(cfg.seed);
  }

  void analysis(int32_t *out) {
    if (!cfg.analyse) throw EngineError(CA_ERR_STATE, "not an analysis trainer");
    /* the eight result words at the head of each slot's request area (mcts.h co_analyse_finish): one strided copy */
    std::vector<uint32_t> rows((size_t)G * 8);
    rt_d2h_2d(rows.data(), 32, req.p, (size_t)spe * CO_STATE_STRIDE * 4, 32, (size_t)G, stream);
    rt_sync(stream);
    fetch_games();
    for (int g = 0; g < G; ++g) {
      const uint32_t *r = &rows[(size_t)g * 8];
      int32_t *o = out + (size_t)g * 8;
      if (!host_games[g].done || r[7] != 1u) throw EngineError(CA_ERR_STATE, "analysis: search of position " + std::to_string(g) + " is not finished");
      const int res = (int)r[1];
      o[0] = (int32_t)r[0];
      o[1] = res == CO_RESULT_LOSS || res == CO_RESULT_DRAW;   /* Node::terminal, node.cpp:96-98 */
      o[2] = res == CO_RESULT_DRAW || res == CO_DEDUCED_DRAW;  /* Node::drawn */
      o[3] = (int32_t)r[2];
      o[4] = (int32_t)r[3];
      o[5] = (int32_t)r[4];
      o[6] = (int32_t)r[5];
      o[7] = (int32_t)r[6];
    }
  }

  /* DockerMC::chooseMove on searches that have not ended (the reference's loop leaves on a time limit,
   * choose_move.pyx:110-117, and then calls chooseMove unconditionally, :199): every unfinished position
   * chooses on its tree as it stands; evaluations still pending are never received (trainmc.cpp:110-137 does
   * not look at searched_).  Before the first iteration the root is created first, as the constructor does. */
  void finish_analysis() {
    if (!cfg.analyse) throw EngineError(CA_ERR_STATE, "ca_trainer_finish: not an analysis trainer");
    if (iterations == 0) do_iteration(nullptr, nullptr, -1);
    P.to_play = -1;
    P.iteration = trainer_iteration;
    P.force_choose = 1;
    RT_LAUNCH(co_k_mcts_step, ((R) + CO_K3_WAVES - 1) / CO_K3_WAVES, CO_WAVE * CO_K3_WAVES, stream, P);
    P.force_choose = 0;
    ++mcts_launches;
    scan_valid = false;
    host_games_valid = false;
    pack(-1);
    check_errors();
  }

  /* ------------------------------------------------------------ fused mode */
  void set_net(int slot, int kind, const float *weights, size_t n) {
    if (slot < 0 || slot > 1) throw EngineError(CA_ERR_ARG, "net slot must be 0 or 1");
    std::unique_ptr<CoNet> fresh;
    try {
      fresh.reset(co_net_create(kind, weights, n, (size_t)R * spe, stream));
    } catch (const std::invalid_argument &e) { /* weights outside the kind's operand range (nn.h range_exceeded) */
      throw EngineError(CA_ERR_ARG, e.what());
    }
    if (!fresh) throw EngineError(CA_ERR_ARG, "unknown net kind or bad weight count");
    if (slot == 0 && !pools.empty()) {
      /* Does the new network change whether fused training keeps an evaluation cache?  The tables are part of the
       * pools, and in the middle of a generation the pending leaves point into them (pend_src): refuse that. */
      std::unique_ptr<CoNet> old = std::move(nets[0]);
      nets[0] = std::move(fresh);
      const bool toggles = (pools[0].cache.hdr != nullptr) != use_cache();
      if (toggles && iterations > 0 && !finished) {
        nets[0] = std::move(old);
        throw EngineError(CA_ERR_STATE, "ca_trainer_set_net: this network turns the evaluation cache " +
                                            std::string(pools[0].cache.hdr ? "off" : "on") +
                                            " in the middle of a generation; finish or reset the generation first");
      }
      if (toggles) free_pools(); /* rebuilt with / without tables */
    } else {
      nets[slot] = std::move(fresh);
    }
    /* entries filled by the previous network must not serve the new one: the next run empties the tables (the value
     * elements the pending leaves point to stay: those rows WERE evaluated by the network in place when they were queued) */
    if (slot == 0) cache_clean = false;
  }

  /* nn.h range_exceeded: an f16x3 network met an operand beyond fp16's range -- its outputs since are NaN or wrong */
  bool net_used = false; /* a network launch has been queued since the range flags were last read */
  void check_net_range() {
    if (!net_used) return; /* (no copy and stream wait per slot when nothing ran) */
    /* the device flag is sticky for the life of the network object: net_used is cleared only when every slot is clean, so
     * that every later check (export, score, write samples) raises again until set_net replaces the slot */
    for (int slot = 0; slot < 2; ++slot)
      if (nets[slot] && nets[slot]->range_exceeded(stream))
        throw EngineError(CA_ERR_ENGINE, std::string("network slot ") + std::to_string(slot) +
                                             ": an activation left the fp16 range of the f16x3 kernels (|x| > 65504); the evaluations "
                                             "are not valid -- use the float32-equivalent x6 kind of the same network");
    net_used = false;
  }

  /* host rows in, host results out (ca_trainer_net_forward): persistent device buffers; the rows travel as
   * the caller holds them (70 floats) and are widened to the kernels' 80-float rows on the device */
  void net_forward_host(int slot, const float *states, int32_t n, float *evals, float *probs) {
    if (slot < 0 || slot > 1 || !nets[slot]) throw EngineError(CA_ERR_STATE, "net slot not set");
    CoNet *net = nets[slot].get();
    if (n < 0 || (size_t)n > net->max_rows()) throw EngineError(CA_ERR_ARG, "net_forward: more rows than num_games*searches_per_eval");
    if (n == 0) return;
    ensure(fw_in70, (size_t)n * CO_GAME_STATE_SIZE);
    ensure(fw_in, (size_t)n * CO_STATE_STRIDE);
    ensure(fw_ev, (size_t)n);
    ensure(fw_pr, (size_t)n * CO_NUM_MOVES);
    ensure(fw_rows, 1);
    rt_h2d(fw_in70.p, states, (size_t)n * CO_GAME_STATE_SIZE * 4, stream);
    rt_h2d(fw_rows.p, &n, 4, stream);
    const int nb = n * CO_STATE_STRIDE / CO_WAVE + 1 < 1024 ? n * CO_STATE_STRIDE / CO_WAVE + 1 : 1024;
    RT_LAUNCH(co_k_expand_rows, nb, CO_WAVE, stream, (const float *)fw_in70.p, fw_in.p, (int)n, nb);
    net_used = true;
    net->forward(fw_in.p, n, fw_rows.p, fw_ev.p, fw_pr.p, stream);
    rt_d2h(evals, fw_ev.p, (size_t)n * 4, stream);
    rt_d2h(probs, fw_pr.p, (size_t)n * CO_NUM_MOVES * 4, stream);
    rt_sync(stream);
    check_net_range();
  }

  void net_forward_rows(int slot, const float *d_in, int32_t rows_cap, const int32_t *d_rows, float *d_eval,
                        float *d_probs) {
    net_used = true;
    nets[slot]->forward(d_in, rows_cap, d_rows, d_eval, d_probs, stream);
  }

  /* the pools of a fused run: streams, events, the host words the counters are copied to, and -- with the evaluation
   * cache -- ONE table for all of them plus every pool's index and counter arrays */
  void pools_create(int npools) {
    free_pools();
    pools.resize(npools);
    for (int p = 0; p < npools; ++p) {
      Pool &q = pools[p];
#ifdef CO_EXP_CU_MASK /* diagnostic build (profiles/r06_coresident.md): every pool's stream on its own share of the compute units */
      rt_stream_create_masked(&q.st, p, npools, CO_EXP_CU_MASK);
#else
      rt_stream_create(&q.st);
#endif
      q.lo = (int)((int64_t)R * p / npools);
      q.n = (int)((int64_t)R * (p + 1) / npools) - q.lo;
      q.row_base = q.lo * spe;
      for (int w = 0; w < 2; ++w) {
        for (auto &e : q.ev[w]) rt_event_create(&e);
        rt_event_create(&q.polled[w]);
      }
      rt_host_alloc((void **)&q.word, 32);
      memset(q.word, 0, 32);
      memset(&q.cache, 0, sizeof q.cache);
      rt_event_create(&q.quiet);
      if (use_cache()) {
        if (p == 0) {
          /* ONE table for all pools: a power of two of at least 8192 entries per slot (a 4096-game generation at 400
           * simulations asks for ~4100 distinct positions per game), within 1/6 of the free device memory */
          size_t want = (size_t)R * 8192, n = 1;
          while (n < want) n <<= 1;
          const size_t per = 16 + CO_CACHE_VAL_FLOATS * 4;
          while (n > 1024 && n * per > rt_mem_free() / 6) n >>= 1;
          if (cfg.eval_cache > 0) { /* given */
            n = (size_t)1 << (cfg.eval_cache < 6 ? 6 : cfg.eval_cache > 30 ? 30 : cfg.eval_cache);
            if (n * per > rt_mem_free() / 2)
              throw EngineError(CA_ERR_ARG, "ca_config.eval_cache: a table of 2^" + std::to_string(cfg.eval_cache) +
                                                " entries does not fit in the free device memory");
          }
          q.c_entries = n;
          rt_malloc((void **)&q.c_hdr, n * 16, q.st);
          /* table values + one scratch element per request row of every pool */
          rt_malloc((void **)&q.c_val, (n + (size_t)R * spe) * CO_CACHE_VAL_FLOATS * 4, q.st);
          rt_malloc((void **)&q.c_done, 4 * CO_MAX_POOLS, q.st);
        } else {
          q.c_entries = pools[0].c_entries;
          q.c_hdr = pools[0].c_hdr;
          q.c_val = pools[0].c_val;
          q.c_done = pools[0].c_done;
        }
        const size_t rows = (size_t)q.n * spe;
        rt_malloc((void **)&q.c_in_idx, rows * 4, q.st);
        rt_malloc((void **)&q.c_out_idx, rows * 4, q.st);
        rt_malloc((void **)&q.c_count, 32, q.st);
        rt_malloc((void **)&q.c_totals, 16, q.st);
        q.cache.hdr = q.c_hdr;
        q.cache.val = q.c_val;
        q.cache.mask = (uint32_t)(q.c_entries - 1);
        q.cache.scratch_base = (uint32_t)q.row_base;
        q.cache.pool_bits = (uint32_t)p << CO_CACHE_POOL_SHIFT;
        q.cache.done = q.c_done;
        q.cache.in_idx = q.c_in_idx;
        q.cache.out_idx = q.c_out_idx;
        q.cache.count = q.c_count;
        q.cache.totals = q.c_totals;
        rt_sync(q.st);
      }
    }
    cache_clean = true; /* freshly zeroed */
  }

  /* a run starts: an emptied table when the generation does (or a new network came), the pools' host-side state.
   * Returns whether the table was emptied here. */
  bool pools_begin_run() {
    const bool cache_clean_now = use_cache() && !cache_clean;
    if (use_cache() && !cache_clean) {
      /* a generation starts with an empty table: nothing evaluated in an earlier generation is carried over */
      for (auto &q : pools) {
        if (&q == &pools[0]) {
          rt_memset(q.c_hdr, 0, q.c_entries * 16, q.st);
          rt_memset(q.c_done, 0, 4 * CO_MAX_POOLS, q.st); /* (the iteration count starts again with the generation) */
        }
        if (iterations == 0) { /* (in mid-generation -- a new network, set_net -- the rows evaluated so far stay counted: ADVICE round 4) */
          rt_memset(q.c_count, 0, 32, q.st);
          rt_memset(q.c_totals, 0, 16, q.st);
        }
        rt_sync(q.st);
      }
      cache_clean = true;
    }
    for (auto &q : pools) {
      q.finished = false;
      q.running = q.n;
      q.idle = 0;
      q.launched[0] = q.launched[1] = 0;
      q.timed[0] = q.timed[1] = 0;
      q.word_iter[0] = q.word_iter[1] = 0;
      q.first_start = P.stagger_div > 0 ? (P.game_base + q.lo) / P.stagger_div : 0;
      if (cache_clean_now) q.c_inserted_est = 0;
    }
    return cache_clean_now;
  }

  /* The host never waits for the window it has just queued: at the end of window w it queues
   * an asynchronous copy of each pool's counter and then reads the copy made at the end of
   * window w-1, so every stream always holds at least one window of work.  A pool is
   * therefore seen to be finished one window late; the launches in between find no running
   * game and no batch row. */
  void pool_collect(Pool &q, int parity, int poll, std::string &failure) {
    if (!q.launched[parity]) return;
    rt_event_sync(q.polled[parity]);
    unsigned long long c = q.word[parity];
    const unsigned long long evaluated = q.cache.hdr ? q.word[2 + parity] & 0xFFFFFFFFull : c & 0xFFFFFFFFull;
    q.running = (int)((c >> 32) & 0xFFFFFFull); /* (bits 56..: games of the iteration that held their leaves back, mcts.h co_step_tail) */
    const bool holding = (c >> 56) != 0;
    if (q.timed[parity]) {
      /* one iteration per window is timed (three event records per launch pair cost 1-4 % of
       * the wall time); its batch size is the counter word just read */
      mcts_timed_ms += rt_event_elapsed_ms(q.ev[parity][0], q.ev[parity][1]);
      pack_timed_ms += rt_event_elapsed_ms(q.ev[parity][1], q.ev[parity][2]); /* cache probe (nothing without a cache) */
      nn_timed_ms += rt_event_elapsed_ms(q.ev[parity][2], q.ev[parity][3]);
      nn_timed_rows += (int64_t)evaluated; /* rows the network kernel worked on */
      q.c_inserted_est += (double)evaluated * poll; /* every evaluated row takes a table entry */
      ++timed_launches;
      q.timed[parity] = 0;
    }
    q.launched[parity] = 0;
    q.finished = ((c >> 32) & 0xFFFFFFull) == 0;
    if (!q.finished && (c & 0xFFFFFFFFull) == 0 && !holding) {
      /* main.pyx:161-163 raises when NO game has a request.  A pool whose first game the
       * staggered start (trainer.cpp:184-186) has not released yet has running games and no
       * rows by construction: that is not the reference's error condition */
      /* (an iteration whose games all stopped at their step budget has no rows either -- `holding`: that is not it) */
      if (q.word_iter[parity] > q.first_start && ++q.idle > 16) failure = "No requests during training";
    } else {
      q.idle = 0;
    }
  }

  /* a run is over (its streams are drained): device times from the timed launches, errors, the rows evaluated, the step
   * budget's statistics */
  void pools_finish_run(const std::string &failure) {
    if (timed_launches > 0) {
      /* device time by kernel family, estimated from the timed launches */
      mcts_ms = mcts_timed_ms * (double)mcts_launches / (double)timed_launches;
      nn_ms = nn_timed_ms * (double)nn_launches / (double)timed_launches;
    }
    if (!failure.empty()) throw EngineError(CA_ERR_ENGINE, failure);
    pack(-1); /* refresh the done flag and the batch description */
    check_errors();
    fetch_games();
    nn_rows = 0;
    for (int g = 0; g < G; ++g) nn_rows += host_games[g].evals; /* every consumed row was requested once */
    nn_rows_evaluated = nn_rows;
    if (use_cache()) {
      nn_rows_evaluated = 0;
      for (auto &q : pools) {
        /* booked by the probe kernels, plus the two iterations whose counters nobody has booked yet (at most one is non-zero) */
        unsigned long long tot[2] = {0, 0};
        uint32_t cnt[8] = {0};
        rt_d2h(tot, q.c_totals, 16, q.st);
        rt_d2h(cnt, q.c_count, 32, q.st);
        rt_sync(q.st);
        nn_rows_evaluated += (int64_t)tot[0] + cnt[0] + cnt[4];
      }
    }
    {
      /* step budget: a pool's word CO_WC_CUTS counts the steps that were cut, CO_WC_BUDGET + {0, 1} hold the last two budgets */
      std::vector<unsigned long long> wc((size_t)CO_WC_WORDS * CO_MAX_POOLS);
      rt_d2h(wc.data(), work_counter.p, wc.size() * 8, stream);
      rt_sync(stream);
      steps_cut = 0;
      for (size_t p = 0; p < pools.size(); ++p) steps_cut += (int64_t)wc[CO_WC_WORDS * p + CO_WC_CUTS];
      step_budget_last = P.step_budget > 0 ? P.step_budget : (int64_t)std::max(wc[CO_WC_BUDGET], wc[CO_WC_BUDGET + 1]) / CO_STEP_UNITS_PER_CONFIG_UNIT;
    }
    if (timed_launches > 0) pack_ms = pack_timed_ms * (double)nn_launches / (double)timed_launches;
  }

  /* The table is half full: start over (returns true).  Between two iterations of EVERY pool, when no entry is pending. */
  bool cache_empty_when_half_full(int npools) {
    if (!pools[0].cache.hdr) return false;
    double taken = 0;
    for (auto &q : pools) taken += q.c_inserted_est;
    if (taken > 0.5 * (double)pools[0].c_entries) {
      /* the table is half full: start over (a long generation asks for far more positions than any table holds;
       * what is asked for again is mostly recent -- the trees of the games in play).  Between two iterations of
       * EVERY pool, when no entry is pending: each stream has the same iterations queued at this point; pool 0's
       * stream waits for the others to get here, empties the table, and the others wait for that. */
      for (int p = 1; p < npools; ++p) {
        rt_event_record(pools[p].quiet, pools[p].st);
        rt_stream_wait(pools[0].st, pools[p].quiet);
      }
      rt_memset(pools[0].c_hdr, 0, pools[0].c_entries * 16, pools[0].st);
      rt_event_record(pools[0].quiet, pools[0].st);
      for (int p = 1; p < npools; ++p) rt_stream_wait(pools[p].st, pools[0].quiet);
      for (auto &q : pools) q.c_inserted_est = 0;
      ++cache_clears;
      return true;
    }
    return false;
  }

  /* one iteration of pool p on its stream: the search launch (with the step's share of the engine parameters: the pool's
   * slice, its counters, the evaluation cache's claim rules) and the network launch behind it; `timed`: with HIP events */
  void pool_queue_iteration(int p, int npools, bool emptied, long long &guard_from, int poll, int parity, int in_window, bool timed) {
    Pool &q = pools[p];
    if (q.finished) return;
    EngineParams pp = P;
    pp.iteration = trainer_iteration;
    /* Deferring the new mover's first searches to the next step (mcts.h co_game_step) balances
     * the waves of a full launch but costs the game one more iteration per ply; once the pool
     * has thinned out, an iteration is as long as its slowest wave anyway and the number of
     * iterations of the longest game is what the generation waits for.  Per-game results do
     * not depend on the choice. */
    pp.defer_handover = q.running * 2 > q.n ? 1 : 0;
    pp.pool_lo = q.lo;
    pp.pool_n = q.n;
    pp.pool_row_base = q.row_base;
    pp.pack_counter = pack_counter.p + CO_PACK_STRIDE * p;
    pp.work_counter = work_counter.p + (size_t)CO_WC_WORDS * p;
    /* (Measured and not done: no automatic budget for a thin pool -- the last eighth of a pool's games, or a trainer of
     * 64 -- on the reasoning that stopping the longest game's steps only adds iterations to its chain: 386.4 against 382.6
     * ms per default generation, 120.7 against 115.3 with the MLP, 95.3 against 93.9 at 64 games.  A thin launch waits for
     * its slowest wavefront like any other.) */
    pp.cache = q.cache; /* (hdr null: no cache) */
    pp.cache.no_claim = emptied ? 1u : 0u;
    if (emptied) guard_from = trainer_iteration;
    pp.cache.guard_pools = 0u;
    if (guard_from >= 0 && trainer_iteration <= guard_from + 2 * poll + 1) {
      /* (the streams are within two windows of each other: the host waits for window w - 1 before it queues w + 1) */
      pp.cache.guard_from = (uint32_t)guard_from;
      /* the OTHER pools only: this pool's own reads of the old contents happened in its launch of iteration
       * guard_from, which stands in front of this launch in its stream -- with its own bit set, every wavefront
       * that runs before the launch's first wave has stored done[p] would lose its claims for nothing */
      for (int p2 = 0; p2 < npools; ++p2)
        if (p2 != p && !pools[p2].finished) pp.cache.guard_pools |= 1u << p2;
    }
    /* The network launch is sized by what the batch can hold: the games still running at the pool's last poll (they
     * only become fewer) times the searches per evaluation.  In a generation's thin tail the throughput kernel is
     * then not launched at all and the small-batch kernel's grid shrinks -- a launch whose workgroups all leave at
     * once still costs their dispatch (for the pixel-major kernel: one 160 KB LDS allocation per 32 rows of capacity). */
    const int cap_rows = (q.running < q.n ? (q.running > 0 ? q.running : 1) : q.n) * spe;
    rt_event_t *e = q.ev[parity];
    if (timed) rt_event_record(e[0], q.st);
    RT_LAUNCH(co_k_mcts_step, ((q.n) + CO_K3_WAVES - 1) / CO_K3_WAVES, CO_WAVE * CO_K3_WAVES, q.st, pp);
    if (timed) rt_event_record(e[1], q.st);
    if (q.cache.hdr) {
      /* the search kernel has resolved every request row to an element of the cache's value array; the network
       * evaluates the rows whose position has no entry yet, reading them through in_idx and writing straight to
       * out_idx */
      if (timed) rt_event_record(e[2], q.st);
      CoNetIO io;
      io.in_idx = q.c_in_idx;
      io.out_idx = q.c_out_idx;
      io.eval_stride = CO_CACHE_VAL_FLOATS;
      io.probs_stride = CO_CACHE_VAL_FLOATS;
      io.alone = npools == 1;
      net_used = true;
      nets[0]->forward(req.p, cap_rows, (const int32_t *)(q.c_count + 4 * (trainer_iteration & 1)), q.c_val, q.c_val + 4, q.st, io);
      if (timed) rt_event_record(e[3], q.st);
    } else {
      if (timed) rt_event_record(e[2], q.st);
      const int32_t *d_rows = (const int32_t *)(pack_counter.p + CO_PACK_STRIDE * p + (trainer_iteration & 1));
      CoNetIO io;
      io.alone = npools == 1;
      io.in_idx = row_idx.p + q.row_base; /* the rows stay where the games wrote them (co_step_tail) */
      net_used = true;
      nets[0]->forward(req.p, cap_rows, d_rows, nn_eval.p + q.row_base, nn_probs.p + (size_t)q.row_base * CO_NUM_MOVES, q.st, io);
      if (timed) rt_event_record(e[3], q.st);
    }
    if (timed) q.timed[parity] = 1;
    q.launched[parity] = in_window + 1;
    ++mcts_launches;
    ++nn_launches;
  }

  /* Fused training as independent pools of games on separate streams:
   * every pool runs the loop of main.pyx:142-168 on its own slice of the game arrays and of
   * the batch buffers; the GPU overlaps one pool's search kernel with another's network
   * kernel and fills launch tails.  Per-game results do not depend on the pooling. */
  bool run_pools(int64_t max_iterations, int npools) {
    const int poll = CO_POOL_POLL;
    if ((int)pools.size() != npools) pools_create(npools);
    const bool cache_clean_now = pools_begin_run();
    rt_sync(stream);
    P.to_play = -1;
    P.row_counter = nullptr;
    P.fused_pack = 1;
    P.defer_handover = 1;
    int64_t it = 0;
    int in_window = 0, window = 0;
    bool all_finished = false;
    std::string failure;
    /* the table was emptied in mid-generation (a new network, set_net): pending leaves point into its old contents */
    bool emptied_before_resuming = cache_clean_now && iterations > 0;
    long long guard_from = -1; /* the no-claim iteration behind the last emptying in mid-generation (EvalCache::guard_from) */
    while (!all_finished && (max_iterations <= 0 || it < max_iterations)) {
      const int parity = window & 1;
      bool emptied = emptied_before_resuming; /* (EvalCache::no_claim) */
      emptied_before_resuming = false;
      if (cache_empty_when_half_full(npools)) emptied = true;
      for (int p = 0; p < npools; ++p)
        pool_queue_iteration(p, npools, emptied, guard_from, poll, parity, in_window,
                             in_window == poll - 1 || (max_iterations > 0 && it + 1 == max_iterations));
      const int counter_slot = trainer_iteration & 1;
      ++trainer_iteration;
      ++iterations;
      ++it;
      ++in_window;
      if (in_window == poll || (max_iterations > 0 && it == max_iterations)) {
        for (auto &q : pools) {
          if (q.finished) continue;
          rt_d2h(&q.word[parity], pack_counter.p + CO_PACK_STRIDE * (&q - &pools[0]) + counter_slot, 8, q.st);
          if (q.cache.hdr) rt_d2h(&q.word[2 + parity], q.c_count + 4 * counter_slot, 4, q.st);
          q.word_iter[parity] = trainer_iteration - 1;
          rt_event_record(q.polled[parity], q.st);
        }
        all_finished = true;
        for (auto &q : pools) {
          if (!q.finished) pool_collect(q, parity ^ 1, poll, failure);
          if (!q.finished) all_finished = false;
        }
        in_window = 0;
        ++window;
        if (!failure.empty()) break;
      }
    }
    /* drain: read what is still in flight (the last window, or both after an iteration cap).  EVERY pool's stream is
     * synchronised here, whichever way the loop ended: that is what allows `guard_from` to be a local of this call -- a
     * later call (an iteration-capped run resumed) starts with no launch of any pool in flight, so no pool can still be
     * reading elements of a table emptied in an earlier call. */
    for (auto &q : pools) {
      rt_sync(q.st);
      for (int w = 0; w < 2; ++w) pool_collect(q, (window + w) & 1, poll, failure);
    }
    P.fused_pack = 0;
    P.defer_handover = 0;
    P.pool_lo = 0;
    P.pool_n = R;
    P.pool_row_base = 0;
    P.pack_counter = pack_counter.p;
    host_games_valid = false;
    scan_valid = false;
    pools_finish_run(failure);
    return finished;
  }

  bool run(int64_t max_iterations) {
    need_positions();
    if (!nets[0]) throw EngineError(CA_ERR_STATE, "ca_trainer_run: no network set (ca_trainer_set_net)");
    if (cfg.testing && !cfg.analyse && !nets[1]) throw EngineError(CA_ERR_STATE, "arena mode needs both networks");
    if (!cfg.testing || cfg.analyse) { /* one network, every slot active: self-play training, or N position searches */
      /* automatic: three pools from 3072 resident games on (round 5, one box, same library, 4096 games x 400: rescnn4 f16x3
       * 417.9 -> 407.6 ms per generation, mlp12x100 f16x3 145.4 -> 138.3 -- with the grouped search a pool's search launch is
       * short enough for a third pool to fit under the other two's network launches), two from 2048, else one */
      int npools = cfg.pools > 0 ? cfg.pools : (R >= 3072 ? 3 : R >= 2048 ? 2 : 1);
      if (npools > CO_MAX_POOLS) npools = CO_MAX_POOLS;
      if (npools > R) npools = R;
      return run_pools(max_iterations, npools);
    }
    /* arena (main.pyx:142-168 with is_testing): one model is served per iteration and hands over
     * when its batch comes back empty.  The model to move lives on the device (arena_state, kept
     * by the two scans of an iteration), both networks are queued every iteration and the idle
     * one finds a row count of zero, so the host only looks every eighth iteration.  Requests
     * are packed by K4 in game order (the reference's request order). */
    const int poll = 8;
    std::vector<rt_event_t> ev(4);
    for (auto &e : ev) rt_event_create(&e);
    P.arena_state = arena_state.p;
    P.row_counter = row_counter.p;
    P.fused_pack = 0;
    P.defer_handover = 0;
    P.to_play = 0;
    int64_t it = 0;
    int32_t st[5] = {0, 0, 0, 0, 0};
    std::string failure;
    while (!finished && (max_iterations <= 0 || it < max_iterations)) {
      const bool timed = (it % poll) == poll - 1 || (max_iterations > 0 && it + 1 == max_iterations);
      P.iteration = trainer_iteration;
      P.scan_phase = 0;
      RT_LAUNCH(co_k_scan, 1, CO_WAVE, stream, P); /* offsets at entry (trainer.cpp:208-215) */
      if (timed) rt_event_record(ev[0], stream);
      RT_LAUNCH(co_k_mcts_step, ((R) + CO_K3_WAVES - 1) / CO_K3_WAVES, CO_WAVE * CO_K3_WAVES, stream, P);
      if (timed) rt_event_record(ev[1], stream);
      P.scan_phase = 1;
      RT_LAUNCH(co_k_scan, 1, CO_WAVE, stream, P);
      RT_LAUNCH(co_k_compact, R, CO_WAVE, stream, P);
      if (timed) rt_event_record(ev[2], stream);
      net_used = true;
      for (int slot = 0; slot < 2; ++slot) /* get_predictions, main.pyx:74-81 */
        nets[slot]->forward(nn_in.p, R * spe, arena_state.p + 3 + slot, nn_eval.p, nn_probs.p, stream);
      if (timed) rt_event_record(ev[3], stream);
      ++iterations;
      ++it;
      ++mcts_launches;
      ++nn_launches;
      if (timed) {
        int32_t d = 0;
        rt_d2h(&d, all_done.p, 4, stream);
        rt_d2h(st, arena_state.p, sizeof st, stream);
        rt_sync(stream);
        finished = d != 0;
        mcts_timed_ms += rt_event_elapsed_ms(ev[0], ev[1]);
        nn_timed_ms += rt_event_elapsed_ms(ev[2], ev[3]);
        pack_ms += rt_event_elapsed_ms(ev[1], ev[2]) * poll;
        nn_timed_rows += st[3] + st[4];
        ++timed_launches;
        if (!finished && st[1] > 4) {
          failure = "arena: no model has requests";
          break;
        }
      }
    }
    rt_sync(stream);
    for (auto &e : ev) rt_event_destroy(e);
    rt_d2h(st, arena_state.p, sizeof st, stream);
    rt_sync(stream);
    P.arena_state = nullptr;
    P.scan_phase = 0;
    P.row_counter = nullptr;
    host_games_valid = false;
    scan_valid = false;
    if (timed_launches > 0) {
      mcts_ms = mcts_timed_ms * (double)mcts_launches / (double)timed_launches;
      nn_ms = nn_timed_ms * (double)nn_launches / (double)timed_launches;
    }
    if (!failure.empty()) throw EngineError(CA_ERR_ENGINE, failure);
    pack(st[0]); /* refresh the done flag and the batch description */
    check_errors();
    unsigned long long rows = 0;
    rt_d2h(&rows, row_counter.p, 8, stream);
    rt_sync(stream);
    nn_rows = (int64_t)rows;
    nn_rows_evaluated = nn_rows;
    return finished;
  }
};

/* ------------------------------------------------------------------- C ABI */
#define CA_GUARD(...)                     \
  try {                                   \
    __VA_ARGS__;                          \
    return CA_OK;                         \
  } catch (const EngineError &e) {        \
    g_last_error = e.what();              \
    return e.code;                        \
  } catch (const std::exception &e) {     \
    g_last_error = e.what();              \
    return CA_ERR_DEVICE;                 \
  }

/* entry points on a trainer select its device first: the caller's thread may have another one current
 * (two trainers on two GPUs in one process; torch.cuda.set_device between calls) */
#define CA_TGUARD(...) CA_GUARD(rt_set_device(t->cfg.device); __VA_ARGS__)

extern "C" int ca_trainer_create(const ca_config *cfg, ca_trainer **out) {
  if (!cfg || !out) {
    g_last_error = "null argument";
    return CA_ERR_ARG;
  }
  *out = nullptr;
  /* the asserts of trainer.cpp:25-34, as errors */
  if (cfg->num_games <= 0 || cfg->max_searches < 0 || cfg->searches_per_eval < 0 || cfg->epsilon < 0.0f ||
      cfg->epsilon > 1.0f || cfg->num_logged != 0 ||
      (cfg->max_searches > 0 && cfg->searches_per_eval > cfg->max_searches) ||
      (cfg->total_games > 0 && cfg->game_base + cfg->num_games > cfg->total_games) || cfg->game_base < 0) {
    g_last_error = "ca_trainer_create: invalid configuration (num_logged must be 0 on device)";
    return CA_ERR_ARG;
  }
  int rc = ca_device_check(cfg->device);
  if (rc != CA_OK) return rc;
  ca_trainer *t = nullptr;
  try {
    t = new ca_trainer();
    t->init(*cfg);
  } catch (const std::exception &e) {
    g_last_error = e.what();
    delete t;
    return CA_ERR_DEVICE;
  }
  *out = t;
  return CA_OK;
}

extern "C" void ca_trainer_destroy(ca_trainer *t) { delete t; }

extern "C" int ca_trainer_num_requests(ca_trainer *t, int to_play, int32_t *out) { CA_TGUARD(*out = t->num_requests(to_play)) }
extern "C" int ca_trainer_num_samples(ca_trainer *t, int32_t *out) { CA_TGUARD(*out = t->num_samples()) }
extern "C" int ca_trainer_score(ca_trainer *t, float *out) { CA_TGUARD(*out = t->score()) }
extern "C" int ca_trainer_avg_mate_length(ca_trainer *t, float *out) { CA_TGUARD(*out = t->avg_mate_length()) }
extern "C" int ca_trainer_write_requests(ca_trainer *t, float *gs, int to_play) { CA_TGUARD(t->write_requests(gs, to_play)) }
extern "C" int ca_trainer_write_samples(ca_trainer *t, float *gs, float *ev, float *pr) { CA_TGUARD(t->write_samples(gs, ev, pr)) }
extern "C" int ca_trainer_write_scores(ca_trainer *t, const char *file) { CA_TGUARD(t->write_scores(file)) }
extern "C" int ca_trainer_do_iteration(ca_trainer *t, const float *ev, const float *pr, int to_play, int32_t *all_done) {
  CA_TGUARD(*all_done = t->do_iteration(ev, pr, to_play) ? 1 : 0)
}
extern "C" int ca_trainer_set_net(ca_trainer *t, int slot, int kind, const float *w, size_t n) { CA_TGUARD(t->set_net(slot, kind, w, n)) }
extern "C" int ca_trainer_run(ca_trainer *t, int64_t max_iterations, int32_t *all_done) {
  CA_TGUARD(*all_done = t->run(max_iterations) ? 1 : 0)
}
extern "C" int ca_trainer_export_samples(ca_trainer *t, float *sp, float *oc) { CA_TGUARD(t->export_samples(sp, oc)) }
extern "C" int ca_trainer_pack_samples_device(ca_trainer *t, void *d_sp, void *d_oc, int32_t cap_rows, int32_t *n_rows) {
  CA_TGUARD(*n_rows = t->pack_samples_device((float *)d_sp, (float *)d_oc, cap_rows))
}
extern "C" int ca_trainer_pin_host(ca_trainer *t, void *p, size_t bytes, int32_t *pinned) {
  CA_TGUARD(*pinned = t->pin_host(p, bytes) ? 1 : 0)
}
extern "C" int ca_trainer_unpin_host(ca_trainer *t, void *p) { CA_TGUARD(t->unpin_host(p)) }
extern "C" int ca_trainer_set_positions(ca_trainer *t, const int32_t *boards, const int32_t *to_play, const int32_t *pieces,
                                        const int32_t *seeds) {
  CA_TGUARD(t->set_positions(boards, to_play, pieces, seeds))
}
extern "C" int ca_trainer_analysis(ca_trainer *t, int32_t *out) { CA_TGUARD(t->analysis(out)) }
extern "C" int ca_trainer_finish(ca_trainer *t) { CA_TGUARD(t->finish_analysis()) }
extern "C" int ca_trainer_set_logging(ca_trainer *t, const char *log_folder, int32_t num_logged) {
  CA_TGUARD(t->set_logging(log_folder, num_logged))
}
extern "C" int ca_trainer_reset(ca_trainer *t, int32_t seed) { CA_TGUARD(t->reset_games(seed)) }
/* ------------------------------------------------------------------ Tourney C ABI */
struct ca_tourney {
  int device = 0;
  uint32_t arena_units = 0;
  int trace = 0;
  std::map<int, PlayerCfg> players;            /* Tourney::players_ (tourney.h:42) */
  std::vector<std::pair<int, int>> matches;    /* addMatch order */
  std::vector<char> match_logging;             /* addMatch's `logging` */
  std::string log_folder;                      /* Tourney::log_folder_ (tourney.h:46) */
  bool seen_done = false;
  std::mt19937 generator;                      /* default constructed: seed 5489 (tourney.h:43) */
  std::vector<uint32_t> seeds;
  std::unique_ptr<ca_trainer> pool;            /* built at the first query after the last addMatch */
  struct PendingNet {
    int kind;
    std::vector<float> w;
  };
  std::map<int, PendingNet> net_specs;         /* fused mode: model id -> network (ca_tourney_set_net) */
  std::map<int, std::unique_ptr<CoNet>> nets;
  bool exact_offsets = false;                  /* ca_tourney_set_exact_offsets */

  /* The loop of rating/tourney.pyx:122-160 with the networks on the GPU: for every model id in
   * ascending order, pack that model's requests (Tourney::writeRequests), evaluate them, iterate
   * its matches (Tourney::doIteration, which reads the evaluations through the reference's offset
   * table).  The evaluation arrays persist between rounds like the driver's, so the result is the
   * one the compat protocol gives with the same networks. */
  bool run(int64_t max_rounds) {
    ca_trainer &p = built();
    std::vector<int> ids;
    for (auto &m : matches)
      for (int pid : {m.first, m.second}) {
        int id = players.at(pid).model_id;
        if (std::find(ids.begin(), ids.end(), id) == ids.end()) ids.push_back(id);
      }
    std::sort(ids.begin(), ids.end());
    for (int id : ids) {
      if (id < 0) continue;
      if (!nets.count(id)) {
        auto it = net_specs.find(id);
        if (it == net_specs.end()) throw EngineError(CA_ERR_STATE, "ca_tourney_run: no network for model id " + std::to_string(id));
        std::unique_ptr<CoNet> n(co_net_create(it->second.kind, it->second.w.data(), it->second.w.size(),
                                               (size_t)p.G * p.spe, p.stream));
        if (!n) throw EngineError(CA_ERR_ARG, "unknown net kind or bad weight count");
        nets[id] = std::move(n);
      }
    }
    int64_t rounds = 0;
    bool done = p.tourney_all_done();
    while (!done && (max_rounds <= 0 || rounds < max_rounds)) {
      for (int id : ids) {
        p.P.to_play = id;
        p.P.iteration = p.trainer_iteration;
        RT_LAUNCH(co_k_scan, 1, CO_WAVE, p.stream, p.P); /* offsets + batch of model `id` */
        if (id >= 0) {
          RT_LAUNCH(co_k_compact, p.G, CO_WAVE, p.stream, p.P);
          nets[id]->forward(p.nn_in.p, p.G * p.spe, p.req_offset.p + p.G, p.nn_eval.p, p.nn_probs.p, p.stream);
          ++p.nn_launches;
        }
        RT_LAUNCH(co_k_mcts_step, ((p.G) + CO_K3_WAVES - 1) / CO_K3_WAVES, CO_WAVE * CO_K3_WAVES, p.stream, p.P);
        ++p.mcts_launches;
        ++p.iterations;
      }
      ++rounds;
      /* the host looks at the all-done flag every eighth round only (a round that finds every match
       * finished launches kernels that return at once), so the queue never runs dry in between */
      if ((rounds & 7) == 0 || (max_rounds > 0 && rounds >= max_rounds)) {
        p.P.to_play = ids.front();
        RT_LAUNCH(co_k_scan, 1, CO_WAVE, p.stream, p.P); /* refresh the all-done flag */
        int32_t d = 0;
        rt_d2h(&d, p.all_done.p, 4, p.stream);
        rt_sync(p.stream);
        done = d != 0;
      }
    }
    p.scan_valid = false;
    p.host_games_valid = false;
    p.check_errors();
    if (done) all_done(); /* (writes the match logs) */
    return done;
  }

  /* Tourney::all_done (tourney.cpp:14-21); the log files of the matches are written the first time it is true */
  bool all_done() {
    ca_trainer &p = built();
    const bool done = p.tourney_all_done();
    if (done && !seen_done) {
      seen_done = true;
      p.maybe_write_logs();
    }
    return done;
  }

  ca_trainer &built() {
    if (pool) return *pool;
    if (matches.empty()) throw EngineError(CA_ERR_STATE, "tourney without matches");
    auto t = std::make_unique<ca_trainer>();
    t->tourney = true;
    ca_config c;
    memset(&c, 0, sizeof c);
    c.num_games = (int32_t)matches.size();
    c.device = device;
    c.testing = 1;
    c.no_stagger = 1;
    c.trace = trace;
    c.arena_units = arena_units;
    c.c_puct = 1.0f;
    c.max_searches = 1;
    c.searches_per_eval = 1;
    for (auto &m : matches) {
      for (int side = 0; side < 2; ++side) {
        const PlayerCfg &p = players.at(side == 0 ? m.first : m.second);
        t->host_pcfg.push_back(p);
        if (!p.random) {
          c.max_searches = std::max(c.max_searches, p.max_searches);
          c.searches_per_eval = std::max(c.searches_per_eval, p.searches_per_eval);
        }
      }
    }
    t->match_seeds = seeds;
    t->init(c);
    /* tourney.cpp:83-96: a match added with logging = true writes <log_folder>/match_<p1>_<p2>_<index>.txt */
    std::vector<int> logged;
    std::vector<std::string> paths;
    for (size_t i = 0; i < matches.size(); ++i)
      if (match_logging[i]) {
        logged.push_back((int)i);
        paths.push_back(log_folder + "/match_" + std::to_string(matches[i].first) + "_" + std::to_string(matches[i].second) + "_" +
                        std::to_string(i) + ".txt");
      }
    if (!logged.empty()) t->set_log_records(logged, paths, true);
    if (exact_offsets) t->P.read_offset = nullptr; /* co_step_row falls back to the writeRequests rows */
    pool = std::move(t);
    return *pool;
  }
};

extern "C" int ca_tourney_create(int device, uint32_t arena_units, int trace, ca_tourney **out) {
  int rc = ca_device_check(device);
  if (rc != CA_OK) return rc;
  CA_GUARD({
    if (!out) throw EngineError(CA_ERR_ARG, "null output pointer");
    auto t = std::make_unique<ca_tourney>();
    t->device = device;
    t->arena_units = arena_units;
    t->trace = trace;
    *out = t.release();
  })
}
extern "C" void ca_tourney_destroy(ca_tourney *t) { delete t; }

extern "C" int ca_tourney_add_player(ca_tourney *t, int32_t player_id, int32_t model_id, int32_t max_searches,
                                     int32_t searches_per_eval, float c_puct, float epsilon, int32_t random) {
  CA_GUARD({
    if (t->pool) throw EngineError(CA_ERR_STATE, "addPlayer after the tournament has started");
    if (!random && (max_searches <= 0 || searches_per_eval <= 0)) throw EngineError(CA_ERR_ARG, "addPlayer: bad search settings");
    PlayerCfg p;
    memset(&p, 0, sizeof p);
    p.player_id = player_id;
    p.model_id = model_id;
    p.max_searches = max_searches;
    p.searches_per_eval = searches_per_eval;
    p.c_puct = c_puct;
    p.epsilon = epsilon;
    p.random = random ? 1 : 0;
    t->players[player_id] = p;
  })
}

extern "C" int ca_tourney_add_match(ca_tourney *t, int32_t player1, int32_t player2, int32_t logging) {
  CA_GUARD({
    if (t->pool) throw EngineError(CA_ERR_STATE, "addMatch after the tournament has started");
    if (!t->players.count(player1) || !t->players.count(player2)) throw EngineError(CA_ERR_ARG, "addMatch: unknown player");
    if (t->players[player1].random && t->players[player2].random)
      throw EngineError(CA_ERR_ARG, "addMatch: at most one random player per match (match.cpp:72)");
    t->matches.emplace_back(player1, player2);
    t->match_logging.push_back(logging ? 1 : 0);
    t->seeds.push_back((uint32_t)t->generator()); /* tourney.cpp:86 */
  })
}
extern "C" int ca_tourney_set_log_folder(ca_tourney *t, const char *log_folder) {
  CA_GUARD({
    if (t->pool) throw EngineError(CA_ERR_STATE, "set_log_folder after the tournament has started");
    t->log_folder = log_folder ? log_folder : "";
  })
}

extern "C" int ca_tourney_set_net(ca_tourney *t, int32_t model_id, int32_t kind, const float *weights, size_t n_floats) {
  CA_GUARD({
    if (model_id < 0) throw EngineError(CA_ERR_ARG, "negative model ids are the dummy ids of random players");
    if (!weights || n_floats == 0) throw EngineError(CA_ERR_ARG, "ca_tourney_set_net: no weights");
    ca_tourney::PendingNet spec;
    spec.kind = kind;
    spec.w.assign(weights, weights + n_floats);
    t->net_specs[model_id] = std::move(spec);
    t->nets.erase(model_id);
  })
}
extern "C" int ca_tourney_set_exact_offsets(ca_tourney *t, int32_t on) {
  CA_GUARD({
    if (t->pool) throw EngineError(CA_ERR_STATE, "set_exact_offsets after the tournament has started");
    t->exact_offsets = on != 0;
  })
}
extern "C" int ca_tourney_run(ca_tourney *t, int64_t max_rounds, int32_t *all_done) {
  CA_GUARD(rt_set_device(t->device); *all_done = t->run(max_rounds) ? 1 : 0)
}
extern "C" int ca_tourney_all_done(ca_tourney *t, int32_t *out) { CA_GUARD(*out = t->all_done() ? 1 : 0) }
extern "C" int ca_tourney_num_requests(ca_tourney *t, int32_t id, int32_t *out) {
  CA_GUARD(rt_set_device(t->device); *out = t->built().tourney_num_requests(id))
}
extern "C" int ca_tourney_write_requests(ca_tourney *t, float *game_states, int32_t id) {
  CA_GUARD(rt_set_device(t->device); t->built().tourney_write_requests(game_states, id))
}
extern "C" int ca_tourney_do_iteration(ca_tourney *t, const float *evaluations, const float *probabilities,
                                       int32_t rows, int32_t id) {
  CA_GUARD(rt_set_device(t->device); t->built().tourney_do_iteration(evaluations, probabilities, rows, id))
}
extern "C" int ca_tourney_num_matches(ca_tourney *t, int32_t *out) { CA_GUARD(*out = (int32_t)t->matches.size()) }
/* out[8] = {player id 1, player id 2, done, result (util.h:57-64, first player's view), side to move, pending
 * requests, plies, error} */
extern "C" int ca_tourney_match_info(ca_tourney *t, int32_t match, int32_t out[8]) {
  CA_GUARD({
    ca_trainer &p = t->built();
    if (match < 0 || match >= p.G) throw EngineError(CA_ERR_ARG, "match index out of range");
    p.fetch_games();
    const GameCtl &gc = p.host_games[match];
    out[0] = t->matches[match].first;
    out[1] = t->matches[match].second;
    out[2] = gc.done;
    out[3] = gc.result;
    out[4] = gc.to_play;
    out[5] = gc.done ? 0 : gc.n_pending;
    out[6] = gc.plies;
    out[7] = gc.error;
  })
}
extern "C" int ca_tourney_match_score(ca_tourney *t, int32_t match, float *out) {
  CA_GUARD({
    ca_trainer &p = t->built();
    if (match < 0 || match >= p.G) throw EngineError(CA_ERR_ARG, "match index out of range");
    p.fetch_games();
    *out = ca_trainer::game_score(p.host_games[match]); /* Match::score, match.cpp:52-58 */
  })
}
/* Tourney::writeScores, tourney.cpp:33-41: "id1 id2 score" per finished match */
extern "C" int ca_tourney_write_scores(ca_tourney *t, const char *filename) {
  CA_GUARD({
    ca_trainer &p = t->built();
    p.fetch_games();
    std::ofstream f(filename);
    if (!f) throw EngineError(CA_ERR_ARG, std::string("cannot open ") + filename);
    for (int g = 0; g < p.G; ++g)
      if (p.host_games[g].done)
        f << t->matches[g].first << ' ' << t->matches[g].second << ' ' << ca_trainer::game_score(p.host_games[g]) << '\n';
  })
}
extern "C" int ca_trainer_trace(ca_trainer *t, int game, int32_t *out, int32_t cap, int32_t *n);
extern "C" int ca_trainer_stats(ca_trainer *t, ca_stats *out);
extern "C" int ca_tourney_trace(ca_tourney *t, int32_t match, int32_t *out, int32_t cap, int32_t *n_out) {
  ca_trainer *p = nullptr;
  try {
    p = &t->built();
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return CA_ERR_STATE;
  }
  return ca_trainer_trace(p, match, out, cap, n_out);
}
extern "C" int ca_tourney_stats(ca_tourney *t, ca_stats *out) {
  ca_trainer *p = nullptr;
  try {
    p = &t->built();
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return CA_ERR_STATE;
  }
  return ca_trainer_stats(p, out);
}

/* diagnostic builds (-DCO_PROF): summed in-kernel cycle stamps, see mcts.h; not in the public header */
extern "C" int ca_trainer_prof(ca_trainer *t, unsigned long long out[2 * CO_NPROF + 24]) {
  CA_TGUARD({
    /* nothing is written to `out` unless this is a stamped build: a caller of the shipped library with a buffer sized for
     * an earlier round's layout gets the error, not an overflow */
    if (!t->prof.p) throw EngineError(CA_ERR_STATE, "not a -DCO_PROF build");
    for (int i = 0; i < 2 * CO_NPROF + 24; ++i) out[i] = 0;
    std::vector<unsigned long long> h((size_t)t->R * CO_NPROF);
    rt_d2h(h.data(), t->prof.p, h.size() * 8, t->stream);
    rt_sync(t->stream);
    for (int g = 0; g < t->R; ++g)
      for (int i = 0; i < CO_NPROF; ++i) out[i] += h[(size_t)g * CO_NPROF + i];
    /* [CO_NPROF ..): clocks, the histogram of wave-step times, then the phase sums of the slow wave-steps alone */
    rt_d2h(out + CO_NPROF, t->prof.p + (size_t)t->R * CO_NPROF, (size_t)(24 + CO_NPROF) * 8, t->stream);
    rt_sync(t->stream);
  })
}

extern "C" int ca_trainer_net_forward(ca_trainer *t, int slot, const float *states, int32_t n, float *evals, float *probs) {
  CA_TGUARD(t->net_forward_host(slot, states, n, evals, probs))
}

/* kernel-only timing of a network on `rows` resident rows (HIP events on the engine's stream);
 * diagnostics and the per-kernel roofline of bench.py */
extern "C" int ca_trainer_net_bench(ca_trainer *t, int slot, const float *states, int32_t rows, int32_t reps, float *ms_per_call) {
  CA_TGUARD({
    if (slot < 0 || slot > 1 || !t->nets[slot]) throw EngineError(CA_ERR_STATE, "net slot not set");
    if ((size_t)rows > t->nets[slot]->max_rows()) throw EngineError(CA_ERR_ARG, "net_bench: too many rows");
    std::vector<float> pad((size_t)rows * CO_STATE_STRIDE, 0.0f);
    for (int r = 0; r < rows; ++r)
      memcpy(&pad[(size_t)r * CO_STATE_STRIDE], states + (size_t)r * CO_GAME_STATE_SIZE, CO_GAME_STATE_SIZE * 4);
    DevBuf<int32_t> d_n;
    d_n.alloc(1, t->stream);
    rt_h2d(t->nn_in.p, pad.data(), pad.size() * 4, t->stream);
    rt_h2d(d_n.p, &rows, 4, t->stream);
    t->net_used = true;
    t->nets[slot]->forward(t->nn_in.p, rows, d_n.p, t->nn_eval.p, t->nn_probs.p, t->stream); /* warm */
    rt_event_t e0, e1;
    rt_event_create(&e0);
    rt_event_create(&e1);
    rt_event_record(e0, t->stream);
    for (int i = 0; i < reps; ++i) t->nets[slot]->forward(t->nn_in.p, rows, d_n.p, t->nn_eval.p, t->nn_probs.p, t->stream);
    rt_event_record(e1, t->stream);
    rt_sync(t->stream);
    *ms_per_call = rt_event_elapsed_ms(e0, e1) / (float)reps;
    rt_event_destroy(e0);
    rt_event_destroy(e1);
  })
}

extern "C" int ca_expand_samples(int device, const float *state_policy, const float *outcome, int32_t n, float *gs, float *ev,
                                 float *pr) {
  /* host-side K7 for gathered shards: same gathers as co_k_write_samples */
  (void)device;
  static const int32_t SS[8][16] = CO_SPACE_SYM_INIT;
  static const int32_t MS[8][96] = CO_MOVE_SYM_INIT;
  for (int32_t i = 0; i < n; ++i) {
    const float *st = state_policy + (size_t)i * CO_SAMPLE_FLOATS;
    const float *pol = st + CO_GAME_STATE_SIZE;
    for (int k = 0; k < 8; ++k) {
      float *g = gs + ((size_t)i * 8 + k) * CO_GAME_STATE_SIZE;
      float *p = pr + ((size_t)i * 8 + k) * CO_NUM_MOVES;
      for (int j = 0; j < 64; ++j) g[j] = st[SS[k][j / 4] * 4 + j % 4];
      for (int j = 64; j < CO_GAME_STATE_SIZE; ++j) g[j] = st[j];
      for (int j = 0; j < CO_NUM_MOVES; ++j) p[j] = pol[MS[k][j]];
      ev[(size_t)i * 8 + k] = outcome[i];
    }
  }
  return CA_OK;
}

extern "C" int ca_trainer_stats(ca_trainer *t, ca_stats *out) {
  CA_TGUARD({
    t->fetch_games();
    memset(out, 0, sizeof *out);
    for (int g = 0; g < t->G; ++g) {
      out->searches += t->host_games[g].searches;
      out->evals += t->host_games[g].evals;
      out->nodes += t->host_games[g].nodes;
      out->plies += t->host_games[g].plies;
    }
    std::vector<TreeCtl> ht((size_t)2 * t->R);
    rt_d2h(ht.data(), t->trees.p, ht.size() * sizeof(TreeCtl), t->stream);
    rt_sync(t->stream);
    for (auto &x : ht) out->peak_arena_units = std::max<int64_t>(out->peak_arena_units, x.peak_units);
    out->iterations = t->iterations;
    out->mcts_ms = t->mcts_ms;
    out->nn_ms = t->nn_ms;
    out->pack_ms = t->pack_ms;
    out->mcts_launches = t->mcts_launches;
    out->nn_launches = t->nn_launches;
    out->nn_rows = t->nn_rows;
    out->pools = t->pools.empty() ? 1 : (int64_t)t->pools.size();
    out->resident_slots = t->R;
    out->nn_rows_evaluated = t->nn_rows_evaluated;
    out->steps_cut = t->steps_cut;
    out->step_budget_last = t->step_budget_last;
    out->timed_launches = t->timed_launches;
    out->nn_timed_rows = t->nn_timed_rows;
    out->mcts_timed_ms = t->mcts_timed_ms;
    out->nn_timed_ms = t->nn_timed_ms;
  })
}

extern "C" int ca_trainer_game_info(ca_trainer *t, int game, int32_t out[8]) {
  CA_TGUARD({
    if (game < 0 || game >= t->G) throw EngineError(CA_ERR_ARG, "game index out of range");
    t->fetch_games();
    const GameCtl &gc = t->host_games[game];
    out[0] = gc.to_play; out[1] = gc.done; out[2] = gc.result; out[3] = gc.n_samples;
    out[4] = gc.done ? 0 : gc.n_pending; out[5] = gc.error; out[6] = gc.mate_turn; out[7] = gc.plies;
  })
}

extern "C" int ca_trainer_trace(ca_trainer *t, int game, int32_t *out, int32_t cap, int32_t *n) {
  CA_TGUARD({
    if (!t->cfg.trace) throw EngineError(CA_ERR_STATE, "trace not enabled");
    if (game < 0 || game >= t->G) throw EngineError(CA_ERR_ARG, "game index out of range");
    t->fetch_games();
    int32_t len = t->host_games[game].trace_len;
    if (len > CO_TRACE_CAP) throw EngineError(CA_ERR_ENGINE, "trace buffer overflow");
    *n = len;
    int32_t c = std::min(len, cap);
    if (out && c > 0) {
      rt_d2h(out, t->trace.p + (size_t)game * CO_TRACE_CAP, (size_t)c * 4, t->stream);
      rt_sync(t->stream);
    }
  })
}

/* ---- stand-alone test entry points */
struct TmpStream {
  rt_stream_t s;
  explicit TmpStream(int device) {
    rt_set_device(device);
    rt_stream_create(&s);
  }
  ~TmpStream() { rt_stream_destroy(s); }
};

extern "C" int ca_rules_legal_moves(int device, const uint64_t *boards, const uint32_t *metas, int32_t n, uint32_t *masks,
                                    int32_t *is_lines) {
  int rc = ca_device_check(device);
  if (rc != CA_OK) return rc;
  CA_GUARD({
    TmpStream ts(device);
    DevBuf<uint64_t> b;
    DevBuf<uint32_t> m, mk;
    DevBuf<int32_t> ln;
    b.alloc(n, ts.s); m.alloc(n, ts.s); mk.alloc((size_t)n * 3, ts.s); ln.alloc(n, ts.s);
    rt_h2d(b.p, boards, (size_t)n * 8, ts.s);
    rt_h2d(m.p, metas, (size_t)n * 4, ts.s);
    RT_LAUNCH(co_k_rules_batch, n, CO_WAVE, ts.s, (const uint64_t *)b.p, (const uint32_t *)m.p, n, mk.p, ln.p);
    rt_d2h(masks, mk.p, (size_t)n * 12, ts.s);
    rt_d2h(is_lines, ln.p, (size_t)n * 4, ts.s);
    rt_sync(ts.s);
  })
}

extern "C" int ca_rules_do_move(int device, uint64_t *boards, uint32_t *metas, const int32_t *moves, int32_t n, float *states) {
  int rc = ca_device_check(device);
  if (rc != CA_OK) return rc;
  CA_GUARD({
    TmpStream ts(device);
    DevBuf<uint64_t> b;
    DevBuf<uint32_t> m;
    DevBuf<int32_t> mv;
    DevBuf<float> st;
    b.alloc(n, ts.s); m.alloc(n, ts.s); mv.alloc(n, ts.s); st.alloc((size_t)n * CO_STATE_STRIDE, ts.s);
    rt_h2d(b.p, boards, (size_t)n * 8, ts.s);
    rt_h2d(m.p, metas, (size_t)n * 4, ts.s);
    rt_h2d(mv.p, moves, (size_t)n * 4, ts.s);
    RT_LAUNCH(co_k_domove_batch, n, CO_WAVE, ts.s, b.p, m.p, (const int32_t *)mv.p, n, st.p);
    std::vector<float> tmp((size_t)n * CO_STATE_STRIDE);
    rt_d2h(boards, b.p, (size_t)n * 8, ts.s);
    rt_d2h(metas, m.p, (size_t)n * 4, ts.s);
    rt_d2h(tmp.data(), st.p, tmp.size() * 4, ts.s);
    rt_sync(ts.s);
    for (int i = 0; i < n; ++i)
      memcpy(states + (size_t)i * CO_GAME_STATE_SIZE, &tmp[(size_t)i * CO_STATE_STRIDE], CO_GAME_STATE_SIZE * 4);
  })
}

extern "C" int ca_rules_rows(int device, uint64_t *boards, uint32_t *metas, const int32_t *moves, int32_t n, uint32_t *masks) {
  int rc = ca_device_check(device);
  if (rc != CA_OK) return rc;
  CA_GUARD({
    TmpStream ts(device);
    DevBuf<uint64_t> b;
    DevBuf<uint32_t> m, mk;
    DevBuf<int32_t> mv;
    b.alloc(n, ts.s); m.alloc(n, ts.s); mk.alloc((size_t)n * 3, ts.s); mv.alloc(n, ts.s);
    rt_h2d(b.p, boards, (size_t)n * 8, ts.s);
    rt_h2d(m.p, metas, (size_t)n * 4, ts.s);
    rt_h2d(mv.p, moves, (size_t)n * 4, ts.s);
    RT_LAUNCH(co_k_rules_rows, (n + 3) / 4, CO_WAVE, ts.s, b.p, m.p, (const int32_t *)mv.p, n, mk.p);
    rt_d2h(boards, b.p, (size_t)n * 8, ts.s);
    rt_d2h(metas, m.p, (size_t)n * 4, ts.s);
    rt_d2h(masks, mk.p, (size_t)n * 12, ts.s);
    rt_sync(ts.s);
  })
}

extern "C" int ca_rng_draw(int device, uint32_t seed, int32_t n, int32_t chunk, uint32_t *out) {
  int rc = ca_device_check(device);
  if (rc != CA_OK) return rc;
  CA_GUARD({
    if (chunk < 1 || chunk > CO_WAVE) throw EngineError(CA_ERR_ARG, "chunk must be 1..64");
    TmpStream ts(device);
    std::vector<uint32_t> x(CO_MT_N);
    x[0] = seed;
    for (int i = 1; i < CO_MT_N; ++i) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
    DevBuf<uint32_t> mt, o;
    DevBuf<int32_t> idx;
    mt.alloc(CO_MT_N, ts.s); o.alloc(n, ts.s); idx.alloc(1, ts.s);
    int32_t i0 = CO_MT_N;
    rt_h2d(mt.p, x.data(), CO_MT_N * 4, ts.s);
    rt_h2d(idx.p, &i0, 4, ts.s);
    RT_LAUNCH(co_k_rng_draw, 1, CO_WAVE, ts.s, mt.p, idx.p, n, chunk, o.p);
    rt_d2h(out, o.p, (size_t)n * 4, ts.s);
    rt_sync(ts.s);
  })
}

extern "C" int ca_fp_probe(int device, const float *in, int32_t n, float *out) {
  int rc = ca_device_check(device);
  if (rc != CA_OK) return rc;
  CA_GUARD({
    TmpStream ts(device);
    DevBuf<float> di, dout;
    di.alloc((size_t)n * 8, ts.s); dout.alloc((size_t)n * 8, ts.s);
    rt_h2d(di.p, in, (size_t)n * 32, ts.s);
    RT_LAUNCH(co_k_fp_probe, (n + CO_WAVE - 1) / CO_WAVE, CO_WAVE, ts.s, (const float *)di.p, n, dout.p);
    rt_d2h(out, dout.p, (size_t)n * 32, ts.s);
    rt_sync(ts.s);
  })
}
