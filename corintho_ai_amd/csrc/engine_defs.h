// engine_defs.h -- HBM layout of the self-play pool (shared by host and kernels).
//
// One process drives one GPU.  G games live in a struct-of-arrays pool:
//
//   games[G]        GameCtl   64 B   turn state of one game (SelfPlayer,
//                                    selfplayer.h:86-111)
//   trees[2G]       TreeCtl   16 B   one per player (TrainMC, trainmc.h:145-188)
//   arena[2G][cap]  uint4            the two search trees of game g, as 16-byte
//                                    units, bump allocated, never freed within a
//                                    game (a re-root only moves the root offset)
//   pend_*          per game         the <= searches_per_eval leaves awaiting a
//                                    network evaluation, each with the path of
//                                    stat slots from the root
//   rng[G][624]     uint32           std::mt19937 state of each game
//   req[G][spe][80] float            leaf states (70 floats, padded to 80)
//   samples[G][44][166] float        (state, policy) per ply
//
// A tree node is a BLOCK of units at offset b of its tree's arena:
//   unit b+0   {board_lo, board_hi, meta, parent_block}
//   unit b+1   {self_slot, denominator(f32), 0, 0}
//   unit b+2+e slot of edge e, e < n_edges, edges in ascending move id:
//                {child_block | NONE, evaluation(f32),
//                 move_id:7 | prior:9 | visits:16 (signed), result:8 | all_visited:8}
//   unit b+2+n (only for a detached root) the root's own stat slot
// The statistics of a node (Node::evaluation_/visits_/result_/all_visited_,
// node.h:150-186) live in the SLOT of the edge that leads to it -- in its
// parent's block -- so that one coalesced 16 B-per-lane load of a block gives
// the PUCT scan everything it needs (trainmc.cpp:540-600); `self_slot` is the
// unit offset of that slot.  meta = pieces[6] (3 bits each) | to_play<<18 |
// depth<<19 (6 bits) | n_edges<<25 (7 bits).
#pragma once
#include <stdint.h>

#include "wave.h" /* uint4 */

#define CO_NONE 0xFFFFFFFFu
#define CO_GAME_STATE_SIZE 70
#define CO_STATE_STRIDE 80 /* request rows are padded to 80 floats (320 B, 16 B aligned) */
#define CO_NUM_MOVES 96
#define CO_NUM_SYMMETRIES 8
#define CO_MAX_PLIES 44
#define CO_SAMPLE_FLOATS (CO_GAME_STATE_SIZE + CO_NUM_MOVES)
#define CO_PATH_MAX 48
#define CO_MT_N 624
#define CO_TRACE_CAP 12288
#define CO_LOG_CAP (64 * 880) /* int32 per logged game: a ply records up to 7 + 5 x 96 + 6 x 64 + 1 + 4 words, a game has fewer than 64 plies (6-bit depth) */
#define CO_ARENA_PAD 160 /* units readable past the last block (whole-wave block loads) */

/* ref: util.h:57-64 */
#define CO_RESULT_NONE 0
#define CO_RESULT_LOSS 1
#define CO_RESULT_DRAW 2
#define CO_RESULT_WIN 3
#define CO_DEDUCED_LOSS 4
#define CO_DEDUCED_DRAW 5
#define CO_DEDUCED_WIN 6

/* error bits in GameCtl.error */
#define CO_ERR_ARENA_FULL 1
#define CO_ERR_PATH_TOO_DEEP 2
#define CO_ERR_TOO_MANY_PLIES 4
#define CO_ERR_INTERNAL 8
#define CO_ERR_STUCK 16

/* ca_config.step_budget = 0: the number of PUCT scans after which a game's step of fused training stops selecting and
 * carries on in the next launch follows the pool's mean (EngineParams::step_budget_k16; measured: DESIGN section 6
 * "K3, round 6") */
#ifndef CO_STEP_BUDGET_K16
#define CO_STEP_BUDGET_K16 24 /* automatic: 1.5 x the smoothed mean of the launches before (measured, clock: 20 / 24 / 28 / 34 -> trained checkpoint 211 / 207 / 210 / 221 ms, random-init MLP 116.2 / 116.8 / 120.2 / 125.4, CNN 383.9 / 383.0 / 385.7 / 388.6) */
#endif
#define CO_STEP_BUDGET_MIN (24 * CO_STEP_UNITS_PER_CONFIG_UNIT) /* 24 us / 24 scans */
/* A pool's words of EngineParams::work_counter, 64 per pool.  The words every wavefront READS (the budget) lie in another
 * 128-byte line than the ones every wavefront adds to with device-scope atomics: in one line with them (and with the next
 * pool's) the read cost 15 % of a generation -- 152 against 132 ms with a budget no step reaches (round 6). */
#define CO_PACK_STRIDE 16 /* words between two pools' pack counters: a line each */
#define CO_WC_WORDS 64
#define CO_WC_BUDGET 16 /* [2] by launch parity */
#define CO_WC_MEAN 18   /* [2] the smoothed mean, in 1/256 scans */
#define CO_WC_CUTS 32   /* steps cut (ca_stats.steps_cut) */

struct GameCtl {
  int32_t to_play;    /* SelfPlayer::to_play_ */
  int32_t done;       /* Trainer::is_done_[i] */
  int32_t result;     /* SelfPlayer::result_ */
  int32_t mate_turn;  /* SelfPlayer::mate_turn_ */
  int32_t n_samples;  /* samples_.size() */
  int32_t parity;     /* SelfPlayer::parity_ */
  int32_t error;
  int32_t n_pending;  /* players_[to_play_].searched_.size() */
  int32_t rng_idx;    /* position in the mt19937 state, 624 = twist first */
  int32_t plies;
  uint32_t searches;  /* counters for the bench: simulations run ... */
  uint32_t evals;     /* ... leaf evaluations consumed ... */
  uint32_t nodes;     /* ... nodes created */
  int32_t trace_len;
  int32_t row_off;    /* fused mode: first row of this game's requests in the compact batch */
  int32_t resume;     /* fused mode: the new mover's first searches of a turn were deferred to the next step */
  /* tournament matches only: Match::root_, the position on the board (match.h:91) */
  uint32_t pos_lo, pos_hi, pos_meta;
  /* the game this slot plays (index within this trainer's games; = the slot index unless the pool recycles
   * slots, see EngineParams::results) */
  int32_t gid;
  /* the search's hint to itself (mcts.h co_mc_do_iteration): running share, in 1/256, of this game's recent simulations that
   * ended in a terminal leaf or a dead end; decides how many simulations are selected together.  Results do not depend on it. */
  int32_t sb_cap;
  /* Fused training, EngineParams::step_budget: the step before this one ran out of its budget of scans and stopped with
   * fewer than searches_per_eval leaves queued -- none of them was submitted, no evaluation is on its way; this step
   * goes on selecting where that one stopped (mcts.h co_mc_do_iteration).  noise_held = generator outputs owed to the
   * leaves queued so far (CoWave::noise_words, carried over).  The game's own sequence of operations is unchanged. */
  int32_t held;       /* bit 0: as described; within a step only: bit 1 this step continues one that was cut, bit 2 it has run a simulation */
  int32_t noise_held;
};

/* one side of a tournament match: Player, match.h:13-31 */
struct PlayerCfg {
  int32_t max_searches, searches_per_eval;
  float c_puct, epsilon;
  int32_t model_id, random, player_id, pad;
};

struct TreeCtl {
  uint32_t root;          /* block offset of the root, CO_NONE = uninitialised */
  int32_t searches_done;  /* TrainMC::searches_done_ */
  uint32_t units_used;    /* bump pointer */
  uint32_t peak_units;
};

/* Evaluation cache of fused training (mcts.h co_cache_resolve), ONE table for all pools of a trainer (round 4; a table
 * per pool until then: 2.3 % of a generation's rows are positions the OTHER pool has evaluated): a network's outputs are a function of
 * the request row alone, and a generation asks for the same positions again and again (the two players' trees of a
 * game overlap, thousands of games leave the same opening: 29 % of the rows of a 4096-game generation with
 * random-init weights, 44 % with a trained checkpoint, tools/exp/dup_rows.py).  The outputs of every evaluated row
 * live in ONE array, `val`: table entries first (open addressing, 16-byte headers {board lo, board hi, reserves |
 * valid, 0} apart), then a scratch row per request row for positions that found no entry.  A request row is
 * resolved to an element of `val`, recorded with the pending leaf it belongs to (pend_src); the network kernel reads
 * the rows that need evaluating through in_idx and writes their outputs straight to out_idx (nn.h CoNetIO); the
 * priors kernel and the search kernel read through pend_src.  Nothing is copied. */
#define CO_CACHE_VAL_FLOATS 100 /* value, 3 pad, 96 priors */
#define CO_CACHE_PROBES 16
#define CO_CACHE_POOL_SHIFT 28        /* bits 28..29 of an entry's claim word: the pool whose network launch writes it */
#define CO_CACHE_POOL_MASK 0x30000000u
struct EvalCache {
  uint32_t *hdr;       /* [mask + 1][4], shared by the pools; null = no cache.  Word 3: iteration of the claim + 1 */
  float *val;          /* [mask + 1 + rows of all pools][CO_CACHE_VAL_FLOATS], shared */
  uint32_t mask;
  uint32_t scratch_base; /* this pool's first scratch element behind the table = its first batch row */
  uint32_t pool_bits;    /* this pool's index << CO_CACHE_POOL_SHIFT */
  uint32_t no_claim;     /* the iteration behind an emptying of the table: no entry is claimed (rows go to scratch).  The
                          * pending leaves of EVERY pool still point at elements of the old contents, which a pool reads
                          * during this iteration -- an entry claimed now could be another pool's network launch writing
                          * into an element a third party has yet to read.  One iteration later nothing points there. */
  /* The pools' streams are only level at the emptying itself; behind it each runs on by itself.  So for a few windows
   * behind an emptying (the host cannot know more than that the streams are within two windows of each other) a claim
   * also needs every pool in guard_pools to be past the no-claim iteration: done[p] > guard_from, i.e. p's search launch
   * of the iteration behind it has started and its priors and search launches of the no-claim iteration -- the last
   * ones that read through pend_src into the OLD contents -- are over.  (a pool with short launches
   * could claim a slot and have it written while a slower pool was still reading the element.)  guard_pools == 0: no guard. */
  uint32_t guard_from;
  uint32_t guard_pools;
  uint32_t *done;      /* [CO_MAX_POOLS] shared: the network launches of pool p's iterations < done[p] have completed
                        * (stored by the first wave of p's next search launch, which stream order puts behind them) */
  int32_t *in_idx;     /* [pool rows] request rows the network evaluates this iteration, compact ... */
  int32_t *out_idx;    /* ... and the element of val each one writes */
  uint32_t *count;     /* [2][4] by iteration parity: {rows to evaluate, 0, 0, 0} */
  unsigned long long *totals; /* [2] rows evaluated in earlier iterations, - */
};

struct EngineParams {
  /* configuration */
  int32_t num_games;   /* SLOTS of the pool: games resident at a time (= the trainer's games unless it recycles) */
  int32_t max_searches;
  int32_t searches_per_eval;
  float c_puct;
  float epsilon;
  int32_t testing;
  int32_t stagger_div; /* 0 = no staggered start, else max(G / max_searches, 1) (trainer.cpp:184-186) */
  int32_t iteration;   /* Trainer::searches_done_ */
  int32_t to_play;     /* -1 training, 0/1 arena model to move */
  int32_t game_base;   /* global index of local game 0 (multi-GPU shard): parity = (base + g) % 2 */
  uint32_t cap_units;  /* arena units per tree */
  int32_t trace_on;
  /* analysis mode (DockerMC, dockermc.cpp / trainmc.cpp:38-45): every slot is ONE position to search --
   * the root is created from gc.pos_* at depth 0, testing = true, and the slot finishes with its
   * first chooseMove; results in the slot's request area (mcts.h co_analyse_finish) */
  int32_t analyse;
  /* analysis mode, ca_trainer_finish: this launch does not search -- every unfinished slot chooses its move on
   * the tree as it stands (DockerMC::chooseMove called after a time limit, choose_move.pyx:110-117 + :199), the
   * evaluations still pending are never received, as in the reference */
  int32_t force_choose;
  /* tournament mode (Match / Tourney): per-match players [2G]; to_play then carries the MODEL id
   * whose matches run (Match::to_play, match.cpp:42-44); read_offset[G] = the reference's
   * offset table of Tourney::doIteration (tourney.cpp:55-62) */
  const PlayerCfg *pcfg;
  int32_t *read_offset;
  /* fused arena (Trainer test mode with both networks on the device): the model to move lives on
   * the device so that the host need not look at every iteration (main.pyx:150-154 flips it when
   * its batch comes back empty).  [0] model of this iteration, [1] consecutive empty batches,
   * [2] model of the next iteration (committed by the next entry scan), [3], [4] batch rows for
   * network slot 0 / 1 (the idle network's count is 0).  Null: `to_play` is the host's. */
  int32_t *arena_state;
  int32_t scan_phase; /* co_k_scan: 0 = offsets at entry of an iteration, 1 = batch after the search */
  /* pool */
  GameCtl *games;
  TreeCtl *trees;
  uint4 *arena;
  uint32_t *pend_leaf;  /* [G][spe] */
  int32_t *pend_depth;  /* [G][spe] */
  uint32_t *pend_path;  /* [G][spe][CO_PATH_MAX] */
  uint4 *pend_key;      /* [G][spe] cache keys of the pending leaves' request rows (fused training with the evaluation cache) */
  int32_t *pend_src;    /* [G][spe] element of the cache's value array that holds the outputs of pending leaf k (resolved at
                         * the end of the step that queued the leaf): K3's receive phase fetches it with the leaf's other words */
  uint32_t *pend_n;     /* [G][spe][4] {(first noise word of the leaf << 8) | legal moves of the leaf, legal-move mask [3]}:
                         * everything the priors pass (mcts.h co_prior_all) needs to fetch the leaf's priors without walking the tree first */
  uint32_t *noise_raw;  /* [G][spe * CO_NUM_MOVES] generator outputs (untempered state words) reserved for the pending
                         * leaves' Dirichlet noise, in request order: drawn when the leaves are queued (mcts.h
                         * co_capture_noise), consumed by the priors pass of the step that receives the evaluations */
  uint32_t *rng;        /* [G][624] */
  float *req;           /* [G][spe][CO_STATE_STRIDE] */
  int32_t *req_offset;  /* [G+1] exclusive prefix of the active games' request counts; [G] = total */
  const float *nn_eval;   /* [rows] compact, row = req_offset[g] + k */
  const float *nn_probs;  /* [rows][96] */
  float *nn_in;           /* [rows][CO_STATE_STRIDE] compact request rows (host protocol, arena; fused training leaves the
                           * rows where the games wrote them, `req`, and hands the network their indices: row_idx) */
  int32_t *row_idx;       /* [rows] fused training: batch row m of this iteration is request row row_idx[m] of `req`
                           * (= slot * searches_per_eval + pending leaf); the network kernels gather through it (nn.h CoNetIO) */
  float *nn_in70;         /* [rows][70] the same rows as Trainer::writeRequests lays them out (compat mode), or null */
  int32_t *ctl;           /* [4] written by co_k_scan: batch rows, all done, OR of the games' error bits, games not done */
  float *samples;       /* [G][CO_MAX_PLIES][166] */
  int32_t *trace;       /* [G][CO_TRACE_CAP] or null */
  /* per-game text logs (Trainer's num_logged, trainer.cpp:243-250): the first num_logged games record what the
   * reference's log prints at every move choice (mcts.h co_log_ply); word 0 of a game's record is its length.  The host
   * writes the files when the games are over (engine.hip write_logs). */
  int32_t *log;         /* [num_logged][CO_LOG_CAP] or null */
  int32_t num_logged;
  const int32_t *log_index; /* tournament: record of match i, or -1 (addMatch's `logging`); null = games 0 .. num_logged - 1 */
  int32_t *all_done;    /* [1] */
  unsigned long long *row_counter; /* [1] rows handed to the network so far */
  /* fused training mode: K3 packs its own requests.  pack_counter[iteration & 1] =
   * (games still running << 32) | rows of this iteration's batch; the other one is
   * cleared for the next iteration.  The network kernels read the low word. */
  int32_t fused_pack;
  int32_t defer_handover; /* fused mode: end a game's step at the hand-over (see mcts.h co_choose_move_and_continue) */
  /* fused training: a game's step stops selecting after this many PUCT scans and carries on in the next launch, its
   * queued leaves held back until there are searches_per_eval of them (or the searches run out) -- a launch lasts as long
   * as its slowest wavefront, and the slowest are steps of 25-30 simulations ten levels deep (half of them ending in
   * terminal leaves, which queue nothing).  0 = no limit.  Per-game results do not depend on it. */
  int32_t step_budget;
  /* step_budget_k16 > 0 (and step_budget == 0): the budget follows the games -- k16 / 16 times the mean number of scans
   * the pool's stepping games made in the launch before this one (early plies: wide shallow trees, a step is ~35 scans;
   * endgames under a trained network: ~70).  work_counter[CO_WC_WORDS] per pool: [0..2] by iteration mod 3, (steps << 32) |
   * scans of the launch -- this launch adds to [iteration % 3], its first wavefront reads [(iteration - 1) % 3], writes the
   * budget of the NEXT launch to [CO_WC_BUDGET + (iteration & 1)] and clears [(iteration + 1) % 3]; every wavefront reads
   * its budget from [CO_WC_BUDGET + ((iteration + 1) & 1)], the word the launch before wrote (mcts.h
   * co_pool_housekeeping); [CO_WC_MEAN ..]: the mean smoothed over ~8 launches, carried from launch to launch the same way. */
  int32_t step_budget_k16;
  unsigned long long *work_counter;
  /* fused training runs the games as independent pools on separate streams, so that one pool's
   * search overlaps another pool's network kernel: this launch covers games [pool_lo, pool_lo +
   * pool_n) and its batch rows start at row pool_row_base of nn_in / nn_eval / nn_probs */
  int32_t pool_lo, pool_n, pool_row_base;
  unsigned long long *pack_counter; /* [2] */
  /* Resident-slot pool (ca_config.resident < num_games, training only): the trainer's `total_local` games are
   * played on num_games slots.  A slot whose game ends stores the game's control block in results[gid], takes
   * the next unstarted game from next_game[0] and seeds its generator from seeds[gid] (the Trainer stream in
   * game order, trainer.cpp:243-255) -- a game's own sequence of operations does not depend on the slot or
   * the moment it starts (every game owns its generator).  Samples and traces are always addressed by gid.
   * results == null: no recycling (slot = game). */
  EvalCache cache;                /* fused training, per pool */
  GameCtl *results;               /* [total_local] */
  unsigned long long *next_game;  /* [1] */
  const uint32_t *seeds;          /* [total_local] */
  int32_t total_local;
  unsigned long long *prof;         /* [G][8] cycle stamps, profiling builds (-DCO_PROF) only */
};
