/* corintho_hip.h -- C ABI of libcorintho_hip.so, the MI355X self-play engine.
 *
 * Drop-in boundary for the reference's hot path.  The reference exposes a C++
 * class consumed at source level by Cython
 * (corintho_ai/python/main.pyx:17-38  `cdef cppclass Trainer`, declared in
 * corintho_ai/cpp/include/trainer.h:17-81).  Each entry point below replaces
 * one member of that class; corintho_ai_amd/cpp/trainer.h wraps them back into
 * a `class Trainer` with the reference's exact signatures, so main.pyx compiles
 * against it unchanged.
 *
 * Conventions: every function returns 0 on success and a negative code on
 * failure; ca_last_error() gives the message (thread local).  All buffers are
 * caller-owned host memory, C-contiguous float32, laid out exactly as the
 * reference lays them out (main.pyx:132-134, 194-198).  No pointer is retained
 * across calls.  One ca_trainer drives one GPU; it is not thread safe (the
 * reference is called with the GIL held).
 */
#ifndef CORINTHO_HIP_H
#define CORINTHO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CA_GAME_STATE_SIZE 70 /* util.h:42 kGameStateSize */
#define CA_NUM_MOVES 96       /* util.h:44 kNumMoves */
#define CA_NUM_SYMMETRIES 8   /* util.h:48 kNumSymmetries */

#define CA_OK 0
#define CA_ERR_ARG -1
#define CA_ERR_DEVICE -2   /* HIP failure, no GPU, extension built for another arch */
#define CA_ERR_ENGINE -3   /* a game reported an engine error (arena full, ...) */
#define CA_ERR_STATE -4    /* call made in the wrong state */
#define CA_ERR_IO -5

typedef struct ca_trainer ca_trainer;

/* Trainer::Trainer arguments (trainer.h:22-25) followed by device options that
 * have no reference counterpart.  Zero-initialise, then fill. */
typedef struct ca_config {
  /* reference ctor arguments, same meaning and defaults as trainer.h:22-25 */
  int32_t num_games;
  int32_t seed;
  int32_t max_searches;       /* default 1600 */
  int32_t searches_per_eval;  /* default 16 */
  float c_puct;               /* default 1.0 */
  float epsilon;              /* default 0.25 */
  int32_t num_logged;         /* must be 0 here: the per-game text logs are switched on by ca_trainer_set_logging */
  int32_t num_threads;        /* accepted and ignored (OpenMP width of the reference) */
  int32_t testing;            /* arena mode: no samples, no opening temperature */
  /* device options */
  int32_t device;             /* HIP device ordinal */
  int32_t no_stagger;         /* 1: start every game at iteration 0 (trainer.cpp:184-186 is a
                                 memory heuristic; per-game results do not depend on it) */
  uint32_t arena_units;       /* 16-byte units per search tree; 0 = default from max_searches */
  int32_t trace;              /* keep a per-ply trace for the parity tests */
  /* sharding (multi-GPU): this trainer owns games [game_base, game_base+num_games) of a
   * generation of total_games games; seeds are drawn from the Trainer stream in global
   * order and parity = global index % 2 (trainer.cpp:243-255).  0/0 = unsharded. */
  int32_t game_base;
  int32_t total_games;
  /* fused training: number of independent game pools run on separate HIP streams of the same
   * GPU (one pool's search overlaps another's network kernel); 0 = automatic */
  int32_t pools;
  /* analysis mode (DockerMC, dockermc.h:13-51): every "game" is ONE position to search,
   * given by ca_trainer_set_positions; testing is implied; a slot is finished by its first chooseMove */
  int32_t analyse;
  /* Resident slots (training mode only).  The reference holds the trees of all num_games games and staggers their
   * starts to bound the memory (trainer.cpp:184-186).  Here `resident` slots hold the games in play; a slot whose
   * game ends takes the next game (its generator seeded from the Trainer stream by game index, trainer.cpp:243-255,
   * so no game's result depends on the slot or moment it starts), and the finished game's tree memory is given back
   * whole.  0 = automatic (all games resident when their trees fit in 5/8 of the free device memory, else as many
   * slots as fit); >= num_games or < 0 = all resident.  With fewer slots than games the staggered start is off. */
  int32_t resident;
  /* Evaluation cache of fused training (ca_trainer_run): a request row whose position was evaluated earlier in the same
   * generation receives the stored outputs instead of a second evaluation -- bit for bit what the network kernel would
   * write again, since a row's outputs depend on the row only.  Emptied at the start of every generation (and whenever
   * it is half full).  0 = on, table sized from the pool and the free memory; n > 0 = on with 2^n entries (one table for all pools of the trainer);
   * negative = off.  (The reference evaluates every request, main.pyx:70-83; results are identical either way.) */
  int32_t eval_cache;
  /* Fused training: a game's step (the body of Trainer::doIteration's loop for that game, trainer.cpp:175-196) stops
   * selecting once it has run for this long and goes on in the next iteration; the leaves it has queued are held back until
   * the batch of searches_per_eval is complete (or the move's searches run out), so the game performs exactly the
   * reference's sequence of operations -- only spread over more iterations.  It bounds what a launch waits for: its
   * slowest game (endgame positions under a trained network: simulations ten levels deep, half of them ending in
   * terminal leaves that queue nothing).  0 = automatic (1.6 x the running mean of the pool's steps), n > 0 = n
   * microseconds, -1 = no limit (below -1, diagnostic: automatic with the factor -n / 16).  Results are identical either
   * way; the number of iterations of a generation is not. */
  int32_t step_budget;
} ca_config;

const char *ca_last_error(void);
/* 0 if a usable gfx950 device is visible */
int ca_device_check(int device);

/* Trainer::Trainer (trainer.cpp:18-37) / ~Trainer */
int ca_trainer_create(const ca_config *cfg, ca_trainer **out);
void ca_trainer_destroy(ca_trainer *t);

/* Trainer::initialize, trainer.cpp:243-250: the first num_logged games of the generation write
 * `<log_folder>/game_<i>.txt` (i = game_base + index), the text SelfPlayer prints at every move choice
 * (selfplayer.cpp:124-232; main line node.cpp:197-240).  Call before the first iteration.  The search kernel records
 * the numbers; the files are written when the last game is over (the reference writes as it goes), byte for byte the
 * reference's text.  A file that cannot be opened is skipped silently, as the reference's ofstream is.  Self-play and
 * arena trainers only. */
int ca_trainer_set_logging(ca_trainer *t, const char *log_folder, int32_t num_logged);

/* int Trainer::num_requests(int to_play) -- trainer.cpp:39-49 */
int ca_trainer_num_requests(ca_trainer *t, int to_play, int32_t *out);
/* int Trainer::num_samples() -- trainer.cpp:51-57 */
int ca_trainer_num_samples(ca_trainer *t, int32_t *out);
/* float Trainer::score() -- trainer.cpp:59-68 */
int ca_trainer_score(ca_trainer *t, float *out);
/* float Trainer::avg_mate_length() -- trainer.cpp:70-77 */
int ca_trainer_avg_mate_length(ca_trainer *t, float *out);
/* void Trainer::writeRequests(float *game_states, int to_play) -- trainer.cpp:79-101
 * game_states: [num_requests(to_play)][70] */
int ca_trainer_write_requests(ca_trainer *t, float *game_states, int to_play);
/* void Trainer::writeSamples(float*, float*, float*) -- trainer.cpp:103-113
 * [n*8][70], [n*8], [n*8][96] with n = num_samples() */
int ca_trainer_write_samples(ca_trainer *t, float *game_states, float *eval_samples, float *prob_samples);
/* void Trainer::writeScores(const std::string &file) -- trainer.cpp:115-162 */
int ca_trainer_write_scores(ca_trainer *t, const char *file);
/* bool Trainer::doIteration(float eval[], float probs[], int to_play) -- trainer.cpp:164-236
 * evaluations: [num_requests], probabilities: [num_requests][96] for the rows last
 * written by writeRequests(to_play); ignored on the first call.  *all_done = return value. */
int ca_trainer_do_iteration(ca_trainer *t, const float *evaluations, const float *probabilities, int to_play,
                            int32_t *all_done);

/* Optional, for the reference protocol above: page-lock a caller-owned array (the three arrays the
 * reference's play loop allocates once and passes to every call, main.pyx:132-134) so that the
 * per-iteration copies from / to it are direct DMA at PCIe speed instead of staged copies from
 * pageable memory.  The array must stay allocated until ca_trainer_unpin_host or
 * ca_trainer_destroy.  *pinned = 0 if the runtime refused (the calls still work, slower). */
int ca_trainer_pin_host(ca_trainer *t, void *p, size_t bytes, int32_t *pinned);
int ca_trainer_unpin_host(ca_trainer *t, void *p);

/* ---------------- analysis of N positions (replaces N x `class DockerMC`, dockermc.h:13-51) ----------------
 * A trainer created with ca_config.analyse = 1 searches num_games independent positions, one wavefront
 * each -- what docker/choose_move.pyx does for one position with one DockerMC, batched.  Position i is the
 * DockerMC constructor's (seed, board[64] (bit = cell*4 + {base, column, capital, frozen}), to_play,
 * pieces[6]) (dockermc.cpp:11-19, game.cpp:14-26); max_searches, searches_per_eval, c_puct, epsilon come
 * from ca_config.  Call once, before the first iteration.  Then either protocol applies: the reference's
 * (ca_trainer_do_iteration / num_requests / write_requests with to_play = -1: DockerMC::doIteration /
 * num_requests / writeRequests over all positions at once) or the fused one (ca_trainer_set_net +
 * ca_trainer_run).  A position whose search is over has chosen its move (DockerMC::chooseMove). */
int ca_trainer_set_positions(ca_trainer *t, const int32_t *boards /* [n][64] */, const int32_t *to_play /* [n] */,
                             const int32_t *pieces /* [n][6] */, const int32_t *seeds /* [n] */);
/* out[i] = {move (chooseMove; -1: the given position was already terminal), done (DockerMC::done of the
 * position after the move), drawn (DockerMC::drawn), nodes (num_nodes), evaluation bits (float eval()),
 * legal-move mask of the new position [3] (getLegalMoves)}: what choose_move.pyx:206-221 reads */
int ca_trainer_analysis(ca_trainer *t, int32_t *out /* [n][8] */);
/* DockerMC::chooseMove (dockermc.cpp:48-50) on searches that have NOT ended: docker/choose_move.pyx:110-117 leaves
 * its loop on a time limit as well as on doIteration returning true, and calls chooseMove either way (:199).
 * Every position that has not finished chooses its move on its tree as it stands (TrainMC::chooseMove,
 * trainmc.cpp:110-137; evaluations still pending are not received); afterwards ca_trainer_analysis reports all
 * positions.  Before any iteration, the root is created first (the TrainMC constructor's createRoot). */
int ca_trainer_finish(ca_trainer *t);

/* ---------------- fused mode (no reference counterpart; opt-in) ----------------
 * The network runs on the device, so the play loop of main.pyx:123-187 never
 * leaves the GPU.  Weights are a flat float32 buffer in the layout documented in
 * corintho_ai_amd/nets.py for each kind. */
#define CA_NET_MLP12X100 1 /* the reference architecture, wrapper.py:256-271 */
#define CA_NET_RESCNN4 2   /* the north-star 4-block residual CNN, fp32 MFMA */
#define CA_NET_RESCNN4_X3 3 /* the same network and weights; 3x3 convolutions at bf16x3 split precision
                              (bf16 MFMA, fp32 accumulate; within 2e-5 of the fp32 result) */
#define CA_NET_MLP12X100_X3 4 /* the reference architecture and weights; dense layers at bf16x3 split precision */
/* float32-equivalent arithmetic on the bf16 matrix pipe ("bf16x6"): both operands of every matrix product as
 * THREE bf16 terms whose sum is the float32 value, the six products down to 2^-16 of the leading one kept (the
 * dropped ones are below one float32 rounding of the product), fp32 accumulation.  Error against a float64
 * evaluation = that of the fp32-MFMA kinds 1 and 2 (tests/test_net_precision.py). */
#define CA_NET_RESCNN4_X6 5
#define CA_NET_MLP12X100_X6 6
/* "f16x3": both operands of every matrix product as TWO fp16 terms, x = fp16(x) + fp16(x - fp16(x)) -- 22 significand
 * bits each -- and the three products w0 x0 + w0 x1 + w1 x0 on the fp16 matrix pipe, fp32 accumulation: the dropped
 * terms are 2^-22 of a product.  Float32-class results at half the matrix work of bf16x6; error against float64
 * measured beside the other kinds in tests/test_net_precision.py. */
#define CA_NET_RESCNN4_H3 8
#define CA_NET_MLP12X100_H3 9
/* slot 0 = best model (training, and arena `to_play == 1`), slot 1 = new model (arena
 * `to_play == 0`), as get_predictions chooses them (main.pyx:70-83) */
int ca_trainer_set_net(ca_trainer *t, int slot, int kind, const float *weights, size_t n_floats);
/* Run the whole generation on the device.  max_iterations 0 = until done. */
int ca_trainer_run(ca_trainer *t, int64_t max_iterations, int32_t *all_done);
/* One network evaluation of host rows through the device kernels (numerics tests):
 * states [n][70] -> evals [n], probs [n][96] */
int ca_trainer_net_forward(ca_trainer *t, int slot, const float *states, int32_t n, float *evals, float *probs);

/* Kernel-only time of one network evaluation of `rows` resident rows (average of `reps`
 * launches, HIP events) -- diagnostics */
int ca_trainer_net_bench(ca_trainer *t, int slot, const float *states, int32_t rows, int32_t reps, float *ms_per_call);

/* Un-augmented samples for the multi-GPU gather: n = num_samples() rows of
 * (state[70], policy[96]) + outcome[n] in game order; the x8 symmetry expansion
 * is applied after the gather by ca_expand_samples. */
int ca_trainer_export_samples(ca_trainer *t, float *state_policy /* [n][166] */, float *outcome /* [n] */);
/* The same rows packed on the device into caller-owned DEVICE buffers
 * ([cap_rows][166] and [cap_rows] float32), ready for an RCCL all-gather. */
int ca_trainer_pack_samples_device(ca_trainer *t, void *d_state_policy, void *d_outcome, int32_t cap_rows,
                                   int32_t *n_rows);
/* Start a new generation in the same pool: Trainer::initialize (trainer.cpp:238-256)
 * with a new seed, without reallocating the device buffers. */
int ca_trainer_reset(ca_trainer *t, int32_t seed);
int ca_expand_samples(int device, const float *state_policy, const float *outcome, int32_t n, float *game_states,
                      float *eval_samples, float *prob_samples);

/* ---------------- introspection (tests, bench) ---------------- */
typedef struct ca_stats {
  int64_t searches;      /* simulations run */
  int64_t evals;         /* leaf evaluations consumed */
  int64_t nodes;         /* nodes created */
  int64_t plies;
  int64_t iterations;    /* doIteration calls / fused steps */
  int64_t peak_arena_units; /* high-water mark over all trees */
  double mcts_ms, nn_ms, pack_ms; /* device time by kernel family (fused mode, HIP events) */
  int64_t mcts_launches, nn_launches;
  int64_t nn_rows;       /* rows evaluated by the network kernels */
  int64_t pools;         /* pools the last ca_trainer_run used (1 in arena mode) */
  /* fused training times one iteration per pool and window of 8 with HIP events (mcts_ms / nn_ms
   * above are then estimates: timed sums scaled by launches / timed launches); exact figures of
   * the timed launches: */
  int64_t timed_launches; /* search + network launch pairs that carried events */
  int64_t nn_timed_rows;  /* batch rows of those network launches */
  double mcts_timed_ms, nn_timed_ms;
  int64_t resident_slots; /* slots of the pool (= num_games unless it recycles, ca_config.resident) */
  int64_t nn_rows_evaluated; /* rows the network kernels worked on; nn_rows - this = rows served by the evaluation cache */
  int64_t steps_cut;         /* ca_config.step_budget: game steps of the last ca_trainer_run that stopped at their budget and went on in the next iteration */
  int64_t step_budget_last;  /* the budget (microseconds) the first pool's last launch worked under; 0 = none */
} ca_stats;
int ca_trainer_stats(ca_trainer *t, ca_stats *out);
/* per-game: {to_play, done, result, n_samples, n_pending, error, mate_turn, plies} */
int ca_trainer_game_info(ca_trainer *t, int game, int32_t out[8]);
/* per-ply trace of one game, same record format as the oracle's; returns words via *n */
int ca_trainer_trace(ca_trainer *t, int game, int32_t *out, int32_t cap, int32_t *n);
/* diagnostic builds of the library (-DCO_PROF) only: summed in-kernel cycle stamps of the search kernel
 * (slots documented in csrc/mcts.h); the shipped build returns CA_ERR_STATE */
int ca_trainer_prof(ca_trainer *t, unsigned long long out[88]);

/* ---- Tourney: replaces `class Tourney` of corintho_ai/cpp/include/tourney.h:13-46
 * consumed by corintho_ai/rating/tourney.pyx:15-31.  One match = one slot of a device pool
 * (two search trees, one wavefront); the pool is built at the first query after the last
 * ca_tourney_add_match.  Model ids are the caller's (negative = the dummy ids of random
 * players, tourney.pyx:131-134). ---- */
typedef struct ca_tourney ca_tourney;
/* Tourney(num_threads, log_folder), tourney.h:15 (threads have no device meaning; the folder: ca_tourney_set_log_folder) */
int ca_tourney_create(int device, uint32_t arena_units, int trace, ca_tourney **out);
/* Tourney::log_folder_ (tourney.h:46): where the matches added with logging = true write
 * `match_<player1>_<player2>_<index>.txt` (tourney.cpp:90-95; the text of match.cpp:78-190).  Before the first query.
 * The files are written when the last match is over (the reference writes them as the matches go); one that cannot
 * be opened is skipped silently, as the reference's ofstream is. */
int ca_tourney_set_log_folder(ca_tourney *t, const char *log_folder);
void ca_tourney_destroy(ca_tourney *t);
/* Tourney::addPlayer, tourney.cpp:72-78 */
int ca_tourney_add_player(ca_tourney *t, int32_t player_id, int32_t model_id, int32_t max_searches,
                          int32_t searches_per_eval, float c_puct, float epsilon, int32_t random);
/* Tourney::addMatch, tourney.cpp:80-96: the match's generator is seeded with the next output of
 * the tourney's default-constructed std::mt19937; `logging`: the match writes its text log (ca_tourney_set_log_folder) */
int ca_tourney_add_match(ca_tourney *t, int32_t player1, int32_t player2, int32_t logging);
/* Tourney::all_done, tourney.cpp:14-21 */
int ca_tourney_all_done(ca_tourney *t, int32_t *out);
/* Tourney::num_requests(id), tourney.cpp:23-31 */
int ca_tourney_num_requests(ca_tourney *t, int32_t id, int32_t *out);
/* Tourney::writeRequests(game_states, id), tourney.cpp:43-51: [num_requests(id)][70] */
int ca_tourney_write_requests(ca_tourney *t, float *game_states, int32_t id);
/* Tourney::doIteration(eval, probs, id), tourney.cpp:53-70.  `rows` = rows of the two caller
 * arrays (eval[rows], probs[rows][96]): matches read them through the reference's own offset
 * table (tourney.cpp:55-62), which can differ from the writeRequests order */
int ca_tourney_do_iteration(ca_tourney *t, const float *evaluations, const float *probabilities, int32_t rows,
                            int32_t id);
/* Fused mode (not in the reference): the networks run on the GPU too.  ca_tourney_set_net
 * gives model `model_id` its network (kinds and weight layouts as ca_trainer_set_net);
 * ca_tourney_run plays the loop of rating/tourney.pyx:122-160 on the device, model ids in
 * ascending order, until every match is done or `max_rounds` rounds have run (0 = no limit). */
int ca_tourney_set_net(ca_tourney *t, int32_t model_id, int32_t kind, const float *weights, size_t n_floats);
int ca_tourney_run(ca_tourney *t, int64_t max_rounds, int32_t *all_done);
/* Diagnostic (not in the reference): 1 = every match reads its evaluations at the rows Tourney::writeRequests
 * gave it, instead of through Tourney::doIteration's own offset table (tourney.cpp:55-62, SURVEY 8a quirk 10,
 * which hands most matches the rows of OTHER matches).  Before the first query; default 0 = the reference's table. */
int ca_tourney_set_exact_offsets(ca_tourney *t, int32_t on);
/* Tourney::writeScores, tourney.cpp:33-41 */
int ca_tourney_write_scores(ca_tourney *t, const char *filename);
int ca_tourney_num_matches(ca_tourney *t, int32_t *out);
/* out[8] = {player id 1, player id 2, done, result for the first player, side to move, pending requests, plies, error} */
int ca_tourney_match_info(ca_tourney *t, int32_t match, int32_t out[8]);
/* Match::score, match.cpp:52-58 */
int ca_tourney_match_score(ca_tourney *t, int32_t match, float *out);
int ca_tourney_trace(ca_tourney *t, int32_t match, int32_t *out, int32_t cap, int32_t *n);
int ca_tourney_stats(ca_tourney *t, ca_stats *out);

/* rule layer on a batch of positions (one wavefront each): legal-move masks + is_lines */
int ca_rules_legal_moves(int device, const uint64_t *boards, const uint32_t *metas, int32_t n, uint32_t *masks /* [n][3] */,
                         int32_t *is_lines);
/* apply moves[i] (or -1 for none) and expand the 70-float state */
int ca_rules_do_move(int device, uint64_t *boards, uint32_t *metas, const int32_t *moves, int32_t n,
                     float *states /* [n][70] */);
/* the same two through the search's four-positions-per-wavefront rule layer (csrc/rules.h co_do_move_lane,
 * co_legal_moves_rows): apply moves[i] (or -1 for none), then the legal-move mask of the new position */
int ca_rules_rows(int device, uint64_t *boards, uint32_t *metas, const int32_t *moves, int32_t n, uint32_t *masks /* [n][3] */);
/* std::mt19937 through the device draw path: n outputs for `seed`, drawn `chunk` at a time */
int ca_rng_draw(int device, uint32_t seed, int32_t n, int32_t chunk, uint32_t *out);
/* floating-point contract probe, see kernels.h co_k_fp_probe: in [n][8] -> out [n][8] */
int ca_fp_probe(int device, const float *in, int32_t n, float *out);

#ifdef __cplusplus
}
#endif
#endif
