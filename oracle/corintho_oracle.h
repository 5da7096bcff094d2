/* corintho_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's self-play hot path
 * (maxjiang216/corintho-ai: corintho_ai/cpp/src/{game,move,node,trainmc,
 * selfplayer,trainer}.cpp).  Every function cites the reference lines it
 * follows.  It keeps the reference's data structures (pointer-linked 64-byte
 * style nodes, sorted sibling lists, per-node edge arrays, new/delete per
 * node) on purpose: the product under corintho_ai_amd/ uses a different
 * layout, so agreement between the two is meaningful.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under corintho_ai_amd/ links or calls it.
 *
 * Parity status: the game-rule layer is pinned against the reference's own
 * known-answer tests (tests/cpp/{game,move,node}_test.cpp, restated in
 * tests/test_oracle_reference_tests.py).  For the search layer the reference
 * holds only property tests (no golden vectors) and the reference itself is
 * unbuildable in this image (it needs Microsoft GSL headers, which are
 * absent, and stand-ins are not allowed): search parity is pinned by those
 * property tests and by line-by-line restatement -- "parity unpinned" beyond
 * that.  See DESIGN.md.
 */
#ifndef CORINTHO_ORACLE_H
#define CORINTHO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CO_GAME_STATE_SIZE 70
#define CO_NUM_MOVES 96
#define CO_NUM_SYMMETRIES 8

/* ---- game-rule layer (game.cpp, move.cpp) ---- */
/* pieces[6]: P1 base/column/capital, P2 base/column/capital (game.h:127-131) */
int co_legal_moves(uint64_t board, const int8_t pieces[6], int to_play,
                   uint32_t mask_out[3]); /* returns is_lines */
void co_do_move(uint64_t *board, int8_t pieces[6], int *to_play, int move_id);
void co_write_game_state(uint64_t board, const int8_t pieces[6], int to_play,
                         float out[CO_GAME_STATE_SIZE]);
/* node.cpp:256-271: 0 none, 1 loss, 2 draw for the side to move */
int co_terminal_result(uint64_t board, const int8_t pieces[6], int to_play);
/* move codec (move.cpp:11-42): out = {is_place, piece, row_from, col_from,
 * row_to, col_to} */
void co_decode_move(int move_id, int out[6]);
int co_encode_place(int row, int col, int piece);
int co_encode_move(int r0, int c0, int r1, int c1);
/* tables */
const uint32_t *co_line_breakers(void);   /* [102][3] */
const float *co_gamma_samples(void);      /* [1024]  */
const int32_t *co_space_symmetries(void); /* [8][16] */
const int32_t *co_move_symmetries(void);  /* [8][96] */

/* ---- mt19937 (libstdc++ std::mt19937) ---- */
typedef struct co_mt19937 co_mt19937;
co_mt19937 *co_mt_create(uint32_t seed);
uint32_t co_mt_next(co_mt19937 *g);
void co_mt_destroy(co_mt19937 *g);

/* ---- Trainer (trainer.h:22-53) ---- */
typedef struct co_trainer co_trainer;
co_trainer *co_trainer_create(int num_games, int seed, int max_searches,
                              int searches_per_eval, float c_puct,
                              float epsilon, int num_threads, int testing);
/* games [first, first + num_games) of a Trainer of total_games games: same seeds (drawn in global
 * index order, trainer.cpp:243-255), parity, colours and stagger rule as the full Trainer -- the
 * slice the multi-GPU tests replay a shard on */
co_trainer *co_trainer_create_slice(int total_games, int first, int num_games, int seed, int max_searches,
                                    int searches_per_eval, float c_puct, float epsilon, int num_threads, int testing);
/* per-game text logs of the first num_logged games, as Trainer::initialize sets them up (trainer.cpp:243-250) */
int co_trainer_set_logging(co_trainer *t, const char *log_folder, int num_logged);
void co_trainer_destroy(co_trainer *t);
int co_trainer_num_requests(const co_trainer *t, int to_play);
int co_trainer_num_samples(const co_trainer *t);
float co_trainer_score(const co_trainer *t);
float co_trainer_avg_mate_length(const co_trainer *t);
void co_trainer_write_requests(const co_trainer *t, float *game_states,
                               int to_play);
/* Trainer::writeScores, trainer.cpp:115-162; returns 0 if the file cannot be opened */
int co_trainer_write_scores(const co_trainer *t, const char *filename);
void co_trainer_write_samples(const co_trainer *t, float *game_states,
                              float *eval_samples, float *prob_samples);
int co_trainer_do_iteration(co_trainer *t, const float *eval,
                            const float *probs, int to_play);
/* Disable the staggered start (trainer.cpp:184-186).  Per-game results do not
 * depend on it; the device engine's fused mode runs without it. */
void co_trainer_set_stagger(co_trainer *t, int on);

/* ---- introspection used by the parity tests ---- */
/* per game: result code (util.h:57-64) or 0 while running */
int co_trainer_game_result(const co_trainer *t, int game);
int co_trainer_game_to_play(const co_trainer *t, int game);
int co_trainer_game_num_requests(const co_trainer *t, int game);
int co_trainer_game_num_samples(const co_trainer *t, int game);
int co_trainer_game_done(const co_trainer *t, int game);
/* Per-ply trace, appended at every chooseMove (selfplayer.cpp:234-244):
 *   [to_play, depth, root_visits, root_result, root_eval_bits, n_children,
 *    {move, visits, eval_bits, result, all_visited} x n_children, choice]
 * Returns the number of int32 words; copies min(cap, words). */
void co_trainer_enable_trace(co_trainer *t, int on);
int co_trainer_trace(const co_trainer *t, int game, int32_t *out, int cap);
/* counters summed over games: [searches, leaf_evals, nodes_created, plies] */
void co_trainer_counters(const co_trainer *t, int64_t out[4]);

/* ---- DockerMC (dockermc.h:13-51): single-position search from an arbitrary position ---- */
typedef struct co_dockermc co_dockermc;
co_dockermc *co_dockermc_create(int seed, int max_searches, int searches_per_eval, float c_puct, float epsilon,
                                const int32_t board[64], int to_play, const int32_t pieces[6]);
void co_dockermc_destroy(co_dockermc *d);
float co_dockermc_eval(const co_dockermc *d);
int co_dockermc_num_requests(const co_dockermc *d);
int co_dockermc_num_nodes(const co_dockermc *d);
int co_dockermc_done(const co_dockermc *d);
int co_dockermc_drawn(const co_dockermc *d);
void co_dockermc_write_requests(const co_dockermc *d, float *game_states);
void co_dockermc_get_legal_moves(const co_dockermc *d, int32_t legal_moves[CO_NUM_MOVES]);
int co_dockermc_choose_move(co_dockermc *d);
int co_dockermc_do_iteration(co_dockermc *d, const float *eval, const float *probs);

/* ---- Match / Tourney (match.h, tourney.h:13-46) ---- */
typedef struct co_tourney co_tourney;
co_tourney *co_tourney_create(int num_threads);
void co_tourney_destroy(co_tourney *t);
int co_tourney_add_player(co_tourney *t, int player_id, int model_id, int max_searches, int searches_per_eval,
                          float c_puct, float epsilon, int random);
int co_tourney_add_match(co_tourney *t, int player1, int player2); /* returns the match index */
/* addMatch(player1, player2, logging = true) of a Tourney(num_threads, log_folder): `<log_folder>/match_<p1>_<p2>_<index>.txt` */
int co_tourney_add_match_logged(co_tourney *t, int player1, int player2, const char *log_folder);
int co_tourney_all_done(const co_tourney *t);
int co_tourney_num_requests(const co_tourney *t, int id);
void co_tourney_write_requests(const co_tourney *t, float *game_states, int id);
void co_tourney_do_iteration(co_tourney *t, const float *eval, const float *probs, int id);
void co_tourney_set_exact_offsets(co_tourney *t, int on); /* diagnostic, not in the reference */
int co_tourney_write_scores(const co_tourney *t, const char *filename);
int co_tourney_num_matches(const co_tourney *t);
int co_tourney_match_done(const co_tourney *t, int i);
float co_tourney_match_score(const co_tourney *t, int i);
int co_tourney_match_result(const co_tourney *t, int i);
int co_tourney_match_to_play(const co_tourney *t, int i);
int co_tourney_match_num_requests(const co_tourney *t, int i);
/* per-ply trace as co_trainer_trace; a random player's ply is [-2, choice] */
void co_tourney_enable_trace(co_tourney *t, int on);
int co_tourney_trace(const co_tourney *t, int i, int32_t *out, int cap);
void co_tourney_counters(const co_tourney *t, int64_t out[4]);
/* std::uniform_int_distribution<int32_t>(0, n - 1) on the generator, as libstdc++ computes it */
uint32_t co_uniform_below(co_mt19937 *g, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
