/* corintho_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See corintho_oracle.h for scope and parity status.  "ref:" comments cite
 * /root/reference/corintho_ai/cpp/... file:line.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fopenmp -fPIC -shared
 *        (no -march=native, no -ffast-math: the reference's x86-64 build has
 *        no FMA and parity depends on that -- SURVEY 8c).
 */
#include "corintho_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "tables.inc"

/* ------------------------------------------------------------------ tables */
static const uint32_t LINE_BREAKERS[CO_NUM_LINES][3] = CO_LINE_BREAKERS_INIT;
static const uint32_t GAMMA_BITS[CO_NUM_GAMMA] = CO_GAMMA_BITS_INIT;
static const int32_t SPACE_SYM[8][16] = CO_SPACE_SYM_INIT;
static const int32_t MOVE_SYM[8][96] = CO_MOVE_SYM_INIT;

static inline float gamma_sample(uint32_t i) {
  float f;
  memcpy(&f, &GAMMA_BITS[i], 4);
  return f;
}

const uint32_t *co_line_breakers(void) { return &LINE_BREAKERS[0][0]; }
const float *co_gamma_samples(void) { return (const float *)GAMMA_BITS; }
const int32_t *co_space_symmetries(void) { return &SPACE_SYM[0][0]; }
const int32_t *co_move_symmetries(void) { return &MOVE_SYM[0][0]; }

/* ref: util.h:57-64 */
enum {
  kResultNone = 0,
  kResultLoss = 1,
  kResultDraw = 2,
  kResultWin = 3,
  kDeducedLoss = 4,
  kDeducedDraw = 5,
  kDeducedWin = 6
};
/* ref: util.h:67-82 */
enum { RL = 0, RR = 1, RB = 2, CU = 3, CD = 4, CB = 5 };
enum { D0U = 0, D0D = 1, D0B = 2, D1U = 3, D1D = 4, D1B = 5, S0 = 6, S1 = 7, S2 = 8, S3 = 9 };
enum { kBase = 0, kColumn = 1, kCapital = 2, kFrozen = 3 };

/* ------------------------------------------------------------------- moves */
typedef struct {
  int is_place, piece, r0, c0, r1, c1;
} move_t;

/* ref: move.cpp:11-42 */
static move_t decode_move(int id) {
  move_t m;
  m.is_place = id >= 48;
  m.piece = 0;
  m.r0 = m.c0 = -1;
  if (m.is_place) {
    m.piece = (id - 48) / 16;
    m.r1 = (id % 16) / 4;
    m.c1 = id % 4;
    return m;
  }
  if (id < 12) { /* right */
    m.r0 = id / 3; m.c0 = id % 3; m.r1 = id / 3; m.c1 = id % 3 + 1;
    return m;
  }
  if (id < 24) { /* down */
    m.r0 = (id - 12) / 4; m.c0 = id % 4; m.r1 = (id - 12) / 4 + 1; m.c1 = id % 4;
    return m;
  }
  if (id < 36) { /* left */
    m.r0 = (id - 24) / 3; m.c0 = id % 3 + 1; m.r1 = (id - 24) / 3; m.c1 = id % 3;
    return m;
  }
  /* up */
  m.r0 = (id - 36) / 4 + 1; m.c0 = id % 4; m.r1 = (id - 36) / 4; m.c1 = id % 4;
  return m;
}

/* ref: move.cpp:80-84 */
int co_encode_place(int row, int col, int piece) { return 48 + piece * 16 + row * 4 + col; }

/* ref: move.cpp:86-108 */
int co_encode_move(int r0, int c0, int r1, int c1) {
  if (c0 < c1) return r0 * 3 + c0;
  if (r0 < r1) return 12 + r0 * 4 + c0;
  if (c0 > c1) return 24 + r0 * 3 + (c0 - 1);
  return 36 + (r0 - 1) * 4 + c0;
}

void co_decode_move(int move_id, int out[6]) {
  move_t m = decode_move(move_id);
  out[0] = m.is_place; out[1] = m.piece; out[2] = m.r0; out[3] = m.c0; out[4] = m.r1; out[5] = m.c1;
}

/* -------------------------------------------------------------------- game */
/* ref: game.h:127-135 */
typedef struct {
  uint64_t board;
  int8_t pieces[6];
  int8_t to_play;
} game_t;

static void game_init(game_t *g) {
  g->board = 0;
  for (int i = 0; i < 6; ++i) g->pieces[i] = 4;
  g->to_play = 0;
}

typedef struct { uint32_t w[3]; } mask96;
static inline int m_test(const mask96 *m, int i) { return (m->w[i >> 5] >> (i & 31)) & 1u; }
static inline void m_clear(mask96 *m, int i) { m->w[i >> 5] &= ~(1u << (i & 31)); }
static inline int m_count(const mask96 *m) {
  return __builtin_popcount(m->w[0]) + __builtin_popcount(m->w[1]) + __builtin_popcount(m->w[2]);
}

/* ref: game.cpp:141-150 */
static inline int g_board(const game_t *g, int row, int col, int k) {
  return (int)((g->board >> (row * 16 + col * 4 + k)) & 1u);
}
static inline void g_set(game_t *g, int row, int col, int k, int state) {
  uint64_t bit = 1ull << (row * 16 + col * 4 + k);
  if (state) g->board |= bit; else g->board &= ~bit;
}
/* ref: game.cpp:152-156 */
static inline int g_empty(const game_t *g, int r, int c) {
  return !(g_board(g, r, c, kBase) || g_board(g, r, c, kColumn) || g_board(g, r, c, kCapital));
}
/* ref: game.cpp:158-168 */
static int g_top(const game_t *g, int r, int c) {
  for (int p = 2; p >= 0; --p)
    if (g_board(g, r, c, p)) return p;
  return -1;
}
/* ref: game.cpp:170-180 */
static int g_bottom(const game_t *g, int r, int c) {
  for (int p = 0; p < 3; ++p)
    if (g_board(g, r, c, p)) return p;
  return 3;
}

/* ref: game.cpp:193-220 */
static int can_place(const game_t *g, const move_t *m) {
  if (g->pieces[g->to_play * 3 + m->piece] == 0) return 0;
  if (g_empty(g, m->r1, m->c1)) return 1;
  if (g_board(g, m->r1, m->c1, kFrozen)) return 0;
  if (m->piece == kBase) return 0;
  if (m->piece == kColumn)
    return !(g_board(g, m->r1, m->c1, kColumn) || g_board(g, m->r1, m->c1, kCapital));
  return !(g_board(g, m->r1, m->c1, kCapital) ||
           (g_board(g, m->r1, m->c1, kBase) && !g_board(g, m->r1, m->c1, kColumn)));
}

/* ref: game.cpp:222-232 */
static int can_move(const game_t *g, const move_t *m) {
  if (g_empty(g, m->r0, m->c0) || g_empty(g, m->r1, m->c1)) return 0;
  if (g_board(g, m->r0, m->c0, kFrozen) || g_board(g, m->r1, m->c1, kFrozen)) return 0;
  return g_bottom(g, m->r0, m->c0) - g_top(g, m->r1, m->c1) == 1;
}

/* ref: game.cpp:234-242 */
static int is_legal_move(const game_t *g, int id) {
  move_t m = decode_move(id);
  return m.is_place ? can_place(g, &m) : can_move(g, &m);
}

/* ref: game.cpp:244-247 */
static void apply_line(int line, mask96 *legal) {
  legal->w[0] &= LINE_BREAKERS[line][0];
  legal->w[1] &= LINE_BREAKERS[line][1];
  legal->w[2] &= LINE_BREAKERS[line][2];
}

/* Space{a, b, flip}: util.h:24-33 -- (row,col) = flip ? (b,a) : (a,b) */
#define SP_R(a, b, flip) ((flip) ? (b) : (a))
#define SP_C(a, b, flip) ((flip) ? (a) : (b))

/* ref: game.cpp:249-315 */
static int apply_row_col_lines(const game_t *g, mask96 *legal, int is_col) {
  for (int i = 0; i < 4; ++i) {
    int top0 = g_top(g, SP_R(i, 0, is_col), SP_C(i, 0, is_col));
    int top1 = g_top(g, SP_R(i, 1, is_col), SP_C(i, 1, is_col));
    int top2 = g_top(g, SP_R(i, 2, is_col), SP_C(i, 2, is_col));
    int top3 = g_top(g, SP_R(i, 3, is_col), SP_C(i, 3, is_col));
    if (top1 == -1 || top2 == -1) continue;
    if (top0 == top1 && top1 == top2 && top2 == top3) {
      apply_line((is_col ? CB : RB) * 12 + i * 3 + top0, legal);
      return 1;
    }
    static const int extend_coords[2] = {3, 0};
    for (int e = 0; e < 2; ++e) {
      int ec = extend_coords[e];
      if (top1 == top2 && ((ec == 3 && top0 == top1) || (ec == 0 && top2 == top3))) {
        if (is_col && ec == 0) apply_line(CD * 12 + i * 3 + top1, legal);
        else if (is_col && ec == 3) apply_line(CU * 12 + i * 3 + top1, legal);
        else if (ec == 0) apply_line(RR * 12 + i * 3 + top1, legal);
        else apply_line(RL * 12 + i * 3 + top1, legal);
        if (top1 == 2) {
          /* ref: game.cpp:280-309 -- note Space{k, ec, isCol}: the line index i
           * does not appear; every k in 0..3 of the extension coordinate is
           * treated. */
#define CELL_R(k) SP_R((k), ec, is_col)
#define CELL_C(k) SP_C((k), ec, is_col)
#define CLR(a, b) m_clear(legal, co_encode_move(CELL_R(a), CELL_C(a), CELL_R(b), CELL_C(b)))
          if (!g_board(g, CELL_R(0), CELL_C(0), kCapital)) CLR(0, 1);
          if (!g_board(g, CELL_R(1), CELL_C(1), kCapital)) { CLR(1, 0); CLR(1, 2); }
          if (!g_board(g, CELL_R(2), CELL_C(2), kCapital)) { CLR(2, 1); CLR(2, 3); }
          if (!g_board(g, CELL_R(3), CELL_C(3), kCapital)) CLR(3, 2);
#undef CLR
#undef CELL_R
#undef CELL_C
        }
        return 1;
      }
    }
  }
  return 0;
}

/* ref: game.cpp:317-360 */
static int apply_long_diag_lines(const game_t *g, mask96 *legal) {
  for (int flip = 0; flip < 2; ++flip) {
    int top0 = g_top(g, 0, flip ? 3 : 0);
    int top1 = g_top(g, 1, flip ? 2 : 1);
    int top2 = g_top(g, 2, flip ? 1 : 2);
    int top3 = g_top(g, 3, flip ? 0 : 3);
    if (top1 == -1 || top2 == -1) continue;
    if (top0 == top1 && top1 == top2 && top2 == top3) {
      apply_line(72 + (flip ? D1B : D0B) * 3 + top1, legal);
      return 1;
    }
    if (top0 == top1 && top1 == top2) {
      apply_line(72 + (flip ? D1U : D0U) * 3 + top1, legal);
      return 1;
    }
    if (top1 == top2 && top2 == top3) {
      apply_line(72 + (flip ? D1D : D0D) * 3 + top1, legal);
      return 1;
    }
  }
  return 0;
}

/* ref: game.cpp:362-391 */
static int apply_short_diag_lines(const game_t *g, mask96 *legal) {
  int top1 = g_top(g, 1, 1);
  if (top1 != -1 && top1 == g_top(g, 0, 2) && top1 == g_top(g, 2, 0)) {
    apply_line(72 + S0 * 3 + top1, legal);
    return 1;
  }
  top1 = g_top(g, 1, 2);
  if (top1 != -1 && top1 == g_top(g, 0, 1) && top1 == g_top(g, 2, 3)) {
    apply_line(72 + S1 * 3 + top1, legal);
    return 1;
  }
  top1 = g_top(g, 2, 2);
  if (top1 != -1 && top1 == g_top(g, 1, 3) && top1 == g_top(g, 3, 1)) {
    apply_line(72 + S2 * 3 + top1, legal);
    return 1;
  }
  top1 = g_top(g, 2, 1);
  if (top1 != -1 && top1 == g_top(g, 1, 0) && top1 == g_top(g, 3, 2)) {
    apply_line(72 + S3 * 3 + top1, legal);
    return 1;
  }
  return 0;
}

/* ref: game.cpp:28-43 (+ applyLines :393-405) */
static int get_legal_moves(const game_t *g, mask96 *legal) {
  legal->w[0] = legal->w[1] = legal->w[2] = 0xFFFFFFFFu;
  int is_lines = 0;
  is_lines |= apply_row_col_lines(g, legal, 0);
  is_lines |= apply_row_col_lines(g, legal, 1);
  is_lines |= apply_long_diag_lines(g, legal);
  is_lines |= apply_short_diag_lines(g, legal);
  for (int i = 0; i < CO_NUM_MOVES; ++i)
    if (m_test(legal, i) && !is_legal_move(g, i)) m_clear(legal, i);
  return is_lines;
}

/* ref: game.cpp:45-58 */
static void write_game_state(const game_t *g, float *out) {
  for (int i = 0; i < 64; ++i) out[i] = ((g->board >> i) & 1u) ? 1.0f : 0.0f;
  for (int i = 0; i < 6; ++i)
    out[64 + i] = (float)((double)(float)g->pieces[(g->to_play * 3 + i) % 6] * 0.25);
}

/* ref: game.cpp:60-96 */
static void do_move(game_t *g, int id) {
  move_t m = decode_move(id);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) g_set(g, r, c, kFrozen, 0);
  if (m.is_place) {
    --g->pieces[g->to_play * 3 + m.piece];
    g_set(g, m.r1, m.c1, m.piece, 1);
    g_set(g, m.r1, m.c1, kFrozen, 1);
  } else {
    for (int p = 0; p < 3; ++p) {
      g_set(g, m.r1, m.c1, p, g_board(g, m.r0, m.c0, p) || g_board(g, m.r1, m.c1, p));
      g_set(g, m.r0, m.c0, p, 0);
    }
    g_set(g, m.r1, m.c1, kFrozen, 1);
  }
  g->to_play = (int8_t)(1 - g->to_play);
}

static game_t make_game(uint64_t board, const int8_t pieces[6], int to_play) {
  game_t g;
  g.board = board;
  memcpy(g.pieces, pieces, 6);
  g.to_play = (int8_t)to_play;
  return g;
}

int co_legal_moves(uint64_t board, const int8_t pieces[6], int to_play, uint32_t mask_out[3]) {
  game_t g = make_game(board, pieces, to_play);
  mask96 m;
  int l = get_legal_moves(&g, &m);
  mask_out[0] = m.w[0]; mask_out[1] = m.w[1]; mask_out[2] = m.w[2];
  return l;
}

void co_do_move(uint64_t *board, int8_t pieces[6], int *to_play, int move_id) {
  game_t g = make_game(*board, pieces, *to_play);
  do_move(&g, move_id);
  *board = g.board;
  memcpy(pieces, g.pieces, 6);
  *to_play = g.to_play;
}

void co_write_game_state(uint64_t board, const int8_t pieces[6], int to_play, float out[CO_GAME_STATE_SIZE]) {
  game_t g = make_game(board, pieces, to_play);
  write_game_state(&g, out);
}

int co_terminal_result(uint64_t board, const int8_t pieces[6], int to_play) {
  game_t g = make_game(board, pieces, to_play);
  mask96 m;
  int lines = get_legal_moves(&g, &m);
  if (m_count(&m) != 0) return kResultNone;
  return lines ? kResultLoss : kResultDraw;
}

/* ----------------------------------------------------------------- mt19937 */
struct co_mt19937 {
  uint32_t mt[624];
  int idx;
};

static void mt_seed(co_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

static uint32_t mt_next(co_mt19937 *g) {
  if (g->idx >= 624) {
    uint32_t *x = g->mt;
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (x[i] & 0x80000000u) | (x[(i + 1) % 624] & 0x7fffffffu);
      x[i] = x[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

co_mt19937 *co_mt_create(uint32_t seed) {
  co_mt19937 *g = (co_mt19937 *)malloc(sizeof *g);
  mt_seed(g, seed);
  return g;
}
uint32_t co_mt_next(co_mt19937 *g) { return mt_next(g); }
void co_mt_destroy(co_mt19937 *g) { free(g); }

/* -------------------------------------------------------------------- node */
/* ref: node.h:24-187 */
typedef struct node {
  game_t game;
  struct node *parent, *next_sibling, *first_child;
  uint16_t *edges; /* move_id : 7 (low bits), probability : 9 */
  float evaluation;
  float denominator;
  int16_t visits;
  int8_t result;
  int8_t child_id;
  int8_t num_legal_moves;
  int8_t depth;
  uint8_t all_visited;
} node_t;

typedef struct {
  int64_t searches, leaf_evals, nodes_created, plies;
} counters_t;

static inline int e_move(uint16_t e) { return e & 127; }
static inline int e_prob(uint16_t e) { return e >> 7; }

/* ref: node.cpp:256-283 */
static void initialize_edges(node_t *n) {
  mask96 legal;
  int is_lines = get_legal_moves(&n->game, &legal);
  n->num_legal_moves = (int8_t)m_count(&legal);
  if (n->num_legal_moves == 0) {
    n->result = is_lines ? kResultLoss : kResultDraw;
    return;
  }
  n->edges = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)n->num_legal_moves);
  int k = 0;
  for (int i = 0; i < CO_NUM_MOVES; ++i)
    if (m_test(&legal, i)) n->edges[k++] = (uint16_t)i; /* Edge(i, 0) */
}

static node_t *node_alloc(counters_t *ctr) {
  node_t *n = (node_t *)calloc(1, sizeof *n);
  n->visits = 1;          /* node.h:164 */
  n->result = kResultNone;
  n->all_visited = 1;     /* node.h:186 */
  if (ctr) ctr->nodes_created++;
  return n;
}

/* ref: node.cpp:14-17 */
static node_t *node_new_start(counters_t *ctr) {
  node_t *n = node_alloc(ctr);
  game_init(&n->game);
  initialize_edges(n);
  return n;
}

/* ref: node.cpp:25-29 */
static node_t *node_new_from_game(const game_t *g, int depth, counters_t *ctr) {
  node_t *n = node_alloc(ctr);
  n->game = *g;
  n->depth = (int8_t)depth;
  initialize_edges(n);
  return n;
}

/* ref: node.cpp:31-39 */
static node_t *node_new_child(const game_t *g, node_t *parent, node_t *next_sibling, int move_id, int depth,
                              counters_t *ctr) {
  node_t *n = node_alloc(ctr);
  n->game = *g;
  n->parent = parent;
  n->next_sibling = next_sibling;
  n->child_id = (int8_t)move_id;
  n->depth = (int8_t)depth;
  do_move(&n->game, move_id);
  initialize_edges(n);
  return n;
}

/* ref: node.cpp:19-23 (recursive delete of siblings and children) */
static void node_delete(node_t *n) {
  while (n) {
    node_t *sib = n->next_sibling;
    node_delete(n->first_child);
    free(n->edges);
    free(n);
    n = sib;
  }
}

static inline int n_terminal(const node_t *n) { return n->result == kResultLoss || n->result == kResultDraw; }
static inline int n_known(const node_t *n) { return n->result != kResultNone; }
static inline int n_won(const node_t *n) { return n->result == kDeducedWin; }
static inline int n_lost(const node_t *n) { return n->result == kResultLoss || n->result == kDeducedLoss; }
static inline int n_drawn(const node_t *n) { return n->result == kResultDraw || n->result == kDeducedDraw; }
/* ref: node.cpp:90-94 */
static inline float n_probability(const node_t *n, int i) { return (float)e_prob(n->edges[i]) * n->denominator; }
static inline int n_move_id(const node_t *n, int i) { return e_move(n->edges[i]); }

/* ----------------------------------------------------------------- TrainMC */
/* ref: trainmc.h:145-188 */
typedef struct {
  node_t *root, *cur;
  int searches_done, max_searches, searches_per_eval;
  float c_puct, epsilon;
  node_t **searched;
  int n_searched;
  float *to_eval;
  int testing;
  co_mt19937 *generator;
  counters_t *ctr;
} trainmc_t;

static void mc_init(trainmc_t *mc, co_mt19937 *gen, float *to_eval, int max_searches, int spe, float c_puct,
                    float epsilon, int testing, counters_t *ctr) {
  memset(mc, 0, sizeof *mc);
  mc->max_searches = max_searches;
  mc->searches_per_eval = spe;
  mc->c_puct = c_puct;
  mc->epsilon = epsilon;
  mc->to_eval = to_eval;
  mc->testing = testing;
  mc->generator = gen;
  mc->searched = (node_t **)malloc(sizeof(node_t *) * (size_t)(spe > 0 ? spe : 1));
  mc->ctr = ctr;
}

static void mc_free(trainmc_t *mc) {
  node_delete(mc->root);
  mc->root = mc->cur = NULL;
  free(mc->searched);
  mc->searched = NULL;
}

/* ref: trainmc.cpp:78-83 */
static void mc_null_root(trainmc_t *mc) {
  node_delete(mc->root);
  mc->root = NULL;
  mc->cur = NULL;
}

static void mc_request(trainmc_t *mc, node_t *n) {
  write_game_state(&n->game, mc->to_eval + mc->n_searched * CO_GAME_STATE_SIZE);
  mc->searched[mc->n_searched++] = n;
}

/* ref: trainmc.cpp:206-210 */
static void mc_create_root(trainmc_t *mc, const game_t *g, int depth) {
  mc->root = node_new_from_game(g, depth, mc->ctr);
  mc->cur = mc->root;
}

/* ref: trainmc.cpp:212-234 */
static void get_filtered_probs(trainmc_t *mc, const float *probs, float *filtered) {
  node_t *cur = mc->cur;
  int edge_index = 0;
  float sum = 0.0f;
  for (int j = 0; j < CO_NUM_MOVES; ++j) {
    if (edge_index < cur->num_legal_moves && n_move_id(cur, edge_index) == j) {
      filtered[edge_index] = probs[j];
      sum += filtered[edge_index];
      ++edge_index;
      if (edge_index == cur->num_legal_moves) break;
    }
  }
  /* 1.0 / sum * (1 - epsilon_): double / float, times float(1 - eps) */
  float one_minus = (float)1 - mc->epsilon;
  float scalar = (float)(1.0 / (double)sum * (double)one_minus);
  for (int j = 0; j < cur->num_legal_moves; ++j) filtered[j] *= scalar;
}

/* ref: trainmc.cpp:236-246 */
static void generate_dirichlet(trainmc_t *mc, float *dirichlet) {
  node_t *cur = mc->cur;
  float sum = 0.0f;
  for (int i = 0; i < cur->num_legal_moves; ++i) {
    dirichlet[i] = gamma_sample(mt_next(mc->generator) % CO_NUM_GAMMA);
    sum += dirichlet[i];
  }
  float scalar = (float)(1.0 / (double)sum * (double)mc->epsilon);
  for (int i = 0; i < cur->num_legal_moves; ++i) dirichlet[i] *= scalar;
}

/* ref: trainmc.cpp:248-267 */
static void set_probs(trainmc_t *mc, const float *filtered, const float *dirichlet) {
  node_t *cur = mc->cur;
  float weighted[CO_NUM_MOVES];
  float max_prob = 0.0f;
  for (int j = 0; j < cur->num_legal_moves; ++j) {
    weighted[j] = filtered[j] + dirichlet[j];
    max_prob = weighted[j] > max_prob ? weighted[j] : max_prob; /* std::max(w, max) */
  }
  float denom = 511.0f / max_prob;
  int32_t final_sum = 0;
  for (int j = 0; j < cur->num_legal_moves; ++j) {
    float x = weighted[j] * denom;
    long r = lround((double)x);
    int32_t prob = (int32_t)r;
    if (prob < 1) prob = 1;
    cur->edges[j] = (uint16_t)((cur->edges[j] & 127) | ((prob & 511) << 7));
    final_sum += prob;
  }
  cur->denominator = (float)(1.0 / (double)(float)final_sum);
}

/* ref: trainmc.cpp:269-296 */
static void receive_eval(trainmc_t *mc, const float *eval, const float *probs) {
  for (int i = 0; i < mc->n_searched; ++i) {
    mc->cur = mc->searched[i];
    float filtered[CO_NUM_MOVES];
    float dirichlet[CO_NUM_MOVES];
    get_filtered_probs(mc, probs + CO_NUM_MOVES * i, filtered);
    generate_dirichlet(mc, dirichlet);
    set_probs(mc, filtered, dirichlet);
    float cur_eval = eval[i];
    while (mc->cur->parent != NULL) {
      mc->cur->evaluation += (float)((double)cur_eval - 1.0);
      mc->cur->all_visited = 0;
      cur_eval = (float)((double)cur_eval * -1.0);
      mc->cur = mc->cur->parent;
    }
    mc->cur->evaluation += (float)((double)cur_eval - 1.0);
    if (mc->ctr) mc->ctr->leaf_evals++;
  }
  mc->root->all_visited = 0;
  mc->n_searched = 0;
}

/* ref: trainmc.cpp:298-308 -- max_prob is an int32_t in the reference */
static int choose_high_prob_move(const trainmc_t *mc) {
  int32_t max_prob = 0;
  int choice = 0;
  for (int i = 0; i < mc->root->num_legal_moves; ++i) {
    if (n_probability(mc->root, i) > (float)max_prob) {
      max_prob = (int32_t)n_probability(mc->root, i);
      choice = n_move_id(mc->root, i);
    }
  }
  return choice;
}

/* ref: trainmc.cpp:475-495 */
static void move_down(trainmc_t *mc, node_t *prev) {
  node_t *new_root;
  if (prev == NULL) {
    new_root = mc->root->first_child;
    mc->root->first_child = new_root->next_sibling;
  } else {
    new_root = prev->next_sibling;
    prev->next_sibling = new_root->next_sibling;
  }
  new_root->next_sibling = NULL;
  new_root->parent = NULL;
  node_delete(mc->root);
  mc->root = new_root;
  mc->cur = mc->root;
  mc->searches_done = 0;
}

/* ref: trainmc.cpp:310-335 */
static int choose_move_won(trainmc_t *mc, float *prob_sample) {
  node_t *cur = mc->root->first_child, *prev = NULL, *best_prev = NULL;
  int choice = 0;
  while (cur != NULL) {
    if (n_lost(cur)) {
      choice = cur->child_id;
      best_prev = prev;
      break;
    }
    prev = cur;
    cur = cur->next_sibling;
  }
  if (prob_sample) prob_sample[choice] = 1.0f;
  move_down(mc, best_prev);
  return choice;
}

/* ref: trainmc.cpp:337-361 */
static int choose_move_lost_drawn(trainmc_t *mc, float *prob_sample) {
  int max_visits = 0;
  node_t *cur = mc->root->first_child, *prev = NULL, *best_prev = NULL;
  int choice = 0;
  while (cur != NULL) {
    if (cur->visits > max_visits && (n_lost(mc->root) || !n_won(cur))) {
      choice = cur->child_id;
      best_prev = prev;
      max_visits = cur->visits;
    }
    prev = cur;
    cur = cur->next_sibling;
  }
  if (prob_sample) prob_sample[choice] = 1.0f;
  move_down(mc, best_prev);
  return choice;
}

/* fresh tree after "1 search or all losing moves" (trainmc.cpp:398-407, 455-464) */
static void reset_tree_to_child(trainmc_t *mc, int choice) {
  node_t *new_root = node_new_child(&mc->root->game, NULL, NULL, choice, mc->root->depth + 1, mc->ctr);
  node_delete(mc->root);
  mc->root = new_root;
  mc->cur = mc->root;
  mc->searches_done = 0;
}

/* ref: trainmc.cpp:363-427 */
static int choose_move_opening(trainmc_t *mc, float *prob_sample) {
  node_t *best_prev = NULL;
  int choice = choose_high_prob_move(mc);
  int32_t visits = 0;
  node_t *cur = mc->root->first_child;
  while (cur != NULL) {
    if (!n_won(cur)) visits += cur->visits;
    cur = cur->next_sibling;
  }
  float denominator = (float)(1.0 / (double)(float)visits);
  cur = mc->root->first_child;
  if (prob_sample != NULL) {
    while (cur != NULL) {
      if (!n_won(cur)) prob_sample[cur->child_id] = (float)cur->visits * denominator;
      best_prev = cur;
      cur = cur->next_sibling;
    }
  }
  if (visits == 0) {
    prob_sample[choice] = 1.0f;
    reset_tree_to_child(mc, choice);
    return choice;
  }
  int32_t target = (int32_t)(mt_next(mc->generator) % (uint32_t)visits);
  int32_t total = 0;
  cur = mc->root->first_child;
  best_prev = NULL;
  while (cur != NULL) {
    if (!n_won(cur)) {
      total += cur->visits;
      if (total > target) {
        choice = cur->child_id;
        break;
      }
    }
    best_prev = cur;
    cur = cur->next_sibling;
  }
  move_down(mc, best_prev);
  return choice;
}

/* ref: trainmc.cpp:429-473 */
static int choose_move_normal(trainmc_t *mc, float *prob_sample) {
  int max_visits = 0;
  float max_eval = 0.0f;
  node_t *cur = mc->root->first_child, *prev = NULL, *best_prev = NULL;
  int choice = choose_high_prob_move(mc);
  while (cur != NULL) {
    if (!n_won(cur)) {
      float eval = cur->evaluation;
      if (cur->result == kResultDraw || cur->result == kDeducedDraw) eval = 0.0f;
      if (cur->visits > max_visits || (cur->visits == max_visits && eval > max_eval)) {
        choice = cur->child_id;
        best_prev = prev;
        max_visits = cur->visits;
        max_eval = eval;
      }
    }
    prev = cur;
    cur = cur->next_sibling;
  }
  if (prob_sample != NULL) prob_sample[choice] = 1.0f;
  if (max_visits == 0) {
    reset_tree_to_child(mc, choice);
    return choice;
  }
  move_down(mc, best_prev);
  return choice;
}

/* ref: trainmc.cpp:110-137 */
static int mc_choose_move(trainmc_t *mc, float *game_state, float *prob_sample) {
  if (!mc->testing) {
    write_game_state(&mc->root->game, game_state);
    memset(prob_sample, 0, CO_NUM_MOVES * sizeof(float));
  }
  if (n_won(mc->root)) return choose_move_won(mc, prob_sample);
  if (n_lost(mc->root) || n_drawn(mc->root)) return choose_move_lost_drawn(mc, prob_sample);
  if (mc->root->depth < 6 && !mc->testing) return choose_move_opening(mc, prob_sample);
  return choose_move_normal(mc, prob_sample);
}

/* ref: trainmc.cpp:497-538 */
static void propagate_terminal(trainmc_t *mc) {
  node_t *cur = mc->cur;
  while (cur != mc->root) {
    if (n_lost(cur)) {
      cur = cur->parent;
      cur->result = kDeducedWin;
    } else {
      cur = cur->parent;
      node_t *cur_child = cur->first_child;
      int has_draw = 0;
      int edge_index = 0;
      while (cur_child != NULL) {
        if (cur_child->child_id != n_move_id(cur, edge_index) || !n_known(cur_child)) return;
        if (n_drawn(cur)) has_draw = 1; /* sic: tests the parent (trainmc.cpp:518) */
        cur_child = cur_child->next_sibling;
        ++edge_index;
      }
      if (edge_index < cur->num_legal_moves) return;
      cur->result = has_draw ? kDeducedDraw : kDeducedLoss;
    }
  }
}

typedef struct {
  int type; /* 0 visited, 1 new, 2 none */
  int choice;
  node_t *node;
} choose_next_t;

/* ref: trainmc.cpp:540-600 */
static choose_next_t choose_next(trainmc_t *mc) {
  node_t *cur = mc->cur;
  float max_eval = -INFINITY;
  int choice = 0;
  node_t *cur_child = cur->first_child;
  int edge_index = 0;
  node_t *prev = NULL, *best_prev = NULL;
  /* c_puct_ * sqrt(static_cast<float>(visits)): ::sqrt(double) is selected */
  float v_sqrt = (float)((double)mc->c_puct * sqrt((double)(float)cur->visits));
  while (cur_child != NULL || edge_index < cur->num_legal_moves) {
    float u = -INFINITY;
    if (cur_child != NULL && cur_child->child_id == n_move_id(cur, edge_index)) {
      if ((!n_known(cur_child) || n_drawn(cur_child)) && !cur_child->all_visited) {
        if (n_drawn(cur_child)) {
          u = n_probability(cur, edge_index) * v_sqrt;
        } else {
          float pv = n_probability(cur, edge_index) * v_sqrt;
          double a = -1.0 * (double)cur_child->evaluation / (double)(float)cur_child->visits;
          double b = (double)pv / ((double)(float)cur_child->visits + 1.0);
          u = (float)(a + b);
        }
      }
      prev = cur_child;
      cur_child = cur_child->next_sibling;
    } else {
      u = n_probability(cur, edge_index) * v_sqrt;
    }
    if (u > max_eval) {
      best_prev = prev;
      max_eval = u;
      choice = n_move_id(cur, edge_index);
    }
    ++edge_index;
  }
  choose_next_t out;
  if (max_eval == -INFINITY) {
    out.type = 2; out.choice = -1; out.node = NULL;
    return out;
  }
  if (best_prev == NULL || best_prev->child_id != choice) {
    out.type = 1; out.choice = choice; out.node = best_prev;
    return out;
  }
  out.type = 0; out.choice = choice; out.node = best_prev;
  return out;
}

/* ref: trainmc.cpp:602-696 */
static void mc_search(trainmc_t *mc) {
  mc->cur = mc->root;
  ++mc->searches_done;
  if (mc->ctr) mc->ctr->searches++;
  while (!n_terminal(mc->cur)) {
    choose_next_t res = choose_next(mc);
    ++mc->cur->visits;
    mc->cur->evaluation += 1.0f;
    if (res.type == 2) {
      mc->cur->all_visited = 1;
      while (mc->cur->parent != NULL) {
        --mc->cur->visits;
        mc->cur->evaluation -= 1.0f;
        mc->cur = mc->cur->parent;
      }
      --mc->cur->visits;
      mc->cur->evaluation -= 1.0f;
      --mc->searches_done;
      if (mc->ctr) mc->ctr->searches--;
      return;
    }
    if (res.type == 1 && res.node == NULL) {
      mc->cur->first_child =
          node_new_child(&mc->cur->game, mc->cur, mc->cur->first_child, res.choice, mc->cur->depth + 1, mc->ctr);
      mc->cur = mc->cur->first_child;
      break;
    }
    if (res.type == 1) {
      res.node->next_sibling =
          node_new_child(&mc->cur->game, mc->cur, res.node->next_sibling, res.choice, mc->cur->depth + 1, mc->ctr);
      mc->cur = res.node->next_sibling;
      break;
    }
    mc->cur = res.node;
  }
  if (n_terminal(mc->cur)) {
    propagate_terminal(mc);
    float cur_eval = -1.0f;
    if (n_drawn(mc->cur)) cur_eval = 0.0f;
    mc->cur->evaluation = cur_eval;
    while (mc->cur->parent != NULL) {
      mc->cur = mc->cur->parent;
      mc->cur->evaluation += (float)((double)cur_eval - 1.0);
      cur_eval = (float)((double)cur_eval * -1.0);
    }
  } else {
    mc->cur->evaluation = 1.0f;
    mc_request(mc, mc->cur);
  }
  mc->cur = mc->root;
}

/* ref: trainmc.cpp:139-178 */
static int mc_do_iteration(trainmc_t *mc, const float *eval, const float *probs) {
  if (mc->root == NULL) {
    mc->root = node_new_start(mc->ctr);
    mc->cur = mc->root;
    mc->searches_done = 1;
    mc_request(mc, mc->cur);
    return 0;
  }
  if (mc->searches_done == 0 && mc->root->visits == 1 && mc->root->all_visited) {
    mc->searches_done = 1;
    mc_request(mc, mc->cur);
    return 0;
  }
  if (mc->n_searched > 0) receive_eval(mc, eval, probs);
  while (mc->n_searched < mc->searches_per_eval && mc->searches_done < mc->max_searches &&
         !n_known(mc->root) && !mc->root->all_visited) {
    mc_search(mc);
  }
  return (mc->searches_done == mc->max_searches || n_known(mc->root)) && mc->n_searched == 0;
}

/* ref: trainmc.cpp:180-204 */
static int mc_receive_opponent_move(trainmc_t *mc, int move_choice, const game_t *game, int depth) {
  node_t *cur = mc->root->first_child, *prev = NULL;
  while (cur != NULL) {
    if (cur->child_id == move_choice) {
      move_down(mc, prev);
      return 0;
    }
    prev = cur;
    cur = cur->next_sibling;
  }
  node_delete(mc->root);
  mc->root = NULL;
  mc_create_root(mc, game, depth);
  mc_request(mc, mc->cur);
  mc->searches_done = 1;
  return 1;
}

/* -------------------------------------------------------------- SelfPlayer */
typedef struct {
  float game_state[CO_GAME_STATE_SIZE];
  float probabilities[CO_NUM_MOVES];
} sample_t;

/* ref: selfplayer.h:28-112 */
typedef struct {
  co_mt19937 generator;
  float *to_eval;
  trainmc_t players[2];
  int to_play;
  sample_t *samples;
  int n_samples, cap_samples;
  int8_t result;
  int mate_turn;
  int parity;
  int testing;
  counters_t ctr;
  /* trace */
  int trace_on;
  int32_t *trace;
  int n_trace, cap_trace;
  /* per-game text log (selfplayer.h:99-102): the file, and whether writeEval has switched the stream to
   * std::fixed << std::setprecision(6) yet -- the manipulators stick to the stream, so every float written
   * afterwards, also the ones printed with plain operator<<, comes out in that format */
  FILE *log;
  int log_fixed;
} selfplayer_t;

static void sp_init(selfplayer_t *sp, uint32_t seed, int max_searches, int spe, float c_puct, float epsilon,
                    int testing, int parity) {
  memset(sp, 0, sizeof *sp);
  mt_seed(&sp->generator, seed);
  sp->to_eval = (float *)calloc((size_t)CO_GAME_STATE_SIZE * (size_t)max_searches, sizeof(float));
  for (int p = 0; p < 2; ++p)
    mc_init(&sp->players[p], &sp->generator, sp->to_eval, max_searches, spe, c_puct, epsilon, testing, &sp->ctr);
  sp->parity = parity;
  sp->testing = testing;
  sp->result = kResultNone;
}

static void sp_free(selfplayer_t *sp) {
  if (sp->log) fclose(sp->log);
  mc_free(&sp->players[0]);
  mc_free(&sp->players[1]);
  free(sp->to_eval);
  free(sp->samples);
  free(sp->trace);
}

static void trace_push(selfplayer_t *sp, int32_t v) {
  if (sp->n_trace == sp->cap_trace) {
    sp->cap_trace = sp->cap_trace ? sp->cap_trace * 2 : 256;
    sp->trace = (int32_t *)realloc(sp->trace, sizeof(int32_t) * (size_t)sp->cap_trace);
  }
  sp->trace[sp->n_trace++] = v;
}

static int32_t fbits(float f) {
  int32_t b;
  memcpy(&b, &f, 4);
  return b;
}

static void trace_root(selfplayer_t *sp) {
  const node_t *r = sp->players[sp->to_play].root;
  trace_push(sp, sp->to_play);
  trace_push(sp, r->depth);
  trace_push(sp, r->visits);
  trace_push(sp, r->result);
  trace_push(sp, fbits(r->evaluation));
  int n = 0;
  for (const node_t *c = r->first_child; c; c = c->next_sibling) ++n;
  trace_push(sp, n);
  for (const node_t *c = r->first_child; c; c = c->next_sibling) {
    trace_push(sp, c->child_id);
    trace_push(sp, c->visits);
    trace_push(sp, fbits(c->evaluation));
    trace_push(sp, c->result);
    trace_push(sp, c->all_visited);
  }
}

/* ---- per-game text logs: selfplayer.cpp:124-204, node.cpp:197-254, game.cpp:98-139, move.cpp:56-78, util.cpp:5-25.
 * A C++ ostream prints a float with operator<< as printf's %g would (precision 6) until a manipulator changes
 * the stream; writeEval's std::fixed << std::setprecision(6) does, for good. */
static void lg_float(FILE *f, int *fixed, float v) { fprintf(f, *fixed ? "%.6f" : "%g", (double)v); }

static const char *str_result(int r) { /* util.cpp:5-25 */
  switch (r) {
    case kResultLoss: return "L";
    case kResultDraw: return "D";
    case kResultWin: return "W";
    case kDeducedLoss: return "DL";
    case kDeducedDraw: return "DD";
    case kDeducedWin: return "DW";
    default: return "N";
  }
}

static void log_move(FILE *f, int id) { /* move.cpp:56-78 */
  move_t m = decode_move(id);
  if (m.is_place) {
    fprintf(f, "%c%c%d", m.piece == 0 ? 'B' : m.piece == 1 ? 'C' : 'A', 'a' + m.c1, 4 - m.r1);
  } else {
    char d = m.c1 < m.c0 ? 'L' : m.c1 > m.c0 ? 'R' : m.r1 < m.r0 ? 'U' : 'D';
    fprintf(f, "%c%d%c", 'a' + m.c0, 4 - m.r0, d);
  }
}

static void log_game(FILE *f, const game_t *g) { /* game.cpp:98-139 */
  for (int row = 0; row < 4; ++row) {
    for (int col = 0; col < 4; ++col) {
      fputc(g_board(g, row, col, 0) ? 'B' : ' ', f);
      fputc(g_board(g, row, col, 1) ? 'C' : ' ', f);
      fputc(g_board(g, row, col, 2) ? 'A' : ' ', f);
      fputc(g_board(g, row, col, 3) ? '#' : ' ', f);
      if (col < 3) fputc('|', f);
    }
    if (row < 3) fputs("\n-------------------\n", f);
  }
  fputc('\n', f);
  for (int player = 0; player < 2; ++player)
    fprintf(f, "Player %d: B: %d C: %d A: %d\n", player + 1, (int)g->pieces[player * 3 + 0], (int)g->pieces[player * 3 + 1],
            (int)g->pieces[player * 3 + 2]);
  fprintf(f, "Player %d to play", g->to_play + 1);
}

/* selfplayer.cpp:124-134 = match.cpp:78-88 */
static void log_eval(FILE *f, int *fixed, const node_t *n) {
  if (n->result != kResultNone) {
    fputs(str_result(n->result), f);
    return;
  }
  *fixed = 1;
  lg_float(f, fixed, n->evaluation / (float)n->visits);
}

/* node.cpp:197-240: the line of most-visited children (a lost child ends the choice at once) */
static void log_main_line(FILE *f, int *fixed, const node_t *n) {
  const node_t *cur = n->first_child, *best = NULL;
  int max_visits = 0, edge_index = 0;
  float max_eval = 0.0f, prob = 0.0f;
  while (cur != NULL) {
    if (n_move_id(n, edge_index) == cur->child_id) {
      if (cur->result == kDeducedLoss || cur->result == kResultLoss) {
        best = cur;
        max_visits = cur->visits;
        prob = n_probability(n, edge_index);
        break;
      }
      if (cur->visits > max_visits || (cur->visits == max_visits && cur->evaluation > max_eval)) {
        best = cur;
        max_visits = cur->visits;
        max_eval = cur->evaluation;
        prob = n_probability(n, edge_index);
      }
      cur = cur->next_sibling;
    }
    ++edge_index;
  }
  if (best != NULL) {
    fprintf(f, "%d. ", (int)best->depth);
    log_move(f, best->child_id);
    fprintf(f, " V: %d E: ", max_visits);
    if (best->result != kResultNone) fputs(str_result(best->result), f);
    else lg_float(f, fixed, max_eval / (float)max_visits);
    fputs(" p: ", f);
    lg_float(f, fixed, prob);
    fputc('\t', f);
    log_main_line(f, fixed, best);
  }
}

typedef struct {
  int visits;
  float evaluation, probability;
  int move;
  const node_t *node;
} log_move_data;

static int log_move_cmp(const void *pa, const void *pb) { /* selfplayer.cpp:164-173 = match.cpp:121-131 */
  const log_move_data *a = (const log_move_data *)pa, *b = (const log_move_data *)pb;
  if (a->visits != b->visits) return a->visits > b->visits ? -1 : 1;
  if (a->evaluation != b->evaluation) return a->evaluation > b->evaluation ? -1 : 1;
  if (a->probability != b->probability) return a->probability > b->probability ? -1 : 1;
  return a->move < b->move ? -1 : a->move > b->move ? 1 : 0;
}

/* selfplayer.cpp:136-183 = match.cpp:90-139 */
static void log_moves(FILE *f, int *fixed, const node_t *root) {
  fputs("LEGAL MOVES:\n", f);
  log_main_line(f, fixed, root);
  fputc('\n', f);
  log_move_data moves[CO_NUM_MOVES];
  int n = 0, edge_index = 0;
  for (const node_t *cur = root->first_child; cur != NULL;) {
    if (cur->child_id == n_move_id(root, edge_index)) {
      moves[n].visits = cur->visits;
      moves[n].evaluation = cur->evaluation / (float)cur->visits;
      moves[n].probability = n_probability(root, edge_index);
      moves[n].move = cur->child_id;
      moves[n].node = cur;
      ++n;
      cur = cur->next_sibling;
    }
    ++edge_index;
  }
  qsort(moves, (size_t)n, sizeof moves[0], log_move_cmp); /* (a total order: std::sort gives the same sequence) */
  for (int i = 1; i < n; ++i) { /* the first one is in the main line already */
    log_move(f, moves[i].move);
    fprintf(f, " V: %d E: ", moves[i].visits);
    log_eval(f, fixed, moves[i].node);
    fputs(" P: ", f);
    lg_float(f, fixed, moves[i].probability);
    fputc('\t', f);
  }
  fputc('\n', f);
}

/* selfplayer.cpp:185-197 = match.cpp:141-153 */
static void log_pre_move_of(FILE *f, int *fixed, const node_t *root, int to_play) {
  fprintf(f, "TURN %d\nPLAYER %d TO PLAY\nVISITS: %d\n", (int)root->depth, to_play + 1, (int)root->visits);
  fputs("POSITION EVALUATION: ", f);
  log_eval(f, fixed, root);
  fputc('\n', f);
  log_moves(f, fixed, root);
}

/* selfplayer.cpp:199-204 = match.cpp:155-159 */
static void log_move_choice_of(FILE *f, int choice, const game_t *after) {
  fputs("CHOSE MOVE ", f);
  log_move(f, choice);
  fputs("\nNEW POSITION:\n", f);
  log_game(f, after);
  fputs("\n\n", f);
}

static void log_pre_move(selfplayer_t *sp) { log_pre_move_of(sp->log, &sp->log_fixed, sp->players[sp->to_play].root, sp->to_play); }
static void log_move_choice(selfplayer_t *sp, int choice) { log_move_choice_of(sp->log, choice, &sp->players[sp->to_play].root->game); }

/* ref: selfplayer.cpp:206-232 */
static void sp_end_game(selfplayer_t *sp) {
  if (sp->players[sp->to_play].root->result == kResultDraw) sp->result = kResultDraw;
  else if (sp->to_play == 1) sp->result = kResultLoss;
  else sp->result = kResultWin;
  if (sp->log) {
    if (sp->result == kResultDraw) fputs("GAME IS DRAWN.\n", sp->log);
    else fprintf(sp->log, "PLAYER %d WON!\n", sp->to_play + 1);
    fclose(sp->log);
    sp->log = NULL;
  }
  mc_null_root(&sp->players[0]);
  mc_null_root(&sp->players[1]);
  free(sp->to_eval);
  sp->to_eval = NULL;
}

/* ref: selfplayer.cpp:234-244 */
static int sp_choose_move(selfplayer_t *sp) {
  int choice;
  if (sp->trace_on) trace_root(sp);
  if (!sp->testing) {
    if (sp->n_samples == sp->cap_samples) {
      sp->cap_samples = sp->cap_samples ? sp->cap_samples * 2 : 32;
      sp->samples = (sample_t *)realloc(sp->samples, sizeof(sample_t) * (size_t)sp->cap_samples);
    }
    sample_t *s = &sp->samples[sp->n_samples];
    choice = mc_choose_move(&sp->players[sp->to_play], s->game_state, s->probabilities);
    sp->n_samples++;
  } else {
    choice = mc_choose_move(&sp->players[sp->to_play], NULL, NULL);
  }
  if (sp->trace_on) trace_push(sp, choice);
  sp->ctr.plies++;
  return choice;
}

/* ref: selfplayer.cpp:246-291 */
static int sp_choose_move_and_continue(selfplayer_t *sp) {
  int need_eval = 0;
  while (!need_eval) {
    if (sp->log) log_pre_move(sp);
    if (n_known(sp->players[sp->to_play].root) && sp->mate_turn == 0) sp->mate_turn = sp->n_samples + 1;
    int choice = sp_choose_move(sp);
    if (sp->log) log_move_choice(sp, choice);
    if (n_terminal(sp->players[sp->to_play].root)) {
      sp_end_game(sp);
      return 1;
    }
    sp->to_play = 1 - sp->to_play;
    trainmc_t *me = &sp->players[sp->to_play];
    trainmc_t *opp = &sp->players[1 - sp->to_play];
    if (me->root == NULL) {
      mc_create_root(me, &opp->root->game, opp->root->depth);
      return mc_do_iteration(me, NULL, NULL);
    }
    need_eval = mc_receive_opponent_move(me, choice, &opp->root->game, opp->root->depth);
    if (!need_eval) need_eval = !mc_do_iteration(me, NULL, NULL);
  }
  return 0;
}

/* ref: selfplayer.cpp:115-122 */
static int sp_do_iteration(selfplayer_t *sp, const float *eval, const float *probs) {
  int done = mc_do_iteration(&sp->players[sp->to_play], eval, probs);
  if (done) return sp_choose_move_and_continue(sp);
  return 0;
}

static inline int sp_num_requests(const selfplayer_t *sp) { return sp->players[sp->to_play].n_searched; }

/* ref: selfplayer.cpp:57-64 */
static float sp_score(const selfplayer_t *sp) {
  if (sp->result == kResultLoss) return 0.0f;
  if (sp->result == kResultWin) return 1.0f;
  return 0.5f;
}

/* ref: selfplayer.cpp:66-71 */
static int sp_mate_length(const selfplayer_t *sp) {
  if (sp->mate_turn == 0) return 0;
  return sp->n_samples - sp->mate_turn + 1;
}

/* ref: selfplayer.cpp:79-113 */
static void sp_write_samples(const selfplayer_t *sp, float *game_states, float *eval_samples, float *prob_samples) {
  float evaluation = 1.0f;
  if (sp->result == kResultDraw) evaluation = 0.0f;
  for (int i = sp->n_samples - 1; i >= 0; --i) {
    for (int k = 0; k < CO_NUM_SYMMETRIES; ++k) {
      float *gs = game_states + (size_t)i * CO_GAME_STATE_SIZE * CO_NUM_SYMMETRIES + (size_t)k * CO_GAME_STATE_SIZE;
      for (int j = 0; j < 64; ++j) gs[j] = sp->samples[i].game_state[SPACE_SYM[k][j / 4] * 4 + j % 4];
      for (int j = 64; j < CO_GAME_STATE_SIZE; ++j) gs[j] = sp->samples[i].game_state[j];
      eval_samples[i * CO_NUM_SYMMETRIES + k] = evaluation;
      float *ps = prob_samples + (size_t)i * CO_NUM_MOVES * CO_NUM_SYMMETRIES + (size_t)k * CO_NUM_MOVES;
      for (int j = 0; j < CO_NUM_MOVES; ++j) ps[j] = sp->samples[i].probabilities[MOVE_SYM[k][j]];
    }
    evaluation = (float)((double)evaluation * -1.0);
  }
}

/* ----------------------------------------------------------------- Trainer */
struct co_trainer {
  selfplayer_t *games;
  uint8_t *is_done;
  int num_games;
  int max_searches, searches_per_eval, num_threads;
  int searches_done;
  int stagger;
  /* a SLICE of a larger Trainer (test convenience, no reference counterpart): games
   * [game_base, game_base + num_games) of a Trainer of total_games games -- seeds, parity, colour in
   * score() and the stagger rule follow the GLOBAL game index, so the slice plays exactly the games
   * the full Trainer would (games are independent, trainer.cpp:243-255) */
  int game_base, total_games;
  co_mt19937 generator;
};

/* ref: trainer.cpp:18-37, 238-256 */
co_trainer *co_trainer_create(int num_games, int seed, int max_searches, int searches_per_eval, float c_puct,
                              float epsilon, int num_threads, int testing) {
  return co_trainer_create_slice(num_games, 0, num_games, seed, max_searches, searches_per_eval, c_puct, epsilon,
                                 num_threads, testing);
}

co_trainer *co_trainer_create_slice(int total_games, int first, int num_games, int seed, int max_searches,
                                    int searches_per_eval, float c_puct, float epsilon, int num_threads, int testing) {
  if (num_games <= 0 || max_searches <= 0 || searches_per_eval <= 0 || first < 0 || first + num_games > total_games)
    return NULL;
  co_trainer *t = (co_trainer *)calloc(1, sizeof *t);
  t->num_games = num_games;
  t->game_base = first;
  t->total_games = total_games;
  t->max_searches = max_searches;
  t->searches_per_eval = searches_per_eval;
  t->num_threads = num_threads > 0 ? num_threads : 1;
  t->stagger = 1;
  mt_seed(&t->generator, (uint32_t)seed);
  t->games = (selfplayer_t *)calloc((size_t)num_games, sizeof(selfplayer_t));
  t->is_done = (uint8_t *)calloc((size_t)num_games, 1);
  for (int i = 0; i < first; ++i) (void)mt_next(&t->generator); /* the seeds of the games before the slice */
  for (int i = 0; i < num_games; ++i)
    sp_init(&t->games[i], mt_next(&t->generator), max_searches, searches_per_eval, c_puct, epsilon, testing,
            (first + i) % 2);
  return t;
}

/* ref: trainer.cpp:243-250: the first num_logged games write `<log_folder>/game_<i>.txt` (a file that cannot be
 * opened is no error there either: an unopened ofstream swallows its output).  Call before the first iteration;
 * returns the number of files opened. */
int co_trainer_set_logging(co_trainer *t, const char *log_folder, int num_logged) {
  int opened = 0;
  if (t->searches_done > 0 || num_logged < 0 || num_logged > t->num_games) return -1;
  for (int i = 0; i < num_logged; ++i) {
    char path[4096];
    snprintf(path, sizeof path, "%s/game_%d.txt", log_folder, t->game_base + i);
    t->games[i].log = fopen(path, "w");
    t->games[i].log_fixed = 0;
    opened += t->games[i].log != NULL;
  }
  return opened;
}

void co_trainer_destroy(co_trainer *t) {
  if (!t) return;
  for (int i = 0; i < t->num_games; ++i) sp_free(&t->games[i]);
  free(t->games);
  free(t->is_done);
  free(t);
}

void co_trainer_set_stagger(co_trainer *t, int on) { t->stagger = on; }

static inline int active_for(const co_trainer *t, int i, int to_play) {
  return t->games[i].to_play == (to_play + t->games[i].parity) % 2;
}

/* ref: trainer.cpp:39-49 */
int co_trainer_num_requests(const co_trainer *t, int to_play) {
  int n = 0;
  for (int i = 0; i < t->num_games; ++i)
    if (!t->is_done[i] && ((to_play != 0 && to_play != 1) || active_for(t, i, to_play))) n += sp_num_requests(&t->games[i]);
  return n;
}

/* ref: trainer.cpp:51-57 */
int co_trainer_num_samples(const co_trainer *t) {
  int n = 0;
  for (int i = 0; i < t->num_games; ++i) n += t->games[i].n_samples;
  return n;
}

/* ref: trainer.cpp:59-68 */
float co_trainer_score(const co_trainer *t) {
  float score = 0;
  const int b = t->game_base & 1; /* local index of the first game with an even global index */
  for (int i = b; i < t->num_games; i += 2) score += sp_score(&t->games[i]);
  for (int i = 1 - b; i < t->num_games; i += 2) score = (float)((double)score + (1.0 - (double)sp_score(&t->games[i])));
  return score / (float)(size_t)t->num_games;
}

/* ref: trainer.cpp:115-162.  `file << float` prints like "%g". */
int co_trainer_write_scores(const co_trainer *t, const char *filename) {
  size_t n = (size_t)t->num_games;
  float *scores = (float *)calloc(n ? n : 1, sizeof(float));
  for (size_t i = 0; i < n; i += 2) scores[i] = sp_score(&t->games[i]);
  for (size_t i = 1; i < n; i += 2) scores[i] = (float)(1.0 - (double)sp_score(&t->games[i]));
  FILE *f = fopen(filename, "w");
  if (!f) {
    free(scores);
    return 0;
  }
  static const char *who[2] = {"First", "Second"};
  for (int side = 0; side < 2; ++side) {
    int32_t wins = 0, draws = 0;
    for (size_t i = (size_t)side; i < n; i += 2) {
      if (scores[i] == 1.0) ++wins;
      else if (scores[i] == 0.5) ++draws;
    }
    size_t half = n / 2;
    fprintf(f, "%s player wins: %d / %zu = %g\n", who[side], wins, half, (double)((float)wins / (float)half));
    fprintf(f, "%s player draws: %d / %zu = %g\n", who[side], draws, half, (double)((float)draws / (float)half));
    fprintf(f, "%s player losses: %zu / %zu = %g\n", who[side], half - (size_t)wins - (size_t)draws, half,
            (double)((float)(half - (size_t)wins - (size_t)draws) / (float)half));
  }
  fclose(f);
  free(scores);
  return 1;
}

/* ref: trainer.cpp:70-77 */
float co_trainer_avg_mate_length(const co_trainer *t) {
  int32_t total = 0;
  for (int i = 0; i < t->num_games; ++i) total += sp_mate_length(&t->games[i]);
  return (float)total / (float)(size_t)t->num_games;
}

/* ref: trainer.cpp:79-101 (SelfPlayer::writeRequests selfplayer.cpp:73-77) */
void co_trainer_write_requests(const co_trainer *t, float *game_states, int to_play) {
  int offset = 0;
  int test = (to_play == 0 || to_play == 1);
  for (int i = 0; i < t->num_games; ++i) {
    if (t->is_done[i]) continue;
    if (test && !active_for(t, i, to_play)) continue;
    int n = sp_num_requests(&t->games[i]);
    memcpy(game_states + (size_t)offset * CO_GAME_STATE_SIZE, t->games[i].to_eval,
           sizeof(float) * (size_t)n * CO_GAME_STATE_SIZE);
    offset += n;
  }
}

/* ref: trainer.cpp:103-113 */
void co_trainer_write_samples(const co_trainer *t, float *game_states, float *eval_samples, float *prob_samples) {
  int offset = 0;
  for (int i = 0; i < t->num_games; ++i) {
    sp_write_samples(&t->games[i], game_states + (size_t)offset * CO_GAME_STATE_SIZE * CO_NUM_SYMMETRIES,
                     eval_samples + (size_t)offset * CO_NUM_SYMMETRIES,
                     prob_samples + (size_t)offset * CO_NUM_MOVES * CO_NUM_SYMMETRIES);
    offset += t->games[i].n_samples;
  }
}

/* ref: trainer.cpp:164-236 */
int co_trainer_do_iteration(co_trainer *t, const float *eval, const float *probs, int to_play) {
  int G = t->num_games;
  int *offsets = (int *)calloc((size_t)G, sizeof(int));
  if (to_play != 0 && to_play != 1) {
    int offset = 0;
    for (int i = 1; i < G; ++i) {
      offset += sp_num_requests(&t->games[i - 1]);
      offsets[i] = offset;
    }
    size_t div = (size_t)t->total_games / (size_t)t->max_searches;
    if (div < 1) div = 1;
#ifdef _OPENMP
    omp_set_num_threads(t->num_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < G; ++i) {
      if (!t->is_done[i]) {
        if (!t->stagger || (size_t)(t->game_base + i) / div <= (size_t)t->searches_done) {
          int done = sp_do_iteration(&t->games[i], eval + offsets[i], probs + (size_t)CO_NUM_MOVES * offsets[i]);
          if (done) t->is_done[i] = 1;
        }
      }
    }
    ++t->searches_done;
  } else {
    /* ref: trainer.cpp:205-235 */
    int offset = 0;
    for (int i = 1; i < G; ++i) {
      if (active_for(t, i - 1, to_play) && !t->is_done[i - 1]) offset += sp_num_requests(&t->games[i - 1]);
      offsets[i] = offset;
    }
#ifdef _OPENMP
    omp_set_num_threads(t->num_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < G; ++i) {
      if (active_for(t, i, to_play) && !t->is_done[i]) {
        int done = sp_do_iteration(&t->games[i], eval + offsets[i], probs + (size_t)CO_NUM_MOVES * offsets[i]);
        if (done) t->is_done[i] = 1;
      }
    }
  }
  free(offsets);
  for (int i = 0; i < G; ++i)
    if (!t->is_done[i]) return 0;
  return 1;
}

/* ------------------------------------------------------------ introspection */
int co_trainer_game_result(const co_trainer *t, int game) { return t->games[game].result; }
int co_trainer_game_to_play(const co_trainer *t, int game) { return t->games[game].to_play; }
int co_trainer_game_num_requests(const co_trainer *t, int game) {
  return t->is_done[game] ? 0 : sp_num_requests(&t->games[game]);
}
int co_trainer_game_num_samples(const co_trainer *t, int game) { return t->games[game].n_samples; }
int co_trainer_game_done(const co_trainer *t, int game) { return t->is_done[game]; }

void co_trainer_enable_trace(co_trainer *t, int on) {
  for (int i = 0; i < t->num_games; ++i) t->games[i].trace_on = on;
}

int co_trainer_trace(const co_trainer *t, int game, int32_t *out, int cap) {
  const selfplayer_t *sp = &t->games[game];
  int n = sp->n_trace < cap ? sp->n_trace : cap;
  if (out && n > 0) memcpy(out, sp->trace, sizeof(int32_t) * (size_t)n);
  return sp->n_trace;
}

void co_trainer_counters(const co_trainer *t, int64_t out[4]) {
  out[0] = out[1] = out[2] = out[3] = 0;
  for (int i = 0; i < t->num_games; ++i) {
    out[0] += t->games[i].ctr.searches;
    out[1] += t->games[i].ctr.leaf_evals;
    out[2] += t->games[i].ctr.nodes_created;
    out[3] += t->games[i].ctr.plies;
  }
}

/* ================================================================== DockerMC
 * ref: dockermc.h:13-51, dockermc.cpp:1-53 -- a TrainMC started from an arbitrary position
 * (trainmc.cpp:38-45: testing = true, createRoot(Game{board, to_play, pieces}, 0)) that owns its
 * generator and request buffer; the web app's single-position search (docker/choose_move.pyx). */
struct co_dockermc {
  co_mt19937 generator;
  float *to_eval;
  trainmc_t mc;
  counters_t ctr;
};

/* ref: game.cpp:14-26 */
co_dockermc *co_dockermc_create(int seed, int max_searches, int searches_per_eval, float c_puct, float epsilon,
                                const int32_t board[64], int to_play, const int32_t pieces[6]) {
  if (max_searches <= 0 || searches_per_eval <= 0) return NULL;
  co_dockermc *d = (co_dockermc *)calloc(1, sizeof *d);
  mt_seed(&d->generator, (uint32_t)seed);
  d->to_eval = (float *)calloc((size_t)searches_per_eval * CO_GAME_STATE_SIZE, sizeof(float));
  mc_init(&d->mc, &d->generator, d->to_eval, max_searches, searches_per_eval, c_puct, epsilon, 1, &d->ctr);
  uint64_t b = 0;
  for (int i = 0; i < 64; ++i)
    if (board[i] != 0) b |= 1ull << i;
  int8_t pc[6];
  for (int i = 0; i < 6; ++i) pc[i] = (int8_t)pieces[i];
  game_t g = make_game(b, pc, to_play);
  mc_create_root(&d->mc, &g, 0);
  return d;
}

void co_dockermc_destroy(co_dockermc *d) {
  if (!d) return;
  mc_free(&d->mc);
  free(d->to_eval);
  free(d);
}

/* ref: node.cpp:179-187 */
static int32_t count_nodes(const node_t *n) {
  int32_t counter = 1;
  for (const node_t *c = n->first_child; c != NULL; c = c->next_sibling) counter += count_nodes(c);
  return counter;
}

float co_dockermc_eval(const co_dockermc *d) { return d->mc.root->evaluation; }           /* trainmc.cpp:51-53 */
int co_dockermc_num_requests(const co_dockermc *d) { return d->mc.n_searched; }          /* :55-57 */
int co_dockermc_num_nodes(const co_dockermc *d) { return d->mc.root ? count_nodes(d->mc.root) : 0; } /* :67-72 */
int co_dockermc_done(const co_dockermc *d) { return n_terminal(d->mc.root); }            /* :74-76 */
int co_dockermc_drawn(const co_dockermc *d) { return n_drawn(d->mc.root); }              /* :78-80 */
/* ref: trainmc.cpp:89-94 */
void co_dockermc_write_requests(const co_dockermc *d, float *game_states) {
  for (int i = 0; i < d->mc.n_searched; ++i) write_game_state(&d->mc.searched[i]->game, game_states + (size_t)i * CO_GAME_STATE_SIZE);
}
/* ref: trainmc.cpp:96-108 */
void co_dockermc_get_legal_moves(const co_dockermc *d, int32_t legal_moves[CO_NUM_MOVES]) {
  mask96 legal;
  get_legal_moves(&d->mc.root->game, &legal);
  for (int i = 0; i < CO_NUM_MOVES; ++i) legal_moves[i] = m_test(&legal, i) ? 1 : 0;
}
/* ref: dockermc.cpp:47-49 (TrainMC::chooseMove with null sample pointers: testing_ is true) */
int co_dockermc_choose_move(co_dockermc *d) { return mc_choose_move(&d->mc, NULL, NULL); }
/* ref: dockermc.cpp:51-53 */
int co_dockermc_do_iteration(co_dockermc *d, const float *eval, const float *probs) {
  return mc_do_iteration(&d->mc, eval, probs);
}

/* ================================================================== Match / Tourney
 * The "next" row of SURVEY 8f.1: tournaments between any number of models
 * (ref: match.h:13-103, match.cpp, tourney.h, tourney.cpp; Python driver rating/tourney.pyx). */

/* ref: match.h:13-31 */
typedef struct {
  int player_id, model_id, max_searches, searches_per_eval;
  float c_puct, epsilon;
  int random;
} player_t;

/* std::uniform_int_distribution<int32_t>(0, n - 1)(std::mt19937 &) as libstdc++ (GCC >= 11,
 * bits/uniform_int_dist.h) computes it for a 32-bit generator: Lemire's nearly divisionless
 * method on one 64-bit product per draw; n == 1 still consumes a draw.  Checked against the
 * C++ library itself by tests/test_tourney.py.  (match.cpp:199-200) */
static uint32_t uniform_below(co_mt19937 *g, uint32_t n) {
  uint64_t product = (uint64_t)mt_next(g) * (uint64_t)n;
  uint32_t low = (uint32_t)product;
  if (low < n) {
    uint32_t threshold = (uint32_t)(0u - n) % n;
    while (low < threshold) {
      product = (uint64_t)mt_next(g) * (uint64_t)n;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}
uint32_t co_uniform_below(co_mt19937 *g, uint32_t n) { return uniform_below(g, n); }

/* ref: match.h:34-103 */
typedef struct {
  co_mt19937 generator;
  float *to_eval;
  trainmc_t players[2];
  int is_random[2]; /* players_[i] == nullptr */
  int ids[2], model_ids[2];
  node_t *root; /* Match::root_: the position on the board */
  int to_play;
  int8_t result;
  counters_t ctr, root_ctr;
  int trace_on;
  int32_t *trace;
  int n_trace, cap_trace;
  FILE *log; /* Match::log_file_ (match.h:98-101) */
  int log_fixed;
} match_t;

static void match_trace_push(match_t *m, int32_t v) {
  if (!m->trace_on) return;
  if (m->n_trace == m->cap_trace) {
    m->cap_trace = m->cap_trace ? m->cap_trace * 2 : 256;
    m->trace = (int32_t *)realloc(m->trace, sizeof(int32_t) * (size_t)m->cap_trace);
  }
  m->trace[m->n_trace++] = v;
}

/* ref: match.cpp:15-35 */
static match_t *match_new(uint32_t seed, const player_t *p1, const player_t *p2) {
  match_t *m = (match_t *)calloc(1, sizeof *m);
  mt_seed(&m->generator, seed);
  int cap = p1->max_searches > p2->max_searches ? p1->max_searches : p2->max_searches;
  m->to_eval = (float *)calloc((size_t)CO_GAME_STATE_SIZE * (size_t)cap, sizeof(float));
  const player_t *pp[2] = {p1, p2};
  for (int i = 0; i < 2; ++i) {
    m->is_random[i] = pp[i]->random;
    m->ids[i] = pp[i]->player_id;
    m->model_ids[i] = pp[i]->model_id;
    if (!pp[i]->random)
      mc_init(&m->players[i], &m->generator, m->to_eval, pp[i]->max_searches, pp[i]->searches_per_eval,
              pp[i]->c_puct, pp[i]->epsilon, 1, &m->ctr);
  }
  m->root = node_new_start(&m->root_ctr);
  m->result = kResultNone;
  return m;
}

static void match_free(match_t *m) {
  for (int i = 0; i < 2; ++i)
    if (!m->is_random[i]) mc_free(&m->players[i]);
  if (m->root) node_delete(m->root);
  free(m->to_eval);
  free(m->trace);
  if (m->log) fclose(m->log);
  free(m);
}

/* ref: match.cpp:42-50 */
static int match_to_play(const match_t *m) { return m->model_ids[m->to_play]; }
static int match_num_requests(const match_t *m) {
  if (m->is_random[m->to_play]) return 0;
  return m->players[m->to_play].n_searched;
}

/* ref: match.cpp:52-58 */
static float match_score(const match_t *m) {
  if (m->result == kResultLoss) return 0.0f;
  if (m->result == kResultWin) return 1.0f;
  return 0.5f;
}

/* ref: match.cpp:163-190 */
static void match_end_game(match_t *m) {
  if (m->root->result == kResultDraw) m->result = kResultDraw;
  else if (m->to_play == 1) m->result = kResultLoss;
  else m->result = kResultWin;
  if (m->log) { /* match.cpp:173-179, 190 */
    if (m->result == kResultDraw) fputs("GAME IS DRAWN.\n", m->log);
    else fprintf(m->log, "PLAYER %d WON!\n", m->to_play + 1);
    fclose(m->log);
    m->log = NULL;
  }
  for (int i = 0; i < 2; ++i)
    if (!m->is_random[i]) mc_null_root(&m->players[i]);
  free(m->to_eval);
  m->to_eval = NULL;
}

/* ref: match.cpp:192-205 */
static int match_choose_move(match_t *m) {
  int choice;
  if (m->is_random[m->to_play]) {
    int n = m->root->num_legal_moves;
    choice = n_move_id(m->root, (int)uniform_below(&m->generator, (uint32_t)n));
    match_trace_push(m, -2);
  } else {
    const node_t *r = m->players[m->to_play].root;
    if (m->trace_on) {
      match_trace_push(m, m->to_play);
      match_trace_push(m, r->depth);
      match_trace_push(m, r->visits);
      match_trace_push(m, r->result);
      match_trace_push(m, fbits(r->evaluation));
      int n = 0;
      for (const node_t *c = r->first_child; c; c = c->next_sibling) ++n;
      match_trace_push(m, n);
      for (const node_t *c = r->first_child; c; c = c->next_sibling) {
        match_trace_push(m, c->child_id);
        match_trace_push(m, c->visits);
        match_trace_push(m, fbits(c->evaluation));
        match_trace_push(m, c->result);
        match_trace_push(m, c->all_visited);
      }
    }
    choice = mc_choose_move(&m->players[m->to_play], NULL, NULL);
  }
  match_trace_push(m, choice);
  m->ctr.plies++;
  return choice;
}

/* ref: match.cpp:207-251 */
static int match_choose_move_and_continue(match_t *m) {
  int need_eval = 0;
  while (!need_eval) {
    if (m->log && !m->is_random[m->to_play]) log_pre_move_of(m->log, &m->log_fixed, m->players[m->to_play].root, m->to_play);
    int choice = match_choose_move(m);
    node_t *next = node_new_child(&m->root->game, NULL, NULL, choice, m->root->depth + 1, &m->root_ctr);
    node_delete(m->root);
    m->root = next;
    if (m->log) log_move_choice_of(m->log, choice, &m->root->game);
    if (n_terminal(m->root)) {
      match_end_game(m);
      return 1;
    }
    m->to_play = 1 - m->to_play;
    if (m->is_random[m->to_play]) continue;
    trainmc_t *me = &m->players[m->to_play];
    if (me->root == NULL) {
      mc_create_root(me, &m->root->game, m->root->depth);
      return mc_do_iteration(me, NULL, NULL);
    }
    need_eval = mc_receive_opponent_move(me, choice, &m->root->game, m->root->depth);
    if (!need_eval) need_eval = !mc_do_iteration(me, NULL, NULL);
  }
  return 0;
}

/* ref: match.cpp:67-79 */
static int match_do_iteration(match_t *m, const float *eval, const float *probs) {
  if (m->is_random[m->to_play]) return match_choose_move_and_continue(m);
  int done = mc_do_iteration(&m->players[m->to_play], eval, probs);
  if (done) return match_choose_move_and_continue(m);
  return 0;
}

/* ref: tourney.h:13-46 */
#define CO_TOURNEY_MAX_PLAYERS 1024
struct co_tourney {
  match_t **matches;
  uint8_t *is_done;
  int n_matches, cap_matches;
  player_t players[CO_TOURNEY_MAX_PLAYERS];
  uint8_t has_player[CO_TOURNEY_MAX_PLAYERS];
  co_mt19937 generator; /* default constructed: seed 5489 (tourney.h:43) */
  int num_threads;
  int trace_on;
  int exact_offsets; /* co_tourney_set_exact_offsets */
};

co_tourney *co_tourney_create(int num_threads) {
  co_tourney *t = (co_tourney *)calloc(1, sizeof *t);
  mt_seed(&t->generator, 5489u);
  t->num_threads = num_threads > 0 ? num_threads : 1;
  return t;
}

void co_tourney_destroy(co_tourney *t) {
  if (!t) return;
  for (int i = 0; i < t->n_matches; ++i) match_free(t->matches[i]);
  free(t->matches);
  free(t->is_done);
  free(t);
}

/* ref: tourney.cpp:72-78 */
int co_tourney_add_player(co_tourney *t, int player_id, int model_id, int max_searches, int searches_per_eval,
                          float c_puct, float epsilon, int random) {
  if (player_id < 0 || player_id >= CO_TOURNEY_MAX_PLAYERS) return -1;
  player_t p = {player_id, model_id, max_searches, searches_per_eval, c_puct, epsilon, random};
  t->players[player_id] = p;
  t->has_player[player_id] = 1;
  return 0;
}

/* ref: tourney.cpp:80-96.  log_folder non-null = addMatch(..., logging = true) of a Tourney built with that folder: the
 * match writes `<log_folder>/match_<player1>_<player2>_<index>.txt` (an unopenable file is skipped silently, like the
 * reference's ofstream) */
int co_tourney_add_match_logged(co_tourney *t, int player1, int player2, const char *log_folder);
int co_tourney_add_match(co_tourney *t, int player1, int player2) { return co_tourney_add_match_logged(t, player1, player2, NULL); }
int co_tourney_add_match_logged(co_tourney *t, int player1, int player2, const char *log_folder) {
  if (player1 < 0 || player1 >= CO_TOURNEY_MAX_PLAYERS || !t->has_player[player1]) return -1;
  if (player2 < 0 || player2 >= CO_TOURNEY_MAX_PLAYERS || !t->has_player[player2]) return -1;
  if (t->n_matches == t->cap_matches) {
    t->cap_matches = t->cap_matches ? t->cap_matches * 2 : 64;
    t->matches = (match_t **)realloc(t->matches, sizeof(match_t *) * (size_t)t->cap_matches);
    t->is_done = (uint8_t *)realloc(t->is_done, (size_t)t->cap_matches);
  }
  match_t *m = match_new(mt_next(&t->generator), &t->players[player1], &t->players[player2]);
  m->trace_on = t->trace_on;
  if (log_folder) {
    char path[4096];
    snprintf(path, sizeof path, "%s/match_%d_%d_%d.txt", log_folder, player1, player2, t->n_matches);
    m->log = fopen(path, "w");
  }
  t->matches[t->n_matches] = m;
  t->is_done[t->n_matches] = 0;
  return t->n_matches++;
}

/* ref: tourney.cpp:14-21 */
int co_tourney_all_done(const co_tourney *t) {
  for (int i = 0; i < t->n_matches; ++i)
    if (!t->is_done[i]) return 0;
  return 1;
}

/* ref: tourney.cpp:23-31 */
int co_tourney_num_requests(const co_tourney *t, int id) {
  int count = 0;
  for (int i = 0; i < t->n_matches; ++i)
    if (!t->is_done[i] && match_to_play(t->matches[i]) == id) count += match_num_requests(t->matches[i]);
  return count;
}

/* ref: tourney.cpp:43-51 (Match::writeRequests match.cpp:60-65) */
void co_tourney_write_requests(const co_tourney *t, float *game_states, int id) {
  int offset = 0;
  for (int i = 0; i < t->n_matches; ++i) {
    if (!t->is_done[i] && match_to_play(t->matches[i]) == id) {
      int n = match_num_requests(t->matches[i]);
      memcpy(game_states + (size_t)offset * CO_GAME_STATE_SIZE, t->matches[i]->to_eval,
             sizeof(float) * (size_t)n * CO_GAME_STATE_SIZE);
      offset += n;
    }
  }
}

/* ref: tourney.cpp:53-70.  The offset table is the reference's: match i
 * reads from the running sum of the request counts of matches j - 1 for the ACTIVE j <= i,
 * which equals its row in writeRequests only while every unfinished match waits for `id`. */
void co_tourney_do_iteration(co_tourney *t, const float *eval, const float *probs, int id) {
  int n = t->n_matches;
  int *offsets = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
  int offset = 0;
  for (int i = 1; i < n; ++i) {
    if (!t->is_done[i] && match_to_play(t->matches[i]) == id) offset += match_num_requests(t->matches[i - 1]);
    offsets[i] = offset;
  }
  if (t->exact_offsets) {
    /* diagnostic: every match reads the rows co_tourney_write_requests gave it */
    offset = 0;
    for (int i = 0; i < n; ++i) {
      offsets[i] = offset;
      if (!t->is_done[i] && match_to_play(t->matches[i]) == id) offset += match_num_requests(t->matches[i]);
    }
  }
#ifdef _OPENMP
  omp_set_num_threads(t->num_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < n; ++i) {
    if (!t->is_done[i] && match_to_play(t->matches[i]) == id) {
      int done = match_do_iteration(t->matches[i], eval + offsets[i], probs + (size_t)offsets[i] * CO_NUM_MOVES);
      if (done) t->is_done[i] = 1;
    }
  }
  free(offsets);
}

/* Not in the reference: 1 = matches read their own rows instead of through the table above.  The reference's
 * own tournament results (rating/results.txt) are reproduced with this switch on and not with the table of
 * tourney.cpp:55-62 (tests/test_reference_results.py). */
void co_tourney_set_exact_offsets(co_tourney *t, int on) { t->exact_offsets = on != 0; }

/* ref: tourney.cpp:33-41: one line "id1 id2 score" per finished match */
int co_tourney_write_scores(const co_tourney *t, const char *filename) {
  FILE *f = fopen(filename, "w");
  if (!f) return -1;
  for (int i = 0; i < t->n_matches; ++i)
    if (t->is_done[i]) fprintf(f, "%d %d %g\n", t->matches[i]->ids[0], t->matches[i]->ids[1], (double)match_score(t->matches[i]));
  fclose(f);
  return 0;
}

/* introspection for the parity tests */
int co_tourney_num_matches(const co_tourney *t) { return t->n_matches; }
int co_tourney_match_done(const co_tourney *t, int i) { return t->is_done[i]; }
float co_tourney_match_score(const co_tourney *t, int i) { return match_score(t->matches[i]); }
int co_tourney_match_result(const co_tourney *t, int i) { return t->matches[i]->result; }
int co_tourney_match_to_play(const co_tourney *t, int i) { return t->matches[i]->to_play; }
int co_tourney_match_num_requests(const co_tourney *t, int i) {
  return t->is_done[i] ? 0 : match_num_requests(t->matches[i]);
}
void co_tourney_enable_trace(co_tourney *t, int on) {
  t->trace_on = on;
  for (int i = 0; i < t->n_matches; ++i) t->matches[i]->trace_on = on;
}
int co_tourney_trace(const co_tourney *t, int i, int32_t *out, int cap) {
  const match_t *m = t->matches[i];
  int n = m->n_trace < cap ? m->n_trace : cap;
  if (out && n > 0) memcpy(out, m->trace, sizeof(int32_t) * (size_t)n);
  return m->n_trace;
}
void co_tourney_counters(const co_tourney *t, int64_t out[4]) {
  out[0] = out[1] = out[2] = out[3] = 0;
  for (int i = 0; i < t->n_matches; ++i) {
    out[0] += t->matches[i]->ctr.searches;
    out[1] += t->matches[i]->ctr.leaf_evals;
    out[2] += t->matches[i]->ctr.nodes_created;
    out[3] += t->matches[i]->ctr.plies;
  }
}
