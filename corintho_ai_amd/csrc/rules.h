// rules.h -- Corintho rules on a packed bitboard, one position per wavefront.
//
// Position = 64-bit board (bit = row*16 + col*4 + {0 base, 1 column, 2 capital,
// 3 frozen}, game.cpp:141-150) + meta word (six 3-bit reserve counters, side to
// move).  Follows game.cpp:28-43 (getLegalMoves), :60-96 (doMove), :45-58
// (writeGameState), :193-232 (canPlace/canMove), :249-405 (line detection) and
// node.cpp:256-271 (terminal result).  Legal-move generation is lane parallel:
// lanes test the 34 candidate lines and the 96 moves at once; everything a
// caller sees is wave-uniform.
#pragma once
#include "engine_defs.h"
#include "tables.inc"
#include "wave.h"

CO_CONST uint32_t CO_LINE_BREAKERS[CO_NUM_LINES][3] = CO_LINE_BREAKERS_INIT;

#define CO_META_TO_PLAY(meta) (((meta) >> 18) & 1u)
#define CO_META_DEPTH(meta) (((meta) >> 19) & 63u)
#define CO_META_NEDGES(meta) (((meta) >> 25) & 127u)
#define CO_META_PIECE(meta, i) (((meta) >> (3 * (i))) & 7u)
#define CO_META_START 0x24924u /* 4,4,4,4,4,4 pieces, first player to move, depth 0 */

CO_DEV uint32_t co_meta_make(uint32_t meta_game, int depth, int n_edges) {
  return (meta_game & 0x7FFFFu) | ((uint32_t)depth << 19) | ((uint32_t)n_edges << 25);
}

CO_DEV int co_nib_top(uint32_t nib) { return (nib & 4u) ? 2 : (nib & 2u) ? 1 : (nib & 1u) ? 0 : -1; }
CO_DEV int co_nib_bottom(uint32_t nib) { return (nib & 1u) ? 0 : (nib & 2u) ? 1 : (nib & 4u) ? 2 : 3; }
CO_DEV uint32_t co_nib(uint64_t board, int cell) { return (uint32_t)(board >> (4 * cell)) & 15u; }

/* move id -> (is_place, piece, from cell, to cell); move.cpp:11-42 */
CO_DEV void co_decode_move(int id, int *is_place, int *piece, int *from, int *to) {
  if (id >= 48) {
    *is_place = 1;
    *piece = (id - 48) / 16;
    *from = -1;
    *to = id % 16;
    return;
  }
  *is_place = 0;
  *piece = 0;
  if (id < 12) {
    *from = (id / 3) * 4 + id % 3;
    *to = *from + 1;
  } else if (id < 24) {
    *from = id - 12;
    *to = *from + 4;
  } else if (id < 36) {
    *from = ((id - 24) / 3) * 4 + id % 3 + 1;
    *to = *from - 1;
  } else {
    *from = id - 36 + 4;
    *to = *from - 4;
  }
}

/* Per-lane constants of the rule layer: functions of the lane index alone, computed once per wavefront
 * (co_lanes_init) and kept in registers across the simulations of a step.
 *   candidate line j = lane < 34, in the reference's scan order: lanes 0-11 rows (i = j / 3; long, left triple, right
 *   triple), 12-23 columns, 24-29 long diagonals (main: long, upper, lower; anti: ...), 30-33 short diagonals.  Every
 *   candidate is an arithmetic progression of cells: line_mask = its cells, line_base = its first line_breakers index
 *   (+ the top piece);
 *   move id = lane < 48: the nibble shifts (4 x cell) of its source and destination cells (move.cpp:11-42);
 *   placements: id 48 + cell on lanes 48-63 and ids 64 + lane on lanes 0-31 all look at cell lane & 15. */
struct CoLanes {
  LV(uint32_t, plane_hi);  /* the board bit this lane contributes to the plane ballot (piece lane >> 4 of cell lane & 15): */
  LV(uint32_t, plane_sh);  /* ... in the high word?  and its position there */
  LV(uint32_t, line_mask);
  LV(int, line_base);
  LV(uint32_t, from_hi);   /* move id = lane < 48: is the source cell's nibble in the high word of the board, */
  LV(uint32_t, from_sh);   /* ... and its shift there */
  LV(uint32_t, to_hi);
  LV(uint32_t, to_sh);
};

CO_DEV void co_lanes_init(CoLanes &K) {
  FOR_LANES {
    const int pbit = (lane & 15) * 4 + (lane >> 4);
    L(K.plane_hi) = (uint32_t)(pbit >> 5);
    L(K.plane_sh) = (uint32_t)(pbit & 31);
    const int j = lane;
    int c0 = 0, step = 1, count = 0, base = 0;
    if (j < 12) {
      int i = j / 3, k = j % 3;
      c0 = 4 * i + (k == 2 ? 1 : 0); step = 1; count = k == 0 ? 4 : 3;
      base = (k == 0 ? 24 : k == 1 ? 0 : 12) + 3 * i;
    } else if (j < 24) {
      int i = (j - 12) / 3, k = (j - 12) % 3;
      c0 = i + (k == 2 ? 4 : 0); step = 4; count = k == 0 ? 4 : 3;
      base = (k == 0 ? 60 : k == 1 ? 36 : 48) + 3 * i;
    } else if (j < 30) {
      int f = (j - 24) / 3, k = (j - 24) % 3;
      step = f ? 3 : 5;
      c0 = (f ? 3 : 0) + (k == 2 ? step : 0); count = k == 0 ? 4 : 3;
      base = 72 + (f ? 9 : 0) + (k == 0 ? 6 : k == 1 ? 0 : 3);
    } else if (j < 34) {
      int s = j - 30;
      c0 = s == 0 ? 2 : s == 1 ? 1 : s == 2 ? 7 : 4; step = (s & 1) ? 5 : 3; count = 3;
      base = 90 + 3 * s;
    }
    uint32_t M = 0u;
    if (count) {
      M = (1u << c0) | (1u << (c0 + step)) | (1u << (c0 + 2 * step));
      if (count == 4) M |= 1u << (c0 + 3 * step);
    }
    L(K.line_mask) = M;
    L(K.line_base) = base;
    int is_place, piece, from, to;
    co_decode_move(lane < 48 ? lane : 0, &is_place, &piece, &from, &to);
    L(K.from_hi) = (uint32_t)(from >> 3);
    L(K.from_sh) = (uint32_t)(4 * (from & 7));
    L(K.to_hi) = (uint32_t)(to >> 3);
    L(K.to_sh) = (uint32_t)(4 * (to & 7));
  }
}

/* game.cpp:28-43.  out[3] = 96-bit legal mask; returns is_lines.  Uniform in, uniform out; inside:
 *   the four 16-cell bit planes (base, column, capital, frozen) are ONE ballot -- lane l tests piece l >> 4 of cell
 *   l & 15;
 *   the 34 candidate lines are tested one per lane against the top-piece planes (game.cpp:249-405);
 *   placements (canPlace, game.cpp:193-220) are a dozen bit operations on the planes;
 *   stack moves (canMove, game.cpp:222-232): lane id < 48 looks at the nibbles of ITS move's two cells -- two compares
 *   per lane, whose masks are the legality bits.
 * (Round 3 computed planes and legality as ~450 scalar instructions per position -- the largest single cost of a
 * simulation.) */
CO_DEV int co_legal_moves(uint64_t board, uint32_t meta, uint32_t out[3], const CoLanes &K) {
  const uint32_t blo = (uint32_t)board, bhi = (uint32_t)(board >> 32);
  LV(int, pbit);
  LV(int, mv1);
  LV(int, mv2);
  FOR_LANES {
    L(pbit) = (int)(((L(K.plane_hi) ? bhi : blo) >> L(K.plane_sh)) & 1u);
    /* a nibble = {base, column, capital, frozen} of a cell.  a -> b is legal iff both are non-empty, neither is frozen and
     * bottom(a) - top(b) == 1: a column-bottomed stack (nibble 2 or 6) on a bare base (1), or a bare capital (4) on a
     * column-topped stack (2 or 3) -- as one byte i = a | b << 4: i in {0x12, 0x16} or i in {0x24, 0x34} */
    const uint32_t na = ((L(K.from_hi) ? bhi : blo) >> L(K.from_sh)) & 15u, nb = ((L(K.to_hi) ? bhi : blo) >> L(K.to_sh)) & 15u;
    const uint32_t i = na | (nb << 4);
    L(mv1) = lane < 48 && (i & 0xFBu) == 0x12u;
    L(mv2) = lane < 48 && (i & 0xEFu) == 0x24u;
  }
  const uint64_t planes = WAVE_BALLOT(pbit);
  const uint64_t moves = WAVE_BALLOT(mv1) | WAVE_BALLOT(mv2);
  const uint32_t B = (uint32_t)planes & 0xFFFFu, C = (uint32_t)(planes >> 16) & 0xFFFFu, A = (uint32_t)(planes >> 32) & 0xFFFFu;
  const uint32_t F = (uint32_t)(planes >> 48);
  const uint32_t T2 = A, T1 = C & ~A, T0 = B & ~C & ~A; /* top piece (game.cpp:158-168) */
  /* ---- lines: a candidate is a line iff all its cells lie in one top-piece plane */
  LV(int, match);
  LV(int, ctop);
  FOR_LANES {
    const uint32_t M = L(K.line_mask);
    const int t = (T0 & M) == M ? 0 : (T1 & M) == M ? 1 : (T2 & M) == M ? 2 : -1;
    L(match) = M != 0u && t >= 0;
    L(ctop) = t;
  }
  const uint64_t cand = WAVE_BALLOT(match);
  uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
  int is_lines = 0;
  if (cand) {
    is_lines = 1;
    /* one line per category, first match in scan order (game.cpp:265,310,330-356,368-388) */
#pragma unroll
    for (int cat = 0; cat < 4; ++cat) {
      const uint64_t cmask = cat == 0 ? 0xFFFull : cat == 1 ? (0xFFFull << 12) : cat == 2 ? (0x3Full << 24) : (0xFull << 30);
      uint64_t c = cand & cmask;
      if (!c) continue;
      int j = co_ffs64(c) - 1;
      int t = WAVE_BCAST(ctop, j);
      int line = WAVE_BCAST(K.line_base, j) + t;
      m0 &= CO_LINE_BREAKERS[line][0];
      m1 &= CO_LINE_BREAKERS[line][1];
      m2 &= CO_LINE_BREAKERS[line][2];
      if (cat < 2 && t == 2) {
        int k = (cat == 0 ? j : j - 12) % 3;
        if (k != 0) {
          /* capital triple in a row/column: game.cpp:280-309.  ec = extension
           * coordinate; the cells Space{kk, ec, isCol}, kk = 0..3, are visited. */
          int is_col = cat == 1;
          int ec = k == 1 ? 3 : 0;
          for (int kk = 0; kk < 4; ++kk) {
            int cell = is_col ? (ec * 4 + kk) : (kk * 4 + ec);
            if ((A >> cell) & 1u) continue;
            /* moves kk -> kk-1 and kk -> kk+1 along that line of cells */
            if (kk > 0) {
              int id = is_col ? (24 + ec * 3 + (kk - 1)) : (36 + (kk - 1) * 4 + ec);
              if (id < 32) m0 &= ~(1u << id); else m1 &= ~(1u << (id - 32));
            }
            if (kk < 3) {
              int id = is_col ? (ec * 3 + kk) : (12 + kk * 4 + ec);
              if (id < 32) m0 &= ~(1u << id); else m1 &= ~(1u << (id - 32));
            }
          }
        }
      }
    }
  }
  /* ---- placements on the planes (canPlace, game.cpp:193-220): an empty cell takes anything (frozen or not: the test for
   * empty comes first); a frozen one nothing else; a column goes on a bare base, a capital on a column-topped stack */
  const uint32_t E = ~(B | C | A) & 0xFFFFu;
  const uint32_t mine = CO_META_TO_PLAY(meta) ? meta >> 9 : meta; /* the mover's three counters in the low nine bits */
  const uint32_t pb = (mine & 7u) ? E : 0u;
  const uint32_t pc = (mine & 0x38u) ? (E | (T0 & ~F)) : 0u;
  const uint32_t pa = (mine & 0x1C0u) ? (E | (T1 & ~F)) : 0u;
  out[0] = m0 & (uint32_t)moves;
  out[1] = m1 & ((uint32_t)(moves >> 32) | (pb << 16));
  out[2] = m2 & (pc | (pa << 16));
  return is_lines;
}

/* game.cpp:60-96.  Uniform.  The cells of a stack move are lane `id`'s constants (no decoding, no division). */
CO_DEV void co_do_move(uint64_t *board, uint32_t *meta, int id, const CoLanes &K) {
  uint64_t b = *board & ~0x8888888888888888ull;
  uint32_t m = *meta;
  if (id >= 48) {
    const uint32_t piece = (uint32_t)(id - 48) >> 4, to4 = 4u * ((uint32_t)id & 15u);
    m -= 1u << (3u * (CO_META_TO_PLAY(m) * 3u + piece));
    b |= (uint64_t)((1u << piece) | 8u) << to4;
  } else {
    const uint32_t from4 = WAVE_BCAST(K.from_sh, id) + 32u * WAVE_BCAST(K.from_hi, id);
    const uint32_t to4 = WAVE_BCAST(K.to_sh, id) + 32u * WAVE_BCAST(K.to_hi, id);
    const uint64_t src = (b >> from4) & 7ull;
    b &= ~(15ull << from4);
    b |= (src | 8ull) << to4;
  }
  *board = b;
  *meta = m ^ (1u << 18);
}

/* game.cpp:45-58: lanes 0..63 write the board bits, lanes 0..5 the reserves (the mover's first).
 * `row` has room for CO_STATE_STRIDE floats; the padding is zeroed. */
CO_DEV void co_write_state(uint64_t board, uint32_t meta, float *row) {
  /* the six counters in the mover's order: the second player's view swaps the two groups of nine bits */
  const uint32_t pc = meta & 0x3FFFFu;
  const uint32_t rot = CO_META_TO_PLAY(meta) ? ((pc >> 9) | (pc << 9)) & 0x3FFFFu : pc;
  FOR_LANES {
    row[lane] = (float)(uint32_t)((board >> lane) & 1ull);
    if (lane < CO_STATE_STRIDE - 64) row[64 + lane] = lane < 6 ? (float)((rot >> (3 * lane)) & 7u) * 0.25f : 0.0f;
  }
}
