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

/* every 4th bit of x (bits 4c + k, k given by the caller's shift) gathered into bits 0..15 */
CO_DEV uint32_t co_plane(uint64_t x) {
  x &= 0x1111111111111111ull;
  x = (x | (x >> 3)) & 0x0303030303030303ull;
  x = (x | (x >> 6)) & 0x000F000F000F000Full;
  x = (x | (x >> 12)) & 0x000000FF000000FFull;
  x = (x | (x >> 24)) & 0xFFFFull;
  return (uint32_t)x;
}

/* game.cpp:28-43.  out[3] = 96-bit legal mask; returns is_lines.  Uniform.
 * The board is split into four 16-cell bit planes (base, column, capital, frozen); basic
 * legality of all 96 moves (game.cpp:193-242) is then ~80 wave-uniform bit operations on
 * those planes -- scalar work, no per-lane code -- and only the 34 candidate lines are
 * tested one per lane. */
CO_DEV int co_legal_moves(uint64_t board, uint32_t meta, uint32_t out[3]) {
  const uint32_t B = co_plane(board), C = co_plane(board >> 1), A = co_plane(board >> 2), F = co_plane(board >> 3);
  const uint32_t N = B | C | A;             /* non-empty */
  const uint32_t E = ~N & 0xFFFFu;          /* empty */
  const uint32_t T2 = A, T1 = C & ~A, T0 = B & ~C & ~A; /* top piece (game.cpp:158-168) */
  /* ---- the 34 candidate lines, in the reference's scan order: lanes 0-11 rows (i = j/3;
   * long, left triple, right triple), 12-23 columns, 24-29 long diagonals (main: long, upper,
   * lower; anti: ...), 30-33 short diagonals.  Every candidate is an arithmetic progression
   * of cells; it is a line iff all its cells lie in one top-piece plane. */
  LV(int, match);
  LV(int, ctop);
  LV(int, cbase);
  FOR_LANES {
    int j = lane;
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
    uint32_t M = (1u << c0) | (1u << (c0 + step)) | (1u << (c0 + 2 * step));
    if (count == 4) M |= 1u << (c0 + 3 * step);
    int t = (T0 & M) == M ? 0 : (T1 & M) == M ? 1 : (T2 & M) == M ? 2 : -1;
    L(match) = count != 0 && t >= 0;
    L(ctop) = t;
    L(cbase) = base;
  }
  uint64_t cand = WAVE_BALLOT(match);
  uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
  int is_lines = 0;
  /* one line per category, first match in scan order (game.cpp:265,310,330-356,368-388) */
#pragma unroll
  for (int cat = 0; cat < 4; ++cat) {
    const uint64_t cmask = cat == 0 ? 0xFFFull : cat == 1 ? (0xFFFull << 12) : cat == 2 ? (0x3Full << 24) : (0xFull << 30);
    uint64_t c = cand & cmask;
    if (!c) continue;
    is_lines = 1;
    int j = co_ffs64(c) - 1;
    int t = WAVE_BCAST(ctop, j);
    int line = WAVE_BCAST(cbase, j) + t;
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
  /* ---- basic legality of all 96 moves on the planes.
   * place (canPlace, game.cpp:193-220): an empty cell takes anything; a frozen one nothing;
   * a column goes on a bare base; a capital on anything without a capital except a bare base */
  const uint32_t NF = N & ~F;
  const uint32_t tp3 = CO_META_TO_PLAY(meta) * 3;
  uint32_t pb = CO_META_PIECE(meta, tp3 + 0) ? E : 0u;
  uint32_t pc = CO_META_PIECE(meta, tp3 + 1) ? (E | (NF & B & ~C & ~A)) : 0u;
  uint32_t pa = CO_META_PIECE(meta, tp3 + 2) ? (E | (NF & ~A & ~(B & ~C))) : 0u;
  /* move a -> b (canMove, game.cpp:222-232): both non-empty, neither frozen,
   * bottom(a) - top(b) == 1, i.e. (bottom 1 on top 0) or (bottom 2 on top 1) */
  const uint32_t Bo1 = ~B & C, Bo2 = ~B & ~C & A;
  const uint32_t R = NF & (NF >> 1) & ((Bo1 & (T0 >> 1)) | (Bo2 & (T1 >> 1))) & 0x7777u; /* from col < 3 */
  const uint32_t D = NF & (NF >> 4) & ((Bo1 & (T0 >> 4)) | (Bo2 & (T1 >> 4))) & 0x0FFFu; /* from row < 3 */
  const uint32_t Lf = NF & (NF << 1) & ((Bo1 & (T0 << 1)) | (Bo2 & (T1 << 1))) & 0xEEEEu; /* from col > 0 */
  const uint32_t U = NF & (NF << 4) & ((Bo1 & (T0 << 4)) | (Bo2 & (T1 << 4))) & 0xFFF0u;  /* from row > 0 */
  /* move ids (move.cpp:11-42): right r*3+c, down 12+cell, left 24+r*3+(c-1), up 36+cell-4 */
  const uint32_t rid = (R & 7u) | (((R >> 4) & 7u) << 3) | (((R >> 8) & 7u) << 6) | (((R >> 12) & 7u) << 9);
  const uint32_t lid = ((Lf >> 1) & 7u) | (((Lf >> 5) & 7u) << 3) | (((Lf >> 9) & 7u) << 6) | (((Lf >> 13) & 7u) << 9);
  const uint32_t b0 = rid | (D << 12) | (lid << 24);
  const uint32_t b1 = (lid >> 8) | ((U >> 4) << 4) | (pb << 16);
  const uint32_t b2 = pc | (pa << 16);
  out[0] = m0 & b0;
  out[1] = m1 & b1;
  out[2] = m2 & b2;
  return is_lines;
}

/* game.cpp:60-96.  Uniform. */
CO_DEV void co_do_move(uint64_t *board, uint32_t *meta, int id) {
  int is_place, piece, from, to;
  co_decode_move(id, &is_place, &piece, &from, &to);
  uint64_t b = *board & ~0x8888888888888888ull;
  uint32_t m = *meta;
  uint32_t tp = CO_META_TO_PLAY(m);
  if (is_place) {
    m -= 1u << (3 * (tp * 3 + piece));
    b |= (uint64_t)((1u << piece) | 8u) << (4 * to);
  } else {
    uint64_t src = (b >> (4 * from)) & 7ull;
    b |= src << (4 * to);
    b &= ~(15ull << (4 * from));
    b |= 8ull << (4 * to);
  }
  *board = b;
  *meta = m ^ (1u << 18);
}

/* game.cpp:45-58: lanes 0..63 write the board bits, lanes 0..5 the reserves.
 * `row` has room for CO_STATE_STRIDE floats; the padding is zeroed. */
CO_DEV void co_write_state(uint64_t board, uint32_t meta, float *row) {
  FOR_LANES {
    row[lane] = ((board >> lane) & 1ull) ? 1.0f : 0.0f;
    if (lane < CO_STATE_STRIDE - 64) {
      float v = 0.0f;
      if (lane < 6) {
        uint32_t tp = CO_META_TO_PLAY(meta);
        v = (float)CO_META_PIECE(meta, (tp * 3 + lane) % 6) * 0.25f;
      }
      row[64 + lane] = v;
    }
  }
}
