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

/* ---- the rule layer for FOUR positions at a time, one per row of 16 lanes (round 5; mcts.h co_search_rows): position,
 * legal-move mask and everything in between are row-uniform values in vector registers; a lane's column is a cell.
 * Same rules, same citations as above; checked against co_legal_moves on the rules corpus and by every parity test. */

/* candidate line j (the reference's scan order, co_lanes_init): cells | first line_breakers index << 16 */
CO_CONST uint32_t CO_LINE_CAND[48] = {
    0x18000Fu, 0x7u, 0xC000Eu, 0x1B00F0u, 0x30070u, 0xF00E0u, 0x1E0F00u, 0x60700u, 0x120E00u, 0x21F000u, 0x97000u, 0x15E000u,
    0x3C1111u, 0x240111u, 0x301110u, 0x3F2222u, 0x270222u, 0x332220u, 0x424444u, 0x2A0444u, 0x364440u, 0x458888u, 0x2D0888u,
    0x398880u, 0x4E8421u, 0x480421u, 0x4B8420u, 0x571248u, 0x510248u, 0x541240u, 0x5A0124u, 0x5D0842u, 0x602480u, 0x634210u,
    0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};

/* game.cpp:60-96 on per-lane values (no cross-lane traffic) */
CO_DEV void co_do_move_lane(uint64_t *board, uint32_t *meta, int id) {
  uint64_t b = *board & ~0x8888888888888888ull;
  uint32_t m = *meta;
  if (id >= 48) {
    const uint32_t piece = (uint32_t)(id - 48) >> 4, to4 = 4u * ((uint32_t)id & 15u);
    m -= 1u << (3u * (CO_META_TO_PLAY(m) * 3u + piece));
    b |= (uint64_t)((1u << piece) | 8u) << to4;
  } else {
    int is_place, piece, from, to;
    co_decode_move(id, &is_place, &piece, &from, &to);
    const uint32_t from4 = 4u * (uint32_t)from, to4 = 4u * (uint32_t)to;
    const uint64_t src = (b >> from4) & 7ull;
    b &= ~(15ull << from4);
    b |= (src | 8ull) << to4;
  }
  *board = b;
  *meta = m ^ (1u << 18);
}

/* bits {4 i + sh .. 4 i + sh + 2}, i = 0..3, of a 16-cell mask as 12 consecutive bits */
CO_DEV uint32_t co_pack3(uint32_t m, int sh) {
  return ((m >> sh) & 7u) | (((m >> (4 + sh)) & 7u) << 3) | (((m >> (8 + sh)) & 7u) << 6) | (((m >> (12 + sh)) & 7u) << 9);
}

/* line_breakers (util.h:85-637) in the workgroup's LDS: the four categories' masks are four DEPENDENT fetches otherwise.
 * Every wavefront writes the same words before it reads any (no workgroup barrier is needed, nor wanted). */
#define CO_LB_WORDS (CO_NUM_LINES * 3)
CO_DEV void co_line_breakers_to_lds(uint32_t *lb) {
  FOR_LANES {
    const uint32_t *flat = &CO_LINE_BREAKERS[0][0];
#pragma unroll
    for (int k = 0; k < (CO_LB_WORDS + CO_WAVE - 1) / CO_WAVE; ++k) {
      const int i = k * CO_WAVE + lane;
      if (i < CO_LB_WORDS) lb[i] = flat[i];
    }
  }
  WAVE_SYNC();
}

/* game.cpp:28-43 for the four rows' positions; o0..o2 = the rows' 96-bit legal masks.  `on` = rows that hold a position;
 * lb = co_line_breakers_to_lds's copy. */
CO_DEV void co_legal_moves_rows(LVP(uint32_t, blo), LVP(uint32_t, bhi), LVP(uint32_t, meta), LVP(int, on), LVP(uint32_t, o0),
                                LVP(uint32_t, o1), LVP(uint32_t, o2), LVP(int, lines), const uint32_t *lb) {
  LV(int, pb);
  LV(int, pc);
  LV(int, pa);
  LV(int, pf);
  LV(uint32_t, cd0);
  LV(uint32_t, cd1);
  LV(uint32_t, cd2);
  FOR_LANES_HOT {
    const int c = lane & 15;
    const uint32_t nib = ((c < 8 ? L(blo) : L(bhi)) >> (4 * (c & 7))) & 15u;
    L(pb) = (int)(nib & 1u);
    L(pc) = (int)((nib >> 1) & 1u);
    L(pa) = (int)((nib >> 2) & 1u);
    L(pf) = (int)((nib >> 3) & 1u);
    L(cd0) = CO_LINE_CAND[c]; /* this lane's three candidate lines: c, c + 16, c + 32 */
    L(cd1) = CO_LINE_CAND[c + 16];
    L(cd2) = CO_LINE_CAND[c + 32];
  }
  LV(uint32_t, B);
  LV(uint32_t, C);
  LV(uint32_t, A);
  LV(uint32_t, F);
  ROW_BALLOT(B, pb);
  ROW_BALLOT(C, pc);
  ROW_BALLOT(A, pa);
  ROW_BALLOT(F, pf);
  /* ---- lines: a candidate is a line iff all its cells lie in one top-piece plane (game.cpp:249-405) */
  LV(int, mt0);
  LV(int, mt1);
  LV(int, mt2);
  LV(uint32_t, pk); /* tops of this lane's candidates (2 bits each) | their line_breakers bases (7 bits each) */
  FOR_LANES_HOT {
    const uint32_t T2 = L(A), T1 = L(C) & ~L(A), T0 = L(B) & ~L(C) & ~L(A);
    uint32_t p = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t cd = k == 0 ? L(cd0) : k == 1 ? L(cd1) : L(cd2);
      const uint32_t M = cd & 0xFFFFu;
      const uint32_t tt = (T0 & M) == M ? 0u : (T1 & M) == M ? 1u : (T2 & M) == M ? 2u : 3u;
      const int hit = M != 0u && tt < 3u && L(on);
      if (k == 0) L(mt0) = hit;
      else if (k == 1) L(mt1) = hit;
      else L(mt2) = hit;
      p |= (tt & 3u) << (2 * k);
      p |= (cd >> 16) << (6 + 7 * k);
    }
    L(pk) = p;
  }
  LV(uint32_t, cb0);
  LV(uint32_t, cb1);
  LV(uint32_t, cb2);
  ROW_BALLOT(cb0, mt0);
  ROW_BALLOT(cb1, mt1);
  ROW_BALLOT(cb2, mt2);
  LV(uint32_t, m0);
  LV(uint32_t, m1);
  LV(uint32_t, m2);
  LV(int, anyl);
  FOR_LANES_HOT {
    L(m0) = L(m1) = L(m2) = 0xFFFFFFFFu;
    L(anyl) = (L(cb0) | L(cb1) | L(cb2)) != 0u;
    L(lines) = L(anyl); /* is_lines (game.cpp:28-43): the mover of a position without legal moves has lost iff there is a line */
  }
  if (WAVE_BALLOT(anyl)) {
    /* one line per category, first match in scan order (game.cpp:265,310,330-356,368-388) */
#pragma unroll
    for (int cat = 0; cat < 4; ++cat) {
      LV(int, jj); /* the category's first matching candidate, or -1 */
      LV(int, col);
      FOR_LANES_HOT {
        const uint32_t lo = L(cb0) | (L(cb1) << 16);
        const uint32_t cm = cat == 0 ? (lo & 0xFFFu) : cat == 1 ? (lo & 0xFFF000u) : cat == 2 ? (lo & 0x3F000000u) : 0u;
        int j = -1;
        if (cat < 3) {
          if (cm) j = co_ffs64((uint64_t)cm) - 1;
        } else {
          const uint32_t c3 = (lo >> 30) | ((L(cb2) & 3u) << 2);
          if (c3) j = 30 + co_ffs64((uint64_t)c3) - 1;
        }
        L(jj) = j;
        L(col) = j & 15;
      }
      LV(int, has);
      FOR_LANES_HOT { L(has) = L(jj) >= 0; }
      if (!WAVE_BALLOT(has)) continue;
      LV(uint32_t, sp);
      ROW_SHFL_U32(sp, pk, col);
      FOR_LANES_HOT {
        if (L(jj) >= 0) {
          const int j = L(jj), k = j >> 4;
          const int tt = (int)((L(sp) >> (2 * k)) & 3u);
          const int line = (int)((L(sp) >> (6 + 7 * k)) & 127u) + tt;
          L(m0) &= lb[3 * line];
          L(m1) &= lb[3 * line + 1];
          L(m2) &= lb[3 * line + 2];
          if (cat < 2 && tt == 2) {
            const int kq = (cat == 0 ? j : j - 12) % 3;
            if (kq != 0) {
              /* capital triple in a row/column: game.cpp:280-309 */
              const int is_col = cat == 1;
              const int ec = kq == 1 ? 3 : 0;
              for (int kk = 0; kk < 4; ++kk) {
                const int cell = is_col ? (ec * 4 + kk) : (kk * 4 + ec);
                if ((L(A) >> cell) & 1u) continue;
                if (kk > 0) {
                  const int id = is_col ? (24 + ec * 3 + (kk - 1)) : (36 + (kk - 1) * 4 + ec);
                  if (id < 32) L(m0) &= ~(1u << id);
                  else L(m1) &= ~(1u << (id - 32));
                }
                if (kk < 3) {
                  const int id = is_col ? (ec * 3 + kk) : (12 + kk * 4 + ec);
                  if (id < 32) L(m0) &= ~(1u << id);
                  else L(m1) &= ~(1u << (id - 32));
                }
              }
            }
          }
        }
      }
    }
  }
  /* ---- stack moves (canMove, game.cpp:222-232) on the planes: a column-bottomed stack without a base onto a bare base,
   * a bare capital onto a column-topped stack, neither frozen; then placements (canPlace, game.cpp:193-220) */
  FOR_LANES_HOT {
    const uint32_t b = L(B), c = L(C), a = L(A), f = L(F);
    const uint32_t T1 = c & ~a, T0 = b & ~c & ~a;
    const uint32_t s1 = c & ~b & ~f, d1 = b & ~c & ~a & ~f, s2 = a & ~b & ~c & ~f, d2 = c & ~a & ~f;
    const uint32_t Rm = ((s1 & (d1 >> 1)) | (s2 & (d2 >> 1))) & 0x7777u;
    const uint32_t Dm = ((s1 & (d1 >> 4)) | (s2 & (d2 >> 4))) & 0x0FFFu;
    const uint32_t Lm = ((s1 & (d1 << 1)) | (s2 & (d2 << 1))) & 0xEEEEu;
    const uint32_t Um = ((s1 & (d1 << 4)) | (s2 & (d2 << 4))) & 0xFFF0u;
    const uint32_t mv_lo = co_pack3(Rm, 0) | (Dm << 12) | (co_pack3(Lm, 1) << 24); /* ids 0 .. 31 (left moves 24 .. 35 straddle) */
    const uint32_t mv_hi = (co_pack3(Lm, 1) >> 8) | ((Um >> 4) << 4);             /* ids 32 .. 47 */
    const uint32_t E = ~(b | c | a) & 0xFFFFu;
    const uint32_t mine = CO_META_TO_PLAY(L(meta)) ? L(meta) >> 9 : L(meta);
    const uint32_t qb = (mine & 7u) ? E : 0u;
    const uint32_t qc = (mine & 0x38u) ? (E | (T0 & ~f)) : 0u;
    const uint32_t qa = (mine & 0x1C0u) ? (E | (T1 & ~f)) : 0u;
    L(o0) = L(m0) & mv_lo;
    L(o1) = L(m1) & (mv_hi | (qb << 16));
    L(o2) = L(m2) & (qc | (qa << 16));
  }
}

/* co_legal_moves with its per-lane constants made on the spot: for callers off the hot path (roots, the sequential
 * simulation), so that the constants do not live in registers across the whole step */
CO_DEV int co_legal_moves1(uint64_t board, uint32_t meta, uint32_t out[3]) {
  CoLanes K;
  co_lanes_init(K);
  return co_legal_moves(board, meta, out, K);
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
