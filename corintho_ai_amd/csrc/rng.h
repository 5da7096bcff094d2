// rng.h -- std::mt19937 per game, state in HBM, lane-parallel twist and draw.
//
// The reference gives every game one std::mt19937 shared by its two searchers
// (selfplayer.cpp:23-28) and consumes raw 32-bit outputs: one per legal move
// of every evaluated leaf (trainmc.cpp:236-246) and one per opening ply
// (trainmc.cpp:407).  The stream must be reproduced exactly.
// MT19937-32: n 624, m 397, a 0x9908B0DF, tempering (11; 7,0x9D2C5680;
// 15,0xEFC60000; 18).  The twist's data dependences (x[i] needs new x[i-227]
// for i >= 227 and old x[i+1]) allow ten 64-lane chunks in index order.
#pragma once
#include "engine_defs.h"
#include "wave.h"

CO_DEV uint32_t co_mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* std::mt19937(seed): x[0] = seed, x[i] = 1812433253 (x[i-1] ^ (x[i-1] >> 30)) + i -- a serial recurrence, run by
 * one lane (a slot of the resident pool starting its next game, once per game) */
CO_DEV void co_mt_seed(uint32_t *mt, uint32_t seed) {
  FOR_LANES {
    if (lane == 0) {
      uint32_t x = seed;
      mt[0] = x;
      for (int i = 1; i < CO_MT_N; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        mt[i] = x;
      }
    }
  }
  WAVE_SYNC();
}

/* mt = this game's 624 state words */
CO_DEV void co_mt_twist(uint32_t *mt) {
  for (int c = 0; c < (CO_MT_N + CO_WAVE - 1) / CO_WAVE; ++c) {
    LV(uint32_t, nv);
    FOR_LANES {
      int i = c * CO_WAVE + lane;
      if (i < CO_MT_N) {
        int i1 = i + 1 == CO_MT_N ? 0 : i + 1;
        int im = i + 397 >= CO_MT_N ? i + 397 - CO_MT_N : i + 397;
        uint32_t y = (mt[i] & 0x80000000u) | (mt[i1] & 0x7fffffffu);
        L(nv) = mt[im] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
    }
    WAVE_SYNC();
    FOR_LANES {
      int i = c * CO_WAVE + lane;
      if (i < CO_MT_N) mt[i] = L(nv);
    }
    WAVE_SYNC();
  }
}

/* The same through a staging copy in the wavefront's LDS (st: CO_MT_STAGE words).  The twist above is ten dependent
 * trips to memory (a chunk reads what the chunk before it stored), 25-40 k cycles at the end of three steps out of four
 * (round 5 stamps: a fifth of the search kernel's wave time); here the old state is fetched in one trip, the ten dependent
 * passes run in LDS, and the new words are stored behind them -- and stay in st for the caller. */
#define CO_MT_STAGE 640
/* the state -> st */
CO_DEV void co_mt_stage_load(const uint32_t *mt, uint32_t *st) {
  LV(uint32_t, x[10]);
  FOR_LANES {
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const int i = c * CO_WAVE + lane;
      L(x[c]) = mt[i < CO_MT_N ? i : 0];
    }
  }
  FOR_LANES {
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const int i = c * CO_WAVE + lane;
      if (i < CO_MT_N) st[i] = L(x[c]);
    }
  }
  WAVE_SYNC();
}
/* st holds the state: twist it there, then store the new state to mt.  (The stores come BEHIND the ten passes, in one
 * run: issued inside the passes, between the LDS reads each pass waits for, every one of them held the wave for
 * 1.6 k cycles -- 8 k in a wave that was slow anyway; round 5, tools/prof_phases.py: 20 k cycles per twist against 5 k.) */
CO_DEV void co_mt_twist_staged(uint32_t *mt, uint32_t *st) {
  for (int c = 0; c < (CO_MT_N + CO_WAVE - 1) / CO_WAVE; ++c) {
    LV(uint32_t, nv);
    FOR_LANES {
      int i = c * CO_WAVE + lane;
      if (i < CO_MT_N) {
        int i1 = i + 1 == CO_MT_N ? 0 : i + 1;
        int im = i + 397 >= CO_MT_N ? i + 397 - CO_MT_N : i + 397;
        uint32_t y = (st[i] & 0x80000000u) | (st[i1] & 0x7fffffffu);
        L(nv) = st[im] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
    }
    WAVE_SYNC();
    FOR_LANES {
      int i = c * CO_WAVE + lane;
      if (i < CO_MT_N) st[i] = L(nv);
    }
    WAVE_SYNC();
  }
  LV(uint32_t, x[10]);
  FOR_LANES {
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const int i = c * CO_WAVE + lane;
      L(x[c]) = st[i < CO_MT_N ? i : 0];
    }
  }
  FOR_LANES {
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const int i = c * CO_WAVE + lane;
      if (i < CO_MT_N) mt[i] = L(x[c]);
    }
  }
}

/* Draw `n` (<= 64) consecutive outputs: lane j < n receives output number j.
 * `idx` is the game's position in the state (uniform, updated). */
#define CO_MT_DRAW(mt, idx, n, outvar)                          \
  do {                                                          \
    int _done = 0;                                              \
    while (_done < (n)) {                                       \
      if ((idx) >= CO_MT_N) {                                   \
        co_mt_twist(mt);                                        \
        (idx) = 0;                                              \
      }                                                         \
      int _take = (n)-_done;                                    \
      if (_take > CO_MT_N - (idx)) _take = CO_MT_N - (idx);     \
      FOR_LANES {                                               \
        int _j = lane - _done;                                  \
        if (_j >= 0 && _j < _take) L(outvar) = co_mt_temper((mt)[(idx) + _j]); \
      }                                                         \
      (idx) += _take;                                           \
      _done += _take;                                           \
    }                                                           \
  } while (0)

/* one output, uniform */
CO_DEV uint32_t co_mt_next(uint32_t *mt, int *idx) {
  if (*idx >= CO_MT_N) {
    co_mt_twist(mt);
    *idx = 0;
  }
  uint32_t y = co_mt_temper(mt[*idx]);
  *idx += 1;
  return y;
}
