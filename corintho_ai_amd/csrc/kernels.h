// kernels.h -- kernel entry points of the self-play pool (wave style, see wave.h).
//   K3 co_k_mcts_step   one wavefront per game: Trainer::doIteration body
//   K4 co_k_scan        request offsets (Trainer::doIteration offsets[],
//                       trainer.cpp:169-174 / :208-215) + all-done flag
//   K4 co_k_compact     Trainer::writeRequests (trainer.cpp:79-101): per-game
//                       request rows -> one game-major batch
//   K7 co_k_write_samples  SelfPlayer::writeSamples (selfplayer.cpp:79-113)
//   test kernels for the rule layer and the floating-point contract
#pragma once
#include "mcts.h"

CO_CONST int32_t CO_SPACE_SYM[8][16] = CO_SPACE_SYM_INIT;
CO_CONST int32_t CO_MOVE_SYM[8][96] = CO_MOVE_SYM_INIT;

#ifdef CO_EMU
#define CO_K3_WAVES 1
#define CO_K3_WAVE_IN_BLOCK 0
#define CO_K3_BOUNDS
#else
/* the launch has exactly CO_K3_WPB wavefronts per workgroup: say so, or the register budget is that of a 1024-thread workgroup */
#define CO_K3_BOUNDS __launch_bounds__(CO_K3_WPB * 64)
#define CO_K3_WAVES CO_K3_WPB
/* wave-uniform by construction: tell the compiler, or the game index and everything derived from it live in vector registers */
#define CO_K3_WAVE_IN_BLOCK (CO_K3_WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0)
#endif
CO_K3_BOUNDS CO_KERNEL co_k_mcts_step(EngineParams P) {
  const int i = CO_BLOCK_IDX * CO_K3_WAVES + CO_K3_WAVE_IN_BLOCK;
  const int g = P.pool_lo + i;
  if (i < P.pool_n && g < P.num_games) co_mcts_step_wave(P, g);
}

/* is game g part of the batch of model `tp` (trainer.cpp:42-46, 84-98)? */
CO_DEV int co_game_active(const EngineParams &P, int g, const GameCtl &gc, int tp) {
  if (gc.done) return 0;
  if (P.pcfg) return P.pcfg[2 * g + gc.to_play].model_id == tp; /* tourney.cpp:26, 46, 66 */
  if (tp == 0 || tp == 1) return gc.to_play == (tp + gc.parity) % 2;
  return 1;
}

/* single wavefront: lane l owns a contiguous chunk of games */
CO_KERNEL co_k_scan(EngineParams P) {
  int G = P.num_games;
  /* fused arena: the entry scan works for the model staged by the previous iteration */
  const int tp = P.arena_state ? P.arena_state[P.scan_phase == 0 ? 2 : 0] : P.to_play;
  int chunk = (G + CO_WAVE - 1) / CO_WAVE;
  LV(int, csum);
  LV(int, nd);
  LV(int, er);
  FOR_LANES {
    int s = 0, notdone = 0, err = 0;
    for (int i = 0; i < chunk; ++i) {
      int g = lane * chunk + i;
      if (g < G) {
        GameCtl gc = P.games[g];
        if (co_game_active(P, g, gc, tp)) s += gc.n_pending;
        notdone += !gc.done;
        err |= gc.error;
      }
    }
    L(csum) = s;
    L(nd) = notdone;
    L(er) = err != 0;
  }
  int not_done = WAVE_SUM_I32(nd);
  int any_error = WAVE_BALLOT(er) != 0;
  /* exclusive scan of the 64 chunk sums (uniform serial loop: 64 adds) */
  WAVE_SHARED(int, cbase, CO_WAVE + 1);
  FOR_LANES { cbase[lane + 1] = L(csum); }
  WAVE_SYNC();
  int run = 0;
  LV(int, mybase);
  FOR_LANES { L(mybase) = 0; }
  for (int l = 0; l < CO_WAVE; ++l) {
    int v = cbase[l + 1];
    FOR_LANES {
      if (lane == l) L(mybase) = run;
    }
    run += v;
  }
  FOR_LANES {
    int s = L(mybase);
    for (int i = 0; i < chunk; ++i) {
      int g = lane * chunk + i;
      if (g < G) {
        GameCtl gc = P.games[g];
        P.req_offset[g] = s;
        if (co_game_active(P, g, gc, tp)) s += gc.n_pending;
      }
    }
    if (lane == 0 && P.read_offset) {
      /* Tourney::doIteration's own table (tourney.cpp:55-62): match i reads at
       * the running sum of num_requests(i - 1) over the ACTIVE i, not at its row of writeRequests */
      int offset = 0;
      P.read_offset[0] = 0;
      for (int i = 1; i < G; ++i) {
        if (co_game_active(P, i, P.games[i], tp)) offset += P.games[i - 1].n_pending;
        P.read_offset[i] = offset;
      }
    }
    if (lane == 0) {
      P.req_offset[G] = run;
      P.all_done[0] = not_done == 0;
      if (P.ctl) {
        P.ctl[0] = run;
        P.ctl[1] = not_done == 0;
        P.ctl[2] = any_error;
        P.ctl[3] = not_done;
      }
      if (P.row_counter && (!P.arena_state || P.scan_phase == 1)) P.row_counter[0] += (unsigned long long)run;
      if (P.arena_state) {
        if (P.scan_phase == 0) {
          P.arena_state[0] = tp; /* commit: the search, the compaction and the network of this iteration see it */
        } else {
          P.arena_state[3 + (tp == 0 ? 1 : 0)] = run; /* get_predictions, main.pyx:74-81: slot 1 serves model 0 */
          P.arena_state[3 + (tp == 0 ? 0 : 1)] = 0;
          /* main.pyx:150-154: a model without requests hands over to the other one */
          if (not_done != 0 && run == 0) {
            P.arena_state[2] = 1 - tp;
            P.arena_state[1] += 1;
          } else {
            P.arena_state[2] = tp;
            P.arena_state[1] = 0;
          }
        }
      }
    }
  }
}

CO_KERNEL co_k_compact(EngineParams P) {
  int g = CO_BLOCK_IDX;
  if (g >= P.num_games) return;
  GameCtl gc = P.games[g];
  if (!co_game_active(P, g, gc, P.arena_state ? P.arena_state[0] : P.to_play)) return;
  int n = gc.n_pending;
  const float *src = P.req + (size_t)g * P.searches_per_eval * CO_STATE_STRIDE;
  float *dst = P.nn_in + (size_t)P.req_offset[g] * CO_STATE_STRIDE;
  int total = n * CO_STATE_STRIDE;
  FOR_LANES {
    for (int i = lane; i < total; i += CO_WAVE) dst[i] = src[i];
  }
  if (P.nn_in70) {
    /* the rows as the caller's array holds them: 70 floats each, no padding (trainer.cpp:79-101) */
    float *d70 = P.nn_in70 + (size_t)P.req_offset[g] * CO_GAME_STATE_SIZE;
    int t70 = n * CO_GAME_STATE_SIZE;
    FOR_LANES {
      for (int i = lane; i < t70; i += CO_WAVE) d70[i] = src[(i / CO_GAME_STATE_SIZE) * CO_STATE_STRIDE + i % CO_GAME_STATE_SIZE];
    }
  }
}

/* rows of 70 floats (the caller's layout) -> rows of CO_STATE_STRIDE floats (the network kernels' input) */
CO_KERNEL co_k_expand_rows(const float *in70, float *out80, int rows, int nblocks) {
  FOR_LANES {
    int total = rows * CO_STATE_STRIDE;
    for (int i = CO_BLOCK_IDX * CO_WAVE + lane; i < total; i += CO_WAVE * nblocks) {
      int r = i / CO_STATE_STRIDE, c = i % CO_STATE_STRIDE;
      out80[i] = c < CO_GAME_STATE_SIZE ? in70[(size_t)r * CO_GAME_STATE_SIZE + c] : 0.0f;
    }
  }
}

/* one wavefront per GAME (not slot); sample_offset[g] = number of plies of games < g, meta[g] = plies | result << 8 */
CO_KERNEL co_k_write_samples(EngineParams P, int n_games, const int32_t *sample_offset, const int32_t *meta, float *game_states,
                             float *eval_samples, float *prob_samples) {
  int g = CO_BLOCK_IDX;
  if (g >= n_games) return;
  const int n = meta[g] & 255, result = meta[g] >> 8;
  size_t off = (size_t)sample_offset[g];
  const float *smp = P.samples + (size_t)g * CO_MAX_PLIES * CO_SAMPLE_FLOATS;
  for (int i = n - 1; i >= 0; --i) {
    /* the last mover wins unless the game is drawn; sign alternates backwards */
    float evaluation = result == CO_RESULT_DRAW ? 0.0f : 1.0f;
    if ((n - 1 - i) & 1) evaluation = (float)((double)evaluation * -1.0);
    const float *st = smp + (size_t)i * CO_SAMPLE_FLOATS;
    const float *pol = st + CO_GAME_STATE_SIZE;
    for (int k = 0; k < CO_NUM_SYMMETRIES; ++k) {
      float *gs = game_states + ((off + i) * CO_NUM_SYMMETRIES + k) * CO_GAME_STATE_SIZE;
      float *ps = prob_samples + ((off + i) * CO_NUM_SYMMETRIES + k) * CO_NUM_MOVES;
      FOR_LANES {
        gs[lane] = st[CO_SPACE_SYM[k][lane / 4] * 4 + lane % 4];
        if (lane < CO_GAME_STATE_SIZE - 64) gs[64 + lane] = st[64 + lane];
        ps[lane] = pol[CO_MOVE_SYM[k][lane]];
        if (lane < CO_NUM_MOVES - 64) ps[64 + lane] = pol[CO_MOVE_SYM[k][64 + lane]];
        if (lane == 0) eval_samples[(off + i) * CO_NUM_SYMMETRIES + k] = evaluation;
      }
    }
  }
}

/* un-augmented (state[70], policy[96]) rows + outcome, game-major, for the
 * multi-GPU gather (the x8 expansion happens after it) */
CO_KERNEL co_k_pack_samples(EngineParams P, int n_games, const int32_t *sample_offset, const int32_t *meta, float *state_policy,
                            float *outcome) {
  int g = CO_BLOCK_IDX;
  if (g >= n_games) return;
  const int n = meta[g] & 255, result = meta[g] >> 8;
  size_t off = (size_t)sample_offset[g];
  const float *smp = P.samples + (size_t)g * CO_MAX_PLIES * CO_SAMPLE_FLOATS;
  int total = n * CO_SAMPLE_FLOATS;
  FOR_LANES {
    for (int i = lane; i < total; i += CO_WAVE) state_policy[off * CO_SAMPLE_FLOATS + i] = smp[i];
    for (int i = lane; i < n; i += CO_WAVE) {
      float e = result == CO_RESULT_DRAW ? 0.0f : 1.0f;
      if ((n - 1 - i) & 1) e = (float)((double)e * -1.0);
      outcome[off + i] = e;
    }
  }
}

/* ---- test kernels: the rule layer on a batch of positions, one per wavefront */
CO_KERNEL co_k_rules_batch(const uint64_t *boards, const uint32_t *metas, int n, uint32_t *masks, int32_t *lines) {
  int i = CO_BLOCK_IDX;
  if (i >= n) return;
  uint32_t lm[3];
  CoLanes K;
  co_lanes_init(K);
  int l = co_legal_moves(boards[i], metas[i], lm, K);
  FOR_LANES {
    if (lane < 3) masks[i * 3 + lane] = lm[lane];
    if (lane == 0) lines[i] = l;
  }
}

CO_KERNEL co_k_domove_batch(uint64_t *boards, uint32_t *metas, const int32_t *moves, int n, float *states) {
  int i = CO_BLOCK_IDX;
  if (i >= n) return;
  uint64_t b = boards[i];
  uint32_t m = metas[i];
  CoLanes K;
  co_lanes_init(K);
  if (moves[i] >= 0) co_do_move(&b, &m, moves[i], K);
  WAVE_SYNC();
  FOR_LANES {
    if (lane == 0) {
      boards[i] = b;
      metas[i] = m;
    }
  }
  co_write_state(b, m, states + (size_t)i * CO_STATE_STRIDE);
}

/* the four-positions-per-wavefront rule layer (rules.h co_do_move_lane, co_legal_moves_rows): position 4 b + r in row r */
CO_KERNEL co_k_rules_rows(uint64_t *boards, uint32_t *metas, const int32_t *moves, int n, uint32_t *masks) {
  const int base = CO_BLOCK_IDX * 4;
  LV(uint32_t, blo);
  LV(uint32_t, bhi);
  LV(uint32_t, mt);
  LV(int, on);
  FOR_LANES {
    const int i = base + (lane >> 4);
    L(on) = i < n;
    uint64_t b = L(on) ? boards[i] : 0ull;
    uint32_t m = L(on) ? metas[i] : 0u;
    if (L(on) && moves[i] >= 0) co_do_move_lane(&b, &m, moves[i]);
    L(blo) = (uint32_t)b;
    L(bhi) = (uint32_t)(b >> 32);
    L(mt) = m;
  }
  LV(uint32_t, o0);
  LV(uint32_t, o1);
  LV(uint32_t, o2);
  WG_SHARED(uint32_t, lb, CO_LB_WORDS);
  co_line_breakers_to_lds(lb);
  LV(int, lines);
  co_legal_moves_rows(blo, bhi, mt, on, o0, o1, o2, lines, lb);
  FOR_LANES {
    const int i = base + (lane >> 4);
    if (L(on) && (lane & 15) == 0) {
      boards[i] = (uint64_t)L(blo) | ((uint64_t)L(bhi) << 32);
      metas[i] = L(mt);
      masks[i * 3 + 0] = L(o0);
      masks[i * 3 + 1] = L(o1);
      masks[i * 3 + 2] = L(o2);
    }
  }
}

/* mt19937: game 0's generator, `n` outputs through the wave draw path */
CO_KERNEL co_k_rng_draw(uint32_t *mt, int32_t *idx_io, int n, int chunk, uint32_t *out) {
  int idx = idx_io[0];
  int done = 0;
  while (done < n) {
    int cnt = n - done < chunk ? n - done : chunk;
    LV(uint32_t, r);
    CO_MT_DRAW(mt, idx, cnt, r);
    FOR_LANES {
      if (lane < cnt) out[done + lane] = L(r);
    }
    done += cnt;
  }
  FOR_LANES {
    if (lane == 0) idx_io[0] = idx;
  }
}

/* The floating-point expressions whose exact rounding the search depends on
 * (trainmc.cpp:230,242,257,266,549,565-568).  One lane per input row:
 *   in[i]  = {c_puct, visits, eval, prob9, denom, cvisits, sum, eps}
 *   out[i] = {v_sqrt, u_visited, u_unvisited, scalar, dscalar, 511/x, 1/(float)n} */
CO_KERNEL co_k_fp_probe(const float *in, int n, float *out) {
  FOR_LANES {
    int i = CO_BLOCK_IDX * CO_WAVE + lane;
    if (i < n) {
      const float *x = in + (size_t)i * 8;
      float c_puct = x[0], visits = x[1], eval = x[2], p9 = x[3], denom = x[4], cv = x[5], sum = x[6], eps = x[7];
      float v_sqrt = (float)((double)c_puct * co_sqrt_f64((double)visits));
      float prob = p9 * denom;
      float pv = prob * v_sqrt;
      /* the search's own quotient routine where it applies (integer counts up to 2^15) */
      int icv = (int)cv;
      double a, b;
      if ((float)icv == cv && icv >= 1 && icv < 32768) {
        a = co_div_small(-(double)eval, cv);
        b = co_div_small((double)pv, cv + 1.0f);
      } else {
        a = -1.0 * (double)eval / (double)cv;
        b = (double)pv / ((double)cv + 1.0);
      }
      float one_minus = (float)1 - eps;
      float *o = out + (size_t)i * 8;
      o[0] = v_sqrt;
      o[1] = (float)(a + b);
      o[2] = pv;
      o[3] = (float)(1.0 / (double)sum * (double)one_minus);
      o[4] = (float)(1.0 / (double)sum * (double)eps);
      o[5] = 511.0f / sum;
      o[6] = (float)(1.0 / (double)(float)(int)cv);
      float xq = p9 * (511.0f / sum);
      float fl = __builtin_truncf(xq);
      int q = (int)fl;
      if (xq - fl >= 0.5f) q += 1;
      o[7] = (float)q;
    }
  }
}
