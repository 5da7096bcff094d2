// mcts.h -- one wavefront advances one game: select / expand / backup, prior
// quantisation with Dirichlet noise, move choice, re-rooting and the hand-over
// between the two players' trees.
//
// Semantics follow the reference line by line (file:line cited at each
// function); the data layout does not (engine_defs.h).  All floating-point
// expressions keep the reference's mixed float/double evaluation order
// (SURVEY 7 "hard parts" 1): this file must be compiled with
// -ffp-contract=off and correctly rounded fp32 division.
#pragma once
#include "engine_defs.h"
#include "rng.h"
#include "rules.h"
#include "wave.h"

#ifndef CO_EMU
#pragma clang fp contract(off)
#endif

CO_CONST uint32_t CO_GAMMA_BITS[CO_NUM_GAMMA] = CO_GAMMA_BITS_INIT;

#define CO_NEG_INF (-__builtin_huge_valf())

/* In-kernel phase stamps (diagnostic build only: hipcc -DCO_PROF, tools/prof_phases.py).  A wavefront keeps ONE running
 * clock (w.tph): CO_PH(slot) charges the cycles since the previous stamp to `slot`, so the slots add up to the wave's
 * whole step; the sums live in registers and are written once, at the end of the step (a stamp is an s_memtime and a
 * wait for it, ~50 cycles).  CO_PH_MEM(slot) first waits for the wave's outstanding memory operations, so that a phase
 * that ends with loads in flight is charged their latency (it also removes the overlap the product build has).
 * Slots: 0 backup: indices, 1 backup: slots fetched, 2 move choice / hand-over, 3 backup: sums and stores,
 *   4 steps (count), 5 simulations (count), 6 evaluations received (count), 7 whole step (cycles, measured apart),
 *   8 PUCT scan, 9 virtual-loss store + path, 10 expansion: slot + path update, 11 descent: block fetch waited for,
 *   12 terminal leaf, 13 (unused), 14 expansion: legal moves, 15 expansion: node stores, 16 expansion: doMove,
 *   17 request: state row, 18 request: pending-leaf records, 19 simulation start (root copy), 20 loop control + root
 *   load, 21 step tail (noise, batch rows, cache), 22 wave set-up, 23 levels scanned (count), 24 expansions (count) */
#define CO_NPROF 32
#if defined(CO_PROF) && !defined(CO_EMU)
#define CO_CLK() __builtin_amdgcn_s_memtime()
#define CO_PROF_ADD(w, slot, v) ((w).pacc[slot] += (v))
#define CO_PH(slot)                               \
  do {                                            \
    unsigned long long now_ = CO_CLK();           \
    w.pacc[slot] += now_ - w.tph;                 \
    w.tph = now_;                                 \
  } while (0)
#ifdef CO_PROF_LIGHT /* stamps that wait for nothing: a phase is charged the waits the PRODUCT has in it */
#define CO_PH_MEM(slot) CO_PH(slot)
#else
#define CO_PH_MEM(slot)                                               \
  do {                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       \
    CO_PH(slot);                                                      \
  } while (0)
#endif
#else
#define CO_CLK() 0ull
#define CO_PROF_ADD(w, slot, v) ((void)0)
#define CO_PH(slot) ((void)0)
#define CO_PH_MEM(slot) ((void)0)
#endif

/* What a step's budget is counted in (EngineParams::step_budget).  The device build measures TIME -- ticks of the 100 MHz
 * real-time counter since the step began: what a launch waits for is its slowest wavefront's time, whatever made it slow
 * (depth, terminal leaves and their propagation, a SIMD shared with another pool's kernel) -- and the emulation build, which
 * has no clock and must stay deterministic, counts PUCT scans (1 scan ~ 1.6 us).  Either way ONE scalar lives across the
 * step: the deadline (device) or the scans so far minus the budget (emulation). */
#if !defined(CO_EMU) && !defined(CO_BUDGET_SCANS)
#define CO_STEP_CLOCK() ((uint32_t)__builtin_amdgcn_s_memrealtime())
#define CO_STEP_UNITS_PER_CONFIG_UNIT 100 /* ca_config.step_budget > 0 is in microseconds */
#define CO_STEP_WORK(w, n) ((void)0)
#define CO_STEP_START(budget) ((int)(CO_STEP_CLOCK() + (uint32_t)(budget)))
#define CO_STEP_SPENT(w) ((int)(CO_STEP_CLOCK() - (uint32_t)(w).work) >= 0)
#define CO_STEP_DONE(w, budget) ((int)(CO_STEP_CLOCK() - ((uint32_t)(w).work - (uint32_t)(budget))))
#else
#define CO_STEP_UNITS_PER_CONFIG_UNIT 1 /* ca_config.step_budget > 0 is in scans */
#define CO_STEP_WORK(w, n) ((w).work += (n))
#define CO_STEP_START(budget) (-(budget))
#define CO_STEP_SPENT(w) ((w).work >= 0)
#define CO_STEP_DONE(w, budget) ((w).work + (budget))
#endif

struct CoTree {
  uint4 *A;      /* arena of this tree */
  TreeCtl tc;    /* register copy, written back at the end of the step */
  uint32_t cap;
};

struct CoWave {
  int g;
  GameCtl gc;
#if defined(CO_PROF) && !defined(CO_EMU)
  unsigned long long pacc[CO_NPROF], tph;
#endif
  /* the tree of the player to move and the opponent's; swapped on hand-over so that
   * no register-resident state is indexed dynamically (that would spill to scratch) */
  CoTree me, opp;
  uint32_t *mt;
  const uint32_t *gamma; /* LDS: gamma_samples (util.h), CO_NUM_GAMMA words */
  const uint32_t *lb; /* LDS: line_breakers (rules.h co_line_breakers_to_lds) */
  int mt_staged;      /* mt_stage holds the generator's current state */
  uint32_t *mt_stage; /* LDS, CO_MT_STAGE words: staging copy of the generator state for a twist (rng.h co_mt_twist_lds) */
  /* per-game views */
  uint32_t *pend_leaf;
  int32_t *pend_depth;
  uint32_t *pend_path;
  uint32_t *pend_n;
  uint4 *pend_key;      /* cache keys of the pending leaves' rows, or null */
  const float *cval;    /* evaluation cache: outputs live in cval[csrc[k] * CO_CACHE_VAL_FLOATS] for pending leaf k; null = in place */
  const int32_t *csrc;
  uint32_t *noise_raw;
  int noise_words; /* generator outputs owed to the leaves queued so far (in this step, and in the steps before it that were
                    * cut short with their leaves held back: GameCtl::held) */
  int work; /* CO_STEP_*: the step's deadline (device), or its PUCT scans so far MINUS its budget (emulation) -- the step stops
             * selecting when the clock passes it / when it reaches zero (one live scalar instead of two: with the budget kept
             * beside a count the kernel spills 170 more SGPRs) */
  /* the records of up to CO_PRE pending leaves, in the wavefront's LDS: pre[f * CO_PRE + k] = field f (CO_PRE_*) of leaf
   * pre_c0 + k, k < pre_n.  Requested with the game's other first loads (co_mcts_step_wave) -- in registers until
   * co_receive_eval they cost 44 spilled registers in the search -- and read by co_receive_eval */
  int pre_n, pre_c0;
  uint32_t *pre;
  float *req;
  float *samples;
  int32_t *trace;
  int32_t *log; /* this game's text-log record (EngineParams::log), or null */
  unsigned long long *prof;
  /* config */
  int max_searches, spe, testing, trace_on, defer_handover, analyse, force_choose;
  float c_puct, epsilon;
  const PlayerCfg *pc; /* tournament match: the two players' settings, else null */
};

/* ---- slot helpers.  slot = {child, eval bits, mp | visits<<16, result | all_visited<<8} */
CO_DEV int co_slot_visits(uint4 s) { return (int)(int16_t)(s.z >> 16); }
CO_DEV uint4 co_slot_set_visits(uint4 s, int v) {
  s.z = (s.z & 0xFFFFu) | ((uint32_t)(uint16_t)(int16_t)v << 16);
  return s;
}
CO_DEV int co_slot_result(uint4 s) { return (int)(s.w & 0xFFu); }
CO_DEV int co_slot_all_visited(uint4 s) { return (int)((s.w >> 8) & 1u); }
CO_DEV uint4 co_slot_set_result(uint4 s, int r) {
  s.w = (s.w & ~0xFFu) | (uint32_t)r;
  return s;
}
CO_DEV uint4 co_slot_set_all_visited(uint4 s, int a) {
  s.w = (s.w & ~0x100u) | ((uint32_t)(a & 1) << 8);
  return s;
}
/* node.cpp:96-118 */
CO_DEV int co_res_terminal(int r) { return r == CO_RESULT_LOSS || r == CO_RESULT_DRAW; }
CO_DEV int co_res_known(int r) { return r != CO_RESULT_NONE; }
CO_DEV int co_res_won(int r) { return r == CO_DEDUCED_WIN; }
CO_DEV int co_res_lost(int r) { return r == CO_RESULT_LOSS || r == CO_DEDUCED_LOSS; }
CO_DEV int co_res_drawn(int r) { return r == CO_RESULT_DRAW || r == CO_DEDUCED_DRAW; }

/* uniform 16-byte read / single-lane write of one unit */
CO_DEV uint4 co_load_unit(const uint4 *A, uint32_t off) { return A[off]; }
CO_DEV void co_store_unit(uint4 *A, uint32_t off, uint4 v) {
  FOR_LANES {
    if (lane == 0) A[off] = v;
  }
  WAVE_SYNC();
}

CO_DEV void co_trace_push(CoWave &w, int32_t v) {
  if (!w.trace_on) return;
  if (w.gc.trace_len < CO_TRACE_CAP) {
    int at = w.gc.trace_len;
    FOR_LANES {
      if (lane == 0) w.trace[at] = v;
    }
  }
  w.gc.trace_len++;
}

/* Write the node of a position whose legal moves (lm, n of them) are known -- Node ctors node.cpp:14-39 +
 * initializeEdges node.cpp:256-283: bump allocation, header, edge slots in ascending move id.  self_slot == CO_NONE
 * makes a detached root that carries its own stat slot.  Returns the block offset (CO_NONE on arena overflow). */
CO_DEV uint32_t co_emit_node(CoWave &w, CoTree &t, uint64_t board, uint32_t meta_game, int depth, uint32_t parent,
                             uint32_t self_slot, const uint32_t lm[3], int n, int res) {
  const int n01 = co_popc32(lm[0]) + co_popc32(lm[1]);
  uint32_t units = 2u + (uint32_t)n + (self_slot == CO_NONE ? 1u : 0u);
  if (t.tc.units_used + units > t.cap) {
    w.gc.error |= CO_ERR_ARENA_FULL;
    return CO_NONE;
  }
  uint32_t b = t.tc.units_used;
  t.tc.units_used += units;
  if (t.tc.units_used > t.tc.peak_units) t.tc.peak_units = t.tc.units_used;
  w.gc.nodes++;
  uint32_t own = self_slot == CO_NONE ? b + 2u + (uint32_t)n : self_slot;
  uint32_t meta = co_meta_make(meta_game, depth, n);
  uint4 *A = t.A;
  FOR_LANES {
    if (lane == 0) A[b] = make_uint4((uint32_t)board, (uint32_t)(board >> 32), meta, parent);
    if (lane == 1) A[b + 1] = make_uint4(own, 0u, 0u, 0u);
    if (lane == 2 && self_slot == CO_NONE)
      A[own] = make_uint4(CO_NONE, 0u, 1u << 16, (uint32_t)res | 0x100u); /* visits 1, all_visited (node.h:164,186) */
    /* edges in ascending move id: rank = number of legal moves below this id */
    const uint64_t lm01 = (uint64_t)lm[0] | ((uint64_t)lm[1] << 32);
    if ((lm01 >> lane) & 1ull) A[b + 2 + LANE_RANK64(lm01)] = make_uint4(CO_NONE, 0u, (uint32_t)lane, 0u);
    if (((uint64_t)lm[2] >> lane) & 1ull) A[b + 2 + n01 + LANE_RANK64((uint64_t)lm[2])] = make_uint4(CO_NONE, 0u, (uint32_t)(64 + lane), 0u);
  }
  WAVE_SYNC();
  CO_PH(15);
  return b;
}

/* Create the node for position (board, meta_game): legal moves (game.cpp:28-43), terminal result (node.cpp:256-271),
 * then co_emit_node.  *res_out = kResultLoss / kResultDraw / kResultNone. */
CO_DEV uint32_t co_create_node(CoWave &w, CoTree &t, uint64_t board, uint32_t meta_game, int depth, uint32_t parent,
                               uint32_t self_slot, int *res_out, int *n_out = (int *)0, uint32_t *lm_out = (uint32_t *)0) {
  uint32_t lm[3];
  int is_lines = co_legal_moves1(board, meta_game, lm);
  CO_PH(14);
  if (lm_out) {
    lm_out[0] = lm[0];
    lm_out[1] = lm[1];
    lm_out[2] = lm[2];
  }
  int n = co_popc32(lm[0]) + co_popc32(lm[1]) + co_popc32(lm[2]);
  int res = CO_RESULT_NONE;
  if (n == 0) res = is_lines ? CO_RESULT_LOSS : CO_RESULT_DRAW;
  *res_out = res;
  if (n_out) *n_out = n;
  return co_emit_node(w, t, board, meta_game, depth, parent, self_slot, lm, n, res);
}

/* A leaf asks for a network evaluation: TrainMC writes the state into to_eval_
 * and records the node in searched_ (trainmc.cpp:684-692, 150-153). */
CO_DEV void co_request(CoWave &w, uint64_t board, uint32_t meta, uint32_t leaf, int n_edges, const uint32_t lm[3], int D,
                       const uint32_t *path_slot) {
  int k = w.gc.n_pending;
  co_write_state(board, meta, w.req + (size_t)k * CO_STATE_STRIDE);
  CO_PH(17);
  uint32_t *pp = w.pend_path + (size_t)k * CO_PATH_MAX;
  const uint32_t pn = ((uint32_t)w.noise_words << 8) | (uint32_t)n_edges;
  w.noise_words += n_edges; /* one generator output per legal move (trainmc.cpp:236-246) */
  FOR_LANES {
    if (lane == 0) {
      w.pend_leaf[k] = leaf;
      w.pend_depth[k] = D;
    }
    if (lane < 4) w.pend_n[4 * k + lane] = lane == 0 ? pn : lm[lane == 1 ? 0 : lane == 2 ? 1 : 2];
    if (lane <= D && lane < CO_PATH_MAX) pp[lane] = path_slot[lane];
    if (lane == 0 && w.pend_key) {
      /* the request row as a key: the board and the reserves in the order Game::writeGameState lays them out (the
       * mover's first, game.cpp:53-57) -- two positions with the same key have the same 70 floats */
      const uint32_t pc = meta & 0x3FFFFu;
      const uint32_t rot = ((meta >> 18) & 1u) ? ((pc >> 9) | (pc << 9)) & 0x3FFFFu : pc;
      w.pend_key[k] = make_uint4((uint32_t)board, (uint32_t)(board >> 32), rot | 0x80000000u, 0u);
    }
  }
  WAVE_SYNC();
  w.gc.n_pending = k + 1;
  CO_PH(18);
}

/* ---- receiveEval (trainmc.cpp:269-296), split in two.
 *
 * What a leaf's evaluation does to the tree has two parts.  (1) The leaf's own priors:
 * getFilteredProbs (trainmc.cpp:212-234), generateDirichlet (:236-246), setProbs (:248-267) -- a
 * function of the network's row for that leaf, of the leaf's legal moves and of the generator
 * outputs owed to it, and of nothing else.  (2) The backup of the value along the path to the root
 * (:280-295), which touches slots shared with other leaves and must run in request order.
 * Round 1 ran both inside the game's wavefront, leaf after leaf: 37 % of the search kernel's wave
 * cycles, serial per game, with ~28 of 64 lanes busy.  Part (1) is independent per LEAF: rounds 2-4 ran
 * it as a kernel of its own in front of the search launch (one wavefront per pending leaf); since round 5
 * it is back inside the step, FOUR LEAVES AT A TIME, one per 16-lane row of the game's wavefront
 * (co_prior_all / co_prior_rows below, called from co_receive_eval in front of part (2)) -- no launch of
 * its own, and no second kernel that has to agree with this one on which games step.
 *
 * The generator: a game's two searchers share one std::mt19937 (selfplayer.cpp:23-28) and the
 * reference draws a leaf's noise when the leaf's evaluation is received.  Between queueing a leaf
 * and receiving its evaluation the game draws nothing else (a move is only chosen with no request
 * pending, trainmc.cpp:139-178), so the outputs a leaf will consume are fixed when it is queued:
 * co_capture_noise copies them (raw state words, tempered by the consumer) into the game's noise
 * buffer at the end of the step and advances the generator -- the stream is consumed in the
 * reference's order, leaf by leaf in request order. */

/* the generator outputs owed to the leaves queued in this step -> noise_raw[0 .. noise_words) */
CO_DEV void co_capture_noise(CoWave &w) {
  int done = 0;
  const int total = w.noise_words;
  int staged = w.mt_staged; /* the state is in w.mt_stage: fetched when the step began (co_mcts_step_wave), in front of
                             * the step's stores -- a fetch here would stand behind all of them -- or by a twist below */
  while (done < total) {
    if (w.gc.rng_idx >= CO_MT_N) {
      if (!staged) co_mt_stage_load(w.mt, w.mt_stage);
      CO_PH(30);
      co_mt_twist_staged(w.mt, w.mt_stage);
      CO_PH(31);
      w.gc.rng_idx = 0;
      staged = 1;
    }
    int cnt = total - done;
    if (cnt > CO_MT_N - w.gc.rng_idx) cnt = CO_MT_N - w.gc.rng_idx;
    uint32_t *dst = w.noise_raw + done;
    if (staged) {
      /* (LDS reads first, then the stores in one run: see co_mt_twist_staged) */
      const uint32_t *src = w.mt_stage + w.gc.rng_idx;
      for (int base = 0; base < cnt; base += 5 * CO_WAVE) {
        LV(uint32_t, v[5]);
        FOR_LANES {
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int i = base + j * CO_WAVE + lane;
            L(v[j]) = src[i < cnt ? i : 0];
          }
        }
        FOR_LANES {
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int i = base + j * CO_WAVE + lane;
            if (i < cnt) dst[i] = L(v[j]);
          }
        }
      }
    } else {
      const uint32_t *src = w.mt + w.gc.rng_idx;
      /* four chunks of 64 words at a time, loads first: their latencies overlap (a step owes ~450 words) */
      for (int base = 0; base < cnt; base += 4 * CO_WAVE) {
        LV(uint32_t, v[4]);
        FOR_LANES {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = base + j * CO_WAVE + lane;
            L(v[j]) = src[i < cnt ? i : 0];
          }
        }
        FOR_LANES {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = base + j * CO_WAVE + lane;
            if (i < cnt) dst[i] = L(v[j]);
          }
        }
      }
    }
    WAVE_SYNC();
    w.gc.rng_idx += cnt;
    done += cnt;
  }
  w.mt_staged = staged;
}

/* one output for the game's own draws (opening moves, trainmc.cpp:407; a random player, match.cpp:199-200), through
 * memory; a twist there leaves the staging copy behind */
CO_DEV uint32_t co_wave_mt_next(CoWave &w) {
  if (w.gc.rng_idx >= CO_MT_N) w.mt_staged = 0;
  return co_mt_next(w.mt, &w.gc.rng_idx);
}

/* Part (1) for FOUR leaves at a time, one per row of the wavefront (round 5): the same arithmetic as co_prior_leaf --
 * which the separate priors kernel of rounds 2-4 ran, one wavefront per leaf, as a launch of its own in front of every
 * search launch: 19-50 us on each pool's chain, most of it waiting for CUs beside the network kernels.  In row form it is
 * ~550 instructions per four leaves inside the game's own step.
 *   A lane looks at move ids column + 16 q (q < 6): legal?  its rank among the legal moves = its edge; the network's prior
 *   of the move (coalesced: 16 consecutive floats of the row), the generator output of the edge -> its gamma sample
 *   (table in LDS).  Illegal moves carry +0.0: the two SEQUENTIAL float sums of the reference's loops over edges
 *   (trainmc.cpp:219-229, 238-241) then run over all 96 move ids in order -- x + 0.0 == x -- as 96 v_add_f32 with a
 *   row_newbcast source each (ROW_SEQ_SUM16), the two chains interleaved.  Weights, row maximum, 9-bit quantisation, the
 *   integer sum and the denominator as in co_prior_leaf. */
#define CO_SB_ROWS 4 /* rows of a wavefront */
/* (-DCO_PROF_PRIORS: the pass split over six stamp slots that are empty otherwise, every stamp behind a full wait) */
#if defined(CO_PROF_PRIORS) && defined(CO_PROF) && !defined(CO_EMU)
#define CO_PPH(slot)                                            \
  do {                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    CO_PH(slot);                                                \
  } while (0)
#else
#define CO_PPH(slot) ((void)0)
#endif
#define CO_PRE 16
#define CO_PRE_LEAF 0
#define CO_PRE_NOFF 1
#define CO_PRE_L0 2
#define CO_PRE_L1 3
#define CO_PRE_L2 4
#define CO_PRE_ROFF 5 /* where the leaf's row of priors starts, in floats from cval / from the step's probs */
#define CO_PRE_DEPTH 6
#define CO_PRE_WORDS (7 * CO_PRE)
/* the records of pending leaves c0 .. c0 + nc - 1 (nc <= CO_PRE) -> w.pre */
CO_DEV void co_pending_records(CoWave &w, int c0, int nc) {
  const uint32_t *pend_leaf = w.pend_leaf;
  const uint4 *pend_n = (const uint4 *)w.pend_n;
  const int32_t *pend_depth = w.pend_depth;
  const int has_cval = w.cval != 0;
  const int32_t *csrc = w.csrc;
  uint32_t *pre = w.pre;
  FOR_LANES_HOT {
    if (lane < nc) {
      const int k = c0 + lane;
      const uint32_t lf = pend_leaf[k];
      const uint4 pn = pend_n[k];
      const int dp = pend_depth[k];
      const uint32_t ro = has_cval ? (uint32_t)csrc[k] * (uint32_t)CO_CACHE_VAL_FLOATS + 4u : (uint32_t)k * (uint32_t)CO_NUM_MOVES;
      pre[CO_PRE_LEAF * CO_PRE + lane] = lf;
      pre[CO_PRE_NOFF * CO_PRE + lane] = pn.x >> 8;
      pre[CO_PRE_L0 * CO_PRE + lane] = pn.y;
      pre[CO_PRE_L1 * CO_PRE + lane] = pn.z;
      pre[CO_PRE_L2 * CO_PRE + lane] = pn.w;
      pre[CO_PRE_ROFF * CO_PRE + lane] = ro;
      pre[CO_PRE_DEPTH * CO_PRE + lane] = (uint32_t)dp;
    }
  }
  WAVE_SYNC();
  w.pre_c0 = c0;
  w.pre_n = nc;
}

/* is move id column + 16 q legal in the leaf whose 96-bit mask is (l0, l1, l2), and which edge is it */
CO_DEV int co_prior_legal(uint32_t l0, uint32_t l1, uint32_t l2, int q, int c) {
  const uint32_t wd = q < 2 ? l0 : q < 4 ? l1 : l2;
  return (int)((wd >> ((16 * q + c) & 31)) & 1u);
}
CO_DEV int co_prior_rank(uint32_t l0, uint32_t l1, uint32_t l2, int q, int c) {
  const uint32_t wd = q < 2 ? l0 : q < 4 ? l1 : l2;
  const int bit = (16 * q + c) & 31;
  const int before = q < 2 ? 0 : q < 4 ? co_popc32(l0) : co_popc32(l0) + co_popc32(l1);
  return before + co_popc32(wd & ((1u << bit) - 1u));
}
/* The passes of a step, four leaves each.  Round 5 stamps (tools/prof_phases.py -DCO_PROF_PRIORS) put a pass at ~9 k
 * cycles: two trips to memory one behind the other (the leaf's record, then the row of priors and the generator words it
 * points to), the sequential sums, and the stores of the quantised priors -- behind which the NEXT pass's loads waited, the
 * counter of outstanding memory operations being retired in order.  So:
 *   - the records of ALL pending leaves are fetched once, lane k = leaf k, and a pass takes its four by a shuffle;
 *   - the loads of pass p + 1 are issued BEFORE the stores of pass p (two register sets, X and Y). */
#define CO_PRIOR_FETCH(K0, on_, leaf_, l0_, l1_, l2_, pq_, rw_)                                                          \
  {                                                                                                                      \
    LV(uint32_t, noff_);                                                                                                 \
    LV(uint32_t, roff_);                                                                                                 \
    FOR_LANES_HOT {                                                                                                      \
      int i = (K0)-c0 + (lane >> 4);                                                                                     \
      i = i < nc ? i : 0;                                                                                                \
      L(leaf_) = pre[CO_PRE_LEAF * CO_PRE + i];                                                                          \
      L(noff_) = pre[CO_PRE_NOFF * CO_PRE + i];                                                                          \
      L(roff_) = pre[CO_PRE_ROFF * CO_PRE + i];                                                                          \
      L(l0_) = pre[CO_PRE_L0 * CO_PRE + i];                                                                              \
      L(l1_) = pre[CO_PRE_L1 * CO_PRE + i];                                                                              \
      L(l2_) = pre[CO_PRE_L2 * CO_PRE + i];                                                                              \
    }                                                                                                                    \
    FOR_LANES_HOT {                                                                                                      \
      const int c = lane & 15;                                                                                           \
      const int has = (K0) + (lane >> 4) < n;                                                                            \
      L(on_) = has;                                                                                                      \
      L(l0_) = has ? L(l0_) : 0u;                                                                                        \
      L(l1_) = has ? L(l1_) : 0u;                                                                                        \
      L(l2_) = has ? L(l2_) : 0u;                                                                                        \
      const float *row = rbase + L(roff_);                                                                               \
      const uint32_t *raw = noise_raw + L(noff_);                                                                        \
      _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                                                    \
        const int legal = co_prior_legal(L(l0_), L(l1_), L(l2_), q, c);                                                  \
        L(pq_[q]) = row[16 * q + c]; /* (all loads first: their latencies overlap) */                                    \
        L(rw_[q]) = raw[legal ? co_prior_rank(L(l0_), L(l1_), L(l2_), q, c) : 0];                                        \
      }                                                                                                                  \
    }                                                                                                                    \
  }
CO_DEV void co_prior_all(CoWave &w, CoTree &t, int c0, int nc, const float *probs) {
  uint4 *A = t.A;
  const uint32_t *noise_raw = w.noise_raw;
  const uint32_t *gam = w.gamma;
  const float *rbase = w.cval ? w.cval : probs;
  const float eps = w.epsilon;
  const int n = c0 + nc;
  const uint32_t *pre = w.pre;
  { /* leaves c0 .. c0 + nc - 1, whose records are in w.pre */
    CO_PPH(19);
    LV(int, on);
    LV(uint32_t, leaf);
    LV(uint32_t, l0); /* the leaf's legal moves; all zero in a row without a leaf */
    LV(uint32_t, l1);
    LV(uint32_t, l2);
    LV(float, pq[6]);
    LV(uint32_t, rw[6]);
    CO_PRIOR_FETCH(c0, on, leaf, l0, l1, l2, pq, rw)
    for (int k0 = c0; k0 < c0 + nc; k0 += CO_SB_ROWS) {
      const int more = k0 + CO_SB_ROWS < c0 + nc;
      CO_PPH(16);
      LV(float, gq[6]);
      FOR_LANES_HOT {
        const int c = lane & 15;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int legal = co_prior_legal(L(l0), L(l1), L(l2), q, c);
          const float g = co_u2f(gam[co_mt_temper(L(rw[q])) % CO_NUM_GAMMA]);
          L(gq[q]) = legal ? g : 0.0f;
          L(pq[q]) = legal ? L(pq[q]) : 0.0f;
        }
      }
      /* a block of sixteen move ids none of the four leaves has a legal move in (the stack moves of an early position:
       * three of the six blocks) adds sixteen zeros: skipped */
      int live[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        LV(int, lq);
        FOR_LANES_HOT {
          const uint32_t m = q < 2 ? L(l0) : q < 4 ? L(l1) : L(l2);
          L(lq) = ((q & 1) ? m >> 16 : m & 0xFFFFu) != 0u;
        }
        live[q] = WAVE_BALLOT(lq) != 0ull;
      }
      CO_PPH(10);
      LV(float, sum);
      LV(float, dsum);
      FOR_LANES_HOT { L(sum) = L(dsum) = 0.0f; }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (!live[q]) continue;
        ROW_SEQ_SUM16_2(sum, pq[q], dsum, gq[q]);
      }
      CO_PPH(25);
      LV(float, mxl);
      FOR_LANES_HOT {
        const int c = lane & 15;
        const float one_minus = (float)1 - eps;
        const float scalar = (float)(1.0 / (double)L(sum) * (double)one_minus);
        const float dscalar = (float)(1.0 / (double)L(dsum) * (double)eps);
        float mx = 0.0f; /* weights are >= 0, as the reference's max_prob start value */
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const float a = L(pq[q]) * scalar;
          const float d = L(gq[q]) * dscalar;
          const float x = a + d;
          L(pq[q]) = x; /* the move's weight from here on */
          if (co_prior_legal(L(l0), L(l1), L(l2), q, c)) mx = x > mx ? x : mx;
        }
        L(mxl) = mx;
      }
      LV(float, mxr);
      ROW_MAX_F32(mxr, mxl);
      CO_PPH(29);
      /* the next pass's loads, in front of this pass's stores */
      LV(int, on_y);
      LV(uint32_t, leaf_y);
      LV(uint32_t, l0_y);
      LV(uint32_t, l1_y);
      LV(uint32_t, l2_y);
      LV(float, pq_y[6]);
      if (more) CO_PRIOR_FETCH(k0 + CO_SB_ROWS, on_y, leaf_y, l0_y, l1_y, l2_y, pq_y, rw)
      LV(int, qs);
      FOR_LANES_HOT {
        const int c = lane & 15;
        const float denom = 511.0f / L(mxr);
        int sm = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          if (co_prior_legal(L(l0), L(l1), L(l2), q, c)) {
            const float x = L(pq[q]) * denom;
            /* lround: half away from zero (x >= 0 here; NaN/neg handled like max(1, .)) */
            const float fl = __builtin_truncf(x);
            int qq = (int)fl;
            if (x - fl >= 0.5f) qq += 1;
            if (!(qq >= 1)) qq = 1;
            sm += qq;
            const uint32_t edge = (uint32_t)co_prior_rank(L(l0), L(l1), L(l2), q, c);
            A[L(leaf) + 2u + edge].z = (uint32_t)(16 * q + c) | ((uint32_t)(qq & 511) << 7);
          }
        }
        L(qs) = sm;
      }
      LV(int, fs);
      ROW_SUM_I32(fs, qs);
      FOR_LANES_HOT {
        if (L(on) && (lane & 15) == 0) {
          const float denominator = (float)(1.0 / (double)(float)L(fs));
          A[L(leaf) + 1u].y = co_f2u(denominator);
        }
      }
      if (more) {
        FOR_LANES_HOT {
          L(on) = L(on_y);
          L(leaf) = L(leaf_y);
          L(l0) = L(l0_y);
          L(l1) = L(l1_y);
          L(l2) = L(l2_y);
#pragma unroll
          for (int q = 0; q < 6; ++q) L(pq[q]) = L(pq_y[q]);
        }
      }
      CO_PPH(18);
    }
  }
}

/* Part (2): the backups of up to CO_RB pending leaves at once.  The node k levels above a leaf
 * receives eval*(-1)^k - 1; every node on the path becomes searchable again (all_visited := false).
 * The slot of a shared ancestor must receive the leaves' contributions in request order (float
 * addition is not associative): the slots of all paths are fetched together, and leaf k starts from
 * the value left by the latest earlier leaf of the batch that touched the same slot (register
 * forwarding) instead of re-reading memory. */
#define CO_RB 8
/* the paths of leaves k0 .. k0 + nb - 1 -> at[k][lane = level] (all CO_PATH_MAX entries whatever the leaf's depth: the
 * loads do not wait for the depths) */
CO_DEV void co_backup_paths(CoWave &w, int k0, int nb, LVPA(uint32_t, at, CO_RB)) {
#pragma unroll
  for (int k = 0; k < CO_RB; ++k) {
    const uint32_t *pp = w.pend_path + (size_t)(k < nb ? k0 + k : k0) * CO_PATH_MAX;
    FOR_LANES { L(at[k]) = pp[lane < CO_PATH_MAX ? lane : 0]; }
  }
}
/* evl: the evaluation (bits) of leaf c0 + k in lane k, its depth in w.pre */
CO_DEV void co_backup_batch(CoWave &w, CoTree &t, int c0, int k0, int nb, LVP(uint32_t, evl), LVPA(uint32_t, at, CO_RB)) {
  uint4 *A = t.A;
  LV(uint32_t, ny[CO_RB]);
  LV(uint32_t, nw[CO_RB]);
  LV(uint32_t, nx[CO_RB]); /* (the slot's other two words, so that a slot goes back as ONE 16-byte store) */
  LV(uint32_t, nz[CO_RB]);
  LV(int, on[CO_RB]);
#pragma unroll
  for (int k = 0; k < CO_RB; ++k) {
    const int D = (int)w.pre[CO_PRE_DEPTH * CO_PRE + (k < nb ? k0 - c0 + k : k0 - c0)];
    FOR_LANES { L(on[k]) = (k < nb && lane <= D); }
  }
  CO_PH_MEM(0);
#pragma unroll
  for (int k = 0; k < CO_RB; ++k) {
    FOR_LANES {
      if (!L(on[k])) L(at[k]) = 0u;
      uint4 sl = A[L(at[k])]; /* unconditional: inactive lanes read unit 0 */
      L(nx[k]) = sl.x;
      L(ny[k]) = sl.y;
      L(nz[k]) = sl.z;
      L(nw[k]) = sl.w & ~0x100u; /* all_visited := false */
    }
  }
  CO_PH_MEM(1);
#pragma unroll
  for (int k = 0; k < CO_RB; ++k) {
    const int D = (int)w.pre[CO_PRE_DEPTH * CO_PRE + (k < nb ? k0 - c0 + k : k0 - c0)];
    const float leaf_eval = k < nb ? co_u2f(WAVE_BCAST(evl, k0 - c0 + k)) : 0.0f;
    FOR_LANES {
      if (L(on[k])) {
        uint32_t cur = L(ny[k]);
#pragma unroll
        for (int j = 0; j < k; ++j)
          if (L(on[j]) && L(at[j]) == L(at[k])) cur = L(ny[j]);
        int kk = D - lane;
        float ce = (kk & 1) ? (float)((double)leaf_eval * -1.0) : leaf_eval;
        float add = (float)((double)ce - 1.0);
        L(ny[k]) = co_f2u(co_u2f(cur) + add);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CO_RB; ++k) {
    FOR_LANES {
      if (L(on[k])) A[L(at[k])] = make_uint4(L(nx[k]), L(ny[k]), L(nz[k]), L(nw[k]));
    }
  }
  WAVE_SYNC();
  CO_PH(3);
  w.gc.evals += (uint32_t)nb;
}

/* trainmc.cpp:269-296: the priors of the pending leaves (co_prior_all, four leaves per pass), then the backups in
 * request order */
CO_DEV void co_receive_eval(CoWave &w, CoTree &t, const float *eval, const float *probs) {
  const int n = w.gc.n_pending;
  for (int c0 = 0; c0 < n; c0 += CO_PRE) { /* (one round: searches_per_eval is 16, or 1) */
    const int nc = n - c0 < CO_PRE ? n - c0 : CO_PRE;
    /* what the backups need of the leaves' records, requested beside the priors' own first loads: depth and evaluation
     * of leaf c0 + k in lane k (one trip for the step instead of two dependent scalar loads per leaf behind the priors'
     * stores) */
    if (c0 > 0 || w.pre_n != nc) co_pending_records(w, c0, nc); /* (else: requested when the step began) */
    w.pre_n = 0;
    LV(uint32_t, evl);
    {
      const float *cval = w.cval;
      const uint32_t *pre = w.pre;
      FOR_LANES_HOT {
        const int i = lane < nc ? lane : 0;
        L(evl) = co_f2u(cval ? cval[pre[CO_PRE_ROFF * CO_PRE + i] - 4u] : eval[c0 + i]);
      }
    }
    LV(uint32_t, pa[CO_RB]);
    co_prior_all(w, t, c0, nc, probs);
    WAVE_SYNC();
    CO_PH_MEM(13);
    for (int k0 = c0; k0 < c0 + nc; k0 += CO_RB) {
      const int nb = c0 + nc - k0 < CO_RB ? c0 + nc - k0 : CO_RB;
      /* (requested in front of the priors' passes instead -- 8 or 16 registers across them: +-0 with the first batch's,
       * 4 % slower with both, the product build's allocation gives way; round 5) */
      co_backup_paths(w, k0, nb, pa);
      co_backup_batch(w, t, c0, k0, nb, evl, pa);
    }
  }
  CO_PROF_ADD(w, 6, (unsigned long long)n);
  w.gc.n_pending = 0;
  w.noise_words = 0;
}

/* propagateTerminal, trainmc.cpp:497-538 (including the parent-for-child draw
 * test at :518).  path_* describe root..leaf, D = index of the terminal leaf. */
CO_DEV void co_propagate_terminal(CoTree &t, const uint32_t *path_block, const uint32_t *path_slot, int D) {
  uint4 *A = t.A;
  int d = D;
  while (d > 0) {
    int rc = co_slot_result(co_load_unit(A, path_slot[d]));
    if (co_res_lost(rc)) {
      --d;
      uint4 s = co_load_unit(A, path_slot[d]);
      co_store_unit(A, path_slot[d], co_slot_set_result(s, CO_DEDUCED_WIN));
    } else {
      --d;
      uint32_t pb = path_block[d];
      int n = (int)CO_META_NEDGES(co_load_unit(A, pb).z);
      int all_known = 1;
      for (int base = 0; base < n; base += CO_WAVE) {
        LV(int, bad);
        FOR_LANES {
          int e = base + lane;
          L(bad) = 0;
          if (e < n) {
            uint4 s = A[pb + 2 + e];
            L(bad) = (s.x == CO_NONE) || !co_res_known(co_slot_result(s));
          }
        }
        if (WAVE_BALLOT(bad)) all_known = 0;
      }
      if (!all_known) return;
      uint4 s = co_load_unit(A, path_slot[d]);
      int has_draw = co_res_drawn(co_slot_result(s)); /* the PARENT's own result, as trainmc.cpp:518 */
      co_store_unit(A, path_slot[d], co_slot_set_result(s, has_draw ? CO_DEDUCED_DRAW : CO_DEDUCED_LOSS));
    }
  }
}

/* x / d for a float x and an integer d in [1, 2^15] as the correctly rounded double quotient --
 * what the reference's double division gives (trainmc.cpp:565-568) -- without the full IEEE
 * division sequence (two v_div_scale, v_rcp_f64, seven fma, v_div_fmas, v_div_fixup; the two
 * quotients and the square root of a PUCT scan were ~45 double-rate instructions per level and
 * the largest single cost of a simulation).  r = 1/d to double precision from the float
 * reciprocal and two Newton steps; q0 = RN(x r); e = x - d q0 is exact (d has 16 bits and x is
 * a multiple of ulp(q0)); q = RN(q0 + e r) = RN(x/d (1 + 2^-51 ulp)).  x/d is either a double
 * itself or at least ulp/(2 d) >= 2^-17 ulp away from every rounding boundary, so q = RN(x/d).
 * (A zero quotient may differ in sign; pv >= +0 makes the PUCT sum the same.)  The seed only has
 * to be good to ~20 bits, so the emulation build's 1.0f / d and v_rcp_f32 give the same q. */
CO_DEV double co_recip_small(float df) {
#ifdef CO_EMU
  float rf = 1.0f / df;
#else
  float rf = __builtin_amdgcn_rcpf(df);
#endif
  double d = (double)df;
  double r = (double)rf;
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
/* x / df given r = co_recip_small(df) */
CO_DEV double co_div_with(double x, float df, double r) {
  double d = (double)df;
  double q0 = x * r;
  double rem = __builtin_fma(-d, q0, x);
  return __builtin_fma(rem, r, q0);
}
CO_DEV double co_div_small(double x, float df) { return co_div_with(x, df, co_recip_small(df)); }

/* Copy of the root kept across the simulations of one step, so that a simulation starts without any
 * dependent load: header and stat slot in registers, the first 64 edge slots (the whole PUCT scan of
 * the root for all but the widest positions) in LDS.  Every simulation reads them and changes one:
 * stores to a root edge slot go to memory AND to the copy (co_store_slot); the rare paths that
 * rewrite several path slots in memory (terminal leaf, dead end) drop the copy (valid = 0). */
struct CoRoot {
  uint4 h0, h1, cs;
  int valid;
  int hdr;        /* h0, h1, e0, ne are the root's (they do not change while a step searches): a dropped copy is made good
                   * again by ONE trip to memory -- own slot and edge slots together -- instead of two dependent ones */
  uint4 *ev;      /* LDS: edge slots 0..63 of the root */
  uint32_t e0;    /* unit offset of the root's edge slot 0 */
  uint32_t ne;    /* edge slots held: min(edges, 64) */
};

/* the exploration factor of a PUCT scan (trainmc.cpp:549): float(double(c_puct) * sqrt(double(float(visits)))).
 * (Computing it one step ahead, in the shadow of the descent's block fetch, was measured in round 4: the generation 5 %
 * SLOWER -- two more values live across the scan loop cost more than the hidden chain saves.) */
CO_DEV float co_vsqrt(float c_puct, int visits) { return (float)((double)c_puct * co_sqrt_f64((double)(float)visits)); }

CO_DEV void co_root_load(CoTree &t, CoRoot &rc) {
  if (!rc.hdr) {
    rc.h0 = co_load_unit(t.A, t.tc.root);
    rc.h1 = co_load_unit(t.A, t.tc.root + 1);
    rc.e0 = t.tc.root + 2u;
    uint32_t n = CO_META_NEDGES(rc.h0.z);
    rc.ne = n < (uint32_t)CO_WAVE ? n : (uint32_t)CO_WAVE;
    rc.hdr = 1;
  }
  rc.cs = co_load_unit(t.A, rc.h1.x);
  const uint4 *A = t.A;
  FOR_LANES { rc.ev[lane] = A[rc.e0 + lane]; } /* arena is padded: lanes >= n read unused units */
  WAVE_SYNC();
  rc.valid = 1;
}


/* single-lane store of a stat slot, mirrored into the root copy when it is one of the root's edge slots */
CO_DEV void co_store_slot(uint4 *A, uint32_t slot, uint4 v, CoRoot &rc) {
  const uint32_t idx = slot - rc.e0; /* unsigned: anything outside the root's edges is >= ne */
  FOR_LANES {
    if (lane == 0) {
      A[slot] = v;
      if (idx < rc.ne) rc.ev[idx] = v;
    }
  }
  WAVE_SYNC();
}

/* u of ONE edge, from its slot (the body of chooseNext's loop, trainmc.cpp:553-579), per lane.  Branch-free: every lane
 * evaluates both forms and selects (one f64 pair per level whatever the mix of edges; divergent branches here cost more
 * than they save). */
CO_DEV float co_puct_u(uint4 s, float denom, float v_sqrt) {
  float prob = (float)((s.z >> 7) & 511u) * denom;
  float pv = prob * v_sqrt;
  int r = (int)(s.w & 0xFFu);
  int has_child = s.x != CO_NONE;
  int drawn = (r == CO_RESULT_DRAW) | (r == CO_DEDUCED_DRAW);
  int searchable = ((r == CO_RESULT_NONE) | drawn) & !((s.w >> 8) & 1u);
  int vis = co_slot_visits(s);
  float cv = (float)(vis > 0 ? vis : 1); /* 1 .. 32767 */
  /* (Measured in round 6: with both quotients as single-precision products with v_rcp_f32 -- NOT the reference's bits; the
   * ceiling of any scheme that scans approximately and verifies -- a generation is 1-2.6 % shorter: not what a scan costs.) */
  double a = co_div_small(-(double)co_u2f(s.y), cv);
  double b = co_div_small((double)pv, cv + 1.0f);
  float uv = (float)(a + b);
  float uc = drawn ? pv : uv;           /* visited child (trainmc.cpp:561-569; a drawn one without the /(n + 1): quirk 6) */
  uc = searchable ? uc : CO_NEG_INF;
  return has_child ? uc : pv;           /* unvisited edge (:575-577) */
}

/* a simulation passes through a node: increment_visits + increase_evaluation(1.0) (trainmc.cpp:609-623), on its stat slot */
CO_DEV uint4 co_slot_pass(uint4 s) {
  s = co_slot_set_visits(s, co_slot_visits(s) + 1);
  s.y = co_f2u(co_u2f(s.y) + 1.0f);
  return s;
}

/* One simulation: TrainMC::search (trainmc.cpp:602-696) with chooseNext
 * (:540-600) inlined as the lane-parallel edge scan.  A node's header and its
 * first 64 edge slots are requested together (one memory round trip per level). */
CO_COLD2 void co_search(CoWave &w, CoTree &t, CoRoot &rc) {
  uint4 *A = t.A;
  WAVE_SHARED(uint32_t, path_block, CO_PATH_MAX);
  WAVE_SHARED(uint32_t, path_slot, CO_PATH_MAX);
  uint32_t cur = t.tc.root;
  ++t.tc.searches_done;
  w.gc.searches++;
  uint4 h0 = rc.h0;
  uint4 h1 = rc.h1;
  uint32_t cur_slot = h1.x;
  uint4 cs = rc.cs;
  LV(uint4, ev);
  FOR_LANES { L(ev) = rc.ev[lane]; }
  int D = 0;
  int leaf_n = 0; /* legal moves of the node this simulation creates */
  uint32_t leaf_lm[3] = {0u, 0u, 0u};
  FOR_LANES {
    if (lane == 0) {
      path_block[0] = cur;
      path_slot[0] = cur_slot;
    }
  }
  WAVE_SYNC();
  CO_PH_MEM(19);
  float v_sqrt_next = co_vsqrt(w.c_puct, co_slot_visits(cs));
  while (!co_res_terminal(co_slot_result(cs))) {
    int n = (int)CO_META_NEDGES(h0.z);
    float denom = co_u2f(h1.y);
    int visits = co_slot_visits(cs);
    const float v_sqrt = v_sqrt_next;
    /* ---- chooseNext: u for every edge, strict first maximum */
    float best_u = CO_NEG_INF;
    int best_e = -1;
    uint4 best_slot = make_uint4(0, 0, 0, 0);
    for (int base = 0; base < n; base += CO_WAVE) {
      LV(float, u);
      FOR_LANES {
        int e = base + lane;
        float uu = CO_NEG_INF;
        if (base > 0 && e < n) L(ev) = A[cur + 2 + e];
        uu = co_puct_u(L(ev), denom, v_sqrt);
        uu = e < n ? uu : CO_NEG_INF;
        L(u) = uu;
      }
      float mx = WAVE_MAX_F32(u);
      if (mx > best_u) {
        LV(int, hit);
        FOR_LANES { L(hit) = (L(u) == mx); }
        int le = co_ffs64(WAVE_BALLOT(hit)) - 1;
        best_u = mx;
        best_e = base + le;
        best_slot = WAVE_BCAST(ev, le);
      }
    }
    CO_PH(8);
    CO_PROF_ADD(w, 23, 1ull);
    CO_STEP_WORK(w, 1);
    /* ---- visit the current node (virtual loss on every node of the path) */
    cs = co_slot_set_visits(cs, visits + 1);
    cs.y = co_f2u(co_u2f(cs.y) + 1.0f);
    if (best_e < 0) {
      /* kNone: nothing searchable below; mark, undo the path, un-count the search */
      cs = co_slot_set_all_visited(cs, 1);
      co_store_slot(A, cur_slot, cs, rc);
      FOR_LANES {
        if (lane <= D) {
          uint4 s = A[path_slot[lane]];
          s = co_slot_set_visits(s, co_slot_visits(s) - 1);
          s.y = co_f2u(co_u2f(s.y) - 1.0f);
          A[path_slot[lane]] = s;
        }
      }
      WAVE_SYNC();
      --t.tc.searches_done;
      w.gc.searches--;
      rc.valid = 0;
      return;
    }
    co_store_slot(A, cur_slot, cs, rc);
    if (D == 0) rc.cs = cs;
    if (D + 1 >= CO_PATH_MAX) {
      w.gc.error |= CO_ERR_PATH_TOO_DEEP;
      return;
    }
    uint32_t child_slot = cur + 2u + (uint32_t)best_e;
    CO_PH(9);
    if (best_slot.x == CO_NONE) {
      /* kNew: expand (Node ctor from parent, node.cpp:31-39) */
      uint64_t board = (uint64_t)h0.x | ((uint64_t)h0.y << 32);
      uint32_t meta = h0.z;
      int move = (int)(best_slot.z & 127u);
      co_do_move_lane(&board, &meta, move);
      int res;
      int depth = (int)CO_META_DEPTH(h0.z) + 1;
      CO_PH(16);
      CO_PROF_ADD(w, 24, 1ull);
      uint32_t nb = co_create_node(w, t, board, meta, depth, cur, child_slot, &res, &leaf_n, leaf_lm);
      if (nb == CO_NONE) return;
      if (w.analyse) {
        /* Node::countNodes (node.cpp:179-187) without a traversal: word z of unit b + 1 counts the
         * nodes below block b; a new node adds one to every node of its path */
        FOR_LANES {
          if (lane <= D) A[path_block[lane] + 1].z += 1u;
        }
        WAVE_SYNC();
      }
      cs = make_uint4(nb, 0u, (best_slot.z & 0xFFFFu) | (1u << 16), (uint32_t)res | 0x100u);
      cur = nb;
      cur_slot = child_slot;
      h0 = make_uint4((uint32_t)board, (uint32_t)(board >> 32), co_meta_make(meta, depth, 0), 0u);
      ++D;
      FOR_LANES {
        if (lane == 0) {
          path_block[D] = cur;
          path_slot[D] = cur_slot;
        }
      }
      WAVE_SYNC();
      CO_PH(10);
      break;
    }
    /* kVisited: descend */
    cur = best_slot.x;
    cur_slot = child_slot;
    cs = best_slot;
    h0 = co_load_unit(A, cur);
    h1 = co_load_unit(A, cur + 1);
    FOR_LANES { L(ev) = A[cur + 2 + lane]; }
    v_sqrt_next = co_vsqrt(w.c_puct, co_slot_visits(cs)); /* under the fetch (see co_search_rows) */
    CO_OPAQUE_V(v_sqrt_next);
    CO_PH_MEM(11);
    ++D;
    FOR_LANES {
      if (lane == 0) {
        path_block[D] = cur;
        path_slot[D] = cur_slot;
      }
    }
    WAVE_SYNC();
  }
  int r = co_slot_result(cs);
  if (co_res_terminal(r)) {
    /* trainmc.cpp:663-682.  (For a fresh child the slot is first written here.) */
    float cur_eval = co_res_drawn(r) ? 0.0f : -1.0f;
    cs.y = co_f2u(cur_eval);
    co_store_slot(A, cur_slot, cs, rc);
    co_propagate_terminal(t, path_block, path_slot, D);
    FOR_LANES {
      if (lane < D) {
        int kk = D - lane; /* kk-th ancestor receives eval*(-1)^(kk-1) - 1 */
        float ce = ((kk - 1) & 1) ? (float)((double)cur_eval * -1.0) : cur_eval;
        float add = (float)((double)ce - 1.0);
        uint4 s = A[path_slot[lane]];
        s.y = co_f2u(co_u2f(s.y) + add);
        A[path_slot[lane]] = s;
      }
    }
    WAVE_SYNC();
    rc.valid = 0;
    CO_PH(12);
  } else {
    /* trainmc.cpp:684-692: default +1 evaluation, queue the leaf */
    cs.y = co_f2u(1.0f);
    co_store_slot(A, cur_slot, cs, rc);
    uint64_t board = (uint64_t)h0.x | ((uint64_t)h0.y << 32);
    co_request(w, board, h0.z, cur, leaf_n, leaf_lm, D, path_slot);
  }
}

/* ---- Several simulations of a step at a time (round 5).
 *
 * The reference runs the simulations of a step one after another (trainmc.cpp:169-173), and so did this kernel: every
 * simulation a chain of dependent waits -- root scan, block fetch, scan, store, expansion, request -- ~10 k cycles of a
 * lone wavefront, sixteen times per launch.  But under the virtual loss a simulation that ends the ORDINARY way -- it
 * descends through visited, non-terminal children and creates one new, non-terminal leaf -- changes the tree in a way
 * that is known the moment each of its choices is made: the chosen child's slot gets one visit and +1.0 (:609-623;
 * a new child is born with all_visited set and cannot be chosen again before its evaluation arrives, quirk 3), and nothing
 * else a later scan of the step looks at.  So CO_SB simulations are SELECTED together, level by level:
 *   - every simulation scans its node's edge slots in registers; a slot another simulation of the group has changed at
 *     this level is patched in first (same node <=> same level: it is a tree), in simulation order -- at the root that is
 *     all of them, deeper down only those that chose the same child;
 *   - the blocks of all the children descended to are requested together: one memory round trip per LEVEL of the group;
 *   - nothing is written while selecting.  Then, in simulation order, each leaf is expanded and the simulation's stores are
 *     issued (path slots, node, request) -- the same values in the same order as one after another, so the arena, the
 *     pending-leaf records and the generator are bit for bit what the sequential search leaves.
 * A simulation that turns out NOT to be ordinary (nothing searchable below a node, a terminal child or a terminal new
 * leaf, a node wider than the wavefront, a path beyond CO_SB_DEPTH, a full arena) ends the group in front of it: the
 * simulations before it are committed, it is run by co_search on the committed tree, the ones behind it are selected
 * again in the next group (their selection is discarded -- nothing was written). */
#ifndef CO_SB_PE_ONE
#define CO_SB_PE_ONE 96
#endif
#ifndef CO_SB_PE_TWO
#define CO_SB_PE_TWO 40
#endif
#define CO_SB_DEPTH 12 /* levels a grouped simulation may pass (a level = a lane of its row when it commits: at most 16) */
#if CO_SB != 1 && CO_SB != 4
#error "CO_SB: 1 (one simulation after another) or 4 (one per row of the wavefront)"
#endif

#if defined(CO_SB_STATS) && defined(CO_EMU)
/* emulation-build counters of the grouped search (tools/sb_stats.py): 0 groups, 1 simulations asked for, 2 committed,
 * 3 nothing searchable, 4 terminal child, 5 wide node, 6 too deep, 7 terminal leaf / full arena, 8 sequential simulations,
 * 9 levels of the groups, 10 scans, 11 patches applied (same node, same level), 12 + k: groups that committed k,
 * 20 shared levels below the root, 21 / 22 levels scanned wave-wide for one / two rows, 23 levels in row form, 24 turns taken there */
extern unsigned long long co_sb_stats[32];
extern unsigned long long co_sb_ply[8][8]; /* by game progress (plies / 4): groups, asked, committed, terminal leaves, nothing searchable, levels, sequential, - */
#define CO_SBS(i, v) __atomic_fetch_add(&co_sb_stats[i], (unsigned long long)(v), __ATOMIC_RELAXED)
#define CO_SBP(w, i, v) __atomic_fetch_add(&co_sb_ply[(w).gc.plies / 4 > 7 ? 7 : (w).gc.plies / 4][i], (unsigned long long)(v), __ATOMIC_RELAXED)
#else
#define CO_SBS(i, v) ((void)0)
#define CO_SBP(w, i, v) ((void)0)
#endif

#if CO_SB > 1
/* m = simulations that are certainly due (1 < m <= CO_SB = 4: each adds one pending leaf and one search).  Returns how
 * many were committed; fewer than m: the next one takes co_search.
 *
 * The group's four simulations live in the four ROWS of the wavefront (16 lanes each, the rows of the DPP unit):
 *   root    the root's edge slots in registers (one per lane), scanned once per simulation, the chosen slot patched in
 *           place -- sequential by nature, wave-wide;
 *   below   row j descends simulation j: its node's edge slots as up to four per lane (edge = 16 k + column), row maximum
 *           and first-maximum by DPP, the winner's slot by ds_bpermute -- one instruction stream for the four descents.
 *           Rows that meet in one node take turns in simulation order, the later one's copy patched with what the
 *           earlier one changed;
 *   commit  in simulation order (co_search_commit). */
CO_DEV int co_search_rows(CoWave &w, CoTree &t, CoRoot &rc, int m) {
  uint4 *A = t.A;
  WAVE_SHARED(uint32_t, sb_block, CO_SB * CO_SB_DEPTH);       /* [sim][level] node scanned */
  WAVE_SHARED(uint32_t, sb_slot, CO_SB * (CO_SB_DEPTH + 1));  /* [sim][level] its stat slot; [leaf level] the new child's */
  WAVE_SHARED(uint4, sb_cs, CO_SB * CO_SB_DEPTH);             /* [sim][level] that slot after the pass */
  WAVE_SHARED(uint4, sb_hand, CO_SB * 2);                     /* root -> row j: {child block, child slot, z of the chosen slot, -}, the chosen slot */
  WAVE_SHARED(uint4, sb_leaf, CO_SB * 2);                     /* row j's leaf: {its parent's board lo, hi, meta, block}, {the chosen slot's move | prior, the leaf's slot, -, -} */
  WAVE_SHARED(int, sb_kn, CO_SB);                             /* sim j found nothing searchable below the node of this level (else -1) */
  FOR_LANES_HOT {
    if (lane < CO_SB) sb_kn[lane] = -1;
  }
  WAVE_SYNC();
  int bad = m; /* the first simulation that is not ordinary */
  /* ---- the root, simulation after simulation; and on down for as long as ALL of them take the same child (a forced
   * line, a trained network's favourite: rows that share a node would take turns anyway -- one wave-wide scan per
   * simulation, the chosen slot patched in registers, is half the instructions of a turn in row form) */
  uint32_t shX = t.tc.root; /* the shared node, its own stat slot and header */
  uint32_t shown = rc.h1.x;
  uint4 shh0 = rc.h0;
  int lev0 = 0;             /* its level */
  {
    int n0 = (int)CO_META_NEDGES(rc.h0.z);
    float denom0 = co_u2f(rc.h1.y);
    LV(uint4, ev);
    FOR_LANES_HOT { L(ev) = rc.ev[lane]; }
    uint4 rcs = rc.cs;
    /* What a scan of this node needs of each edge, kept across the group's scans (co_puct_u in two halves): the
     * exploitation term a = -E / V, the reciprocal of V + 1 for the exploration term, the prior, and which form u takes
     * (0 unvisited, 1 visited, 2 drawn, 3 not to be searched).  Only the chosen edge changes between two scans -- one
     * more visit, +1.0: its a is the old reciprocal's quotient, its reciprocal is made anew.  And the exploration factors
     * of the group's scans (visits, visits + 1, ...: one double-precision square root each) come from ONE pass, a lane each. */
    LV(double, ea);
    LV(double, er);
    LV(float, ecv1);
    LV(float, eprob);
    LV(int, emode);
    LV(float, vsl);
    for (;;) {
      const int msel = bad < m ? bad : m;
      uint32_t same_slot = CO_NONE; /* the child every simulation so far has taken, 0 = not one child (or a new one) */
      uint4 first_bs = make_uint4(0u, 0u, 0u, 0u);
      {
        const int v0 = co_slot_visits(rcs);
        FOR_LANES_HOT {
          const uint4 s = L(ev);
          const int r = (int)(s.w & 0xFFu);
          const int has_child = s.x != CO_NONE;
          const int drawn = (r == CO_RESULT_DRAW) | (r == CO_DEDUCED_DRAW);
          const int searchable = ((r == CO_RESULT_NONE) | drawn) & !((s.w >> 8) & 1u);
          const int vis = co_slot_visits(s);
          const float cv = (float)(vis > 0 ? vis : 1);
          L(eprob) = (float)((s.z >> 7) & 511u) * denom0;
          L(ea) = co_div_small(-(double)co_u2f(s.y), cv);
          L(ecv1) = cv + 1.0f;
          L(er) = co_recip_small(L(ecv1));
          L(emode) = !has_child ? 0 : !searchable ? 3 : drawn ? 2 : 1;
          L(vsl) = co_vsqrt(w.c_puct, v0 + (lane < CO_SB ? lane : 0));
        }
      }
      for (int j = 0; j < msel; ++j) {
        CO_SBS(10, 1);
        CO_PROF_ADD(w, 23, 1ull);
        CO_STEP_WORK(w, 1);
        const float v_sqrt = WAVE_BCAST(vsl, j);
        LV(float, u);
        FOR_LANES_HOT {
          const float pv = L(eprob) * v_sqrt;
          const double bq = co_div_with((double)pv, L(ecv1), L(er));
          const float uv = (float)(L(ea) + bq);
          const int md = L(emode);
          const float uu = md == 1 ? uv : md == 3 ? CO_NEG_INF : pv;
          L(u) = lane < n0 ? uu : CO_NEG_INF;
        }
        const float mx = WAVE_MAX_F32(u);
        rcs = co_slot_pass(rcs);
        if (!(mx > CO_NEG_INF)) {
          CO_SBS(3, 1);
          bad = j;
          const uint32_t xo = shown;
          const int lv = lev0;
          FOR_LANES_HOT {
            if (lane == 0) { /* (for the tail below: this node's slot and its value after the pass) */
              sb_slot[j * (CO_SB_DEPTH + 1) + lv] = xo;
              sb_cs[j * CO_SB_DEPTH + lv] = rcs;
              sb_kn[j] = lv;
            }
          }
          break;
        }
        LV(int, hit);
        FOR_LANES_HOT { L(hit) = (L(u) == mx); }
        const int le = co_ffs64(WAVE_BALLOT(hit)) - 1;
        const uint4 bs = WAVE_BCAST(ev, le);
        if (bs.x != CO_NONE && co_res_terminal(co_slot_result(bs))) {
          CO_SBS(4, 1);
          bad = j;
          break;
        }
        const uint32_t child_slot = shX + 2u + (uint32_t)le;
        const uint4 nv = bs.x == CO_NONE ? make_uint4(0u, co_f2u(1.0f), (bs.z & 0xFFFFu) | (1u << 16), 0x100u) : co_slot_pass(bs);
        if (j == 0) {
          same_slot = bs.x == CO_NONE ? 0u : child_slot;
          first_bs = bs;
        } else if (child_slot != same_slot) {
          same_slot = 0u;
        }
        const uint32_t xb = shX, xo = shown;
        const int lv = lev0;
        FOR_LANES_HOT {
          if (lane == le) {
            L(ev) = nv;
            if (L(emode) == 0) {
              L(emode) = 3; /* a new child: all_visited until its evaluation arrives */
            } else {
              L(ea) = co_div_with(-(double)co_u2f(nv.y), L(ecv1), L(er));
              L(ecv1) = L(ecv1) + 1.0f;
              L(er) = co_recip_small(L(ecv1));
            }
          }
          if (lane == 0) {
            sb_block[j * CO_SB_DEPTH + lv] = xb;
            sb_slot[j * (CO_SB_DEPTH + 1) + lv] = xo;
            sb_slot[j * (CO_SB_DEPTH + 1) + lv + 1] = child_slot;
            sb_cs[j * CO_SB_DEPTH + lv] = rcs;
            sb_hand[2 * j] = make_uint4(bs.x, child_slot, bs.z, 0u);
            sb_hand[2 * j + 1] = bs;
          }
        }
      }
      WAVE_SYNC();
      const int left = bad < m ? bad : m;
      if (left < 2 || same_slot == 0u || same_slot == CO_NONE || lev0 + 3 >= CO_SB_DEPTH) break;
      /* every simulation left went to the same visited child: that node is shared as well */
      CO_SBS(20, 1);
      shX = first_bs.x;
      shown = same_slot;
      rcs = first_bs;
      ++lev0;
      CO_SBS(9, 1);
      shh0 = co_load_unit(A, shX);
      denom0 = co_u2f(co_load_unit(A, shX + 1u).y);
      {
        const uint32_t e0 = shX + 2u;
        FOR_LANES_HOT { L(ev) = A[e0 + lane]; }
      }
      n0 = (int)CO_META_NEDGES(shh0.z);
      if (n0 > CO_WAVE) { /* (a node wider than the wavefront: co_search's) */
        CO_SBS(5, 1);
        bad = 0;
        break;
      }
    }
  }
  CO_PH(28);
  /* ---- below the root: row j = simulation j */
  LV(uint32_t, X);
  LV(uint32_t, slot);
  LV(uint4, cs);
  LV(int, act);
  LV(int, leafD);      /* level of the new leaf, 0 = none (yet) */
  {
    const int nsel = bad < m ? bad : m;
    const uint4 rh0 = shh0;
    const uint32_t root = shX;
    FOR_LANES_HOT {
      const int r = lane >> 4;
      const uint4 h = sb_hand[2 * r], c = sb_hand[2 * r + 1];
      const int valid = r < nsel;
      const int isnew = valid && h.x == CO_NONE;
      L(X) = h.x;
      L(slot) = h.y;
      L(cs) = c;
      L(act) = valid && !isnew;
      L(leafD) = isnew ? lev0 + 1 : 0;
      if (isnew && (lane & 15) == 0) { /* a new edge of the shared node: the leaf is one level below it */
        sb_leaf[2 * r] = make_uint4(rh0.x, rh0.y, rh0.z, root);
        sb_leaf[2 * r + 1] = make_uint4(h.z & 0xFFFFu, h.y, 0u, 0u);
      }
    }
  }
  for (int lev = lev0 + 1;; ++lev) {
    if (!WAVE_BALLOT(act)) break;
    CO_SBS(9, 1);
    if (lev + 1 >= CO_SB_DEPTH) { /* too deep for the group's records */
      const int fr = (co_ffs64(WAVE_BALLOT(act)) - 1) >> 4;
      CO_SBS(6, 1);
      if (fr < bad) bad = fr;
      break;
    }
    {
      /* One or two rows left descending (the deep, narrow trees of a trained network: most levels): the row form costs
       * ~300 instructions a level however few rows are busy, a wave-wide scan ~110 -- so these rows are scanned wave-wide,
       * one after the other, their blocks fetched together.  (With the reference's last checkpoint the grouped search was
       * 8 % SLOWER than round 4's sequential one before this path existed) */
      const uint64_t am = WAVE_BALLOT(act);
      const int a0 = (int)(am & 1ull), a1 = (int)((am >> 16) & 1ull), a2 = (int)((am >> 32) & 1ull), a3 = (int)((am >> 48) & 1ull);
      if (a0 + a1 + a2 + a3 <= 2) {
        CO_SBS(20 + (a0 + a1 + a2 + a3), 1);
        const int r0 = a0 ? 0 : a1 ? 1 : a2 ? 2 : 3;
        const int r1 = a0 + a1 + a2 + a3 < 2 ? -1 : (a3 ? 3 : a2 ? 2 : 1); /* (the last active row: with two, the other one) */
        uint32_t wx[2], ws[2];
        uint4 wc[2], wh[2];
        uint32_t wd[2];
        LV(uint4, we[2]);
        int wr[2] = {r0, r1};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = wr[i] < 0 ? r0 : wr[i];
          wx[i] = WAVE_BCAST(X, 16 * r);
          ws[i] = WAVE_BCAST(slot, 16 * r);
          wc[i] = WAVE_BCAST(cs, 16 * r);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          wh[i] = co_load_unit(A, wx[i]);
          wd[i] = co_load_unit(A, wx[i] + 1u).y;
          const uint32_t e0 = wx[i] + 2u;
          FOR_LANES_HOT { L(we[i]) = A[e0 + lane]; }
        }
        /* the exploration factors (a double-precision square root each: ~20 dependent instructions) depend on the rows'
         * own slots only: computed HERE, while the blocks are on their way -- left to the compiler they stand behind the
         * first use of the headers (the wide-node test), i.e. behind the wait */
        float wvs[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          wvs[i] = co_vsqrt(w.c_puct, co_slot_visits(wc[i]));
          CO_OPAQUE_V(wvs[i]);
        }
        CO_PH_MEM(11);
        uint32_t pX = CO_NONE;
        int pe = -1;
        uint4 pv = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = wr[i];
          if (r < 0 || r >= bad) continue;
          CO_SBS(10, 1);
          CO_PROF_ADD(w, 23, 1ull);
          CO_STEP_WORK(w, 1);
          const int n = (int)CO_META_NEDGES(wh[i].z);
          int gone = 0; /* the row's simulation is not ordinary */
          int le = 0;
          uint4 bs = make_uint4(0u, 0u, 0u, 0u);
          if (n > CO_WAVE) {
            CO_SBS(5, 1);
            gone = 1;
          } else {
            const float denom = co_u2f(wd[i]);
            const float v_sqrt = wvs[i];
            LV(float, u);
            FOR_LANES_HOT {
              uint4 sl = L(we[i]);
              if (pe >= 0 && pX == wx[i] && lane == pe) { /* the row before scanned this very node */
                sl = pv;
                CO_SBS(11, 1);
              }
              L(we[i]) = sl;
              const float uu = co_puct_u(sl, denom, v_sqrt);
              L(u) = lane < n ? uu : CO_NEG_INF;
            }
            const float mx = WAVE_MAX_F32(u);
            wc[i] = co_slot_pass(wc[i]);
            if (!(mx > CO_NEG_INF)) {
              CO_SBS(3, 1);
              gone = 1;
              const uint32_t xs = ws[i];
              const uint4 xc = wc[i];
              FOR_LANES_HOT {
                if (lane == 0) {
                  sb_slot[r * (CO_SB_DEPTH + 1) + lev] = xs;
                  sb_cs[r * CO_SB_DEPTH + lev] = xc;
                  sb_kn[r] = lev;
                }
              }
            } else {
              LV(int, hit);
              FOR_LANES_HOT { L(hit) = (L(u) == mx); }
              le = co_ffs64(WAVE_BALLOT(hit)) - 1;
              bs = WAVE_BCAST(we[i], le);
              if (bs.x != CO_NONE && co_res_terminal(co_slot_result(bs))) {
                CO_SBS(4, 1);
                gone = 1;
              }
            }
          }
          if (gone) {
            if (r < bad) bad = r;
            FOR_LANES_HOT {
              if ((lane >> 4) >= r) L(act) = 0;
            }
            continue;
          }
          const uint32_t child_slot = wx[i] + 2u + (uint32_t)le;
          const int isnew = bs.x == CO_NONE;
          pX = wx[i];
          pe = le;
          pv = isnew ? make_uint4(0u, co_f2u(1.0f), (bs.z & 0xFFFFu) | (1u << 16), 0x100u) : co_slot_pass(bs);
          const uint32_t xb = wx[i], xs = ws[i];
          const uint4 xc = wc[i], xh = wh[i];
          FOR_LANES_HOT {
            if (lane == 0) {
              sb_block[r * CO_SB_DEPTH + lev] = xb;
              sb_slot[r * (CO_SB_DEPTH + 1) + lev] = xs;
              sb_slot[r * (CO_SB_DEPTH + 1) + lev + 1] = child_slot;
              sb_cs[r * CO_SB_DEPTH + lev] = xc;
              if (isnew) {
                sb_leaf[2 * r] = make_uint4(xh.x, xh.y, xh.z, xb);
                sb_leaf[2 * r + 1] = make_uint4(bs.z & 0xFFFFu, child_slot, 0u, 0u);
              }
            }
            if ((lane >> 4) == r) {
              if (isnew) {
                L(leafD) = lev + 1;
                L(act) = 0;
              } else {
                L(X) = bs.x;
                L(slot) = child_slot;
                L(cs) = bs;
              }
            }
          }
        }
        WAVE_SYNC();
        CO_PH(8);
        continue;
      }
    }
    CO_SBS(23, 1);
    /* the nodes of this level: header and up to 64 edge slots, four per lane */
    LV(uint4, h0);
    LV(uint32_t, dnb);
    LV(uint4, e0);
    LV(uint4, e1);
    LV(uint4, e2);
    LV(uint4, e3);
    FOR_LANES_HOT {
      if (L(act)) {
        const uint32_t b = L(X);
        const uint32_t c = b + 2u + (uint32_t)(lane & 15);
        L(h0) = A[b];
        L(dnb) = A[b + 1u].y;
        L(e0) = A[c]; /* the arena is padded: slots beyond the node's edges are read and not used */
        L(e1) = A[c + 16u];
        L(e2) = A[c + 32u];
        L(e3) = A[c + 48u];
      }
    }
    LV(float, vs); /* (a row's exploration factor is its node's: once per level, under the fetch -- not once per turn) */
    FOR_LANES_HOT {
      L(vs) = co_vsqrt(w.c_puct, co_slot_visits(L(cs)));
      CO_OPAQUE_V(L(vs));
    }
    CO_PH_MEM(11);
    LV(int, todo);
    LV(int, wide);
    FOR_LANES_HOT {
      L(wide) = L(act) && (int)CO_META_NEDGES(L(h0).z) > CO_WAVE;
      L(todo) = L(act) && !L(wide);
    }
    {
      const uint64_t wm = WAVE_BALLOT(wide);
      if (wm) {
        const int fr = (co_ffs64(wm) - 1) >> 4;
        CO_SBS(5, 1);
        if (fr < bad) bad = fr;
      }
    }
    for (;;) {
      /* rows behind a simulation that is not ordinary are dropped */
      FOR_LANES_HOT {
        if ((lane >> 4) >= bad) {
          L(todo) = 0;
          L(act) = 0;
        }
      }
      const uint64_t tm = WAVE_BALLOT(todo);
      if (!tm) break;
      CO_SBS(24, 1);
      CO_STEP_WORK(w, 2); /* (a turn in row form: the instructions of two to three wave-wide scans) */
      /* a row waits while an earlier row of the same node has yet to scan it */
      const uint32_t x0 = WAVE_BCAST(X, 0), x1 = WAVE_BCAST(X, 16), x2 = WAVE_BCAST(X, 32);
      const int t0 = (int)(tm & 1ull), t1 = (int)((tm >> 16) & 1ull), t2 = (int)((tm >> 32) & 1ull);
      LV(int, ready);
      LV(float, bu);
      LV(uint32_t, be);
      LV(uint32_t, bx);
      LV(uint32_t, by);
      LV(uint32_t, bz);
      LV(uint32_t, bw);
      FOR_LANES_HOT {
        const int r = lane >> 4;
        const int blocked = (r > 0 && t0 && x0 == L(X)) || (r > 1 && t1 && x1 == L(X)) || (r > 2 && t2 && x2 == L(X));
        L(ready) = L(todo) && !blocked;
        L(bu) = CO_NEG_INF;
        L(be) = 255u;
        L(bx) = L(by) = L(bz) = L(bw) = 0u;
        if (L(ready)) {
          CO_SBS(10, (lane & 15) == 0);
          CO_PROF_ADD(w, 23, 0ull);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        LV(int, more);
        FOR_LANES_HOT { L(more) = L(ready) && (int)CO_META_NEDGES(L(h0).z) > 16 * k; }
        if (!WAVE_BALLOT(more)) break;
        FOR_LANES_HOT {
          const uint4 sl = k == 0 ? L(e0) : k == 1 ? L(e1) : k == 2 ? L(e2) : L(e3);
          const int e = 16 * k + (lane & 15);
          float uu = co_puct_u(sl, co_u2f(L(dnb)), L(vs));
          uu = (L(ready) && e < (int)CO_META_NEDGES(L(h0).z)) ? uu : CO_NEG_INF;
          if (uu > L(bu)) { /* strict: a lane keeps its first maximum */
            L(bu) = uu;
            L(be) = (uint32_t)e;
            L(bx) = sl.x;
            L(by) = sl.y;
            L(bz) = sl.z;
            L(bw) = sl.w;
          }
        }
      }
      LV(float, mx);
      ROW_MAX_F32(mx, bu);
      LV(uint32_t, cand);
      FOR_LANES_HOT { L(cand) = (L(bu) == L(mx) && L(mx) > CO_NEG_INF) ? L(be) : 255u; }
      LV(uint32_t, emin);
      ROW_MIN_U32(emin, cand); /* the first maximum in edge order (trainmc.cpp:581: strictly greater) */
      LV(int, col);
      FOR_LANES_HOT { L(col) = (int)(L(emin) & 15u); }
      ROW_SHFL_U32(bx, bx, col);
      ROW_SHFL_U32(by, by, col);
      ROW_SHFL_U32(bz, bz, col);
      ROW_SHFL_U32(bw, bw, col);
      /* what the scan did: this node's own slot, the records, the leaf or the child, the patch for rows that wait */
      LV(int, isbad);
      LV(int, fin);
      LV(uint32_t, pX);
      LV(uint4, nv);
      FOR_LANES_HOT {
        L(isbad) = 0;
        L(fin) = L(ready);
        L(pX) = L(X);
        L(nv) = make_uint4(0u, 0u, 0u, 0u);
        if (L(ready)) {
          const int r = lane >> 4;
          const uint4 bs = make_uint4(L(bx), L(by), L(bz), L(bw));
          const int none = L(emin) == 255u;
          const int isnew = bs.x == CO_NONE;
          const int term = !isnew && co_res_terminal(co_slot_result(bs));
          L(cs) = co_slot_pass(L(cs));
          L(isbad) = none || term;
          const uint32_t child_slot = L(X) + 2u + L(emin);
          if (!L(isbad)) {
            if ((lane & 15) == 0) {
              sb_block[r * CO_SB_DEPTH + lev] = L(X);
              sb_slot[r * (CO_SB_DEPTH + 1) + lev] = L(slot);
              sb_slot[r * (CO_SB_DEPTH + 1) + lev + 1] = child_slot;
              sb_cs[r * CO_SB_DEPTH + lev] = L(cs);
            }
            if (isnew) {
              L(nv) = make_uint4(0u, co_f2u(1.0f), (bs.z & 0xFFFFu) | (1u << 16), 0x100u);
              L(leafD) = lev + 1;
              if ((lane & 15) == 0) {
                sb_leaf[2 * r] = make_uint4(L(h0).x, L(h0).y, L(h0).z, L(X));
                sb_leaf[2 * r + 1] = make_uint4(bs.z & 0xFFFFu, child_slot, 0u, 0u);
              }
              L(act) = 0;
            } else {
              L(nv) = co_slot_pass(bs);
              L(X) = bs.x;
              L(slot) = child_slot;
              L(cs) = bs;
            }
          } else {
            if (none) CO_SBS(3, (lane & 15) == 0);
            else CO_SBS(4, (lane & 15) == 0);
            if (none && (lane & 15) == 0) {
              sb_slot[r * (CO_SB_DEPTH + 1) + lev] = L(slot);
              sb_cs[r * CO_SB_DEPTH + lev] = L(cs);
              sb_kn[r] = lev;
            }
            L(act) = 0;
          }
          L(todo) = 0;
        }
      }
      {
        const uint64_t bm = WAVE_BALLOT(isbad);
        if (bm) {
          const int fr = (co_ffs64(bm) - 1) >> 4;
          if (fr < bad) bad = fr;
        }
      }
      /* rows that wait at a node one of the finished rows has just scanned: take its change */
      const uint64_t fm = WAVE_BALLOT(fin), wm2 = WAVE_BALLOT(todo);
      if (wm2) {
#pragma unroll
        for (int r2 = 0; r2 < CO_SB - 1; ++r2) {
          if (!((fm >> (16 * r2)) & 1ull)) continue;
          const uint32_t px = WAVE_BCAST(pX, 16 * r2);
          const uint32_t pe = WAVE_BCAST(emin, 16 * r2);
          const uint4 pv = WAVE_BCAST(nv, 16 * r2);
          FOR_LANES_HOT {
            if (L(todo) && (lane >> 4) > r2 && L(X) == px && (uint32_t)(lane & 15) == (pe & 15u)) {
              CO_SBS(11, 1);
              const uint32_t k = pe >> 4;
              if (k == 0u) L(e0) = pv;
              else if (k == 1u) L(e1) = pv;
              else if (k == 2u) L(e2) = pv;
              else L(e3) = pv;
            }
          }
        }
      }
    }
    CO_PH(8);
  }
  WAVE_SYNC();
  /* ---- expand: the four leaves at once, one per row (rules.h co_legal_moves_rows) */
  int nsel = bad < m ? bad : m;
  LV(int, on);
  LV(uint32_t, nb0);
  LV(uint32_t, nb1);
  LV(uint32_t, nmeta);
  LV(uint32_t, l0);
  LV(uint32_t, l1);
  LV(uint32_t, l2);
  LV(int, nl); /* legal moves of the row's leaf */
  LV(uint32_t, lz);    /* the chosen slot's move and prior */
  LV(uint32_t, lpar);  /* the leaf's parent block, the leaf's slot, the parent's meta word */
  LV(uint32_t, lslot);
  LV(uint32_t, lmeta);
  FOR_LANES_HOT {
    L(on) = (lane >> 4) < nsel;
    const uint4 la = sb_leaf[2 * (lane >> 4)], lb = sb_leaf[2 * (lane >> 4) + 1];
    L(lz) = lb.x;
    L(lslot) = lb.y;
    L(lpar) = la.w;
    L(lmeta) = la.z;
    uint64_t b = (uint64_t)la.x | ((uint64_t)la.y << 32);
    uint32_t mm = la.z;
    if (L(on)) co_do_move_lane(&b, &mm, (int)(L(lz) & 127u));
    L(nb0) = (uint32_t)b;
    L(nb1) = (uint32_t)(b >> 32);
    L(nmeta) = mm;
  }
  LV(int, lin);
  co_legal_moves_rows(nb0, nb1, nmeta, on, l0, l1, l2, lin, w.lb);
  FOR_LANES_HOT { L(nl) = L(on) ? co_popc32(L(l0)) + co_popc32(L(l1)) + co_popc32(L(l2)) : 0; }
  int term = -1; /* the simulation whose new leaf is terminal: committed behind the others, ends the group */
  {
    /* a terminal leaf ends the group behind it, a full arena in front of it (co_search reports it); the others get
     * their blocks */
    uint32_t used = t.tc.units_used;
    int j = 0;
    for (; j < nsel; ++j) {
      const int n = WAVE_BCAST(nl, 16 * j);
      if (used + 2u + (uint32_t)n > t.cap) break;
      if (n == 0) {
        CO_SBS(7, 1);
        term = j;
        break;
      }
      used += 2u + (uint32_t)n;
    }
    if (j < nsel && term < 0) bad = j; /* (not ordinary, and not handled here) */
    nsel = j;
  }
  CO_PH(14);
  int done = nsel;
  if (nsel > 0) {
    const int n0 = WAVE_BCAST(nl, 0), n1 = WAVE_BCAST(nl, 16), n2 = WAVE_BCAST(nl, 32), n3 = WAVE_BCAST(nl, 48);
    const uint32_t base0 = t.tc.units_used;
    const int noise0 = w.noise_words, k0 = w.gc.n_pending;
    /* ---- commit.  (1) the slots the simulations passed through: every (simulation, level) one lane; a slot several
     * simulations of the group passed takes the LAST one's value (it has the earlier passes in it) */
    const int dl[CO_SB] = {WAVE_BCAST(leafD, 0), WAVE_BCAST(leafD, 16), WAVE_BCAST(leafD, 32), WAVE_BCAST(leafD, 48)};
    FOR_LANES_HOT {
      const int r = lane >> 4, c = lane & 15;
      if (r < nsel && c < L(leafD)) {
        const uint32_t sl = sb_slot[r * (CO_SB_DEPTH + 1) + c];
        int last = 1;
#pragma unroll
        for (int r2 = 1; r2 < CO_SB; ++r2)
          if (r2 > r && r2 < nsel && c < dl[r2] && sb_slot[r2 * (CO_SB_DEPTH + 1) + c] == sl) last = 0;
        if (last) {
          const uint4 v = sb_cs[r * CO_SB_DEPTH + c];
          A[sl] = v;
          const uint32_t idx = sl - rc.e0;
          if (idx < rc.ne) rc.ev[idx] = v;
        }
      }
    }
    for (int j = 0; j < nsel; ++j) rc.cs = co_slot_pass(rc.cs);
    CO_PH(9);
    /* (2) the new nodes: blocks in simulation order (the bump pointer of the sequential search), header, edges in
     * ascending move id */
    LV(uint32_t, nblk);
    LV(int, noff); /* generator outputs owed to the leaves queued before this one */
    FOR_LANES_HOT {
      const int r = lane >> 4, c = lane & 15;
      const uint32_t before = (r > 0 ? 2u + (uint32_t)n0 : 0u) + (r > 1 ? 2u + (uint32_t)n1 : 0u) + (r > 2 ? 2u + (uint32_t)n2 : 0u);
      L(nblk) = base0 + before;
      L(noff) = noise0 + (r > 0 ? n0 : 0) + (r > 1 ? n1 : 0) + (r > 2 ? n2 : 0);
      if (r < nsel) {
        const uint32_t b = L(nblk);
        const int depth = (int)CO_META_DEPTH(L(lmeta)) + 1;
        if (c == 0) A[b] = make_uint4(L(nb0), L(nb1), co_meta_make(L(nmeta), depth, L(nl)), L(lpar));
        if (c == 1) A[b + 1] = make_uint4(L(lslot), 0u, 0u, 0u);
        const int c0 = co_popc32(L(l0)), c01 = c0 + co_popc32(L(l1));
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int id = 16 * q + c;
          const uint32_t wd = q < 2 ? L(l0) : q < 4 ? L(l1) : L(l2);
          const int bit = id & 31;
          if ((wd >> bit) & 1u) {
            const int rank = (q < 2 ? 0 : q < 4 ? c0 : c01) + co_popc32(wd & ((1u << bit) - 1u));
            A[b + 2u + (uint32_t)rank] = make_uint4(CO_NONE, 0u, (uint32_t)id, 0u);
          }
        }
        /* (3) the leaf's own slot, in its parent's block (and in the root copy when the parent is the root) */
        if (c == 2) {
          const uint4 v = make_uint4(b, co_f2u(1.0f), L(lz) | (1u << 16), (uint32_t)CO_RESULT_NONE | 0x100u);
          A[L(lslot)] = v;
          const uint32_t idx = L(lslot) - rc.e0;
          if (idx < rc.ne) rc.ev[idx] = v;
        }
      }
    }
    {
      const uint32_t units = 2u * (uint32_t)nsel + (uint32_t)(n0 + (nsel > 1 ? n1 : 0) + (nsel > 2 ? n2 : 0) + (nsel > 3 ? n3 : 0));
      t.tc.units_used = base0 + units;
      if (t.tc.units_used > t.tc.peak_units) t.tc.peak_units = t.tc.units_used;
      w.gc.nodes += (uint32_t)nsel;
      t.tc.searches_done += nsel;
      w.gc.searches += (uint32_t)nsel;
    }
    CO_PH(15);
    CO_PROF_ADD(w, 24, (unsigned long long)nsel);
    if (w.analyse) {
      /* Node::countNodes without a traversal (co_search): every node of a path counts the new node below it */
      for (int j = 0; j < nsel; ++j) {
        const int D = WAVE_BCAST(leafD, 16 * j);
        FOR_LANES_HOT {
          if (lane < D) A[sb_block[j * CO_SB_DEPTH + lane] + 1].z += 1u;
        }
        WAVE_SYNC();
      }
    }
    /* (4) the requests (co_request): state rows, pending-leaf records, cache keys -- leaf k0 + r from row r */
    {
      float *req = w.req;
      uint32_t *pend_leaf = w.pend_leaf, *pend_path = w.pend_path, *pend_n = w.pend_n;
      int32_t *pend_depth = w.pend_depth;
      uint4 *pend_key = w.pend_key;
      FOR_LANES_HOT {
        const int r = lane >> 4, c = lane & 15;
        if (r < nsel) {
          const int k = k0 + r;
          const uint64_t b = (uint64_t)L(nb0) | ((uint64_t)L(nb1) << 32);
          /* Game::writeGameState (game.cpp:45-58): a lane writes its cell's four floats, lanes 0..3 the reserves (the
           * mover's first) and the padding */
          uint4 *row = (uint4 *)(req + (size_t)k * CO_STATE_STRIDE);
          const uint32_t nib = (uint32_t)(b >> (4 * c)) & 15u;
          const uint32_t one = co_f2u(1.0f);
          row[c] = make_uint4((nib & 1u) ? one : 0u, (nib & 2u) ? one : 0u, (nib & 4u) ? one : 0u, (nib & 8u) ? one : 0u);
          const uint32_t pcs = L(nmeta) & 0x3FFFFu;
          const uint32_t rot = CO_META_TO_PLAY(L(nmeta)) ? ((pcs >> 9) | (pcs << 9)) & 0x3FFFFu : pcs;
          if (c < 4) {
            uint32_t f[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int x = 4 * c + i;
              f[i] = x < 6 ? co_f2u((float)((rot >> (3 * x)) & 7u) * 0.25f) : 0u;
            }
            row[16 + c] = make_uint4(f[0], f[1], f[2], f[3]);
          }
          if (c == 4) {
            pend_leaf[k] = L(nblk);
            pend_depth[k] = L(leafD);
          }
          if (c >= 5 && c < 9) pend_n[4 * k + (c - 5)] = c == 5 ? (((uint32_t)L(noff) << 8) | (uint32_t)L(nl)) : c == 6 ? L(l0) : c == 7 ? L(l1) : L(l2);
          if (c == 9 && pend_key) pend_key[k] = make_uint4(L(nb0), L(nb1), rot | 0x80000000u, 0u);
          if (c <= L(leafD)) pend_path[(size_t)k * CO_PATH_MAX + c] = sb_slot[r * (CO_SB_DEPTH + 1) + c];
        }
      }
      w.gc.n_pending = k0 + nsel;
      w.noise_words = noise0 + n0 + (nsel > 1 ? n1 : 0) + (nsel > 2 ? n2 : 0) + (nsel > 3 ? n3 : 0);
    }
    WAVE_SYNC();
    CO_PH(17);
  }
  if (term >= 0) {
    /* ---- the terminal leaf of simulation `term` (trainmc.cpp:663-682), on the tree the simulations before it have been
     * committed to: its passes, the node, the result carried up (propagateTerminal), the default evaluations of its path
     * replaced by the result.  The root copy is dropped: the caller loads it again and looks at the root's result. */
    const int j = term;
    const int D = WAVE_BCAST(leafD, 16 * j);
    const uint64_t board = (uint64_t)WAVE_BCAST(nb0, 16 * j) | ((uint64_t)WAVE_BCAST(nb1, 16 * j) << 32);
    const uint32_t meta = WAVE_BCAST(nmeta, 16 * j), z = WAVE_BCAST(lz, 16 * j), parent = WAVE_BCAST(lpar, 16 * j);
    const uint32_t child_slot = WAVE_BCAST(lslot, 16 * j);
    const int depth = (int)CO_META_DEPTH(WAVE_BCAST(lmeta, 16 * j)) + 1;
    const int res = WAVE_BCAST(lin, 16 * j) ? CO_RESULT_LOSS : CO_RESULT_DRAW;
    uint32_t *pslot = &sb_slot[j * (CO_SB_DEPTH + 1)];
    uint32_t *pblock = &sb_block[j * CO_SB_DEPTH];
    ++t.tc.searches_done;
    w.gc.searches++;
    FOR_LANES_HOT {
      if (lane < D) {
        const uint32_t sl = pslot[lane];
        const uint4 v = sb_cs[j * CO_SB_DEPTH + lane];
        A[sl] = v;
        const uint32_t idx = sl - rc.e0;
        if (idx < rc.ne) rc.ev[idx] = v;
      }
    }
    WAVE_SYNC();
    const uint32_t none[3] = {0u, 0u, 0u};
    const uint32_t nb = co_emit_node(w, t, board, meta, depth, parent, child_slot, none, 0, res);
    if (nb != CO_NONE) {
      if (w.analyse) {
        FOR_LANES_HOT {
          if (lane < D) A[pblock[lane] + 1].z += 1u;
        }
        WAVE_SYNC();
      }
      const float cur_eval = res == CO_RESULT_DRAW ? 0.0f : -1.0f;
      co_store_slot(A, child_slot, make_uint4(nb, co_f2u(cur_eval), z | (1u << 16), (uint32_t)res | 0x100u), rc);
      co_propagate_terminal(t, pblock, pslot, D);
      FOR_LANES_HOT {
        if (lane < D) {
          int kk = D - lane; /* kk-th ancestor receives eval*(-1)^(kk-1) - 1 */
          float ce = ((kk - 1) & 1) ? (float)((double)cur_eval * -1.0) : cur_eval;
          float add = (float)((double)ce - 1.0);
          uint4 s = A[pslot[lane]];
          s.y = co_f2u(co_u2f(s.y) + add);
          A[pslot[lane]] = s;
        }
      }
      WAVE_SYNC();
      CO_PROF_ADD(w, 5, 1ull);
    }
    rc.valid = 0;
    CO_PH(12);
  }
  int dead_end = 0;
  if (term < 0 && done < m && done == bad) {
    WAVE_SYNC();
    const int D = sb_kn[bad];
    if (D >= 0) {
      /* ---- simulation `bad` found nothing searchable below the node at level D (kNone, trainmc.cpp:625-640), on the tree
       * the simulations before it have been committed to: the node becomes all_visited, the passes along the path are
       * taken back (+1 visit, +1.0 and then -1 visit, -1.0: two float operations, as the reference does them), the search
       * is not counted.  The root copy is dropped. */
      const int j = bad;
      FOR_LANES_HOT {
        if (lane <= D) {
          uint4 v = sb_cs[j * CO_SB_DEPTH + lane];
          if (lane == D) v = co_slot_set_all_visited(v, 1);
          v = co_slot_set_visits(v, co_slot_visits(v) - 1);
          v.y = co_f2u(co_u2f(v.y) - 1.0f);
          A[sb_slot[j * (CO_SB_DEPTH + 1) + lane]] = v;
        }
      }
      WAVE_SYNC();
      rc.valid = 0;
      dead_end = 1;
    }
  }
  CO_SBS(0, 1);
  CO_SBS(1, m);
  CO_SBS(2, done);
  CO_SBP(w, 0, 1);
  CO_SBP(w, 1, m);
  CO_SBP(w, 2, done);
  CO_SBP(w, 3, term >= 0);
  CO_SBP(w, 4, term < 0 && !dead_end && done < m);
  CO_SBS(12 + done, 1);
  CO_PROF_ADD(w, 5, (unsigned long long)done);
  /* simulations committed | 0x100: a terminal leaf behind them ended the group (done here) | 0x200: the group stopped at a
   * simulation that is neither ordinary nor a terminal leaf -- the caller's next simulation takes co_search */
  return done | (term >= 0 || dead_end ? 0x100 : 0) | (term < 0 && !dead_end && done < m ? 0x200 : 0);
}
#endif

/* the root asks for its own evaluation (trainmc.cpp:143-167, 198-202) */
CO_DEV void co_request_root(CoWave &w, CoTree &t) {
  uint4 *A = t.A;
  uint32_t root = t.tc.root;
  uint4 h0 = co_load_unit(A, root);
  uint4 h1 = co_load_unit(A, root + 1);
  WAVE_SHARED(uint32_t, one_path, 1);
  uint32_t self_slot = h1.x;
  FOR_LANES {
    if (lane == 0) one_path[0] = self_slot;
  }
  WAVE_SYNC();
  uint64_t board = (uint64_t)h0.x | ((uint64_t)h0.y << 32);
  uint32_t lm[3];
  co_legal_moves1(board, h0.z, lm); /* (a root asks once per tree; its edges hold the same moves) */
  co_request(w, board, h0.z, root, (int)CO_META_NEDGES(h0.z), lm, 0, one_path);
}

/* TrainMC::doIteration, trainmc.cpp:139-178.  Returns "turn finished". */
CO_DEV int co_mc_do_iteration(CoWave &w, CoTree &t, const float *eval, const float *probs) {
  if (t.tc.root == CO_NONE) {
    int res;
    /* a search from the start position; in analysis mode TrainMC(..., board, to_play, pieces) ->
     * createRoot(Game{...}, 0) (trainmc.cpp:38-45), whose first doIteration asks for the root's evaluation */
    const uint64_t b0 = w.analyse ? ((uint64_t)w.gc.pos_lo | ((uint64_t)w.gc.pos_hi << 32)) : 0ull;
    uint32_t b = co_create_node(w, t, b0, w.analyse ? (w.gc.pos_meta & 0x7FFFFu) : CO_META_START, 0, CO_NONE, CO_NONE, &res);
    if (b == CO_NONE) return 0;
    t.tc.root = b;
    t.tc.searches_done = 1;
    co_request_root(w, t);
    return 0;
  }
  if (t.tc.searches_done == 0) { /* (only then: the root's slot is two dependent trips to memory in front of everything else) */
    const uint4 rs = co_load_unit(t.A, co_load_unit(t.A, t.tc.root + 1).x);
    if (co_slot_visits(rs) == 1 && co_slot_all_visited(rs)) {
      t.tc.searches_done = 1;
      co_request_root(w, t);
      return 0;
    }
  }
  /* (held: the step before stopped at its budget; its leaves were not submitted -- nothing to receive, the search goes on) */
  if (w.gc.n_pending > 0 && !(w.gc.held & 1)) co_receive_eval(w, t, eval, probs);
  w.gc.held = (w.gc.held & 2) | ((w.gc.held & 1) << 1); /* bit 1: this step continues one that was cut (co_step_tail's statistics) */
  WAVE_SHARED(uint4, root_ev, CO_WAVE);
  CoRoot rc;
  rc.valid = 0;
  rc.hdr = 0;
  rc.ev = root_ev;
  rc.e0 = 0u;
  rc.ne = 0u;
#if CO_SB > 1
  /* How many to select together follows from how the position's simulations have been ending: everything behind one that
   * is not ordinary is selected in vain (endgames of a trained network: every second simulation a terminal leaf or a
   * dead end, ten levels down).  pe = running share of such simulations in 1/256 (seven eighths of the old estimate per
   * simulation, kept with the game from step to step): above 3/8 they run one after the other, above 5/32 two at a time. */
  int pe = w.gc.sb_cap > 0 && w.gc.sb_cap <= 256 ? w.gc.sb_cap : 0;
#endif
  /* (bit 2 of gc.held: this call has run a simulation.  A step whose budget is spent before its first one -- a deadline
   * already behind the receive phase -- still makes progress: else it would stop again and again, for ever.  In the
   * game's record because one more scalar kept across the loop costs the kernel 200 spilled registers.) */
  for (;;) {
    if (!(w.gc.n_pending < w.spe && t.tc.searches_done < w.max_searches)) break;
    if (!rc.valid) co_root_load(t, rc);

    if (co_res_known(co_slot_result(rc.cs)) || co_slot_all_visited(rc.cs)) break;
    if (w.gc.error) break;
    if ((w.gc.held & 4) && CO_STEP_SPENT(w)) {
      /* more simulations are due and this step has done its share: the next launch continues here (co_step_tail holds
       * the queued leaves back).  Everything the loop carries is in the tree, in the game's record or in `pe`. */
      w.gc.held |= 1;
      break;
    }
    CO_PH_MEM(20);
#if CO_SB > 1
    int counted = 0;
    {
      const int cap = pe > CO_SB_PE_ONE ? 1 : pe > CO_SB_PE_TWO ? 2 : CO_SB;
      int m = w.spe - w.gc.n_pending;
      if (w.max_searches - t.tc.searches_done < m) m = w.max_searches - t.tc.searches_done;
      if (m > cap) m = cap;
      if (m > 1 && (int)CO_META_NEDGES(rc.h0.z) <= CO_WAVE) {
        const int r = co_search_rows(w, t, rc, m);
        w.gc.held |= 4;
        for (int i = 0; i < (r & 0xFF); ++i) pe -= pe >> 3;
        if (r & 0x300) {
          pe += (256 - pe) >> 3;
          counted = 1;
        }
        if (!(r & 0x200)) continue;
      }
    }
    const int pending_before = w.gc.n_pending;
#endif
    CO_SBS(8, 1);
    CO_SBP(w, 6, 1);
    co_search(w, t, rc);
    w.gc.held |= 4;
    CO_PROF_ADD(w, 5, 1ull);
#if CO_SB > 1
    if (!counted) {
      if (w.gc.n_pending > pending_before) pe -= pe >> 3; /* (it queued a leaf: ordinary) */
      else pe += (256 - pe) >> 3;
    }
#endif
  }
#if CO_SB > 1
  w.gc.sb_cap = pe > 0 ? pe : 0;
#endif
  /* (the root's slot only when the answer depends on it: with leaves pending -- nearly every step -- it does not) */
  if (w.gc.n_pending != 0 || (w.gc.held & 1)) return 0;
  if (t.tc.searches_done == w.max_searches) return 1;
  const uint4 rs = co_load_unit(t.A, co_load_unit(t.A, t.tc.root + 1).x);
  return co_res_known(co_slot_result(rs));
}

/* TrainMC::moveDown, trainmc.cpp:475-495: the chosen child becomes the root.
 * Nothing is freed: the old root's block stays behind (it still holds the new
 * root's stat slot). */
CO_DEV void co_move_down(CoTree &t, uint32_t child_block) {
  uint4 h0 = co_load_unit(t.A, child_block);
  h0.w = CO_NONE; /* null_parent */
  co_store_unit(t.A, child_block, h0);
  t.tc.root = child_block;
  t.tc.searches_done = 0;
}

/* "Reset tree": a fresh root one move below the current one
 * (trainmc.cpp:398-407, 455-464). */
CO_DEV void co_reset_tree_to_child(CoWave &w, CoTree &t, int choice) {
  uint4 h0 = co_load_unit(t.A, t.tc.root);
  uint64_t board = (uint64_t)h0.x | ((uint64_t)h0.y << 32);
  uint32_t meta = h0.z;
  int depth = (int)CO_META_DEPTH(meta) + 1;
  co_do_move_lane(&board, &meta, choice);
  int res;
  uint32_t b = co_create_node(w, t, board, meta, depth, CO_NONE, CO_NONE, &res);
  if (b == CO_NONE) return;
  t.tc.root = b;
  t.tc.searches_done = 0;
}

/* ---- per-game text logs (SelfPlayer::writePreMoveLogs / writeMoveChoice, selfplayer.cpp:185-204).  The device records the
 * numbers, the host prints them (engine.hip write_logs).  One record per move choice:
 *   1, to_play, depth, root visits, root result, root evaluation bits,
 *   nc, nc x {move, visits, evaluation bits, result, probability bits}        the root's children (writeMoves :136-183)
 *   m x {depth, move, visits, result, evaluation bits, probability bits}, -1   the main line (Node::writeMainLine,
 *                                                                              node.cpp:197-240): most visits, then the
 *                                                                              larger evaluation; a lost child at once
 *   [after the choice, co_game_step] move, board low, board high, meta of the new position
 * and for the move of a tournament's random player: 2, move, board low, board high, meta.
 * `ch` is the caller's scratch for one node's slots.  Logged games are few (Trainer's default is 10); their waves spend a
 * few microseconds here once per ply. */
CO_DEV void co_log_put(CoWave &w, int &at, int32_t v) {
  if (at + 1 < CO_LOG_CAP) {
    int32_t *dst = w.log + 1 + at;
    FOR_LANES {
      if (lane == 0) *dst = v;
    }
  }
  ++at;
}
CO_DEV void co_log_commit(CoWave &w, int at) {
  int32_t *dst = w.log;
  FOR_LANES {
    if (lane == 0) *dst = at;
  }
  WAVE_SYNC();
}

CO_DEV void co_log_ply(CoWave &w, CoTree &t, uint4 (&ch)[CO_NUM_MOVES]) {
  uint4 *A = t.A;
  uint32_t node = t.tc.root;
  uint4 h0 = co_load_unit(A, node);
  uint4 h1 = co_load_unit(A, node + 1);
  uint4 rs = co_load_unit(A, h1.x);
  int at = w.log[0];
  co_log_put(w, at, 1);
  co_log_put(w, at, w.gc.to_play);
  co_log_put(w, at, (int)CO_META_DEPTH(h0.z));
  co_log_put(w, at, co_slot_visits(rs));
  co_log_put(w, at, co_slot_result(rs));
  co_log_put(w, at, (int32_t)rs.y);
  int depth = (int)CO_META_DEPTH(h0.z);
  for (int level = 0;; ++level) {
    int n = (int)CO_META_NEDGES(h0.z);
    float denom = co_u2f(h1.y);
    WAVE_SYNC();
    for (int base = 0; base < n; base += CO_WAVE) {
      FOR_LANES {
        int e = base + lane;
        if (e < n) ch[e] = A[node + 2 + e];
      }
    }
    WAVE_SYNC();
    if (level == 0) {
      int nc = 0;
      for (int e = 0; e < n; ++e) nc += ch[e].x != CO_NONE;
      co_log_put(w, at, nc);
      for (int e = 0; e < n; ++e) {
        if (ch[e].x == CO_NONE) continue;
        co_log_put(w, at, (int)(ch[e].z & 127u));
        co_log_put(w, at, co_slot_visits(ch[e]));
        co_log_put(w, at, (int32_t)ch[e].y);
        co_log_put(w, at, co_slot_result(ch[e]));
        co_log_put(w, at, (int32_t)co_f2u((float)((ch[e].z >> 7) & 511u) * denom));
      }
    }
    int best = -1, max_visits = 0;
    float max_eval = 0.0f;
    for (int e = 0; e < n; ++e) {
      if (ch[e].x == CO_NONE) continue;
      int r = co_slot_result(ch[e]), v = co_slot_visits(ch[e]);
      float ev = co_u2f(ch[e].y);
      if (co_res_lost(r)) {
        best = e;
        break;
      }
      if (v > max_visits || (v == max_visits && ev > max_eval)) {
        best = e;
        max_visits = v;
        max_eval = ev;
      }
    }
    if (best < 0) break;
    ++depth;
    co_log_put(w, at, depth);
    co_log_put(w, at, (int)(ch[best].z & 127u));
    co_log_put(w, at, co_slot_visits(ch[best]));
    co_log_put(w, at, co_slot_result(ch[best]));
    co_log_put(w, at, (int32_t)ch[best].y);
    co_log_put(w, at, (int32_t)co_f2u((float)((ch[best].z >> 7) & 511u) * denom));
    node = ch[best].x;
    h0 = co_load_unit(A, node);
    h1 = co_load_unit(A, node + 1);
  }
  co_log_put(w, at, -1);
  co_log_commit(w, at);
}

/* TrainMC::chooseMove and its four variants, trainmc.cpp:110-137, 298-473.
 * sample = this ply's (state[70], policy[96]) row, or null in testing mode. */
CO_COLD int co_choose_move(CoWave &w, CoTree &t, float *sample) {
  uint4 *A = t.A;
  uint32_t root = t.tc.root;
  uint4 h0 = co_load_unit(A, root);
  uint4 h1 = co_load_unit(A, root + 1);
  uint4 rs = co_load_unit(A, h1.x);
  int n = (int)CO_META_NEDGES(h0.z);
  float denom = co_u2f(h1.y);
  int rres = co_slot_result(rs);
  uint64_t board = (uint64_t)h0.x | ((uint64_t)h0.y << 32);
  float *policy = sample ? sample + CO_GAME_STATE_SIZE : (float *)0;
  if (!w.testing) {
    /* state first (70 floats), then a zeroed policy */
    WAVE_SHARED(float, srow, CO_STATE_STRIDE);
    co_write_state(board, h0.z, srow);
    WAVE_SYNC();
    FOR_LANES {
      sample[lane] = srow[lane];
      if (lane < CO_GAME_STATE_SIZE - 64) sample[64 + lane] = srow[64 + lane];
      policy[lane] = 0.0f;
      if (lane < CO_NUM_MOVES - 64) policy[64 + lane] = 0.0f;
    }
    WAVE_SYNC();
  }
  /* children in edge order = the reference's sorted sibling list */
  WAVE_SHARED(uint4, ch, CO_NUM_MOVES);
  if (w.log) co_log_ply(w, t, ch); /* (uses ch for the nodes of the main line) */
  for (int base = 0; base < n; base += CO_WAVE) {
    FOR_LANES {
      int e = base + lane;
      if (e < n) ch[e] = A[root + 2 + e];
    }
  }
  WAVE_SYNC();
  if (w.trace_on) {
    co_trace_push(w, w.gc.to_play);
    co_trace_push(w, (int)CO_META_DEPTH(h0.z));
    co_trace_push(w, co_slot_visits(rs));
    co_trace_push(w, rres);
    co_trace_push(w, (int32_t)rs.y);
    int nc = 0;
    for (int e = 0; e < n; ++e) nc += ch[e].x != CO_NONE;
    co_trace_push(w, nc);
    for (int e = 0; e < n; ++e) {
      if (ch[e].x == CO_NONE) continue;
      co_trace_push(w, (int)(ch[e].z & 127u));
      co_trace_push(w, co_slot_visits(ch[e]));
      co_trace_push(w, (int32_t)ch[e].y);
      co_trace_push(w, co_slot_result(ch[e]));
      co_trace_push(w, co_slot_all_visited(ch[e]));
    }
  }
  int choice = 0;
  int chosen_e = -1;
  if (co_res_won(rres)) {
    /* chooseMoveWon :310-335: first child that is lost */
    for (int e = 0; e < n; ++e) {
      if (ch[e].x != CO_NONE && co_res_lost(co_slot_result(ch[e]))) {
        choice = (int)(ch[e].z & 127u);
        chosen_e = e;
        break;
      }
    }
  } else if (co_res_lost(rres) || co_res_drawn(rres)) {
    /* chooseMoveLostDrawn :337-361 */
    int max_visits = 0;
    for (int e = 0; e < n; ++e) {
      if (ch[e].x == CO_NONE) continue;
      int v = co_slot_visits(ch[e]);
      if (v > max_visits && (co_res_lost(rres) || !co_res_won(co_slot_result(ch[e])))) {
        choice = (int)(ch[e].z & 127u);
        chosen_e = e;
        max_visits = v;
      }
    }
  } else {
    /* chooseHighProbMove :298-308 (int32 max_prob, sic) */
    int32_t max_prob = 0;
    for (int e = 0; e < n; ++e) {
      float p = (float)((ch[e].z >> 7) & 511u) * denom;
      if (p > (float)max_prob) {
        max_prob = (int32_t)p;
        choice = (int)(ch[e].z & 127u);
      }
    }
    if ((int)CO_META_DEPTH(h0.z) < 6 && !w.testing) {
      /* chooseMoveOpening :363-427 */
      int32_t visits = 0;
      for (int e = 0; e < n; ++e)
        if (ch[e].x != CO_NONE && !co_res_won(co_slot_result(ch[e]))) visits += co_slot_visits(ch[e]);
      float denominator = (float)(1.0 / (double)(float)visits);
      FOR_LANES {
        for (int e = lane; e < n; e += CO_WAVE)
          if (ch[e].x != CO_NONE && !co_res_won(co_slot_result(ch[e])))
            policy[ch[e].z & 127u] = (float)co_slot_visits(ch[e]) * denominator;
      }
      WAVE_SYNC();
      if (visits == 0) {
        FOR_LANES {
          if (lane == 0) policy[choice] = 1.0f;
        }
        WAVE_SYNC();
        co_reset_tree_to_child(w, t, choice);
        return choice;
      }
      int32_t target = (int32_t)(co_wave_mt_next(w) % (uint32_t)visits);
      int32_t total = 0;
      for (int e = 0; e < n; ++e) {
        if (ch[e].x == CO_NONE || co_res_won(co_slot_result(ch[e]))) continue;
        total += co_slot_visits(ch[e]);
        if (total > target) {
          choice = (int)(ch[e].z & 127u);
          chosen_e = e;
          break;
        }
      }
      if (chosen_e < 0) {
        w.gc.error |= CO_ERR_INTERNAL;
        return choice;
      }
      co_move_down(t, ch[chosen_e].x);
      return choice;
    }
    /* chooseMoveNormal :429-473 */
    int max_visits = 0;
    float max_eval = 0.0f;
    for (int e = 0; e < n; ++e) {
      if (ch[e].x == CO_NONE) continue;
      int r = co_slot_result(ch[e]);
      if (co_res_won(r)) continue;
      float ev = co_u2f(ch[e].y);
      if (r == CO_RESULT_DRAW || r == CO_DEDUCED_DRAW) ev = 0.0f;
      int v = co_slot_visits(ch[e]);
      if (v > max_visits || (v == max_visits && ev > max_eval)) {
        choice = (int)(ch[e].z & 127u);
        chosen_e = e;
        max_visits = v;
        max_eval = ev;
      }
    }
    if (policy) {
      FOR_LANES {
        if (lane == 0) policy[choice] = 1.0f;
      }
      WAVE_SYNC();
    }
    if (max_visits == 0) {
      co_reset_tree_to_child(w, t, choice);
      return choice;
    }
    co_move_down(t, ch[chosen_e].x);
    return choice;
  }
  /* won / lost / drawn roots */
  if (policy) {
    FOR_LANES {
      if (lane == 0) policy[choice] = 1.0f;
    }
    WAVE_SYNC();
  }
  if (chosen_e < 0) {
    /* the reference would move to the first child here; it cannot happen for a
     * root whose result was deduced from its children */
    w.gc.error |= CO_ERR_INTERNAL;
    return choice;
  }
  co_move_down(t, ch[chosen_e].x);
  return choice;
}

/* TrainMC::receiveOpponentMove, trainmc.cpp:180-204.  (board, meta) is the
 * opponent's new root position.  Returns "needs an evaluation". */
CO_COLD int co_receive_opponent_move(CoWave &w, CoTree &t, int move_choice, uint64_t board, uint32_t meta_game,
                                    int depth) {
  uint4 *A = t.A;
  uint32_t root = t.tc.root;
  int n = (int)CO_META_NEDGES(co_load_unit(A, root).z);
  uint32_t found = CO_NONE;
  for (int base = 0; base < n; base += CO_WAVE) {
    LV(int, hit);
    LV(uint32_t, cb);
    FOR_LANES {
      int e = base + lane;
      L(hit) = 0;
      L(cb) = CO_NONE;
      if (e < n) {
        uint4 s = A[root + 2 + e];
        L(cb) = s.x;
        L(hit) = (s.x != CO_NONE) && ((int)(s.z & 127u) == move_choice);
      }
    }
    uint64_t m = WAVE_BALLOT(hit);
    if (m) found = WAVE_BCAST(cb, co_ffs64(m) - 1);
  }
  if (found != CO_NONE) {
    co_move_down(t, found);
    return 0;
  }
  int res;
  uint32_t b = co_create_node(w, t, board, meta_game, depth, CO_NONE, CO_NONE, &res);
  if (b == CO_NONE) return 1;
  t.tc.root = b;
  co_request_root(w, t);
  t.tc.searches_done = 1;
  return 1;
}

/* Tournament matches (match.h, match.cpp): the two sides search with their own settings */
CO_DEV void co_use_player(CoWave &w, int p) {
  if (!w.pc) return;
  w.max_searches = w.pc[p].max_searches;
  w.spe = w.pc[p].searches_per_eval;
  w.c_puct = w.pc[p].c_puct;
  w.epsilon = w.pc[p].epsilon;
}

/* std::uniform_int_distribution<int32_t>(0, n - 1)(generator_) of match.cpp:199-200 as libstdc++
 * (GCC >= 11) computes it on a 32-bit generator: Lemire's nearly divisionless method, one
 * 64-bit product per draw, a draw consumed even for n == 1 */
CO_DEV uint32_t co_uniform_below(CoWave &w, uint32_t n) {
  unsigned long long product = (unsigned long long)co_wave_mt_next(w) * (unsigned long long)n;
  uint32_t low = (uint32_t)product;
  if (low < n) {
    uint32_t threshold = (0u - n) % n;
    while (low < threshold) {
      product = (unsigned long long)co_wave_mt_next(w) * (unsigned long long)n;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}

/* the k-th (0-based) set bit of a 96-bit legal-move mask = Node::move_id(k) */
CO_DEV int co_nth_move(const uint32_t lm[3], int k) {
  for (int wd = 0; wd < 3; ++wd) {
    int c = co_popc32(lm[wd]);
    if (k < c) {
      uint32_t m = lm[wd];
      for (int i = 0; i < k; ++i) m &= m - 1u;
      return wd * 32 + co_ffs64((uint64_t)m) - 1;
    }
    k -= c;
  }
  return -1;
}

/* Analysis mode: what docker/choose_move.pyx:206-221 reads off the DockerMC after chooseMove(), written as
 * eight words into the slot's request area: {move, Node result of the new root (util.h:57-64), nodes in
 * the kept tree (num_nodes), bits of the new root's summed evaluation (eval), legal-move mask of the new
 * root [3], 1} -- done() = result is a terminal one, drawn() = result is a draw. */
CO_DEV void co_analyse_finish(CoWave &w, CoTree &t, int choice) {
  uint4 h0 = co_load_unit(t.A, t.tc.root);
  uint4 h1 = co_load_unit(t.A, t.tc.root + 1);
  uint4 rs = co_load_unit(t.A, h1.x);
  uint32_t lm[3];
  co_legal_moves1((uint64_t)h0.x | ((uint64_t)h0.y << 32), h0.z, lm);
  uint32_t *out = (uint32_t *)w.req;
  FOR_LANES {
    if (lane == 0) {
      out[0] = (uint32_t)choice;
      out[1] = (uint32_t)co_slot_result(rs);
      out[2] = h1.z + 1u;
      out[3] = rs.y;
      out[4] = lm[0];
      out[5] = lm[1];
      out[6] = lm[2];
      out[7] = 1u;
    }
  }
  WAVE_SYNC();
}

/* One step of a game:
 *   self-play   SelfPlayer::doIteration (selfplayer.cpp:115-122) + chooseMoveAndContinue
 *               (:246-291; chooseMove :234-244, endGame :206-232),
 *   tournament  Match::doIteration (match.cpp:67-79) + chooseMoveAndContinue (:207-251; chooseMove
 *               :192-205, endGame :163-190); gc.pos_* is Match::root_, a random side
 *               (players_[i] == nullptr) has no tree.
 * The reference's mutual recursion doIteration -> chooseMoveAndContinue -> doIteration is
 * written as ONE loop so that the kernel holds a single copy of the search, of the move choice
 * and of the hand-over: inlined along the call chain they were repeated up to seven times
 * (250 KB of instructions in front of a 64 KB instruction cache).  Returns "game over". */
CO_DEV int co_game_step(CoWave &w, const float *eval, const float *probs) {
  const float *ev = eval, *pr = probs;
  if (w.gc.resume) {
    /* continuation of a deferred hand-over: selfplayer.cpp:287-288 */
    w.gc.resume = 0;
    ev = pr = (const float *)0;
  }
  if (w.gc.held & 1) ev = pr = (const float *)0; /* continuation of a step cut at its budget: nothing was submitted */
  int skip_iteration = (w.pc && w.pc[w.gc.to_play].random) /* match.cpp:68-70 */ || w.force_choose;
  int fresh_root = 0;
  for (;;) {
    if (!skip_iteration) {
      int done = co_mc_do_iteration(w, w.me, ev, pr);
      if (w.gc.error) return 0;
      /* `return players_[to_play_]->doIteration()` after createRoot (selfplayer.cpp:281-283,
       * match.cpp:236-239): the new root asks for its evaluation, so this is "not over" */
      if (fresh_root) return done;
      if (!done) return 0;
    }
    skip_iteration = 0;
    ev = pr = (const float *)0;
    /* ---- chooseMoveAndContinue, one ply per pass of the loop */
    CO_PH(20);
    int p = w.gc.to_play;
    int choice, terminal, tres, depth;
    uint64_t board;
    uint32_t meta;
    if (w.pc && w.pc[p].random) {
      board = (uint64_t)w.gc.pos_lo | ((uint64_t)w.gc.pos_hi << 32);
      meta = w.gc.pos_meta;
      uint32_t lm[3];
      co_legal_moves1(board, meta, lm);
      int n = co_popc32(lm[0]) + co_popc32(lm[1]) + co_popc32(lm[2]);
      choice = co_nth_move(lm, (int)co_uniform_below(w, (uint32_t)n));
      co_trace_push(w, -2);
      co_trace_push(w, choice);
      w.gc.plies++;
      co_do_move_lane(&board, &meta, choice);
      int lines = co_legal_moves1(board, meta, lm);
      terminal = (lm[0] | lm[1] | lm[2]) == 0u;
      tres = lines ? CO_RESULT_LOSS : CO_RESULT_DRAW;
      depth = w.gc.plies; /* root_->depth() */
      if (w.log) { /* a random side has no pre-move block (match.cpp:213-221): record type 2 */
        int at = w.log[0];
        co_log_put(w, at, 2);
        co_log_put(w, at, choice);
        co_log_put(w, at, (int32_t)(uint32_t)board);
        co_log_put(w, at, (int32_t)(uint32_t)(board >> 32));
        co_log_put(w, at, (int32_t)meta);
        co_log_commit(w, at);
      }
    } else {
      CoTree &me = w.me;
      if (!w.pc) {
        uint4 rs = co_load_unit(me.A, co_load_unit(me.A, me.tc.root + 1).x);
        if (co_res_known(co_slot_result(rs)) && w.gc.mate_turn == 0) w.gc.mate_turn = w.gc.n_samples + 1;
      }
      float *sample = (float *)0;
      if (!w.testing) {
        if (w.gc.n_samples >= CO_MAX_PLIES) {
          w.gc.error |= CO_ERR_TOO_MANY_PLIES;
          return 0;
        }
        sample = w.samples + (size_t)w.gc.n_samples * CO_SAMPLE_FLOATS;
      }
      choice = co_choose_move(w, me, sample);
      if (w.gc.error) return 0;
      if (w.analyse) {
        co_analyse_finish(w, me, choice);
        w.gc.plies++;
        return 1;
      }
      if (!w.testing) w.gc.n_samples++;
      co_trace_push(w, choice);
      w.gc.plies++;
      /* new root of the mover = the position on the board */
      uint4 h0 = co_load_unit(me.A, me.tc.root);
      uint4 nrs = co_load_unit(me.A, co_load_unit(me.A, me.tc.root + 1).x);
      tres = co_slot_result(nrs);
      terminal = co_res_terminal(tres);
      board = (uint64_t)h0.x | ((uint64_t)h0.y << 32);
      meta = h0.z;
      depth = (int)CO_META_DEPTH(h0.z);
      if (w.log) { /* writeMoveChoice, selfplayer.cpp:199-204 */
        int at = w.log[0];
        co_log_put(w, at, choice);
        co_log_put(w, at, (int32_t)h0.x);
        co_log_put(w, at, (int32_t)h0.y);
        co_log_put(w, at, (int32_t)meta);
        co_log_commit(w, at);
      }
    }
    if (w.pc) {
      w.gc.pos_lo = (uint32_t)board;
      w.gc.pos_hi = (uint32_t)(board >> 32);
      w.gc.pos_meta = meta & 0x7FFFFu; /* reserves and side to move */
    }
    CO_PH_MEM(2);
    if (terminal) {
      if (tres == CO_RESULT_DRAW) w.gc.result = CO_RESULT_DRAW;
      else if (p == 1) w.gc.result = CO_RESULT_LOSS;
      else w.gc.result = CO_RESULT_WIN;
      return 1;
    }
    /* hand over: the opponent becomes the player to move */
    w.gc.to_play = 1 - p;
    {
      CoTree tmp = w.me;
      w.me = w.opp;
      w.opp = tmp;
    }
    co_use_player(w, 1 - p);
    if (w.pc && w.pc[1 - p].random) {
      skip_iteration = 1;
      continue;
    }
    CoTree &nm = w.me; /* the new player to move */
    if (nm.tc.root == CO_NONE) {
      /* first move of the second player: createRoot + doIteration */
      int res;
      uint32_t b = co_create_node(w, nm, board, meta, depth, CO_NONE, CO_NONE, &res);
      if (b == CO_NONE) return 0;
      nm.tc.root = b;
      fresh_root = 1;
      continue;
    }
    if (co_receive_opponent_move(w, nm, choice, board, meta, depth)) return 0; /* needs an evaluation */
    if (w.defer_handover) {
      /* Lock-step scheduling only: games are independent, so the new mover's first searches
       * may as well run in the next step.  Without this, the few games that change turns in
       * a step run up to twice the simulations of the others and every launch waits for
       * them.  The game's own sequence of operations is unchanged. */
      w.gc.resume = 1;
      return 0;
    }
    /* else: `need_eval = !doIteration()` -- search on; a finished turn chooses again */
  }
}

/* Does game g step in this launch?  0 no, 1 yes, 2 not yet released by the staggered start.
 * (trainer.cpp:176-196 training, :216-229 arena, tourney.cpp:63-72) */
CO_DEV int co_step_gate(const EngineParams &P, int g, const GameCtl &gc) {
  if (gc.done || gc.error) return 0;
  const int tp = P.arena_state ? P.arena_state[0] : P.to_play;
  if (P.pcfg) {
    if (P.pcfg[2 * g + gc.to_play].model_id != tp) return 0; /* tourney.cpp:66 */
  } else if (tp == 0 || tp == 1) {
    if (gc.to_play != (tp + gc.parity) % 2) return 0;
  } else if (P.stagger_div > 0) {
    if ((P.game_base + g) / P.stagger_div > P.iteration) return 2;
  }
  return 1;
}

/* first row of game g's evaluations in nn_eval / nn_probs */
CO_DEV int co_step_row(const EngineParams &P, int g, const GameCtl &gc) {
  return P.fused_pack ? gc.row_off : P.read_offset ? P.read_offset[g] : P.req_offset[g];
}

/* ---- evaluation cache (engine_defs.h EvalCache), fused training: the tail of a game's step resolves its new request
 * rows to elements of the cache's value array, ONE LANE per row:
 *   the position has an entry of this pool (evaluated in an earlier iteration, or claimed by another row of this
 *   batch, whose outputs this iteration's network launch writes: stream order)   -> that entry;
 *   an entry of ANOTHER pool: its outputs are there once that pool's network launch of the claiming iteration has
 *   completed, which the pool's next search launch publishes (EvalCache::done)     -> that entry, if published;
 *   else the row is evaluated into its scratch element (nothing is waited for, no second entry is made);
 *   no entry, an empty slot in the probe window                -> the lane claims it, the row is evaluated into it;
 *   neither (the window is full)                               -> the row is evaluated into its scratch element.
 * Rows to evaluate are numbered with one atomic per game.  Correctness does not depend on who wins a race: every
 * path resolves a row to the network kernel's outputs for exactly its position (rows are evaluated independently
 * of their batch; keys are compared in full; an entry is never moved or reused before the table is
 * emptied as a whole, between two iterations).  Header words are accessed with relaxed device-scope atomics only:
 * the XCDs' L2s are not coherent with each other within a launch, and an acquire / release would invalidate /
 * write back a whole L2 per wave.  (Round 3 first had this as a kernel of its own between the search and the
 * network: 36 us per launch beside the other pool's network kernel, most of it waiting for a SIMD.) */
CO_DEV uint32_t co_cache_hash(uint32_t k0, uint32_t k1, uint32_t k2) {
  uint32_t h = k0 * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + k1 * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + k2 * 0xC2B2AE3Du;
  h ^= h >> 16;
  h *= 0x27D4EB2Fu;
  return h ^ (h >> 15);
}

/* request rows [row0, row0 + n) of `req` = the pending leaves 0 .. n - 1 of game g (keys in w.pend_key) */
CO_DEV void co_cache_resolve(const EngineParams &P, CoWave &w, int g, int n, int row0) {
  const EvalCache &C = P.cache;
  const int par = P.iteration & 1;
  int may_claim = !C.no_claim;
  if (may_claim && C.guard_pools) { /* an emptying not long ago: see EvalCache::guard_from */
    for (uint32_t p = 0; p < 4u; ++p)
      if (((C.guard_pools >> p) & 1u) && co_atomic_load_u32(C.done + p) <= C.guard_from) may_claim = 0;
  }
  LV(int, slot);
  LV(int, need);
  FOR_LANES {
    int sl = -1, nd = 0;
    if (lane < n) {
      const uint4 key = w.pend_key[lane];
      const uint32_t h = co_cache_hash(key.x, key.y, key.z);
      nd = 1;
      /* Header = {board lo ^ X, board hi ^ X, reserves | 1 << 31}; X sets the frozen bit of every cell, which no
       * position has (at most one cell is frozen, game.cpp:66-71): a header word that has not been written yet (0)
       * never equals a stored word, so the three words of an entry may become visible in any order -- a reader
       * takes an entry for its position only when all three are there.  The third word is the claim: empty = 0. */
      const uint32_t x0 = key.x ^ 0x88888888u, x1 = key.y ^ 0x88888888u;
      /* The claim word also names the claiming pool (bits the key leaves free), so whose entry it is is known with
       * the claim itself; the iteration of the claim follows in word 3 and matters to the OTHER pools only. */
      const uint32_t mine = key.z | C.pool_bits;
      for (int probe = 0; probe < CO_CACHE_PROBES; ++probe) {
        const uint32_t s = (h + (uint32_t)probe) & C.mask;
        uint32_t *H = C.hdr + (size_t)s * 4;
        uint32_t e0 = co_lane_load_coherent_u32(H + 0), e1 = co_lane_load_coherent_u32(H + 1), e2 = co_lane_load_coherent_u32(H + 2);
        if (e2 == 0u) {
          if (!may_claim) break; /* (EvalCache::no_claim, guard_from: the row is evaluated into its scratch element) */
          e2 = co_lane_cas_u32(H + 2, 0u, mine);
          if (e2 == 0u) {
            co_lane_store_coherent_u32(H + 0, x0);
            co_lane_store_coherent_u32(H + 1, x1);
            co_lane_store_coherent_u32(H + 3, (uint32_t)P.iteration + 1u);
            sl = (int)s; /* ours: this row is evaluated into the entry */
            break;
          }
          if ((e2 & ~CO_CACHE_POOL_MASK) == key.z) { /* taken this instant, perhaps for the same position: look again */
            e0 = co_lane_load_coherent_u32(H + 0);
            e1 = co_lane_load_coherent_u32(H + 1);
          }
        }
        if ((e2 & ~CO_CACHE_POOL_MASK) == key.z && e0 == x0 && e1 == x1) {
          if ((e2 & CO_CACHE_POOL_MASK) == C.pool_bits) {
            sl = (int)s;
            nd = 0;
          } else {
            /* another pool's: usable iff its claiming iteration's network launch is known to be over (a stamp that
             * is not visible yet reads 0: not usable) */
            const uint32_t stamp = co_lane_load_coherent_u32(H + 3);
            const uint32_t dn = co_lane_load_coherent_u32(C.done + ((e2 & CO_CACHE_POOL_MASK) >> CO_CACHE_POOL_SHIFT));
            if (stamp != 0u && stamp <= dn) {
              sl = (int)s;
              nd = 0;
            }
          }
          break;
        }
      }
    }
    L(slot) = sl;
    L(need) = nd;
  }
  const uint64_t nm = WAVE_BALLOT(need);
  const uint32_t cnt = (uint32_t)co_popc64(nm);
  uint32_t base = 0u;
  if (cnt) base = co_atomic_add_u32(C.count + 4 * par, cnt);
  int32_t *psrc = P.pend_src + (size_t)g * P.searches_per_eval;
  FOR_LANES {
    if (lane < n) {
      int src = L(slot);
      if (L(need)) {
        const uint32_t m = base + (uint32_t)co_popc64(nm & ((1ull << lane) - 1ull));
        if (src < 0) src = (int)(C.mask + 1u + C.scratch_base + m); /* this pool's scratch element m of this iteration */
        C.in_idx[m] = row0 + lane;
        C.out_idx[m] = src;
      }
      psrc[lane] = src;
    }
  }
}

/* ---- the step of one game's wavefront (co_k_mcts_step), in pieces.
 * (Round 4 also ran fused training as TWO kernels -- the receive + simulate half as a hot kernel of its own, 62 spilled
 * SGPRs instead of 260 and a third of the instructions, and the once-per-ply work (move choice, logs, re-root, hand-over,
 * slot recycling) as a second launch over a list of the games whose turn had ended.  Measured on the same box,
 * alternating libraries: 4096 games x 400 simulations, mlp12x100 f16x3, 168.0 against 160.5 ms per generation, rescnn4
 * 518.4 against 512.8 -- the second launch per iteration costs more than the spills and the instruction-cache misses
 * of the cold paths, which are not executed in the hot loop anyway.  Removed; commit 28aee16 has it.) */

/* first wave of a pool's launch: clear the counters of the NEXT iteration (nobody reads them before the next launch) */
CO_DEV void co_pool_housekeeping(const EngineParams &P, int g) {
  if (P.fused_pack && g == P.pool_lo) {
    FOR_LANES {
      if (lane == 0) {
        P.pack_counter[(P.iteration + 1) & 1] = 0ull;
        if (P.work_counter) {
          /* step budget that follows the games (EngineParams::step_budget_k16): this launch's first wavefront turns what
           * the steps of the launch before cost (CO_STEP_*: ticks of the real-time counter on the device, scans on the
           * emulation build; complete: a kernel boundary lies between) into the budget of the launch after -- the division
           * is this one wavefront's, every other one reads a finished word */
          const unsigned long long wc = P.work_counter[(P.iteration + 2) % 3];
          const uint32_t steps = (uint32_t)(wc >> 32), scans = (uint32_t)wc;
          /* the mean, smoothed over ~8 launches (in 1/256 units; CO_WC_MEAN + parity carries it from launch to launch): the
           * games of a generation that started together also choose their moves together, and a launch of first steps on
           * fresh roots says nothing about the launch two later */
          unsigned long long sm = P.work_counter[CO_WC_MEAN + ((P.iteration + 1) & 1)];
          if (steps > 0u) {
            const unsigned long long inst = ((unsigned long long)scans << 8) / steps;
            sm = sm ? (7ull * sm + inst) >> 3 : inst;
          }
          uint32_t b = 0u; /* (no limit) */
          if (sm > 0ull && P.step_budget_k16 > 0) {
            b = (uint32_t)((sm * (uint32_t)P.step_budget_k16) >> 12);
            if (b < CO_STEP_BUDGET_MIN) b = CO_STEP_BUDGET_MIN;
          }
          P.work_counter[CO_WC_MEAN + (P.iteration & 1)] = sm;
          P.work_counter[CO_WC_BUDGET + (P.iteration & 1)] = (unsigned long long)b;
          P.work_counter[(P.iteration + 1) % 3] = 0ull;
        }
        if (P.cache.hdr) {
          /* the other parity's counter of rows to evaluate was the previous iteration's (its network launch is
           * over): book it, clear it for the next iteration */
          const int op = (P.iteration & 1) ^ 1;
          P.cache.totals[0] += P.cache.count[4 * op];
          P.cache.count[4 * op] = 0u;
          /* this launch stands behind the pool's network launch of iteration - 1 in its stream: entries the pool
           * claimed in iterations < iteration (stamps <= iteration) hold their outputs -- tell the other pools */
          co_lane_store_coherent_u32(P.cache.done + (P.cache.pool_bits >> CO_CACHE_POOL_SHIFT), (uint32_t)P.iteration);
        }
      }
    }
  }
}

/* The scans a game's step may make in this launch before it stops selecting (EngineParams::step_budget), fused training
 * only.  Automatic: what the first wavefront of the launch before this one made of the launch before that
 * (co_pool_housekeeping).  No limit = a number no step reaches. */
#define CO_STEP_BUDGET_NONE (1 << 28)
CO_DEV int co_step_budget(const EngineParams &P) {
  if (!(P.fused_pack && !P.testing && !P.analyse && !P.pcfg)) return CO_STEP_BUDGET_NONE;
  int b = P.step_budget * CO_STEP_UNITS_PER_CONFIG_UNIT;
  if (P.step_budget == 0 && P.step_budget_k16 > 0 && P.work_counter) b = (int)(uint32_t)P.work_counter[CO_WC_BUDGET + ((P.iteration + 1) & 1)];
  return b > 0 ? b : CO_STEP_BUDGET_NONE;
}

/* the wavefront's view of game slot g.  (Fields a kernel never touches cost nothing: they are never loaded.) */
CO_DEV void co_wave_init(const EngineParams &P, int g, const GameCtl &gc, const TreeCtl &tc0, const TreeCtl &tc1, CoWave &w) {
  w.g = g;
  w.gc = gc;
  {
    const size_t stride = (size_t)P.cap_units + CO_ARENA_PAD;
    const int tm = 2 * g + gc.to_play, to = 2 * g + 1 - gc.to_play;
    w.me.A = P.arena + (size_t)tm * stride;
    w.me.tc = gc.to_play ? tc1 : tc0;
    w.me.cap = P.cap_units;
    w.opp.A = P.arena + (size_t)to * stride;
    w.opp.tc = gc.to_play ? tc0 : tc1;
    w.opp.cap = P.cap_units;
  }
  w.mt = P.rng + (size_t)g * CO_MT_N;
  w.pend_leaf = P.pend_leaf + (size_t)g * P.searches_per_eval;
  w.pend_depth = P.pend_depth + (size_t)g * P.searches_per_eval;
  w.pend_path = P.pend_path + (size_t)g * P.searches_per_eval * CO_PATH_MAX;
  w.pend_n = P.pend_n + (size_t)g * P.searches_per_eval * 4;
  w.pend_key = P.cache.hdr ? P.pend_key + (size_t)g * P.searches_per_eval : (uint4 *)0;
  w.noise_raw = P.noise_raw + (size_t)g * P.searches_per_eval * CO_NUM_MOVES;
  w.noise_words = (gc.held & 1) ? gc.noise_held : 0; /* a step consumes every pending leaf before it queues new ones -- unless they were held back */
  w.work = CO_STEP_START(co_step_budget(P));
  w.req = P.req + (size_t)g * P.searches_per_eval * CO_STATE_STRIDE;
  /* samples and traces belong to the GAME, not to the slot */
  w.samples = P.samples ? P.samples + (size_t)gc.gid * CO_MAX_PLIES * CO_SAMPLE_FLOATS : (float *)0;
  w.trace = P.trace ? P.trace + (size_t)gc.gid * CO_TRACE_CAP : (int32_t *)0;
  w.log = (int32_t *)0;
  if (P.log) {
    const int rec = P.log_index ? P.log_index[gc.gid] : gc.gid < P.num_logged ? gc.gid : -1;
    if (rec >= 0) w.log = P.log + (size_t)rec * CO_LOG_CAP;
  }
  w.prof = P.prof ? P.prof + (size_t)g * CO_NPROF : (unsigned long long *)0;
  w.max_searches = P.max_searches;
  w.spe = P.searches_per_eval;
  w.testing = P.testing;
  w.trace_on = P.trace_on && P.trace;
  w.defer_handover = P.defer_handover;
  w.analyse = P.analyse;
  w.force_choose = P.analyse && P.force_choose;
  w.c_puct = P.c_puct;
  w.epsilon = P.epsilon;
  w.pc = P.pcfg ? P.pcfg + 2 * g : (const PlayerCfg *)0;
  co_use_player(w, gc.to_play);
  w.cval = P.cache.hdr ? P.cache.val : (const float *)0;
  w.csrc = P.cache.hdr ? P.pend_src + (size_t)g * P.searches_per_eval : (const int32_t *)0;
}

/* The slot's game is over: file it, and in a resident-slot pool hand the slot the next unstarted game.  Returns 1 when
 * a fresh game sits in the slot (its first step -- root + request for its evaluation -- is the caller's next pass). */
CO_COLD int co_slot_next_game(const EngineParams &P, CoWave &w) {
  w.gc.done = 1;
  if (!P.results) return 0;
  {
    GameCtl *dst = P.results + w.gc.gid;
    FOR_LANES {
      if (lane == 0) *dst = w.gc;
    }
    WAVE_SYNC();
  }
  const unsigned long long next = co_atomic_add_u64(P.next_game, 1ull);
  if (next >= (unsigned long long)P.total_local) return 0; /* none left: the slot is done */
  /* Trainer::initialize for game `next` (trainer.cpp:243-255) in this slot: a fresh SelfPlayer -- own
   * generator seeded from the Trainer stream by game index, two empty trees (the arena of the finished game
   * is given back whole: the bump pointers return to zero), colour parity by global index */
  const int gid = (int)next;
  if (w.gc.to_play != 0) { /* w.me is the tree of the player to move: player 0 starts */
    CoTree tmp = w.me;
    w.me = w.opp;
    w.opp = tmp;
  }
  GameCtl fresh;
  fresh.to_play = 0; fresh.done = 0; fresh.result = 0; fresh.mate_turn = 0; fresh.n_samples = 0;
  fresh.parity = (P.game_base + gid) % 2; fresh.error = 0; fresh.n_pending = 0; fresh.rng_idx = CO_MT_N;
  fresh.plies = 0; fresh.searches = 0u; fresh.evals = 0u; fresh.nodes = 0u; fresh.trace_len = 0;
  fresh.row_off = 0; fresh.resume = 0; fresh.pos_lo = 0u; fresh.pos_hi = 0u; fresh.pos_meta = CO_META_START;
  fresh.gid = gid;
  fresh.sb_cap = 0;
  fresh.held = 0;
  fresh.noise_held = 0;
  w.gc = fresh;
  w.me.tc.root = CO_NONE; w.me.tc.searches_done = 0; w.me.tc.units_used = 0u;   /* peak_units: high-water of the slot */
  w.opp.tc.root = CO_NONE; w.opp.tc.searches_done = 0; w.opp.tc.units_used = 0u;
  w.noise_words = 0;
  w.samples = P.samples ? P.samples + (size_t)gid * CO_MAX_PLIES * CO_SAMPLE_FLOATS : (float *)0;
  w.trace = P.trace ? P.trace + (size_t)gid * CO_TRACE_CAP : (int32_t *)0;
  w.log = (int32_t *)0; /* logged games are the first ones: they start in their own slots */
  co_mt_seed(w.mt, P.seeds[gid]);
  w.mt_staged = 0;
  return 1;
}

/* End of a game's step.  Trainer::writeRequests fused into the step: reserve rows of the compact batch (any order: a
 * row's evaluation does not depend on its position; the atomic's round trip runs under the noise capture), reserve the
 * generator outputs owed to the queued leaves, resolve the rows against the evaluation cache, copy them to the batch. */
CO_DEV void co_step_tail(const EngineParams &P, CoWave &w, int g, int done) {
  const int packs = P.fused_pack && !w.gc.done && !w.gc.error;
  unsigned long long old = 0ull;
  CO_PH_MEM(25);
  if (packs && P.work_counter) {
    const int scans = CO_STEP_DONE(w, co_step_budget(P)); /* (the budget once more, from where it lies: not kept across the step) */
    /* scans of the launch over the steps BEGUN in it: the pieces of a step that was cut add their scans, not a step -- counted
     * as steps of their own they pull the mean down, the budget follows, more steps are cut: it collapses to its floor */
    if (scans > 0) co_atomic_add_u64_noret(P.work_counter + P.iteration % 3, ((w.gc.held & 2) ? 0ull : 1ull << 32) | (unsigned long long)scans);
  }
  w.gc.held &= 1;
  if (packs && w.gc.held) {
    /* the step stopped at its budget: the game is running and submits nothing; its leaves (request rows, records and keys
     * are in place) wait for the rest of their batch, the generator outputs owed to them are reserved with the others' */
    /* (bits 56..: games that hold their leaves back -- the host tells an iteration without rows from "no game has a request") */
    co_atomic_add_u64(P.pack_counter + (P.iteration & 1), (1ull << 32) | (1ull << 56));
    if (P.work_counter) co_atomic_add_u64_noret(P.work_counter + CO_WC_CUTS, 1ull); /* (ca_stats.steps_cut) */
    w.gc.noise_held = w.noise_words;
    return;
  }
  if (packs) old = co_atomic_add_u64(P.pack_counter + (P.iteration & 1), (1ull << 32) | (unsigned long long)w.gc.n_pending);
  CO_PH_MEM(26);
  if (!done && !w.gc.error && w.gc.n_pending > 0) co_capture_noise(w);
  CO_PH_MEM(27);
  if (packs) {
    const int n = w.gc.n_pending;
    const int base = P.pool_row_base + (int)(unsigned)(old & 0xFFFFFFFFull);
    w.gc.row_off = base;
    /* the rows stay in the game's request area; the network kernels read them through an index array (nn.h CoNetIO):
     * with the evaluation cache the rows it has to evaluate (co_cache_resolve), else every row of the batch.  (Until
     * round 5 the rows were copied into a compact batch here: 5 KB read back and written again at the end of every
     * step, 35 k cycles of the wave's 190 k with two pools in flight.) */
    const int first = g * P.searches_per_eval;
    if (w.pend_key) {
      co_cache_resolve(P, w, g, n, first);
    } else {
      int32_t *ri = P.row_idx + base;
      FOR_LANES {
        if (lane < n) ri[lane] = first + lane;
      }
    }
  }
}

CO_DEV void co_wave_store(const EngineParams &P, const CoWave &w, int g) {
  FOR_LANES {
    if (lane == 0) {
      P.games[g] = w.gc;
      P.trees[2 * g + w.gc.to_play] = w.me.tc;
      P.trees[2 * g + 1 - w.gc.to_play] = w.opp.tc;
    }
  }
}

/* Trainer::doIteration for game g (trainer.cpp:164-236): the body of the
 * `omp parallel for`, one wavefront per game. */
CO_DEV void co_mcts_step_wave(const EngineParams &P, int g) {
  co_pool_housekeeping(P, g);
  GameCtl gc = P.games[g];
  /* both trees' control words, fetched with the game's (not behind it) */
  const TreeCtl tc0 = P.trees[2 * g], tc1 = P.trees[2 * g + 1];
  const int gate = co_step_gate(P, g, gc);
  if (gate != 1) {
    if (gate == 2 && P.fused_pack) co_atomic_add_u64(P.pack_counter + (P.iteration & 1), 1ull << 32); /* still running */
    return;
  }
  CoWave w;
  WAVE_SHARED(uint32_t, mt_stage, CO_MT_STAGE);
  w.mt_stage = mt_stage;
  WAVE_SHARED(uint32_t, pre_rec, CO_PRE_WORDS);
  w.pre = pre_rec;
  WG_SHARED(uint32_t, lb, CO_LB_WORDS);
  co_line_breakers_to_lds(lb);
  w.lb = lb;
  WG_SHARED(uint32_t, gam, CO_NUM_GAMMA);
  { /* (the same way: every wavefront writes the same words before it reads any) */
    FOR_LANES {
#pragma unroll
      for (int i = 0; i < CO_NUM_GAMMA / CO_WAVE; ++i) gam[i * CO_WAVE + lane] = CO_GAMMA_BITS[i * CO_WAVE + lane];
    }
    WAVE_SYNC();
  }
  w.gamma = gam;
#if defined(CO_PROF) && !defined(CO_EMU)
  const unsigned long long t_wave0 = CO_CLK(), t_real0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < CO_NPROF; ++i) w.pacc[i] = 0ull;
  w.tph = t_wave0;
#endif
  co_wave_init(P, g, gc, tc0, tc1, w);
  /* (in front of the generator's state: the counter of outstanding loads retires in order, and the step needs these first) */
  w.pre_n = 0;
  if (w.gc.n_pending > 0 && !(w.gc.held & 1)) co_pending_records(w, 0, w.gc.n_pending < CO_PRE ? w.gc.n_pending : CO_PRE);
  co_mt_stage_load(w.mt, w.mt_stage);
  w.mt_staged = 1;
  CO_PROF_ADD(w, 4, 1ull);
  CO_PH_MEM(22);
  int off = co_step_row(P, g, gc);
  const float *step_eval = P.nn_eval + off, *step_probs = P.nn_probs + (size_t)off * CO_NUM_MOVES;
  int done;
  for (;;) { /* one pass, unless the slot's game ends and the pool hands it the next one */
    done = co_game_step(w, step_eval, step_probs);
    if (!done) break;
    if (!co_slot_next_game(P, w)) break;
    step_eval = step_probs = (const float *)0; /* the first step of a game creates the root and asks for its evaluation */
  }
  CO_PH(20);
  co_step_tail(P, w, g, done);
  CO_PH_MEM(21);
#if defined(CO_PROF) && !defined(CO_EMU)
  w.pacc[7] = CO_CLK() - t_wave0;
  if (w.prof && (threadIdx.x & 63) == 0) {
    for (int i = 0; i < CO_NPROF; ++i) w.prof[i] += w.pacc[i];
    const unsigned long long dt = w.pacc[7];
    unsigned long long *glob = P.prof + (size_t)P.num_games * CO_NPROF;
    if (g == 1) {
      glob[0] += dt;
      glob[1] += __builtin_amdgcn_s_memrealtime() - t_real0;
    }
    atomicMax(glob + 2, dt);
#ifndef CO_PROF_SLOW_FROM
#define CO_PROF_SLOW_FROM 0
#endif
#ifndef CO_PROF_SLOW_CYCLES
#define CO_PROF_SLOW_CYCLES 280000ull
#endif
    if (dt > CO_PROF_SLOW_CYCLES && P.iteration >= CO_PROF_SLOW_FROM) { /* the waves a launch waits for: their phases apart (slot 4 counts them) */
      for (int i = 0; i < CO_NPROF; ++i) atomicAdd(glob + 24 + i, w.pacc[i]);
    }
    int bucket = (int)(dt / 50000ull);
    if (bucket > 15) bucket = 15;
    atomicAdd(glob + 4 + bucket, 1ull);
  }
#endif
  co_wave_store(P, w, g);
}
