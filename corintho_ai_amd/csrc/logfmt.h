// logfmt.h -- host side of the per-game text logs (Trainer's num_logged, trainer.cpp:243-250).
// The search kernel records the numbers of every move choice (mcts.h co_log_ply, co_game_step); this file prints
// them in the reference's layout:
//   SelfPlayer::writePreMoveLogs / writeMoves / writeEval / writeMoveChoice   selfplayer.cpp:124-204
//   (Match's copies of the four, match.cpp:78-159, and Match::endGame :163-190, print the same text)
//   Node::writeMainLine                                                      node.cpp:197-240
//   Game operator<<                                                          game.cpp:98-139
//   Move operator<<                                                          move.cpp:56-78
//   strResult                                                                util.cpp:5-25
// A C++ ostream prints a float as printf's %g would until a manipulator changes the stream; writeEval's
// std::fixed << std::setprecision(6) does, for the rest of the file: `fixed` below.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "engine_defs.h"
#include "rules.h"

struct CoLogWriter {
  FILE *f;
  bool fixed = false;
  explicit CoLogWriter(FILE *out) : f(out) {}

  static float bits(int32_t u) {
    float v;
    memcpy(&v, &u, 4);
    return v;
  }
  void put_float(float v) { fprintf(f, fixed ? "%.6f" : "%g", (double)v); }
  static const char *result_name(int r) {
    switch (r) {
      case CO_RESULT_LOSS: return "L";
      case CO_RESULT_DRAW: return "D";
      case CO_RESULT_WIN: return "W";
      case CO_DEDUCED_LOSS: return "DL";
      case CO_DEDUCED_DRAW: return "DD";
      case CO_DEDUCED_WIN: return "DW";
      default: return "N";
    }
  }
  /* move ids: SURVEY 8a row a2 (move.cpp:11-42) */
  void put_move(int id) {
    if (id >= 48) {
      const int piece = (id - 48) / 16, r = (id % 16) / 4, c = id % 4;
      fprintf(f, "%c%c%d", piece == 0 ? 'B' : piece == 1 ? 'C' : 'A', 'a' + c, 4 - r);
      return;
    }
    int r, c;
    char d;
    if (id < 12) { r = id / 3; c = id % 3; d = 'R'; }
    else if (id < 24) { r = (id - 12) / 4; c = id % 4; d = 'D'; }
    else if (id < 36) { r = (id - 24) / 3; c = id % 3 + 1; d = 'L'; }
    else { r = (id - 36) / 4 + 1; c = id % 4; d = 'U'; }
    fprintf(f, "%c%d%c", 'a' + c, 4 - r, d);
  }
  /* writeEval: a known result by name, else the mean evaluation -- which switches the stream to fixed notation */
  void put_eval(int result, float evaluation, int visits) {
    if (result != CO_RESULT_NONE) {
      fputs(result_name(result), f);
      return;
    }
    fixed = true;
    put_float(evaluation / (float)visits);
  }
  void put_position(uint64_t board, uint32_t meta) {
    for (int row = 0; row < 4; ++row) {
      for (int col = 0; col < 4; ++col) {
        const unsigned cell = (unsigned)(board >> (row * 16 + col * 4)) & 15u;
        fputc(cell & 1u ? 'B' : ' ', f);
        fputc(cell & 2u ? 'C' : ' ', f);
        fputc(cell & 4u ? 'A' : ' ', f);
        fputc(cell & 8u ? '#' : ' ', f);
        if (col < 3) fputc('|', f);
      }
      if (row < 3) fputs("\n-------------------\n", f);
    }
    fputc('\n', f);
    for (int player = 0; player < 2; ++player)
      fprintf(f, "Player %d: B: %d C: %d A: %d\n", player + 1, (int)CO_META_PIECE(meta, player * 3 + 0),
              (int)CO_META_PIECE(meta, player * 3 + 1), (int)CO_META_PIECE(meta, player * 3 + 2));
    fprintf(f, "Player %d to play", (int)CO_META_TO_PLAY(meta) + 1);
  }

  /* writeMoveChoice: {move, board low, board high, meta of the new position} */
  void put_choice(const int32_t *r) {
    fputs("CHOSE MOVE ", f);
    put_move(r[0]);
    fputs("\nNEW POSITION:\n", f);
    put_position((uint64_t)(uint32_t)r[1] | ((uint64_t)(uint32_t)r[2] << 32), (uint32_t)r[3]);
    fputs("\n\n", f);
  }

  struct Child {
    int move, visits, result;
    float evaluation, mean, probability;
  };

  /* one game's record (EngineParams::log without its length word); false = the record is malformed */
  bool write_game(const int32_t *rec, int len, int game_result) {
    int at = 0;
    auto need = [&](int n) { return at + n <= len; };
    while (at < len) {
      if (need(5) && rec[at] == 2) { /* the move of a random player: no pre-move block (match.cpp:213-221) */
        put_choice(rec + at + 1);
        at += 5;
        continue;
      }
      if (!need(7) || rec[at] != 1) return false;
      const int to_play = rec[at + 1], depth = rec[at + 2], visits = rec[at + 3], result = rec[at + 4];
      const float evaluation = bits(rec[at + 5]);
      const int nc = rec[at + 6];
      at += 7;
      if (nc < 0 || nc > CO_NUM_MOVES || !need(5 * nc)) return false;
      std::vector<Child> ch((size_t)nc);
      for (auto &c : ch) {
        c.move = rec[at];
        c.visits = rec[at + 1];
        c.evaluation = bits(rec[at + 2]);
        c.result = rec[at + 3];
        c.probability = bits(rec[at + 4]);
        c.mean = c.evaluation / (float)c.visits;
        at += 5;
      }
      fprintf(f, "TURN %d\nPLAYER %d TO PLAY\nVISITS: %d\nPOSITION EVALUATION: ", depth, to_play + 1, visits);
      put_eval(result, evaluation, visits);
      fputs("\nLEGAL MOVES:\n", f);
      /* main line */
      for (;;) {
        if (!need(1)) return false;
        if (rec[at] == -1) {
          ++at;
          break;
        }
        if (!need(6)) return false;
        fprintf(f, "%d. ", rec[at]);
        put_move(rec[at + 1]);
        fprintf(f, " V: %d E: ", rec[at + 2]);
        if (rec[at + 3] != CO_RESULT_NONE) fputs(result_name(rec[at + 3]), f);
        else put_float(bits(rec[at + 4]) / (float)rec[at + 2]); /* (the stream's notation as it is: no manipulator here) */
        fputs(" p: ", f);
        put_float(bits(rec[at + 5]));
        fputc('\t', f);
        at += 6;
      }
      fputc('\n', f);
      /* the other moves: most visits, then mean evaluation, then prior, then id (selfplayer.cpp:164-173) */
      std::sort(ch.begin(), ch.end(), [](const Child &a, const Child &b) {
        if (a.visits != b.visits) return a.visits > b.visits;
        if (a.mean != b.mean) return a.mean > b.mean;
        if (a.probability != b.probability) return a.probability > b.probability;
        return a.move < b.move;
      });
      for (size_t i = 1; i < ch.size(); ++i) { /* the first one is taken to be in the main line */
        put_move(ch[i].move);
        fprintf(f, " V: %d E: ", ch[i].visits);
        put_eval(ch[i].result, ch[i].evaluation, ch[i].visits);
        fputs(" P: ", f);
        put_float(ch[i].probability);
        fputc('\t', f);
      }
      fputc('\n', f);
      if (!need(4)) return false;
      put_choice(rec + at);
      at += 4;
    }
    /* endGame, selfplayer.cpp:206-232 */
    if (game_result == CO_RESULT_DRAW) fputs("GAME IS DRAWN.\n", f);
    else fprintf(f, "PLAYER %d WON!\n", game_result == CO_RESULT_WIN ? 1 : 2);
    return true;
  }
};
