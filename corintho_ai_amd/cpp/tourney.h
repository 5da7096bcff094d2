// tourney.h -- `class Tourney` with the reference's public interface
// (corintho_ai/cpp/include/tourney.h:13-46), implemented on the MI355X engine through the C ABI
// of include/corintho_hip.h.
//
// Drop-in use from the reference's Cython boundary (corintho_ai/rating/tourney.pyx:15-31):
//     cdef extern from "<repo>/corintho_ai_amd/cpp/tourney.cpp":
//         cdef cppclass Tourney: ...            # declarations unchanged
// and link the extension with -lcorintho_hip.
#ifndef CORINTHO_AMD_TOURNEY_H
#define CORINTHO_AMD_TOURNEY_H

#include <cstdint>
#include <string>

struct ca_tourney;

class Tourney {
 public:
  Tourney(int32_t num_threads, std::string log_folder);
  Tourney(const Tourney &) = delete;
  Tourney &operator=(const Tourney &) = delete;
  ~Tourney();

  bool all_done() const;
  int32_t num_requests(int32_t id) const;
  void writeScores(const std::string &filename) const;
  void writeRequests(float *game_states, int32_t id);
  // The reference reads eval/probs through its own offset table (tourney.cpp:55-62); the arrays
  // must have the capacity the Python driver allocates (tourney.pyx:118-119), which setRows tells
  // the engine once (default: every pending slot of every match).
  void doIteration(float eval[], float probs[], int32_t id);
  void addPlayer(int32_t player_id, int32_t model_id, int32_t max_searches = 1600, int32_t searches_per_eval = 16,
                 float c_puct = 1.0, float epsilon = 0.25, bool random = false);
  void addMatch(int32_t player1, int32_t player2, bool logging = false);

  // ---- addition; not part of the reference interface ----
  void setRows(int32_t rows) { rows_ = rows; }

 private:
  ca_tourney *impl_{nullptr};
  int32_t rows_{0};      // rows of the caller's eval/probs arrays
  int32_t max_rows_{0};  // sum of searches_per_eval over both sides of every match
  int32_t spe_[1024] = {0};
};

#endif
