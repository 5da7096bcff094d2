// dockermc.h -- `class DockerMC` with the reference's public interface
// (corintho_ai/cpp/include/dockermc.h:13-51), implemented on the MI355X engine: an analysis
// trainer with one position (include/corintho_hip.h, ca_config.analyse).
//
// Drop-in use from the reference's Cython boundary (corintho_ai/docker/choose_move.pyx:21-42):
//     cdef extern from "<repo>/corintho_ai_amd/cpp/dockermc.cpp":
//         cdef cppclass DockerMC: ...           # declarations unchanged
// The engine chooses the move itself when the search of a position is over (doIteration returns
// true) and chooseMove() reports that move; when the caller's loop stops earlier (the time limit of
// choose_move.pyx:110-117), chooseMove() makes the engine choose on the tree as it stands
// (ca_trainer_finish), as TrainMC::chooseMove does in the reference.  One position is one wavefront:
// for throughput search many positions at once (corintho_ai_amd/analyse.py, ca_trainer_set_positions).
#ifndef CORINTHO_AMD_DOCKERMC_H
#define CORINTHO_AMD_DOCKERMC_H

#include <cstdint>

struct ca_trainer;

class DockerMC {
 public:
  DockerMC(int32_t seed, int32_t max_searches, int32_t searches_per_eval, float c_puct, float epsilon,
           int32_t board[64], int32_t to_play, int32_t pieces[6]);
  DockerMC(const DockerMC &) = delete;
  DockerMC &operator=(const DockerMC &) = delete;
  ~DockerMC();

  float eval() const;
  int32_t num_requests() const;
  int32_t num_nodes() const;
  bool done() const;
  bool drawn() const;

  void writeRequests(float *game_states) const;
  void getLegalMoves(int32_t legal_moves[96]) const;

  int32_t chooseMove();
  bool doIteration(float eval[] = nullptr, float probs[] = nullptr);

 private:
  void fetch() const;
  ca_trainer *impl_{nullptr};
  bool finished_{false};
  mutable bool have_{false};
  mutable int32_t res_[8]{};
  // the position as given: done() / drawn() / getLegalMoves() before the search (choose_move.pyx:75-86)
  uint32_t mask0_[3]{};
  int32_t lines0_{0};
};

#endif
