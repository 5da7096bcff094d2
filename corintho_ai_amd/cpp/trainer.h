// trainer.h -- `class Trainer` with the reference's exact public interface
// (corintho_ai/cpp/include/trainer.h:17-81), implemented on the MI355X engine
// through the C ABI of include/corintho_hip.h.
//
// Drop-in use from the reference's Cython boundary (corintho_ai/python/main.pyx:17-38):
//     cdef extern from "<repo>/corintho_ai_amd/cpp/trainer.cpp":
//         cdef cppclass Trainer: ...            # declarations unchanged
// and link the extension with -lcorintho_hip.  Failures are
// thrown as std::runtime_error, which Cython's `except +` turns into Python
// exceptions, like the reference's C++ exceptions.
#ifndef CORINTHO_AMD_TRAINER_H
#define CORINTHO_AMD_TRAINER_H

#include <cstdint>
#include <string>

struct ca_trainer;

class Trainer {
 public:
  Trainer() = default;
  Trainer(int32_t num_games, const std::string &log_folder, int32_t seed, int32_t max_searches = 1600,
          int32_t searches_per_eval = 16, float c_puct = 1.0, float epsilon = 0.25, int32_t num_logged = 10,
          int32_t num_threads = 1, bool testing = false);
  Trainer(const Trainer &) = delete;
  Trainer &operator=(const Trainer &) = delete;
  ~Trainer();

  int32_t num_requests(int32_t to_play = -1) const;
  int32_t num_samples() const;
  float score() const;
  float avg_mate_length() const;

  void writeRequests(float *game_states, int32_t to_play = -1) const;
  void writeSamples(float *game_states, float *eval_samples, float *prob_samples) const;
  void writeScores(const std::string &file) const;

  bool doIteration(float eval[], float probs[], int32_t to_play = -1);

  // ---- additions (fused mode); not part of the reference interface ----
  void setNet(int32_t kind, const float *weights, size_t n_floats, int32_t slot = 0);
  bool run(int64_t max_iterations = 0);
  // page-lock a caller array that is passed to every doIteration / writeRequests (main.pyx:132-134) so
  // that its copies are direct DMA; it must outlive this object
  bool pinBuffer(void *p, size_t bytes);

 private:
  ca_trainer *impl_{nullptr};
};

#endif
