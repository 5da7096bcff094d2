// trainer.cpp -- see trainer.h.  Each member forwards to the C entry point that
// replaces the reference member of the same name (trainer.cpp:18-236).
#include "trainer.h"

#include <cstdio>
#include <stdexcept>

#include "../../include/corintho_hip.h"

namespace {
void check(int rc) {
  if (rc != CA_OK) throw std::runtime_error(std::string("corintho_hip: ") + ca_last_error());
}
}  // namespace

Trainer::Trainer(int32_t num_games, const std::string &log_folder, int32_t seed, int32_t max_searches,
                 int32_t searches_per_eval, float c_puct, float epsilon, int32_t num_logged, int32_t num_threads,
                 bool testing) {
  ca_config cfg{};
  cfg.num_games = num_games;
  cfg.seed = seed;
  cfg.max_searches = max_searches;
  cfg.searches_per_eval = searches_per_eval;
  cfg.c_puct = c_puct;
  cfg.epsilon = epsilon;
  cfg.num_logged = 0;  // switched on below (ca_trainer_set_logging)
  cfg.num_threads = num_threads;
  cfg.testing = testing ? 1 : 0;
  check(ca_trainer_create(&cfg, &impl_));
  // trainer.cpp:243-250: the first num_logged games write log_folder/game_<i>.txt
  if (num_logged > 0) {
    const int rc = ca_trainer_set_logging(impl_, log_folder.c_str(), num_logged);
    if (rc != CA_OK) {
      ca_trainer_destroy(impl_);
      impl_ = nullptr;
      check(rc);
    }
  }
}

Trainer::~Trainer() { ca_trainer_destroy(impl_); }

int32_t Trainer::num_requests(int32_t to_play) const {
  int32_t n = 0;
  check(ca_trainer_num_requests(impl_, to_play, &n));
  return n;
}

int32_t Trainer::num_samples() const {
  int32_t n = 0;
  check(ca_trainer_num_samples(impl_, &n));
  return n;
}

float Trainer::score() const {
  float s = 0;
  check(ca_trainer_score(impl_, &s));
  return s;
}

float Trainer::avg_mate_length() const {
  float s = 0;
  check(ca_trainer_avg_mate_length(impl_, &s));
  return s;
}

void Trainer::writeRequests(float *game_states, int32_t to_play) const {
  check(ca_trainer_write_requests(impl_, game_states, to_play));
}

void Trainer::writeSamples(float *game_states, float *eval_samples, float *prob_samples) const {
  check(ca_trainer_write_samples(impl_, game_states, eval_samples, prob_samples));
}

void Trainer::writeScores(const std::string &file) const { check(ca_trainer_write_scores(impl_, file.c_str())); }

bool Trainer::doIteration(float eval[], float probs[], int32_t to_play) {
  int32_t done = 0;
  check(ca_trainer_do_iteration(impl_, eval, probs, to_play, &done));
  return done != 0;
}

void Trainer::setNet(int32_t kind, const float *weights, size_t n_floats, int32_t slot) {
  check(ca_trainer_set_net(impl_, slot, kind, weights, n_floats));
}

bool Trainer::run(int64_t max_iterations) {
  int32_t done = 0;
  check(ca_trainer_run(impl_, max_iterations, &done));
  return done != 0;
}

bool Trainer::pinBuffer(void *p, size_t bytes) {
  int32_t pinned = 0;
  check(ca_trainer_pin_host(impl_, p, bytes, &pinned));
  return pinned != 0;
}
