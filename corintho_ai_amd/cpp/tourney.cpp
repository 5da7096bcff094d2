// tourney.cpp -- see tourney.h.  Each member forwards to the C entry point that replaces the
// reference member of the same name (tourney.cpp:14-96).
#include "tourney.h"

#include <stdexcept>

#include "../../include/corintho_hip.h"

namespace {
void tcheck(int rc) {
  if (rc != CA_OK) throw std::runtime_error(std::string("corintho_hip: ") + ca_last_error());
}
}  // namespace

Tourney::Tourney(int32_t num_threads, std::string log_folder) {
  (void)num_threads;  // host threads of the reference's OpenMP loop; matches run one wavefront each
  tcheck(ca_tourney_create(0, 0, 0, &impl_));
  // tourney.h:46: matches added with logging = true write <log_folder>/match_<p1>_<p2>_<index>.txt
  const int rc = ca_tourney_set_log_folder(impl_, log_folder.c_str());
  if (rc != CA_OK) {
    ca_tourney_destroy(impl_);
    impl_ = nullptr;
    tcheck(rc);
  }
}

Tourney::~Tourney() { ca_tourney_destroy(impl_); }

bool Tourney::all_done() const {
  int32_t d = 0;
  tcheck(ca_tourney_all_done(impl_, &d));
  return d != 0;
}

int32_t Tourney::num_requests(int32_t id) const {
  int32_t n = 0;
  tcheck(ca_tourney_num_requests(impl_, id, &n));
  return n;
}

void Tourney::writeScores(const std::string &filename) const { tcheck(ca_tourney_write_scores(impl_, filename.c_str())); }

void Tourney::writeRequests(float *game_states, int32_t id) { tcheck(ca_tourney_write_requests(impl_, game_states, id)); }

void Tourney::doIteration(float eval[], float probs[], int32_t id) {
  tcheck(ca_tourney_do_iteration(impl_, eval, probs, rows_ > 0 ? rows_ : max_rows_, id));
}

void Tourney::addPlayer(int32_t player_id, int32_t model_id, int32_t max_searches, int32_t searches_per_eval,
                        float c_puct, float epsilon, bool random) {
  tcheck(ca_tourney_add_player(impl_, player_id, model_id, max_searches, searches_per_eval, c_puct, epsilon,
                               random ? 1 : 0));
  if (player_id >= 0 && player_id < 1024) spe_[player_id] = random ? 0 : searches_per_eval;
}

void Tourney::addMatch(int32_t player1, int32_t player2, bool logging) {
  tcheck(ca_tourney_add_match(impl_, player1, player2, logging ? 1 : 0));
  if (player1 >= 0 && player1 < 1024 && player2 >= 0 && player2 < 1024) max_rows_ += spe_[player1] + spe_[player2];
}
