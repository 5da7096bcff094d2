// dockermc.cpp -- see dockermc.h.  Each member forwards to the C entry points that replace the
// reference member of the same name (dockermc.cpp:11-53).
#include "dockermc.h"

#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/corintho_hip.h"

namespace {
void dcheck(int rc) {
  if (rc != CA_OK) throw std::runtime_error(std::string("corintho_hip: ") + ca_last_error());
}
}  // namespace

DockerMC::DockerMC(int32_t seed, int32_t max_searches, int32_t searches_per_eval, float c_puct, float epsilon,
                   int32_t board[64], int32_t to_play, int32_t pieces[6]) {
  ca_config cfg{};
  cfg.num_games = 1;
  cfg.max_searches = max_searches;
  cfg.searches_per_eval = searches_per_eval;
  cfg.c_puct = c_puct;
  cfg.epsilon = epsilon;
  cfg.analyse = 1;
  // a tree gains one node per simulation; ~34 units per node on average, 50 with the widest positions
  const uint64_t units = (uint64_t)max_searches * 64 + 4096;
  cfg.arena_units = units > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (uint32_t)units;
  dcheck(ca_trainer_create(&cfg, &impl_));
  uint64_t b = 0;
  for (int i = 0; i < 64; ++i)
    if (board[i]) b |= 1ull << i;
  uint32_t meta = (uint32_t)to_play << 18;
  for (int i = 0; i < 6; ++i) meta |= (uint32_t)pieces[i] << (3 * i);
  dcheck(ca_rules_legal_moves(0, &b, &meta, 1, mask0_, &lines0_));
  dcheck(ca_trainer_set_positions(impl_, board, &to_play, pieces, &seed));
}

DockerMC::~DockerMC() { ca_trainer_destroy(impl_); }

void DockerMC::fetch() const {
  if (!have_) {
    dcheck(ca_trainer_analysis(impl_, res_));
    have_ = true;
  }
}

bool DockerMC::doIteration(float eval[], float probs[]) {
  int32_t done = 0;
  dcheck(ca_trainer_do_iteration(impl_, eval, probs, -1, &done));
  finished_ = done != 0;
  return finished_;
}

int32_t DockerMC::num_requests() const {
  int32_t n = 0;
  dcheck(ca_trainer_num_requests(impl_, -1, &n));
  return n;
}

void DockerMC::writeRequests(float *game_states) const { dcheck(ca_trainer_write_requests(impl_, game_states, -1)); }

int32_t DockerMC::chooseMove() {
  if (!finished_) {
    // the reference's loop also ends on its time limit (choose_move.pyx:110-117) and then chooses on the tree
    // as it stands; so does the engine (ca_trainer_finish)
    dcheck(ca_trainer_finish(impl_));
    finished_ = true;
    have_ = false;
  }
  fetch();
  return res_[0];
}

// before the search: the position as given; after it: the position after the chosen move
bool DockerMC::done() const {
  if (!finished_) return (mask0_[0] | mask0_[1] | mask0_[2]) == 0u;
  fetch();
  return res_[1] != 0;
}

bool DockerMC::drawn() const {
  if (!finished_) return (mask0_[0] | mask0_[1] | mask0_[2]) == 0u && !lines0_;
  fetch();
  return res_[2] != 0;
}

int32_t DockerMC::num_nodes() const {
  if (!finished_) return 1;
  fetch();
  return res_[3];
}

float DockerMC::eval() const {
  if (!finished_) return 0.0f;
  fetch();
  float f;
  std::memcpy(&f, &res_[4], 4);
  return f;
}

void DockerMC::getLegalMoves(int32_t legal_moves[96]) const {
  uint32_t m[3] = {mask0_[0], mask0_[1], mask0_[2]};
  if (finished_) {
    fetch();
    for (int i = 0; i < 3; ++i) m[i] = (uint32_t)res_[5 + i];
  }
  for (int i = 0; i < 96; ++i) legal_moves[i] = (m[i >> 5] >> (i & 31)) & 1u;
}
