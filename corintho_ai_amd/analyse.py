"""Batched single-position search: N x the reference's `DockerMC` (corintho_ai/cpp/include/dockermc.h:13-51,
the web app's move chooser docker/choose_move.pyx) on the MI355X engine, one wavefront per position.

    a = Analyser(boards, to_play, pieces, seeds, max_searches=400, searches_per_eval=16)
    # the reference protocol, all positions at once (DockerMC::doIteration / num_requests / writeRequests):
    while not a.doIteration(evals, probs):
        n = a.num_requests(); a.writeRequests(game_states); evals[:n], probs[:n] = model(game_states[:n])
    # or with the network on the device:   a.set_net(kind, weights); a.run()
    a.results()   # per position: what choose_move.pyx:206-221 returns

`boards[i]` is the constructor's int32[64] (bit = cell*4 + {base, column, capital, frozen}, game.cpp:14-26),
`pieces[i]` its int32[6], `seeds[i]` the DockerMC seed.  The reference searches until a time limit or
max_searches; a batch has one budget, max_searches (choose_move.pyx:180-181 caps it at 32 760).
"""
import numpy as np

from . import trainer as T


class Analyser:
    def __init__(self, boards, to_play, pieces, seeds, max_searches=1600, searches_per_eval=16, c_puct=1.0, epsilon=0.25,
                 *, device=0, arena_units=0, _cdll=None):
        b = np.ascontiguousarray(boards, dtype=np.int32).reshape(-1, 64)
        self.n = b.shape[0]
        self.searches_per_eval = searches_per_eval
        self._t = T.Trainer(self.n, "", 0, max_searches, searches_per_eval, c_puct, epsilon, 0, 1, True, device=device,
                            stagger=False, arena_units=arena_units, analyse=True, _cdll=_cdll)
        self._t.set_positions(b, to_play, pieces, seeds)

    # DockerMC surface, over all positions
    def doIteration(self, evaluations, probabilities):
        return self._t.doIteration(evaluations, probabilities, -1)

    def num_requests(self):
        return self._t.num_requests(-1)

    def writeRequests(self, game_states):
        self._t.writeRequests(game_states, -1)

    # fused
    def set_net(self, kind, weights):
        self._t.set_net(kind, weights)

    def run(self, max_iterations=0):
        return self._t.run(max_iterations)

    def finish(self):
        """DockerMC::chooseMove for the positions still searching (the reference's loop may stop on a time limit)"""
        self._t.finish()

    def net_forward(self, states):
        return self._t.net_forward(states)

    def stats(self):
        return self._t.stats()

    def results(self):
        """per position, the dictionary of choose_move.pyx:214-221 (or its pre-result, :75-86)"""
        out = []
        for move, done, drawn, nodes, ev_bits, m0, m1, m2 in self._t.analysis():
            if move < 0:
                out.append({"pre-result": "draw" if drawn else "win"})
                continue
            mask = (int(np.uint32(m0)) | (int(np.uint32(m1)) << 32) | (int(np.uint32(m2)) << 64))
            evaluation = float(np.array([ev_bits], np.int32).view(np.float32)[0])
            out.append({"move": int(move), "is_done": bool(done), "has_won": bool(done and not drawn),
                        "legal_moves": [] if done else [i for i in range(96) if mask >> i & 1],
                        "nodes_searched": int(nodes), "evaluation": evaluation / nodes, "eval_sum": evaluation})
        return out

    def close(self):
        self._t.close()
