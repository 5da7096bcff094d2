"""Sample persistence in the reference's on-disk format (corintho_ai/python/main.pyx:189-219
get_samples): three `np.savez_compressed` files whose single array is stored as `arr_0`:
    <folder>/game_states.npz         [n*8, 70] float32
    <folder>/evaluation_labels.npz   [n*8]     float32
    <folder>/probability_labels.npz  [n*8, 96] float32
so the reference's Keras training step (main.pyx:221-283) consumes them unchanged."""
import os

import numpy as np

from .trainer import GAME_STATE_SIZE, NUM_MOVES, NUM_SYMMETRIES


def get_samples(trainer):
    """main.pyx:193-198: allocate the three arrays and let the trainer fill them"""
    n = trainer.num_samples()
    gs = np.zeros((n * NUM_SYMMETRIES, GAME_STATE_SIZE), dtype=np.float32)
    ev = np.zeros(n * NUM_SYMMETRIES, dtype=np.float32)
    pr = np.zeros((n * NUM_SYMMETRIES, NUM_MOVES), dtype=np.float32)
    if n:
        trainer.writeSamples(gs, ev, pr)
    return gs, ev, pr


def save_samples(sample_folder, game_states, eval_labels, prob_labels):
    """main.pyx:200-204"""
    os.makedirs(sample_folder, exist_ok=True)
    np.savez_compressed(os.path.join(sample_folder, "game_states"), game_states)
    np.savez_compressed(os.path.join(sample_folder, "evaluation_labels"), eval_labels)
    np.savez_compressed(os.path.join(sample_folder, "probability_labels"), prob_labels)


def load_samples(sample_folder):
    """the reader side of main.pyx:208-216"""
    out = []
    for name, shape in (("game_states", (-1, GAME_STATE_SIZE)), ("evaluation_labels", (-1,)),
                        ("probability_labels", (-1, NUM_MOVES))):
        with np.load(os.path.join(sample_folder, name + ".npz")) as z:
            out.append(np.reshape(z["arr_0"], shape))
    return tuple(out)


def samples_for_training(trainer, sample_folder, old_training_samples=(), mix_old=False):
    """The whole of get_samples (main.pyx:189-219): fetch this generation's samples, save them, then walk the
    replay window `old_training_samples` (folders of earlier generations, wrapper.py passes the last few).

    The reference loads every old generation and calls np.concatenate on it -- and DISCARDS the result
    (main.pyx:212-214), so what it returns, and trains on, is the current generation alone.  That is
    reproduced by default (the old files are still opened, read and shape-checked like the reference does,
    so a corrupt window fails here too).  mix_old=True returns what the code evidently meant: the current
    samples followed by the window's."""
    game_states, eval_labels, prob_labels = get_samples(trainer)
    save_samples(sample_folder, game_states, eval_labels, prob_labels)
    for cur_path in old_training_samples:
        old_gs, old_ev, old_pr = load_samples(cur_path)
        a = np.concatenate((game_states, old_gs))
        b = np.concatenate((eval_labels, old_ev))
        c = np.concatenate((prob_labels, old_pr))
        if mix_old:
            game_states, eval_labels, prob_labels = a, b, c
    return game_states, eval_labels, prob_labels
