"""Multi-GPU plumbing: one process per GPU, games sharded, no data-path collective
(games are independent, trainer.cpp:243-255; each rank runs its own fused loop).  At the end
of a generation:

  C1  one all-gather of the per-rank sample counts (world x int32), then ONE all-gather of the
      un-augmented samples, every rank contributing max(count) rows of 167 floats
      (state[70] | policy[96] as one block, then the outcomes) -- the x8 symmetry expansion
      happens after the gather, locally;
  C2  one all-reduce of two scalars: the score sum (Trainer::score, trainer.cpp:59-68) and the
      number of games that did not finish.

torch.distributed is the transport only (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).
"""
import numpy as np

SAMPLE_FLOATS = 166  # state[70] + policy[96]
ROW_FLOATS = SAMPLE_FLOATS + 1  # + outcome
MAX_PLIES = 44


def shard(rank, world, games_per_rank):
    """-> (game_base, total_games) keeping seeds/colours on the global game index"""
    return rank * games_per_rank, world * games_per_rank


class SampleGather:
    """Reusable buffers for the per-generation gather.  `on_device=True` packs with the
    engine's kernel straight into CUDA tensors and gathers them over RCCL; otherwise host
    arrays travel (gloo)."""

    def __init__(self, trainer, games_per_rank, on_device=True, group=None):
        import torch
        import torch.distributed as dist

        self.t, self.dist, self.torch, self.group = trainer, dist, torch, group
        self.world = dist.get_world_size(group)
        self.games = games_per_rank
        self.cap = games_per_rank * MAX_PLIES  # upper bound of a rank's rows; the gather moves max(count) of them
        dev = "cuda" if on_device else "cpu"
        self.dev = dev
        self.on_device = on_device
        # The payload buffers are sized AFTER the counts are known (round 5): world x max(count) rows, not
        # world x games x 44 plies -- 963 MB per rank at 8 x 4096 games before, ~400 MB for the ~17 plies games have.
        # They are kept and only grow (a generation's counts vary by a few per cent).
        self.buf = None
        self.all = None
        self.cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.all_cnt = torch.zeros((self.world,), dtype=torch.int32, device=dev)
        self.pair = torch.zeros((2,), dtype=torch.float64, device=dev)
        self.bytes_moved = 0

    def gather(self):
        """all ranks call; returns (counts[world] on the host, max_n, flat tensor): rank r's block is
        flat[r * max_n * 167 : (r + 1) * max_n * 167] = max_n x 166 state/policy floats, then max_n outcomes"""
        n = self.t.num_samples()
        self.cnt[0] = n
        self.dist.all_gather_into_tensor(self.all_cnt, self.cnt, group=self.group)
        counts = self.all_cnt.cpu().numpy().astype(np.int64)
        max_n = int(counts.max())
        if max_n == 0:
            return counts, 0, self.torch.zeros((0,), dtype=self.torch.float32, device=self.dev)
        assert max_n <= self.cap, (max_n, self.cap)
        if self.buf is None or self.buf.numel() < max_n * ROW_FLOATS:
            rows = min(self.cap, max_n + max_n // 8 + 64)  # some headroom, so that the next generations reuse it
            self.buf = self.torch.zeros((rows * ROW_FLOATS,), dtype=self.torch.float32, device=self.dev)
            self.all = self.torch.zeros((self.world * rows * ROW_FLOATS,), dtype=self.torch.float32, device=self.dev)
        sp = self.buf[:max_n * SAMPLE_FLOATS]
        oc = self.buf[max_n * SAMPLE_FLOATS:max_n * ROW_FLOATS]
        if self.on_device:
            got = self.t.pack_samples_device(sp.data_ptr(), oc.data_ptr(), max_n)
        else:
            hsp, hoc = self.t.export_samples()
            got = hsp.shape[0]
            sp[:got * SAMPLE_FLOATS] = self.torch.from_numpy(hsp.reshape(-1))
            oc[:got] = self.torch.from_numpy(hoc)
        assert got == n
        out = self.all[:self.world * max_n * ROW_FLOATS]
        self.dist.all_gather_into_tensor(out, self.buf[:max_n * ROW_FLOATS], group=self.group)
        if self.on_device:
            self.torch.cuda.synchronize()
        self.bytes_moved = self.world * max_n * ROW_FLOATS * 4
        return counts, max_n, out

    def rows(self):
        """concatenated (state_policy, outcome) of the whole generation, in global game
        order = what one Trainer of world x games would export"""
        counts, max_n, flat = self.gather()
        flat = flat.cpu().numpy()
        parts_sp, parts_oc = [], []
        for r in range(self.world):
            blk = flat[r * max_n * ROW_FLOATS:(r + 1) * max_n * ROW_FLOATS]
            c = int(counts[r])
            parts_sp.append(blk[:max_n * SAMPLE_FLOATS].reshape(max_n, SAMPLE_FLOATS)[:c])
            parts_oc.append(blk[max_n * SAMPLE_FLOATS:][:c])
        return np.concatenate(parts_sp), np.concatenate(parts_oc)

    def score_and_unfinished(self, finished=True):
        """C2: (mean score over the whole generation, games not finished anywhere)"""
        self.pair[0] = float(self.t.score()) * self.games
        self.pair[1] = 0.0 if finished else 1.0
        self.dist.all_reduce(self.pair, group=self.group)
        if self.on_device:
            self.torch.cuda.synchronize()
        return float(self.pair[0]) / (self.games * self.world), int(self.pair[1])
