"""Multi-GPU plumbing: one process per GPU, games sharded, ONE collective per
generation -- the all-gather of the un-augmented samples.

torch.distributed is used as the transport only (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  No data-path collective exists: games are
independent (trainer.cpp:243-255), each rank runs its own fused loop.
"""
import numpy as np

SAMPLE_FLOATS = 166  # state[70] + policy[96]
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
        self.cap = games_per_rank * MAX_PLIES
        dev = "cuda" if on_device else "cpu"
        self.on_device = on_device
        self.sp = torch.zeros((self.cap, SAMPLE_FLOATS), dtype=torch.float32, device=dev)
        self.oc = torch.zeros((self.cap,), dtype=torch.float32, device=dev)
        self.all_sp = torch.zeros((self.world * self.cap, SAMPLE_FLOATS), dtype=torch.float32, device=dev)
        self.all_oc = torch.zeros((self.world * self.cap,), dtype=torch.float32, device=dev)
        self.cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.all_cnt = torch.zeros((self.world,), dtype=torch.int32, device=dev)

    def gather(self):
        """all ranks call; returns (counts[world], all_sp, all_oc) -- rank r's rows are
        all_sp[r*cap : r*cap + counts[r]]"""
        if self.on_device:
            n = self.t.pack_samples_device(self.sp.data_ptr(), self.oc.data_ptr(), self.cap)
        else:
            sp, oc = self.t.export_samples()
            n = sp.shape[0]
            self.sp[:n] = self.torch.from_numpy(sp)
            self.oc[:n] = self.torch.from_numpy(oc)
        self.cnt[0] = n
        self.dist.all_gather_into_tensor(self.all_cnt, self.cnt, group=self.group)
        self.dist.all_gather_into_tensor(self.all_sp, self.sp, group=self.group)
        self.dist.all_gather_into_tensor(self.all_oc, self.oc, group=self.group)
        if self.on_device:
            self.torch.cuda.synchronize()
        return self.all_cnt, self.all_sp, self.all_oc

    def rows(self):
        """concatenated (state_policy, outcome) of the whole generation, in global game
        order = what one Trainer of world x games would export"""
        cnt, sp, oc = self.gather()
        cnt = cnt.cpu().numpy()
        sp = sp.cpu().numpy()
        oc = oc.cpu().numpy()
        parts_sp = [sp[r * self.cap:r * self.cap + int(cnt[r])] for r in range(self.world)]
        parts_oc = [oc[r * self.cap:r * self.cap + int(cnt[r])] for r in range(self.world)]
        return np.concatenate(parts_sp), np.concatenate(parts_oc)
