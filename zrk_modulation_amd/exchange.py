"""Per-tick exchange of the compacted detection list between the GPUs of a node.

Entities shard by contiguous index ranges (rank g owns [g*shard, (g+1)*shard)), so the
rank-ordered concatenation of every rank's stable list IS the single-GPU list.  What travels is
the union list zrk_compact packs: element 0 = count, then ((global index << 32) | radar mask) per
detected entity, ascending.  One fixed-size all-gather per tick, no host round trip; a radar's own
list is a stable filter of the gathered union (bit r of the mask).

torch.distributed's "nccl" backend is RCCL on ROCm (xGMI inside a node); the same code runs on
"gloo" with CPU tensors, which is how tests cover world_size > 1 without GPUs.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class DetectionExchange:
    def __init__(self, capacity, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.capacity = int(capacity)
        self.gathered = torch.zeros(self.world, self.capacity + 1, dtype=torch.int64, device=device)
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self._flat = backend == "nccl" or torch.device(device).type == "cpu"

    def all_gather(self, packed, async_op=False):
        """packed: int64[capacity+1] from zrk_compact on this rank.  Fills and returns
        int64[world, capacity+1]; with async_op=True returns the work handle instead (wait() on it
        before reading `gathered` or overwriting `packed`), so the exchange of tick t overlaps the
        sweep of tick t+1."""
        assert packed.numel() == self.capacity + 1
        if self.world == 1:
            self.gathered[0].copy_(packed)
            return None if async_op else self.gathered
        if self._flat:
            work = dist.all_gather_into_tensor(self.gathered.view(-1), packed, group=self.group, async_op=async_op)
        else:                       # backends without the flat form for this device type
            work = dist.all_gather([self.gathered[g] for g in range(self.world)], packed, group=self.group,
                                   async_op=async_op)
        return work if async_op else self.gathered

    def counts(self):
        """Per-rank detection counts (synchronises)."""
        return self.gathered[:, 0].cpu().tolist()

    def overflowed(self):
        return any(c > self.capacity for c in self.counts())

    def merged(self):
        """Global union list in index order: (indices int64, masks int64).  Synchronises."""
        cnt = self.counts()
        parts = [self.gathered[g, 1:1 + min(c, self.capacity)] for g, c in enumerate(cnt)]
        allp = torch.cat(parts) if parts else self.gathered.new_zeros(0)
        return allp >> 32, allp & 0xFFFFFFFF

    def radar_list(self, r):
        """FoundObjectsMessage order for radar r over the whole population (global indices)."""
        idx, mask = self.merged()
        return idx[((mask >> r) & 1).bool()]
