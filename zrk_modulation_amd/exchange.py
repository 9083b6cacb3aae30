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


def union_bits_words(n, R, entries):
    """64-bit words of the wire format for n slots and up to `entries` seen ones (zrk_union_bits_words)."""
    return 2 + (int(n) + 63) // 64 + (int(entries) * (2 if R <= 16 else 4) + 7) // 8


def encode_union_bits(vis, R, words):
    """The wire format zrk_compact_bits writes, from a mask array on the host (tests, CPU ranks): word 0 = count,
    word 1 = n, one bit per slot, then the masks of the seen slots (16 bits each when R <= 16, else 32)."""
    import numpy as np
    vis = np.asarray(vis, np.uint32)
    n = len(vis)
    out = np.zeros(int(words), np.int64)
    seen = np.nonzero(vis)[0]
    nbits = (n + 63) // 64
    bits = np.zeros(nbits * 64, np.uint8)
    bits[seen] = 1
    out[0], out[1] = len(seen), n
    out[2:2 + nbits] = np.packbits(bits.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1).view(np.int64)
    dt = np.uint16 if R <= 16 else np.uint32
    room = out[2 + nbits:].view(dt)
    k = min(len(seen), len(room))
    room[:k] = vis[seen[:k]].astype(dt)
    return out


class PoisonedList(RuntimeError):
    """A rank's collective went out although its list was never handed over (the wait kernel in front of it gave up,
    include/zrk_hot.h: zrk_exchange_wait): that rank wrote count -1 over the list instead of letting an unfinished one
    pass for a tick's detections."""


def check_not_poisoned(gathered):
    """Raises PoisonedList if any rank's buffer carries the poison (count < 0).  Synchronises."""
    bad = torch.nonzero(gathered[:, 0] < 0).reshape(-1).cpu().tolist()
    if bad:
        raise PoisonedList(f"the list of rank(s) {bad} is poisoned: its producer never handed it over to the collective")


def decode_union_bits(gathered, R, offsets, ev_capacity=0, union_only=False):
    """Global union list from gathered wire-format buffers int64[world, words]: (indices int64, masks int64) in index
    order.  union_only: the buffers carry the bitmap alone (no masks behind it): (indices, None).  Synchronises."""
    world, words = gathered.shape
    check_not_poisoned(gathered)
    tail = 1 + int(ev_capacity) if ev_capacity else 0
    idx, msk = [], []
    if union_only:
        shifts = torch.arange(64, dtype=torch.int64, device=gathered.device)
        for g in range(world):
            n = int(gathered[g, 1].item())
            nw = (n + 63) // 64
            bits = ((gathered[g, 2:2 + nw].unsqueeze(1) >> shifts) & 1).reshape(-1)[:n]
            idx.append(torch.nonzero(bits).reshape(-1) + offsets[g])
        return (torch.cat(idx) if idx else gathered.new_zeros(0)), None
    shifts = torch.arange(64, dtype=torch.int64, device=gathered.device)
    for g in range(world):
        c, n = int(gathered[g, 0].item()), int(gathered[g, 1].item())
        nw = (n + 63) // 64
        bits = ((gathered[g, 2:2 + nw].unsqueeze(1) >> shifts) & 1).reshape(-1)[:n]
        slots = torch.nonzero(bits).reshape(-1)
        body = gathered[g, 2 + nw:words - tail]
        room = body.numel() * (4 if R <= 16 else 2)
        k = min(c, room)
        masks = (body.view(torch.int16)[:k].to(torch.int64) & 0xFFFF) if R <= 16 else (body.view(torch.int32)[:k].to(torch.int64) & 0xFFFFFFFF)
        idx.append(slots[:k] + offsets[g])
        msk.append(masks)
    return (torch.cat(idx) if idx else gathered.new_zeros(0)), (torch.cat(msk) if msk else gathered.new_zeros(0))


def decode_events(gathered, ev_capacity):
    """Detonations carried at the tail of gathered wire buffers, rank after rank, each rank's in list order:
    [(global missile index, global target index | -1)].  Synchronises."""
    out = []
    if not ev_capacity:
        return out
    check_not_poisoned(gathered)
    tail = gathered[:, gathered.shape[1] - 1 - int(ev_capacity):].cpu().numpy()
    for g in range(tail.shape[0]):
        k = min(int(tail[g, 0]), int(ev_capacity))
        for w in tail[g, 1:1 + k]:
            w = int(w) & 0xFFFFFFFFFFFFFFFF
            t = w & 0xFFFFFFFF
            out.append((w >> 32, -1 if t == 0xFFFFFFFF else t))
    return out


def rccl_library_path():
    """The librccl this process already holds: PyTorch's own copy ("nccl" in torch.distributed is RCCL on ROCm)."""
    import os
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


class RcclExchange:
    """The exchange issued from the C side (zrk_exchange_*, include/zrk_hot.h): the library's own RCCL communicator and
    stream; zrk_run_ticks_x posts one all-gather per tick behind that tick's compaction, so no Python runs between a
    tick and its collective.  torch.distributed is used once, to hand rank 0's communicator id to the others."""

    def __init__(self, words, device, R, offsets=None, ev_capacity=0, group=None, wire="masks", interest=None):
        """wire "masks": count, n, one bit per slot, the 16- or 32-bit radar masks of the seen slots (`words` from
        union_bits_words(n, R, entries)); wire "union": the bitmap alone (union_bits_words(n, R, 0)): who was seen by
        any radar, a fifth of the bytes -- what crosses xGMI when no consumer on another rank asks which radar saw it.
        interest: the radars (indices) consumers on other ranks listen to -- the reference's command post reads one
        FoundObjectsMessage per radar in its radar_ids (modules/CCP.py:409-417); the list that travels is then the union list
        of THESE radars (a slot is on it when one of them saw it, its mask carries their bits); None: all radars."""
        import ctypes as C
        from . import _lib
        self._C, self.lib = C, _lib.load()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.R, self.ev_capacity = int(R), int(ev_capacity)
        self.union_only = wire == "union"
        self.interest = 0
        for r in (interest or []):
            assert 0 <= int(r) < self.R, "interest: radar indices"
            self.interest |= 1 << int(r)
        self.words = int(words) + (1 + self.ev_capacity if self.ev_capacity else 0)
        self.offsets = list(offsets) if offsets is not None else [0] * self.world
        path = rccl_library_path()
        cpath = path.encode() if path else None
        uid = _lib.ZrkRcclId()
        made = 1
        if self.rank == 0:
            made = 1 if self.lib.zrk_exchange_unique_id(cpath, C.byref(uid)) == 0 else 0
        if self.world > 1:
            # rank 0's id and whether it got one travel together, so that every rank goes on or gives up alike (a rank
            # that raised alone would leave the others waiting in the communicator's rendezvous)
            backend = dist.get_backend(group)
            t = torch.frombuffer(bytearray(bytes(uid) + bytes([made])), dtype=torch.uint8).clone()
            t = t.to(self.device) if backend == "nccl" else t
            dist.broadcast(t, src=0, group=group)
            raw = t.cpu().numpy().tobytes()
            C.memmove(C.byref(uid), raw[:128], 128)
            made = raw[128]
        if not made:
            raise _lib.ZrkError(f"zrk_exchange_unique_id failed on rank 0: RCCL could not be loaded from {path}")
        h = C.c_void_p()
        rc = self.lib.zrk_exchange_create(cpath, C.byref(uid), self.world, self.rank, self.device.index or 0, C.byref(h))
        self.handle = h
        if rc != 0:
            msg = self.lib.zrk_exchange_last_error(h) if h.value else b""
            raise _lib.ZrkError(f"zrk_exchange_create failed ({rc}): {msg.decode() if msg else ''}")
        self._lib_mod = _lib
        self._allocate()

    def _allocate(self):
        self.slots = self._lib_mod.EXCHANGE_SLOTS      # tick t goes through slot t % slots
        self.send = [torch.zeros(self.words, dtype=torch.int64, device=self.device) for _ in range(self.slots)]
        self.recv = [torch.zeros(self.world, self.words, dtype=torch.int64, device=self.device) for _ in range(self.slots)]
        io = self._lib_mod.ZrkExchangeIo()
        io.x = self.handle
        for k in range(self.slots):
            io.send[k], io.recv[k] = self.send[k].data_ptr(), self.recv[k].data_ptr()
        io.words, io.ev_capacity = self.words, self.ev_capacity
        io.interest = self.interest
        self.io = io

    def resize(self, words):
        """New list size (same on every rank), same communicator: waits for what is in flight, then new buffers."""
        self.sync()
        self.words = int(words) + (1 + self.ev_capacity if self.ev_capacity else 0)
        self._allocate()

    def sync(self):
        rc = self.lib.zrk_exchange_sync(self.handle)
        if rc != 0:
            raise RuntimeError(self.lib.zrk_exchange_last_error(self.handle).decode())

    def counts(self, slot):
        self.sync()
        return self.recv[slot][:, 0].cpu().tolist()

    def room(self):
        """Masks a rank's list has room for, given the largest shard seen so far."""
        n = int(max(r[:, 1].max().item() for r in self.recv))
        body = self.words - (1 + self.ev_capacity if self.ev_capacity else 0) - 2 - (n + 63) // 64
        return body * (4 if self.R <= 16 else 2)

    def overflowed(self):
        self.sync()
        for r in self.recv:
            check_not_poisoned(r)
        room = self.room()
        if self.union_only:
            room = 1 << 62                     # (nothing but the bitmap travels: no list to run out of room)
        ev_over = self.ev_capacity and any(int(r[:, self.words - 1 - self.ev_capacity].max().item()) > self.ev_capacity for r in self.recv)
        return any(c > room for r in self.recv for c in r[:, 0].cpu().tolist()) or bool(ev_over)

    def merged(self, slot):
        self.sync()
        return decode_union_bits(self.recv[slot], self.R, self.offsets, self.ev_capacity, self.union_only)

    def info(self):
        """zrk_exchange_info: the communicator's own size, the pattern in use, collectives issued, host waits."""
        st = self._lib_mod.ZrkExchangeStats()
        rc = self.lib.zrk_exchange_info(self.handle, self._C.byref(st))
        if rc != 0:
            raise RuntimeError(f"zrk_exchange_info failed ({rc})")
        return {"world": st.world, "rank": st.rank, "rccl_ranks_seen": st.comm_ranks, "pattern": "direct send/recv" if st.direct else "ncclAllGather",
                "helper_threads": st.helper_threads, "grouped_pairs": bool(st.grouped_pairs), "collectives": st.collectives, "host_waits": st.host_waits, "host_wait_us": st.host_wait_us}

    def events(self, slot):
        self.sync()
        return decode_events(self.recv[slot], self.ev_capacity)

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.zrk_exchange_destroy(self.handle)
            self.handle = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DetectionExchange:
    def __init__(self, capacity, device, group=None, fmt="pairs", offsets=None, R=None, union_only=False):
        """fmt "pairs": buffers of capacity + 1 words, [count, (global index << 32 | mask) ...].
        fmt "bits": buffers of `capacity` WORDS in the wire format of zrk_compact_bits (union_bits_words); the slot
        numbers are local to each rank, `offsets[g]` is added to rank g's when the lists are merged, `R` tells the
        width of the masks."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.capacity = int(capacity)
        self.fmt = fmt
        self.offsets = list(offsets) if offsets is not None else [0] * self.world
        self.R = R
        self.union_only = bool(union_only) and fmt == "bits"       # the bitmap alone (see RcclExchange: wire "union")
        words = self.capacity + 1 if fmt == "pairs" else self.capacity
        self.gathered = torch.zeros(self.world, words, dtype=torch.int64, device=device)
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self._flat = backend == "nccl" or torch.device(device).type == "cpu"

    def all_gather(self, packed, async_op=False):
        """packed: int64[capacity+1] from zrk_compact on this rank.  Fills and returns
        int64[world, capacity+1]; with async_op=True returns the work handle instead (wait() on it
        before reading `gathered` or overwriting `packed`), so the exchange of tick t overlaps the
        sweep of tick t+1."""
        assert packed.numel() == self.gathered.shape[1]
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

    def _room(self, g):
        """Masks rank g's buffer has room for (bits format)."""
        n = int(self.gathered[g, 1].item())
        return (self.gathered.shape[1] - 2 - (n + 63) // 64) * (4 if self.R <= 16 else 2)

    def overflowed(self):
        if self.fmt == "pairs":
            return any(c > self.capacity for c in self.counts())
        if self.union_only:
            return False
        return any(c > self._room(g) for g, c in enumerate(self.counts()))

    def merged(self):
        """Global union list in index order: (indices int64, masks int64).  Synchronises."""
        cnt = self.counts()
        if self.fmt == "pairs":
            parts = [self.gathered[g, 1:1 + min(c, self.capacity)] for g, c in enumerate(cnt)]
            allp = torch.cat(parts) if parts else self.gathered.new_zeros(0)
            return allp >> 32, allp & 0xFFFFFFFF
        if self.union_only:
            return decode_union_bits(self.gathered, self.R, self.offsets, 0, True)
        idx, msk = [], []
        shifts = torch.arange(64, dtype=torch.int64, device=self.gathered.device)
        for g, c in enumerate(cnt):
            n = int(self.gathered[g, 1].item())
            nw = (n + 63) // 64
            words = self.gathered[g, 2:2 + nw]
            bits = ((words.unsqueeze(1) >> shifts) & 1).reshape(-1)[:n]
            slots = torch.nonzero(bits).reshape(-1)
            k = min(c, self._room(g))
            tail = self.gathered[g, 2 + nw:]
            masks = (tail.view(torch.int16)[:k].to(torch.int64) & 0xFFFF) if self.R <= 16 else (tail.view(torch.int32)[:k].to(torch.int64) & 0xFFFFFFFF)
            idx.append(slots[:k] + self.offsets[g])
            msk.append(masks)
        return (torch.cat(idx) if idx else self.gathered.new_zeros(0)), (torch.cat(msk) if msk else self.gathered.new_zeros(0))

    def radar_list(self, r):
        """FoundObjectsMessage order for radar r over the whole population (global indices)."""
        idx, mask = self.merged()
        return idx[((mask >> r) & 1).bool()]
