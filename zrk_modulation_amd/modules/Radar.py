"""SectorRadar on the device (reference modules/Radar.py:12-218).

step() keeps the reference's sequence (:144-205): AllObjectsMessage, visibility gate over the live
objects (one kernel sweep + stable compaction), in-place N(0, 5^2) noise on the detections drawn from
numpy's global legacy stream exactly as `smooth_objects` does (:138-142, so a seeded run reproduces
the reference's positions), FoundObjectsMessage, the two relays, then the scan-state update on the
host (:96-117; scalar per-radar state).
"""
import logging

import numpy as np

from ..engine import scan_mode_code, scan_next
from .BaseModel import BaseModel
from .constants import CCP_ID, RADAR_NOISE_SIGMA, MessageType
from .Messages import AllObjectsMessage, DestroyedMissileId, FoundObjectsMessage, UpdateTargetPosition

logger = logging.getLogger(__name__)


class SectorRadar(BaseModel):
    def __init__(self, manager, id: int, pos: np.ndarray, azimuth_start: float, elevation_start: float,
                 max_distance: float, azimuth_range: float, elevation_range: float, azimuth_speed: float,
                 elevation_speed: float, scan_mode: str = "horizontal"):
        super().__init__(manager, id, pos)
        self.azimuth_start = azimuth_start
        self.elevation_start = elevation_start
        self.max_distance = max_distance
        self.azimuth_range = azimuth_range
        self.elevation_range = elevation_range
        self.azimuth_speed = azimuth_speed
        self.elevation_speed = elevation_speed
        self.scan_mode = scan_mode
        self.current_azimuth = azimuth_start
        self.current_elevation = elevation_start

    # geometry ----------------------------------------------------------------------------------
    def _params(self):
        p = np.asarray(self.pos, dtype=np.float64)
        return (p[0], p[1], p[2], self.max_distance, self.current_azimuth, self.azimuth_range,
                self.current_elevation, self.elevation_range)

    def _sweep(self, objects):
        """Device sweep of the current sector.  Returns (store, det tensor, count, slots ndarray)."""
        store = getattr(objects, "store", None)
        if store is None:
            raise TypeError("SectorRadar needs the ActiveObjectsMessage of a device-backed AirEnv "
                            "(zrk_modulation_amd.modules.AirEnv); plain object lists have no CPU path here")
        store.sweep([self._params()], 0)
        det, cnts = store.compact(1)
        cnt = int(cnts[0].item())
        slots = det[:cnt].cpu().numpy() if cnt else np.zeros(0, np.int32)
        return store, det, cnt, slots

    def find_visible_objects(self, objects):
        """Objects inside range and sector, in input order (reference :44-73)."""
        _, _, _, slots = self._sweep(objects)
        return [objects._handles[s] for s in slots]

    def smooth_objects(self, objects):
        """pos += N(0, 5^2) per axis for every object of the list, in list order, from numpy's global legacy stream
        (reference :138-142): one (k, 3) draw is bit-identical to the reference's k separate draws.  Objects that live
        in a device table are perturbed there in one launch; others on the host."""
        objects = list(objects)
        if not objects:
            return
        noise = np.random.normal(0, RADAR_NOISE_SIGMA, (len(objects), 3))
        bound = [k for k, o in enumerate(objects) if getattr(o, "_store", None) is not None and o._frozen is None]
        stores = {id(objects[k]._store): objects[k]._store for k in bound}
        if len(stores) == 1 and len(bound) == len(objects):
            import torch
            store = next(iter(stores.values()))
            idx = torch.as_tensor(np.asarray([o._slot for o in objects], np.int32), device=store.device)
            store.noise_apply(idx, len(objects), noise)
            return
        for o, nz in zip(objects, noise):
            o.pos = o.pos + nz

    def start(self, objects):
        """The reference's stand-alone test helper (reference :207-218): look, move to the next sector, repeat."""
        num_steps = int(self.azimuth_range / self.azimuth_speed * self.elevation_range / self.elevation_speed)
        seen = []
        for _ in range(num_steps):
            seen.append(self.find_visible_objects(objects))
            self.move_to_next_sector()
        return seen

    def move_to_next_sector_circular(self):
        self.current_azimuth, self.current_elevation = scan_next(
            scan_mode_code(self.scan_mode), self.azimuth_range, self.azimuth_speed, self.elevation_speed,
            self.elevation_start, self.current_azimuth, self.current_elevation)

    def move_to_next_sector(self):
        """Non-circular variant used only by the reference's test helper (reference :75-94)."""
        if self.scan_mode == "horizontal":
            self.current_azimuth += self.azimuth_speed
            if self.current_azimuth >= self.azimuth_start + self.azimuth_range:
                self.current_azimuth = self.azimuth_start
                self.current_elevation += self.elevation_speed
                if self.current_elevation >= self.elevation_start + self.elevation_range:
                    self.current_elevation = self.elevation_start
        elif self.scan_mode == "vertical":
            self.current_elevation += self.elevation_speed
            if self.current_elevation >= self.elevation_start + self.elevation_range:
                self.current_elevation = self.elevation_start
                self.current_azimuth += self.azimuth_speed
                if self.current_azimuth >= self.azimuth_start + self.azimuth_range:
                    self.current_azimuth = self.azimuth_start

    def update_scan_parameters(self, new_azimuth_range=None, new_elevation_range=None, new_azimuth_speed=None,
                               new_elevation_speed=None):
        if new_azimuth_range is not None:
            self.azimuth_range = new_azimuth_range
        if new_elevation_range is not None:
            self.elevation_range = new_elevation_range
        if new_azimuth_speed is not None:
            self.azimuth_speed = new_azimuth_speed
        if new_elevation_speed is not None:
            self.elevation_speed = new_elevation_speed

    # tick --------------------------------------------------------------------------------------
    def step(self):
        mgr = self._manager
        now, dt = mgr.time.get_time(), mgr.time.get_dt()
        objects = mgr.give_messages_by_type(MessageType.ACTIVE_OBJECTS)[0].active_objects

        mgr.add_message(AllObjectsMessage(time=now, sender_id=self.id, receiver_id=CCP_ID, objects=objects))

        store, det, cnt, slots = self._sweep(objects)
        if cnt:
            # the same draws, in the same order, from the same global stream as the reference's
            # per-object np.random.normal(0, 5, 3)
            store.noise_apply(det, cnt, np.random.normal(0, RADAR_NOISE_SIGMA, (cnt, 3)))
        visible = [objects._handles[s] for s in slots]
        logger.debug("radar %s sees %d objects", self.id, cnt)
        found = FoundObjectsMessage(time=now, sender_id=self.id, receiver_id=CCP_ID, visible_objects=visible)
        # (for a command post that works on the device: the same list as table rows, still in HBM -- the compaction's buffer
        # is reused by the next radar, hence the copy)
        found.device_rows, found.device_store = (det[:cnt].clone() if cnt else det[:0]), store
        mgr.add_message(found)

        # relay of target updates to missiles: nobody produces UPDATE_TARGET, kept for shape
        for m in mgr.give_messages_by_type(MessageType.UPDATE_TARGET, step_time=now - dt):
            mgr.add_message(UpdateTargetPosition(time=now, sender_id=self.id, receiver_id=m.missile_id,
                                                 upd_object=m.target))
        # relay of last tick's detonations to the command post
        for m in mgr.give_messages_by_type(MessageType.MISSILE_DETONATE, step_time=now - dt):
            mgr.add_message(DestroyedMissileId(time=now, sender_id=self.id, receiver_id=CCP_ID,
                                               missile_id=m.missile_id, self_detonation=m.self_detonation))
        self.move_to_next_sector_circular()
