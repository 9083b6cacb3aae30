"""The CLOSED loop of the reference, composed from the oracle's pieces: TEST INFRASTRUCTURE ONLY (like everything under
oracle/: imported by tests/ and bench.py's cpu_baseline leg, never by the product package).

`OracleBattery` runs, tick by tick and with the reference's message latencies, what `Manager.run_simulation` makes ALL the
modules do together (reference modules/Manager.py:111-140, modules ordered AirEnv, radars, launchers, command post):

    AirEnv.step                      OracleSim.airenv_step            modules/AirEnv.py:26-53
    SectorRadar.step (each)          OracleSim.radar_step             modules/Radar.py:144-205
    MissileLauncher.step (each)      _launcher_step (below)           modules/MissileLauncher.py:82-138, :58-80
        Missile.step 'ready' -> _launch                               modules/Missile.py:153-160, :104-133
    CombatControlPoint.step          CcpState.step (zo_ccp_step)      modules/CCP.py:368-431

Latencies, all of which follow from who reads which tick's messages:
    tick t      the command post links its detections and asks launcher l for a missile (LAUNCH_COMMAND @t)
    tick t+1    launcher l reads @t: pops a missile from the END of its list, the missile solves against the target's position
                as it stands NOW -- after this tick's radars have added their noise -- (LAUNCH_SUCCESSFUL / LAUNCH_CANCELLED @t+1)
    tick t+2    the launcher reads @t+1: a cancelled missile goes back to the end of its list (BEFORE this tick's requests are
                served: its messages were posted before the command post's), a launched one is announced (LAUNCHED_MISSILE,
                NEW_MISSILE @t+2); the command post, later in the same tick, takes it into its missile dictionary
    tick t+3    AirEnv appends it and steps it for the first time (its fuse timer starts here)
    tick 0      the command post asks for the missile counts, tick 1 it learns them: no launch request before tick 1
The command post never forgets a track (DestroyedMissileId carries a 1-tuple) and never learns of a
cancelled launch.
"""
from __future__ import annotations

import numpy as np

from . import oracle as O


class OracleBattery:
    def __init__(self, cfg, noise_fn, slack_steps=None):
        """cfg: the YAML-schema dictionary the reference loads (main.py:35-149); noise_fn(count) -> (count, 3) draws, or None."""
        self.cfg = cfg
        self.dt_ms = int(cfg["simulation"]["time_step"])
        targets = cfg["air_environment"].get("targets", []) or []
        launchers = cfg.get("missile_launchers", []) or []
        ccp = cfg.get("combat_control_point", {}) or {}
        n_m = sum(len(lc.get("missiles", []) or []) for lc in launchers)
        self.sim = O.OracleSim(self.dt_ms, len(targets) + n_m + 4, n_m + 1)
        for tc in targets:
            self.sim.add_target(tc["id"], tc["position"], tc["velocity"], 0.0)
        for rc in cfg.get("radars", []) or []:
            self.sim.add_radar(rc["id"], rc["position"], rc["azimuth_start"], rc["elevation_start"], rc["max_distance"],
                               rc["azimuth_range"], rc["elevation_range"], rc["azimuth_speed"], rc["elevation_speed"], rc["scan_mode"])
        # launchers in module order; each holds its missiles as a list it pops from the END (MissileLauncher.py:69)
        self.launchers = []
        for lc in launchers:
            stack = []
            for mc in lc.get("missiles", []) or []:
                if len(stack) < lc.get("max_missiles", 5):            # MissileLauncher.add_missile, :42-47
                    self.sim.add_missile(mc["id"], lc["position"], mc.get("velocity", 1000), mc.get("explosion_radius", 50), mc.get("life_time", 60))
                    stack.append(int(mc["id"]))
            self.launchers.append(dict(id=lc["id"], pos=np.asarray(lc["position"], np.float64), stack=stack,
                                       inbox_prev=[], inbox_now=[]))
        # the command post's launchers: the ids it was given, in that order, as far as they exist (main.py:106-110)
        by_id = {l["id"]: k for k, l in enumerate(self.launchers)}
        self.ccp_launchers = [by_id[i] for i in ccp.get("missile_launcher_ids", []) if i in by_id]
        self.post = O.CcpState(self.sim.cap, [self.launchers[k]["pos"] for k in self.ccp_launchers] or np.zeros((0, 3)),
                               np.zeros(len(self.ccp_launchers), np.int32))
        self.noise_fn = noise_fn
        self.slack_steps = 100 if slack_steps is None else slack_steps      # POSSIBLE_TARGET_RADIUS, modules/constants.py:31
        self.tick_no = 0
        self._count_request_posted = False
        self._count_response = None            # (tick it is readable in, counts)

    # one MissileLauncher.step: the messages addressed to it one tick ago, in the order they were posted
    def _launcher_step(self, k, log):
        lch = self.launchers[k]
        sim, t = self.sim, self.sim.time_ms
        for kind, a, b in lch["inbox_prev"]:
            if kind == "count_request":
                self._count_response = self._count_response or {}
                self._count_response[k] = len(lch["stack"])
            elif kind == "ok":                       # LaunchedMissileMessage + MissileToAirEnvMessage (:103-124)
                sim.announce_missile(a)
                slot = sim.n + len(sim._pending_new) - 1        # the row AirEnv gives it next tick (AirEnv.py:42-43: in message order)
                log["new_missile"].append([t, a])
                log["launched_now"].append((a, slot))
            elif kind == "cancel":                   # back to the end of the list (:126-129)
                lch["stack"].append(a)
            elif kind == "request":                  # launch_missile (:58-80): pop, LaunchMissileMessage, missile.step() -> _launch
                if not lch["stack"]:
                    continue
                mid = lch["stack"].pop()
                tgt_slot = a
                log["launch_cmd"].append([t, lch["id"], mid, int(sim.ids[tgt_slot])])
                rc, V, _tt = sim.launch(mid, tgt_slot)
                if rc == 0:
                    log["launch_ok"].append([t, mid, int(sim.ids[tgt_slot])])
                    log["launch_traj"].append(list(V) + list(sim.m_pos0[sim.row_of_missile[mid]]) + [t / 1000])
                    lch["inbox_now"].append(("ok", mid, tgt_slot))
                else:
                    log["launch_cancel"].append([t, mid, rc])
                    lch["inbox_now"].append(("cancel", mid, None))

    def tick(self):
        """One tick of every module; returns what the reference's message dictionary would hold of it."""
        sim = self.sim
        t = sim.time_ms
        now = t / 1000
        log = dict(t=t, detonations=[], found=[], launch_req=[], launch_cmd=[], launch_ok=[], launch_traj=[], launch_cancel=[],
                   new_missile=[], launched_now=[], verdicts=None)
        for ms, ts, self_det in sim.airenv_step():
            log["detonations"].append([t, int(sim.ids[ms]), -1 if ts < 0 else int(sim.ids[ts]), int(self_det)])
        log["active"] = sim.ids[sim.active_slots()].copy()
        for r in range(len(sim.radars)):
            log["found"].append(sim.radar_step(r, self.noise_fn))
        log["radar_state"] = np.array([[rd["caz"], rd["cel"]] for rd in sim.radars])
        for k in range(len(self.launchers)):
            self._launcher_step(k, log)
        # ---- CombatControlPoint.step (modules/CCP.py:368-431) ----
        if self.ccp_launchers or self.cfg.get("combat_control_point"):
            if not self._count_request_posted:       # send_request_msg_to_ml_capacity (:122-136): capacity and launched start at 0
                for k in self.ccp_launchers:
                    self.launchers[k]["inbox_now"].insert(0, ("count_request", None, None))      # (relevance 3: ahead of the rest)
                self._count_request_posted = True
            if self._count_response:                 # get_current_missile_launcher_capacity (:138-146): answered THIS tick
                for j, k in enumerate(self.ccp_launchers):
                    if k in self._count_response:
                        self.post.l_cap[j] = self._count_response[k]
                self._count_response = None
            # check_if_missiles_launched (:160-169): LAUNCHED_MISSILE of this tick, in the order the launchers posted them
            cap = sim.cap
            P = sim.pos_of(np.arange(cap)); Q = sim.prev_of(np.arange(cap))
            none = (sim.prev_valid == 0).astype(np.uint8)
            speed = sim.speed_mod.copy()
            for mid, slot in log["launched_now"]:
                r = sim.row_of_missile[mid]
                P[slot] = sim.m_pos0[r]; none[slot] = 1; speed[slot] = sim.m_v0[r]     # not in the air yet: where it stands, never stepped
                self.post.add_missile(slot, now)
            seq = np.concatenate(log["found"]).astype(np.int32) if log["found"] else np.zeros(0, np.int32)
            rows, verdict, match, launcher = self.post.step(seq, P, Q, none, speed, now, self.slack_steps * self.dt_ms / 1000)
            log["verdicts"] = (rows, verdict, match, launcher)
            # LAUNCH_COMMAND: to the launcher, about the DETECTED object, in processing order (:306-314)
            radar_of_row = {}
            for r, f in enumerate(log["found"]):
                for row in f:
                    radar_of_row.setdefault(int(row), sim.radars[r]["id"])
            for row, l in zip(rows, launcher):
                if l >= 0:
                    k = self.ccp_launchers[l]
                    self.launchers[k]["inbox_now"].append(("request", int(row), None))
                    log["launch_req"].append([t, self.launchers[k]["id"], int(sim.ids[row]), radar_of_row[int(row)]])
        for lch in self.launchers:
            lch["inbox_prev"], lch["inbox_now"] = lch["inbox_now"], []
        sim.end_tick()
        self.tick_no += 1
        return log
