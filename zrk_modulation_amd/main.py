"""Scenario loader: the YAML schema the reference's main.py accepts (reference main.py:30-174),
building the device-backed AirEnv / SectorRadar and the host-side launcher / command post.

    python -m zrk_modulation_amd.main path/to/config.yaml
"""
import logging
import sys
from typing import Any, Dict, Tuple

import numpy as np
import yaml

from .modules.AirEnv import AirEnv
from .modules.AirObject import Trajectory
from .modules.CCP import CombatControlPoint
from .modules.Manager import Manager
from .modules.Missile import Missile
from .modules.MissileLauncher import MissileLauncher
from .modules.Radar import SectorRadar
from .modules.Timer import Timer
from .modules.utils import Target, TargetType

logger = logging.getLogger(__name__)


def load_config(config_path: str) -> Dict[str, Any]:
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


def create_objects_from_config(config: Dict[str, Any], device=None, replay=None) -> Tuple[Manager, Dict[int, object]]:
    """Same construction order as the reference (it fixes module scheduling ties and list order):
    AirEnv, radars, launchers with their missiles, command post, then the targets.
    replay: optional zrk_modulation_amd.replay.ReplayLog for bounded columnar retention of what the GUI replays."""
    manager = Manager(replay=replay)
    by_id: Dict[int, object] = {}

    timer = Timer()
    timer.set_dt(config["simulation"]["time_step"])
    manager.time = timer

    ae_cfg = config["air_environment"]
    air_env = AirEnv(manager, ae_cfg["id"], np.array(ae_cfg["position"]), device=device)
    manager.add_module(air_env)
    by_id[ae_cfg["id"]] = air_env

    for rc in config.get("radars", []) or []:
        radar = SectorRadar(manager, rc["id"], np.array(rc["position"]), rc["azimuth_start"], rc["elevation_start"],
                            rc["max_distance"], rc["azimuth_range"], rc["elevation_range"], rc["azimuth_speed"],
                            rc["elevation_speed"], rc["scan_mode"])
        manager.add_module(radar)
        by_id[rc["id"]] = radar

    for lc in config.get("missile_launchers", []) or []:
        launcher = MissileLauncher(manager, lc["id"], np.array(lc["position"]), lc.get("max_missiles", 5))
        for mc in lc.get("missiles", []) or []:
            launcher.add_missile(Missile(manager, mc["id"], np.array(lc["position"]),
                                         velocity_module=mc.get("velocity", 1000),
                                         detonate_radius=mc.get("explosion_radius", 50),
                                         detonate_period=mc.get("life_time", 60)))
        manager.add_module(launcher)
        by_id[lc["id"]] = launcher

    ccp_cfg = config.get("combat_control_point", {})
    if ccp_cfg:
        launcher_pos = {i: by_id[i].pos for i in ccp_cfg.get("missile_launcher_ids", []) if i in by_id}
        radar_pos = {i: by_id[i].pos for i in ccp_cfg.get("radar_ids", []) if i in by_id}
        ccp = CombatControlPoint(manager, ccp_cfg["id"], missile_launcher_coords=launcher_pos,
                                 radars_coords=radar_pos, position=np.array([0, 0, 0]))
        manager.add_module(ccp)
        by_id[ccp_cfg["id"]] = ccp

    targets = []
    for tc in ae_cfg.get("targets", []) or []:
        pos, vel = np.array(tc["position"]), np.array(tc["velocity"])
        targets.append(Target(manager, tc["id"], pos, Trajectory(velocity=vel, start_pos=pos, start_time=0.0),
                              getattr(TargetType, tc["type"])))
    air_env.add_targets(targets)             # one table append, list order as given (reference main.py:128-147)
    return manager, by_id


def run_simulation_from_config(config_path: str, device=None) -> Manager:
    config = load_config(config_path)
    manager, _ = create_objects_from_config(config, device=device)
    manager.run_simulation(config["simulation"]["duration"])
    logger.info("messages in total: %d", sum(len(v) for v in manager.messages.values()))
    return manager


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    run_simulation_from_config(sys.argv[1] if len(sys.argv) > 1 else "config.yaml")
