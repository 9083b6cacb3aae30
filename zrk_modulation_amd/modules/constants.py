"""Message kinds, well-known ids and tuning constants (reference modules/constants.py:1-38).

The enum values are the wire names other modules filter on, so they are kept verbatim."""
from enum import Enum


class MessageType(Enum):
    # launcher <-> missile
    LAUNCH_MISSILE = "launch_missile"
    LAUNCHED_MISSILE = "launched_missile"
    LAUNCH_COMMAND = "launch_command"
    LAUNCH_CANCELLED = "launch_cancelled"
    LAUNCH_SUCCESSFUL = "launch_successful"
    NEW_MISSILE = "new_missile"
    # magazine bookkeeping
    MISSILE_COUNT_REQUEST = "missile_count_request"
    MISSILE_COUNT_RESPONSE = "missile_count_response"
    # radar / air picture
    ALL_OBJECTS = "all_objects"
    FOUND_OBJECTS = "found_objects"
    ACTIVE_OBJECTS = "active_objects"
    CCP_UPDATE_TARGET = "ccp_upd_target"
    UPDATE_TARGET = "upd_target"
    # missile life cycle
    MISSILE_GET_HIT = "missile_get_hit"
    DESTROYED_MISSILE = "destroyed_missile"
    MISSILE_POS = "missile_pos"
    MISSILE_DETONATE = "missile_detonate"
    # GUI replay
    DRAW_OBJECTS = "draw_objects"


SIMULATION_STEP = 1                 # ms; Timer default

MISSILE_VELOCITY_MODULE = 1600      # m/s
MISSILE_DETONATE_PERIOD = 120       # s
MISSILE_DETONATE_RADIUS = MISSILE_VELOCITY_MODULE / 1000 * SIMULATION_STEP   # m

MIN_DIST_DETECTION = 30             # m (unused by the reference as well)
MAX_DIST_DETECTION = 50000          # m
POSSIBLE_TARGET_RADIUS = 100        # track-association gate, in simulation steps

MISSILE_TYPE_DRAWER = 0
TARGET_TYPE_DRAWER = 1

CCP_ID = 0
DRAWER_ID = 1
MANAGER_ID = 2

RADAR_NOISE_SIGMA = 5               # m; `error` in SectorRadar.smooth_objects (modules/Radar.py:139)
