"""Synthetic scenarios of the benchmark configurations as plain arrays.

Seeded with numpy.random.Generator(PCG64(seed)); nothing here computes on the hot path."""
from __future__ import annotations

import numpy as np

WORKLOADS = {
    # name: (targets, radars, missiles)       BASELINE.json configs[1], configs[2], configs[3]
    "C2": (100_000, 4, 1_000),
    "C3": (1_000_000, 16, 10_000),
    "C4": (10_000_000, 16, 10_000),           # ONE population of 1e7, cut into contiguous shards (strong scaling)
    "C3x4": (4_000_000, 16, 10_000),          # C3's scene at four times the rows: the per-tick footprint no longer fits
                                              # the 256 MiB Infinity Cache, so the sweep's bytes really come from HBM
    "tiny": (4_096, 4, 64),
    "tiny4": (16_384, 4, 64),                 # C4's mechanics at test size
}
# Monte-Carlo ensemble, BASELINE.json configs[4]: (scenarios per GPU, targets, radars, missiles) per scenario
ENSEMBLES = {
    "C5": (128, 10_000, 4, 100),
    "tiny5": (6, 1_000, 3, 10),
}
SEEDS = {"C2": 1236, "C3": 1237, "C3x4": 1237, "C4": 1238, "C5": 1239, "tiny": 1234, "tiny4": 1235, "tiny5": 1233}
STRONG = ("C4", "tiny4")
CHUNK = 62_500                                # rows per independently seeded chunk of a sharded population


def synthetic_targets(n, seed, first_id=1000):
    g = np.random.Generator(np.random.PCG64(seed))
    sp = np.empty((n, 3))
    sp[:, 0] = g.uniform(-60e3, 60e3, n)
    sp[:, 1] = g.uniform(-60e3, 60e3, n)
    sp[:, 2] = g.uniform(100.0, 12e3, n)
    vel = g.normal(0.0, 150.0, (n, 3))
    ids = first_id + np.arange(n, dtype=np.int64)
    return ids, sp, vel, np.zeros(n)


def population_slice(seed, lo, hi, first_id=1000, chunk=CHUNK):
    """Rows [lo, hi) of the one population `seed` defines, whoever asks and however it is cut: the population is
    generated in chunks of `chunk` rows, chunk c from PCG64([seed, c]), so a rank builds its shard without generating
    anybody else's."""
    parts = []
    for c in range(lo // chunk, (hi + chunk - 1) // chunk):
        g = np.random.Generator(np.random.PCG64([seed, c]))
        sp = np.empty((chunk, 3))
        sp[:, 0] = g.uniform(-60e3, 60e3, chunk)
        sp[:, 1] = g.uniform(-60e3, 60e3, chunk)
        sp[:, 2] = g.uniform(100.0, 12e3, chunk)
        vel = g.normal(0.0, 150.0, (chunk, 3))
        a, b = max(lo, c * chunk) - c * chunk, min(hi, (c + 1) * chunk) - c * chunk
        parts.append((sp[a:b], vel[a:b]))
    sp = np.concatenate([p[0] for p in parts]) if parts else np.zeros((0, 3))
    vel = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0, 3))
    ids = first_id + np.arange(lo, hi, dtype=np.int64)
    return ids, sp, vel, np.zeros(hi - lo)


def shard_bounds(n, world, rank):
    """Contiguous index range of rank `rank`: [g*n/G, (g+1)*n/G)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def synthetic_radars(R):
    return [dict(id=10_000 + r, position=[1000.0 * r, 0.0, 0.0], max_distance=50e3, azimuth_start=0.0,
                 elevation_start=0.0, azimuth_range=90.0, elevation_range=45.0, azimuth_speed=10.0,
                 elevation_speed=0.0, scan_mode="horizontal") for r in range(R)]


def missile_targets(n, m):
    """Missile k is launched against target k * floor(n / m)."""
    if m <= 0:
        return np.zeros(0, np.int32)
    return (np.arange(m, dtype=np.int64) * (n // m)).astype(np.int32)


def synthetic_config(n, R, seed, launchers=1, missiles_per_launcher=0, time_step=10, duration=1000, first_id=1000):
    """The same synthetic scene as a configuration in the schema the reference's GUI saves and its main.py loads
    (UI/PolygonEditor.py:522-573, main.py:35-149): feed it to either side's `create_objects_from_config`.
    Key order is the GUI's (`yaml.dump(..., sort_keys=False)`)."""
    ids, sp, vel, _ = synthetic_targets(n, seed, first_id)
    radars = synthetic_radars(R)
    launcher_ids = [20_000 + k for k in range(launchers)]
    return {
        "simulation": {"time_step": int(time_step), "duration": int(duration)},
        "air_environment": {
            "id": 1, "position": [0.0, 0.0, 0.0],
            "targets": [{"id": int(ids[k]), "type": "AIR_PLANE", "position": [float(v) for v in sp[k]],
                         "velocity": [float(v) for v in vel[k]]} for k in range(n)],
        },
        "combat_control_point": {"id": 2, "missile_launcher_ids": launcher_ids, "radar_ids": [r["id"] for r in radars]},
        "missile_launchers": [{
            "id": lid, "position": [0.0, 0.0, 0.0], "max_missiles": int(missiles_per_launcher),
            "missiles": [{"id": 30_000 + 1000 * k + j, "position": [0.0, 0.0, 0.0], "velocity": 1000,
                          "explosion_radius": 150, "life_time": 60} for j in range(missiles_per_launcher)],
        } for k, lid in enumerate(launcher_ids)],
        "radars": [{key: r[key] for key in ("id", "position", "azimuth_start", "elevation_start", "max_distance",
                                            "azimuth_range", "elevation_range", "azimuth_speed", "elevation_speed",
                                            "scan_mode")} for r in radars],
    }
