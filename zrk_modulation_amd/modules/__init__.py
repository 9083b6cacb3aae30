"""Drop-in counterparts of the reference's `modules` package for the hot path.

Same class names, constructor signatures, message types and scheduling contract as the reference
(Ollegorii/ZRK_modulation, modules/*.py), so a scenario is assembled exactly as main.py does.
`AirEnv`, `SectorRadar`, `Target` and `Missile` run their per-tick work on the device through
libzrk_hot.so; `Manager`, `Timer` and the message classes are the (host-side) bus they talk through;
`CombatControlPoint` and `MissileLauncher` are host-side consumers kept behaviour-compatible so the
stock YAML scenarios run end to end.
"""
