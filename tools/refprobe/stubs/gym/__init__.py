"""Import stand-in (probe only): rl/utils/param_noise.py imports gym at module level and never uses it on the paths probed here."""
