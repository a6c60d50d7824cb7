#!/usr/bin/env python3
"""Writes tests/golden/binom_pins.json: scipy.stats.binom.cdf(k, n, p) bit patterns (float.hex) on a grid of the small-integer
arguments the pair test sees (phaser/phaser.py:1649), evaluated by the scipy of the build container -- the one that also ran the
reference for every other golden file.  tests/test_engine_host.py compares engine.binom_cdf_dedup with them bit for bit, so a
different scipy on another box shows up as a failing CPU test rather than as a last-digit difference in a p-value column."""
import json, os, sys
import numpy as np
import scipy
from scipy.stats import binom
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "binom_pins.json")
rng = np.random.default_rng(7)
pins = []
for p in (0.0005, 0.001, 0.0025, 0.01, 0.05):
    ns = list(range(1, 41)) + [50, 64, 100, 127, 255, 500, 1000, 4000]
    for n in ns:
        ks = sorted(set([0, 1, 2, n // 2, n - 1, n] + rng.integers(0, n + 1, 4).tolist()))
        for k in ks:
            if 0 <= k <= n:
                pins.append([int(k), int(n), p, float(binom.cdf(k, n, p)).hex()])
json.dump({"scipy": scipy.__version__, "numpy": np.__version__, "pins": pins}, open(out, "w"))
print(len(pins), "pins ->", out)
