"""Regenerates the `self_consistency` block of prng_kat.json from oracle/prng.py (the public KATs in the
same file are literature values, typed in from SURVEY.md 8c, not generated)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import prng  # noqa: E402

path = os.path.join(os.path.dirname(__file__), "prng_kat.json")
g = json.load(open(path))
g["self_consistency"] = {
    "randint_split42_0_9": prng.randint(prng.split(prng.prng_key(42))[0], (2,), 0, 9).tolist(),
    "choice_key3_10_2": prng.choice_no_replace(prng.prng_key(3), 10, 2).tolist(),
    "normal_key1_5": [float(x) for x in prng.normal(prng.prng_key(1), (5,))],
}
json.dump(g, open(path, "w"), indent=1)
