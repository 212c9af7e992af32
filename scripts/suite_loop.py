"""Repeat a few GPU test selections in ONE process (heap-corruption hunt): python scripts/suite_loop.py <reps> <pytest args...>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
import pytest
reps = int(sys.argv[1])
for i in range(reps):
    rc = pytest.main(["-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + sys.argv[2:])
    print("rep", i, "rc", rc, flush=True)
    if rc != 0:
        sys.exit(int(rc))
