"""Soaks longer than the suite's (tests/test_hip_soak.py: 6 000 steps): the same comparison with the oracle — rewards / done every step, observations (and
the encoding the step's own launch wrote) every 50th, canonical state + RNG every 500th — over 12 000 … 30 000 steps on three scenarios."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import warnings; warnings.simplefilter("ignore")
import test_hip_soak as S
t0 = time.time()
s = S.soak("MarlGrid-3AgentCluttered15x15-v0", 96, 30000, seed0=990000, action_seed=21, encode_in_step=True)
print("headline, 96 envs x 30 000 steps (encode_in_step): episodes per env >= %d, MT wraps >= %.0f, %.0f s" % (s["episodes"].min(), s["wraps"].min(), time.time() - t0))
t0 = time.time()
s = S.soak("Goalcycle-demo-solo-v0", 64, 20000, seed0=5500, action_seed=22, p=[.2, .2, .6])
print("goal cycle solo ('prestige', tile 11: the ready-made gather atlas), 64 envs x 20 000 steps: episodes per env >= %d, %.0f s" % (s["episodes"].min(), time.time() - t0))
t0 = time.time()
s = S.soak("Custom-8AgentCluttered30x30", 24, 12000, seed0=31000, action_seed=23, encode_in_step=True)
print("configs[4]'s scenario, 24 envs x 12 000 steps (encode_in_step): episodes per env >= %d, MT wraps >= %.0f, %.0f s" % (s["episodes"].min(), s["wraps"].min(), time.time() - t0))
