#!/usr/bin/env python
"""debug: where the fused step's RNG state array first differs from the two-launch step's"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import product_envs
name = sys.argv[1] if len(sys.argv) > 1 else "MarlGrid-3AgentCluttered15x15-v0"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
T = int(sys.argv[3]) if len(sys.argv) > 3 else 80
seeds = 31000 + np.arange(B)
e1 = product_envs.build(name, batch_size=B, seeds=seeds, auto_reset=True, fused_step=True, place_obs=False)
e2 = product_envs.build(name, batch_size=B, seeds=seeds, auto_reset=True, fused_step=False, place_obs=False)
e1.reset(); e2.reset()
rng = np.random.RandomState(4)
n = e1.num_agents
found = 0
for t in range(T):
    a = torch.from_numpy(rng.randint(0, 7, size=(B, n)))
    pos_before = e1.mt_pos.clone()
    o1, r1, d1, _ = e1.step(a)
    o2, r2, d2, _ = e2.step(a)
    same = {k: bool(torch.equal(getattr(e1, k), getattr(e2, k))) for k in ("agent_state", "grid_state", "mt_pos", "mt_head", "step_count_t", "mt_state")}
    if not all(same.values()):
        diff = (e1.mt_state != e2.mt_state)
        envs = torch.nonzero(diff.any(dim=1))[:, 0]
        print("step", t, same, "envs with different mt_state:", envs.numel(), envs[:8].tolist())
        for b in envs[:4].tolist():
            w = torch.nonzero(diff[b])[:, 0].tolist()
            print("  env", b, "words", w[:12], "pos before", int(pos_before[b]), "pos after", int(e1.mt_pos[b]), int(e2.mt_pos[b]),
                  "done", bool(d1[b]), "fused", [hex(int(e1.mt_state[b, i]) & 0xFFFFFFFF) for i in w[:6]],
                  "two", [hex(int(e2.mt_state[b, i]) & 0xFFFFFFFF) for i in w[:6]])
        found += 1
        if found >= 3:
            break
        e2.mt_state.copy_(e1.mt_state)      # go on from a common state
print("done", t, "differences found:", found)
