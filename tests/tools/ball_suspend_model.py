#!/usr/bin/env python3
"""Monte-Carlo model of a SUSPENDABLE rejection sampler, to decide whether to build it.

Today every lane that hits something draws its 1 (metal) or 2 (diffuse) points in the unit ball inside ONE wave-level
loop (brt_device.h scatter): the wave iterates until its unluckiest lane is done -- 8.8 iterations per round at 14 of 64
lanes on the headline frame, where the average lane needs 3.3.  A suspendable sampler would leave the loop when at most K
lanes still need a point and let them finish next round beside the new entrants.  But a suspended lane cannot walk in
that next round (its ray is not known yet), so every suspension costs one lane-round of the walk, which is ~3/4 of a
round's instructions.  The model: 63 live lanes, 60 % of the landed rays hit (84 % diffuse), acceptance probability
pi/6, a round = 2350 instruction-equivalents of walk + shading plus 52 per sampler iteration (section profile of the
production kernel).  Result (cost per traced ray segment): K = 2..3 is the optimum at -2 %, K >= 8 LOSES; the verdict's
target of >= 24 lanes per sampler iteration corresponds to K ~ 16: +11 %.  Not built.
"""
import numpy as np


def sim(K, rounds=20000, lanes=63, p_hit=0.6, p_diffuse=0.84, round_cost=2350.0, it_cost=52.0, seed=1):
    rng = np.random.default_rng(seed)
    need = np.zeros(lanes, int)            # outstanding points of the suspended lanes
    cost, useful, iters, sampler_lanes = 0.0, 0, 0, 0
    for _ in range(rounds):
        walking = need == 0
        useful += int(walking.sum())
        hit = walking & (rng.random(lanes) < p_hit)
        need = np.where(hit, np.where(rng.random(lanes) < p_diffuse, 2, 1), need)
        it = 0
        while (need > 0).sum() > K:
            sampler_lanes += int((need > 0).sum())
            need = np.where((need > 0) & (rng.random(lanes) < np.pi / 6), need - 1, need)
            it += 1
        iters += it
        cost += round_cost + it * it_cost
    return cost / useful, iters / rounds, useful / rounds, sampler_lanes / max(1, iters)


if __name__ == "__main__":
    base = None
    for K in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24):
        c, it, u, sl = sim(K)
        base = base or c
        print(f"leave at <= {K:2d} lanes: cost per ray segment {c:6.1f} ({c / base - 1:+6.1%})  sampler iterations / round {it:5.2f} "
              f"at {sl:4.1f} lanes  walking lanes / round {u:5.1f}")
