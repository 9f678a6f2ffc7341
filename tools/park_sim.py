"""CPU-only design tool (round 4): what ONE OR MORE RAYS PARKED PER LANE IN LDS would do to the lane fill of the walk.

A lane's ray in registers (R) steps box by box; with probability P_STOP per step it stops - at a leaf (Q_LEAF of the stops)
or because it has left the tree.  Today a lane that stops idles until the round ends (leaf test in the merged first step of the
next round, refill when REFILL_MIN lanes are free).  With a parked ray (P) the lane exchanges R and P through LDS when R cannot
step and P can.  The Monte Carlo below follows 64 lanes through rounds of 1 merged + b box steps and reports active lanes per
box wave-step / per leaf phase and wave-steps per ray, for: today's walk; one exchange point in the middle of the round; an
exchange after EVERY step (the upper bound of any per-lane parking scheme) with 1 - 3 parked rays per lane.

Result (P_STOP, Q_LEAF from profiles/r03_c2_walk_stats.txt: 13.1 box steps, 0.89 leaf stops and one exit per ray): box wave-steps
per ray fall by 14 % with one exchange per round and by at most 19 - 22 % with an exchange after every step - rays that stay with
their lane cannot fill a wave; together with the measured cost of a wave-step (half of it does not depend on the active lanes:
profiles/r04_lane_limit_probe.txt) and of an LDS state round trip (profiles/r04_lds_roundtrip_probe.txt) nothing is left.
DESIGN.md section 6, "Re-grouping rays between waves through LDS"; profiles/NOTES.md.
"""
import numpy as np
rng = np.random.default_rng(1)
# per ray: sequence of events; each box lane-step ends with prob p: leaf (q) or exit (1-q)
P_STOP, Q_LEAF = (0.89+1.0)/13.1, 0.89/1.89
def simulate(park, b1, b2, refill_min=24, rounds=4000, nl=64):
    # lane state: R in {0 empty,1 step,2 leaf,3 pending}, P in {0 none,1 step,2 leaf,3 done}
    R = np.zeros(nl, int); Pp = np.zeros(nl, int)
    box_steps = box_lanes = leaf_phases = leaf_lanes = rays = swaps = 0
    def step(mask):
        nonlocal R
        stop = mask & (rng.random(nl) < P_STOP)
        leaf = stop & (rng.random(nl) < Q_LEAF)
        R[leaf] = 2; R[stop & ~leaf] = 3
    for _ in range(rounds):
        # service
        idle = (R == 0) | (R == 3)
        if idle.sum() >= refill_min or idle.all():
            rays += idle.sum(); R[idle] = 1
            if park: Pp[Pp == 3] = 0
        if park:   # boundary swap: R step & P leaf -> test P now
            m = (R == 1) & (Pp == 2); R[m] = 2; Pp[m] = 1; swaps += m.any()
        # merged step
        lm = R == 2; bm = R == 1
        if lm.any(): leaf_phases += 1; leaf_lanes += lm.sum()
        if bm.any(): box_steps += 1; box_lanes += bm.sum()
        R[lm] = 1     # (after the test the ray steps on; ignore shadow hits)
        step(bm)
        for k in range(b1 + b2):
            if park and k == b1:
                # mid swap: R leaf/pending & P step -> exchange ; R leaf/pending & P none -> park, R empty
                m = ((R == 2) | (R == 3)) & (Pp == 1)
                newP = np.where(R == 2, 2, 3)
                Pp[m] = newP[m]; R[m] = 1
                m2 = (R == 2) & (Pp == 0)
                Pp[m2] = 2; R[m2] = 0
                swaps += (m.any() or m2.any())
            bm = R == 1
            if not bm.any(): break
            box_steps += 1; box_lanes += bm.sum()
            step(bm)
    return dict(box_fill=box_lanes/box_steps/nl, leaf_fill=leaf_lanes/max(1,leaf_phases)/nl, box_steps_per_ray=box_steps/rays*1.0,
                leaf_phases_per_ray=leaf_phases/rays, swaps_per_round=swaps/rounds, steps_per_round=box_steps/rounds)
print("today     ", simulate(False, 6, 0))
for b1,b2 in ((3,3),(2,4),(4,3),(3,4),(4,4),(2,2)):
    print("park", b1, b2, simulate(True, b1, b2))

def simulate_every(b, refill_min=24, rounds=4000, nl=64, nparked=1):
    # swap after EVERY step; nparked parked rays per lane (list of states)
    R = np.zeros(nl, int); Pp = np.zeros((nparked, nl), int)
    box_steps = box_lanes = leaf_phases = leaf_lanes = rays = 0
    for _ in range(rounds):
        idle = (R == 0) | (R == 3)
        if idle.sum() >= refill_min or idle.all():
            rays += idle.sum(); R[idle] = 1; Pp[Pp == 3] = 0
        for j in range(nparked):
            m = (R == 1) & (Pp[j] == 2); R[m] = 2; Pp[j][m] = 1
        lm = R == 2; bm = R == 1
        if lm.any(): leaf_phases += 1; leaf_lanes += lm.sum()
        if bm.any(): box_steps += 1; box_lanes += bm.sum()
        R[lm] = 1
        stop = bm & (rng.random(nl) < P_STOP); leaf = stop & (rng.random(nl) < Q_LEAF); R[leaf] = 2; R[stop & ~leaf] = 3
        for k in range(b):
            for j in range(nparked):
                m = ((R == 2) | (R == 3)) & (Pp[j] == 1)
                newP = np.where(R == 2, 2, 3); Pp[j][m] = newP[m]; R[m] = 1
                m2 = (R == 2) & (Pp[j] == 0); Pp[j][m2] = 2; R[m2] = 0
            bm = R == 1
            if not bm.any(): break
            box_steps += 1; box_lanes += bm.sum()
            stop = bm & (rng.random(nl) < P_STOP); leaf = stop & (rng.random(nl) < Q_LEAF); R[leaf] = 2; R[stop & ~leaf] = 3
    return dict(box_fill=round(box_lanes/box_steps/nl,3), leaf_fill=round(leaf_lanes/max(1,leaf_phases)/nl,3), box_steps_per_ray=round(box_steps/rays,3), leaf_phases_per_ray=round(leaf_phases/rays,4))
print("swap every step, 1 parked:", simulate_every(6))
print("swap every step, 2 parked:", simulate_every(6, nparked=2))
print("swap every step, 3 parked:", simulate_every(6, nparked=3))
