#!/usr/bin/env python3
"""Monte-Carlo of the walker pool's scheduling (no GPU): how many lanes a burst / an issue pass gets when any lane may
serve any slot (the queues of biased_pool_kernel) and when lane i owns slots i and i+64 (VERDICT r02 next 1a,
"lane-affine ownership").  The per-visit probabilities are the kernel's own counters on rmat10m_biased
(profiles/r03_pool_phases.txt): 15.27 trials per step, 4 trials per visit, 0.31 candidates per step that the masks cannot
decide (entry gather -> table probe -> 2 % accepted), one gather round trip = issue in iteration k, adopt in k+1."""
import random
import sys

random.seed(1)
SLOTS, LANES, ITERS = 128, 64, 20000
P_FIN = 1.0 / 15.27 * (1 + 0.31)          # a trial ends the visit: accepted, or needs a look
P_TEST = 0.31 / 1.31                      # ... and it was a look
P_TEST_ACCEPT = 0.07                      # a looked-at candidate is accepted (then: entry gather)
MIN = 48                                  # a phase runs when this many want it (or nothing else can)

TRIAL, WANT, FLIGHT = 0, 1, 2


def run(affine):
    state = [TRIAL] * SLOTS
    kind = [0] * SLOTS                    # what the slot's gather is: 0 entry, 1 test, 2 member
    holds = []                            # slots whose gather is in flight (issued last iteration)
    lanes_burst = lanes_issue = bursts = issues = steps = 0
    for _ in range(ITERS):
        # adopt
        for s in holds:
            if kind[s] == 0:
                state[s] = TRIAL
                steps += 1
            elif kind[s] == 1:
                kind[s] = 2
                state[s] = WANT
            else:
                if random.random() < P_TEST_ACCEPT:
                    kind[s] = 0
                    state[s] = WANT
                else:
                    state[s] = TRIAL
        holds = []
        want = [s for s in range(SLOTS) if state[s] == WANT]
        trial = [s for s in range(SLOTS) if state[s] == TRIAL]
        # issue
        if affine:
            per_lane = {}
            for s in want:
                per_lane.setdefault(s % LANES, s)
            cand = list(per_lane.values())
        else:
            cand = want[:LANES]
        trial_lanes = len({s % LANES for s in trial}) if affine else min(LANES, len(trial))
        if len(cand) >= MIN or (cand and trial_lanes < MIN):
            for s in cand:
                state[s] = FLIGHT
            holds = cand
            issues += 1
            lanes_issue += len(cand)
        # burst
        if affine:
            per_lane = {}
            for s in trial:
                per_lane.setdefault(s % LANES, s)
            cand = list(per_lane.values())
        else:
            cand = trial[:LANES]
        want_left = [s for s in range(SLOTS) if state[s] == WANT]
        want_lanes = len({s % LANES for s in want_left}) if affine else min(LANES, len(want_left))
        if len(cand) >= MIN or (cand and want_lanes < MIN):
            bursts += 1
            lanes_burst += len(cand)
            for s in cand:
                for _t in range(4):
                    if random.random() < P_FIN:
                        kind[s] = 1 if random.random() < P_TEST else 0
                        state[s] = WANT
                        break
    return lanes_burst / max(1, bursts), bursts / ITERS, lanes_issue / max(1, issues), issues / ITERS, steps / ITERS


for name, aff in (("queues (any lane, any slot)", False), ("lane-affine (lane i: slots i, i+64)", True)):
    lb, bi, li, ii, st = run(aff)
    print("%-38s burst: %.1f lanes, %.2f per iteration; issue/adopt: %.1f lanes, %.2f per iteration; %.1f steps per iteration" % (name, lb, bi, li, ii, st))
