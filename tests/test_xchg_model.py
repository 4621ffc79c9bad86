"""A pure-Python model of the one-shot gradient exchange's inbox protocol (csrc/reduce.hip: xchg_publish / xchg_gather,
hcatgnet_amd/xgmi.py) for world 8 over six steps -- the configuration the driver's scaling tier runs and no one-GPU box can.

What the kernel relies on, stated as the model's rules:
  * inbox of rank p = 2 parities x world writers x n_ext granules {value, stamp}; slot = xgmi.inbox_slot(...);
  * step s (1-based, the optimiser's device counter) uses parity s & 1 and stamps its granules with s;
  * a rank's step-(s + 1) launch starts only after its step-s launch has completed (stream order): every element of
    step s has been published AND gathered;
  * inside a launch, elements are published / gathered in any order, ranks run at any relative speed.
Checked under random schedules: every gather returns exactly the value its writer published for that step (never a stale
or a future granule), the total is the rank-ordered sum on every rank, a publish never overwrites a granule its owner has
not read yet (double buffering suffices), and every rank finishes (no deadlock)."""
import random

import pytest

from hcatgnet_amd.xgmi import inbox_slot

WORLD, STEPS, N_EXT = 8, 6, 5


def _value(rank, step, elem):
    return 1000.0 * step + 10.0 * rank + elem


@pytest.mark.parametrize("seed", range(8))
def test_inbox_protocol_world8_six_steps(seed):
    rng = random.Random(seed)
    inbox = [[(0.0, 0)] * (2 * WORLD * N_EXT) for _ in range(WORLD)]          # stamps 0 = the zero-filled allocation
    consumed = [set() for _ in range(WORLD)]                                   # (owner) -> {(step, writer, elem)} already read
    # per rank: current step, elements still to publish, elements still to gather, per-element running sums
    step = [1] * WORLD
    to_pub = [set(range(N_EXT)) for _ in range(WORLD)]
    to_get = [set(range(N_EXT)) for _ in range(WORLD)]
    totals = [dict() for _ in range(WORLD)]
    done = [False] * WORLD
    guard = 0
    while not all(done):
        guard += 1
        assert guard < 200000, "no progress: the protocol dead-locked"
        # enabled actions: a publish is always enabled; a gather of elem e once e was published by this rank (the thread
        # publishes, then polls) and every peer's granule of this step has arrived
        acts = []
        for r in range(WORLD):
            if done[r]:
                continue
            s, par = step[r], step[r] & 1
            for e in to_pub[r]:
                acts.append(("pub", r, e))
            for e in to_get[r] - to_pub[r]:
                if all(q == r or inbox[r][inbox_slot(par, WORLD, q, N_EXT, e)][1] == s for q in range(WORLD)):
                    acts.append(("get", r, e))
        assert acts, "no enabled action although ranks are unfinished: dead-lock"
        kind, r, e = rng.choice(acts)
        s, par = step[r], step[r] & 1
        if kind == "pub":
            for p in range(WORLD):
                if p == r:
                    continue
                slot = inbox_slot(par, WORLD, r, N_EXT, e)
                old_val, old_stamp = inbox[p][slot]
                # double buffering: what sits here is from step s - 2 (or the initial zero) and its owner has read it
                assert old_stamp in (0, s - 2), (old_stamp, s)
                if old_stamp == s - 2 and s - 2 >= 1:
                    assert (s - 2, r, e) in consumed[p], "a granule was overwritten before its owner read it"
                inbox[p][slot] = (_value(r, s, e), s)
            to_pub[r].discard(e)
        else:
            tot = 0.0
            for q in range(WORLD):                                           # rank order: the same sum on every rank
                if q == r:
                    v = _value(r, s, e)
                else:
                    v, stamp = inbox[r][inbox_slot(par, WORLD, q, N_EXT, e)]
                    assert stamp == s and v == _value(q, s, e)
                    consumed[r].add((s, q, e))
                tot += v
            totals[r][(s, e)] = tot
            to_get[r].discard(e)
        if not to_pub[r] and not to_get[r]:                                    # launch complete: the next step may start
            if step[r] == STEPS:
                done[r] = True
            else:
                step[r] += 1
                to_pub[r], to_get[r] = set(range(N_EXT)), set(range(N_EXT))
    want = {(s, e): sum(_value(q, s, e) for q in range(WORLD)) for s in range(1, STEPS + 1) for e in range(N_EXT)}
    for r in range(WORLD):
        assert totals[r] == want


def test_slots_of_one_inbox_never_collide():
    seen = set()
    for par in range(2):
        for w in range(WORLD):
            for e in range(N_EXT):
                seen.add(inbox_slot(par, WORLD, w, N_EXT, e))
    assert len(seen) == 2 * WORLD * N_EXT and max(seen) == 2 * WORLD * N_EXT - 1
