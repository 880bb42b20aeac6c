"""A model of huff_tail_write's scheduling (jpeggpu_amd/csrc/jg_kernels.hip): roles by ticket, a ready queue per job, writers
that claim from any job's queue. The kernel's argument for "no deadlock" is about an order nothing specifies -- which
workgroups are resident when -- so the model replays it against adversarial schedulers: few slots (down to ONE), resident
workgroups advancing in any order, parts finishing in any order, writers looking and claiming at any time (with the race
between the two). What it checks is what the
GPU tests cannot show by passing: every sequence is written exactly once, never before all the parts that hold its
subsequences are done, and the launch always drains (a writer never waits for a part that has not started).

This is host logic about the protocol, not the product's code path; the GPU tests and the soak run the kernel itself."""
import random


def make_job(rng, seq_len):
    """Parts (runs of whole segments, ~4 sequences each) and sequences of one job: sorted cut points over [0, n)."""
    n = rng.randint(1, 40) * seq_len // rng.randint(1, 3) + rng.randint(1, seq_len)
    cuts = sorted(set([0, n] + [rng.randint(1, n - 1) for _ in range(rng.randint(0, 6))])) if n > 1 else [0, n]
    parts = list(zip(cuts[:-1], cuts[1:]))
    num_seq = (n + seq_len - 1) // seq_len
    return n, parts, num_seq


def run_launch(rng, jobs, seq_len, slots):
    """One launch. A workgroup takes its ticket when it STARTS, so its role follows the start order whichever workgroup of the
    grid the hardware picks next: the model only counts starts."""
    max_parts = max(len(p) for _, p, _ in jobs)
    max_seq = max(s for _, _, s in jobs)
    total_wgs = len(jobs) * (max_parts + max_seq)
    # control words (fuse_init)
    waiting = [[sum(1 for lo, hi in parts if lo <= min((q + 1) * seq_len, n) - 1 and hi > q * seq_len) for q in range(ns)] for n, parts, ns in jobs]
    queue = [[None] * ns for _, _, ns in jobs]
    pushed = [0] * len(jobs)
    claimed = [0] * len(jobs)
    part_done = [[False] * len(p) for _, p, _ in jobs]
    written = [[0] * ns for _, _, ns in jobs]
    ticket = 0
    resident = []  # workgroups that hold a slot: dicts with a role and its state
    next_start = 0
    steps = 0
    while next_start < total_wgs or resident:
        steps += 1
        assert steps < 200000, "the launch does not drain"
        # the dispatcher fills free slots, in ITS order
        while len(resident) < slots and next_start < total_wgs:
            next_start += 1
            role = ticket  # taken at start: atomic counter
            ticket += 1
            num_tail = len(jobs) * max_parts
            if role < num_tail:
                j, p = divmod(role, max_parts)
                if p < len(jobs[j][1]):
                    resident.append({"kind": "part", "job": j, "part": p, "left": rng.randint(1, 6)})
            else:
                r = role - num_tail
                j, q = divmod(r, max_seq)
                if q < jobs[j][2]:  # the static share only says whether there is a sequence for this workgroup
                    resident.append({"kind": "writer", "home": j, "claim": None, "saw": None, "left": rng.randint(1, 4)})
        if not resident:
            continue  # (every workgroup started so far had nothing to do)
        # one resident workgroup makes a step, picked by an adversary
        wg = rng.choice(resident)
        if wg["kind"] == "part":
            wg["left"] -= 1
            if wg["left"] == 0:
                j, p = wg["job"], wg["part"]
                n, parts, ns = jobs[j]
                lo, hi = parts[p]
                part_done[j][p] = True
                for q in range(lo // seq_len, (hi - 1) // seq_len + 1):
                    if q < ns:
                        waiting[j][q] -= 1
                        if waiting[j][q] == 0:
                            i = pushed[j]
                            pushed[j] += 1
                            queue[j][i] = q
                resident.remove(wg)
        else:
            if wg["claim"] is None and wg.get("saw") is None:
                # look at the queues from the home job on: where is something pushed and not yet claimed?
                for k in range(len(jobs)):
                    j = (wg["home"] + k) % len(jobs)
                    if claimed[j] < pushed[j]:
                        wg["saw"] = j
                        break
            elif wg["claim"] is None:
                # ... and claim there, a step later: others may have claimed in between (the kernel's fetch-add behind a load).
                # An index beyond what is pushed will be pushed (it is waited for); beyond the job's sequences: look again.
                j = wg.pop("saw")
                i = claimed[j]
                claimed[j] += 1
                if i < jobs[j][2]:
                    wg["claim"] = (j, i)
            else:
                j, i = wg["claim"]
                q = queue[j][i]
                if q is not None:  # (an entry claimed ahead of its push would be waited for here)
                    wg["left"] -= 1
                    if wg["left"] == 0:
                        n, parts, ns = jobs[j]
                        first, last = q * seq_len, min((q + 1) * seq_len, n) - 1
                        for p, (lo, hi) in enumerate(parts):
                            if lo <= last and hi > first:
                                assert part_done[j][p], "a sequence was written before one of its parts was done"
                        written[j][q] += 1
                        resident.remove(wg)
    for j, (_, _, ns) in enumerate(jobs):
        assert written[j] == [1] * ns, (j, written[j])
        assert pushed[j] == ns


def test_every_sequence_is_written_once_and_the_launch_drains():
    rng = random.Random(20251005)
    for trial in range(150):
        seq_len = rng.choice([3, 8, 255])
        jobs = [make_job(rng, seq_len) for _ in range(rng.randint(1, 5))]
        # from one slot (the hardest case: nothing overlaps) to more slots than workgroups
        for slots in (1, 2, rng.randint(3, 12), 1000):
            run_launch(rng, jobs, seq_len, slots)
