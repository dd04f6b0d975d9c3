"""Record of the id-for-id parity counts the -m gpu tests observe (test infrastructure).

Every test that counts "users identical to the reference id for id" calls record(): the observed count and the
near-tie users go into gpurun_out/parity_counts.json (merged back by gpurun; the file judged is the copy committed
under profiles/rNN/).  The tests assert the counts EXACTLY against the tables below (tests/test_oracle_golden.py:18 does
the same for the CPU oracle); IRS_RECORD_PARITY=1 turns the exact assertion into a recording run (how the tables were
filled the first time a kernel changed)."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out", "parity_counts.json")
RECORD_ONLY = os.environ.get("IRS_RECORD_PARITY", "0") == "1"


def record(key, strict, users, near_tie_users, extra=None):
    try:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        data = {}
        if os.path.exists(OUT):
            with open(OUT) as fh:
                data = json.load(fh)
        data[key] = {"identical_users": int(strict), "users": int(users), "near_tie_users": sorted(int(u) for u in near_tie_users)}
        if extra:
            data[key].update(extra)
        with open(OUT, "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)
    except OSError:
        pass  # a read-only tree must not fail a parity test


def check_exact(key, near_tie_users, expected, users):
    """The observed near-tie user set must be exactly the recorded one (so the identical count is users - len(expected))."""
    got = sorted(int(u) for u in near_tie_users)
    record(key, users - len(got), users, got)
    if RECORD_ONLY:
        return
    exp = sorted(expected)
    assert got == exp, (f"{key}: users whose ranked ids differ from the reference's inside a near-tie run: {got}, recorded {exp} "
                        f"({users - len(got)} of {users} identical id for id)")
