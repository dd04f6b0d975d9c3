"""Comparison of a ranked id list with the reference's recorded ranking (tests/golden/irn_*.npz:
top_ids0 / top_vals / top_gaps written by tests/golden/make_golden.py from the unmodified reference).

The reference ranks with torch float32 logits whose accumulation order differs from the fixed chain this
repo defines, so two items whose REFERENCE scores differ by less than `tau` may swap.
The check is order-exact everywhere else: position p must hold the reference's id at p, or -- only if p lies in
a run of reference positions chained by recorded gaps < tau -- one of the ids of that run."""
import numpy as np


def check_ranked(ours, ref_ids, ref_gaps, tau):
    """ours: [k] ids; ref_ids: [K >= k] reference ids in rank order; ref_gaps[j] = s(rank j) - s(rank j+1).
    Returns True when ours == ref_ids[:k] id for id; raises AssertionError on an unexplained difference."""
    ours = np.asarray(ours)
    k = ours.shape[0]
    K = ref_ids.shape[0]
    assert K >= k and ref_gaps.shape[0] >= K - 1
    if np.array_equal(ours, ref_ids[:k]):
        return True
    # runs of near-tied reference positions
    run = np.zeros(K, dtype=np.int64)
    for j in range(1, K):
        run[j] = run[j - 1] if ref_gaps[j - 1] < tau else run[j - 1] + 1
    assert len(set(ours.tolist())) == k, "duplicate ids in a ranked list"
    for p in range(k):
        if ours[p] == ref_ids[p]:
            continue
        members = ref_ids[run == run[p]]
        assert members.shape[0] > 1, f"position {p}: id {ours[p]} != reference {ref_ids[p]} and no near-tie (gap >= {tau})"
        assert ours[p] in members, f"position {p}: id {ours[p]} is not in the near-tied run {members.tolist()}"
        if run[p] == run[K - 1]:
            assert K > k, "near-tied run reaches the end of the recorded list"
    return False
