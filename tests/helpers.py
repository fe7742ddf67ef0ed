"""Shared test helpers."""
import numpy as np


def random_boards(n, seed, with_bar=True):
    """Arbitrary (not necessarily reachable) positions: random checker placement, bars and borne-off counts,
    heavy stacks, late bear-off boards, both sides on the bar."""
    rng = np.random.RandomState(seed)
    st = np.zeros((n, 28), dtype=np.int32)
    for i in range(n):
        kind = rng.randint(4)
        owner = rng.randint(0, 3, 24)                       # 0 empty, 1 P1, 2 P2
        if kind == 1:                                       # both sides (almost) home: bear-off rules
            owner[:] = 0
            owner[18:] = rng.randint(0, 2, 6); owner[:6] = rng.randint(0, 2, 6) * 2
            if rng.rand() < 0.5:
                owner[rng.randint(6, 18)] = rng.randint(1, 3)
        left = [15, 15]
        if with_bar and kind != 1 and rng.rand() < 0.4:
            st[i, 24] = rng.randint(0, 3); left[0] -= st[i, 24]
        if with_bar and kind != 1 and rng.rand() < 0.4:
            st[i, 25] = rng.randint(0, 3); left[1] -= st[i, 25]
        pts = [np.where(owner == 1)[0], np.where(owner == 2)[0]]
        for side in (0, 1):
            if len(pts[side]) == 0:
                continue
            k = rng.randint(1, left[side] + 1)
            cnt = np.bincount(rng.choice(pts[side], k), minlength=24)
            if kind == 2:                                   # one heavy stack
                cnt[:] = 0; cnt[pts[side][0]] = k
            st[i, :24] += cnt * (1 if side == 0 else -1)
            left[side] -= k
        st[i, 26], st[i, 27] = left[0], left[1]             # the rest is borne off
    return st


