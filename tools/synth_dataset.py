"""Writes a Pixel200K-SHAPED synthetic interaction CSV (the real Pixel200K.csv is not available offline):
200 K users, ~96 K items with a Zipf-like popularity tail, 5..40 interactions per user, unique timestamps per user.
usage: python tools/synth_dataset.py <out_dir> [n_users] [n_items]"""
import os
import sys

import numpy as np


def main(out_dir, n_users=200_000, n_items=96_000, seed=2020):
    rng = np.random.default_rng(seed)
    lens = rng.integers(5, 41, size=n_users)
    total = int(lens.sum())
    ranks = np.arange(1, n_items + 1, dtype=np.float64)
    cdf = np.cumsum(1.0 / ranks); cdf /= cdf[-1]
    perm = rng.permutation(n_items)
    items = perm[np.minimum(np.searchsorted(cdf, rng.random(total)), n_items - 1)]
    users = np.repeat(np.arange(n_users), lens)
    ts = rng.permutation(total * 4)[:total]            # unique timestamps
    order = rng.permutation(total)
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "Pixel200K.csv")
    with open(path, "w") as f:
        f.write("item_id,user_id,timestamp\n")
        np.savetxt(f, np.stack([items[order], users[order], ts[order]], 1), fmt=["v%d", "u%d", "%d"], delimiter=",")
    print(f"wrote {path}: {total} interactions, {n_users} users, <= {n_items} items")


if __name__ == "__main__":
    main(sys.argv[1], *(int(x) for x in sys.argv[2:]))
