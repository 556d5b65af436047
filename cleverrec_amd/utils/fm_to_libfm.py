"""Ratings file -> the comma-separated libfm-ish text RatingPreprocess reads (label,id:val,id:val).

Host-side data tool, the writer half of the FM data path (reference utils/fm_to_libfm.py:15-96, a script
with its paths as module constants; here a function with the same steps plus a small command line).
Same output, byte for byte, for the same input file and the same numpy global-RNG state:

  * `read_csv(header=0, names=...)`: the first line of the file is taken as a header and dropped
    (fm_to_libfm.py:20), as the ranking loader does;
  * user / item ids -> 1-based dense ids in the iteration order of the `set` of unique raw ids
    (fm_to_libfm.py:24-28);
  * rows are emitted in FILE order: the reference sorts the frame by (u_id, time) but then addresses rows
    by index label 0..n-1 (fm_to_libfm.py:30, 49), which undoes the sort;
  * features per row: `u:1`, `(user_nums+i):1` (or the bare ids when is_real_valued is False)
    (fm_to_libfm.py:52-76);
  * `train_test_split(rows, test_size=0.2)` on numpy's global RNG (fm_to_libfm.py:89), written with
    `DataFrame.to_csv(index=0, header=None)` to <dataset>.train.libfm / <dataset>.test.libfm (:92-94).

The reference builds the rows in a 32-process pool of Python loops; the rows here are built column-wise
in one pass (the split only needs the row count, so it is drawn on row numbers: same RNG consumption).
"""
import argparse
import os

import numpy as np
import pandas as pd
from sklearn.model_selection import train_test_split

from .tools import re_index


def fm_to_libfm(data_dir, dataset, file_name, sep='::', is_real_valued=True, test_size=0.2, out_dir=None,
                with_time=True):
    """Writes <out_dir or data_dir/dataset>/<dataset>.{train,test}.libfm; returns (user_nums, item_nums,
    n_train, n_test).  with_time=False takes user,item,rating files (dataset/Ciao/ratings.csv): the time column
    never reaches the output."""
    src = os.path.join(data_dir, dataset, file_name)
    names = ['u_id', 'i_id', 'rating', 'time'][:4 if with_time else 3]
    ratings = pd.read_csv(src, sep=sep, header=0, names=names, usecols=list(range(len(names))),
                          engine='python' if len(sep) > 1 else None)
    if with_time:
        ratings.time = ratings.time.astype(int)

    user_set, item_set = set(ratings.u_id.unique()), set(ratings.i_id.unique())
    user_nums, item_nums = len(user_set), len(item_set)
    user_map, item_map = re_index(user_set), re_index(item_set)
    u = ratings.u_id.map(user_map).to_numpy() + 1          # ids start from 1
    i = ratings.i_id.map(item_map).to_numpy() + 1 + user_nums
    if is_real_valued:
        cols = {0: ratings.rating.to_numpy(), 1: np.char.add(u.astype(str), ':1'), 2: np.char.add(i.astype(str), ':1')}
    else:
        cols = {0: ratings.rating.to_numpy(), 1: u, 2: i}
    rows = pd.DataFrame(cols)

    train_rows, test_rows = train_test_split(np.arange(len(rows)), test_size=test_size)
    out_dir = out_dir or os.path.join(data_dir, dataset)
    os.makedirs(out_dir, exist_ok=True)
    rows.iloc[train_rows].to_csv(os.path.join(out_dir, dataset + '.train.libfm'), index=0, header=None)
    rows.iloc[test_rows].to_csv(os.path.join(out_dir, dataset + '.test.libfm'), index=0, header=None)
    return user_nums, item_nums, len(train_rows), len(test_rows)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--data-dir', default='dataset')
    ap.add_argument('--dataset', default='ml-1m')
    ap.add_argument('--file', default='ratings.dat')
    ap.add_argument('--sep', default='::')
    ap.add_argument('--no-time', action='store_true', help='three-column input (user, item, rating)')
    ap.add_argument('--out-dir', default=None)
    ap.add_argument('--ids-only', action='store_true', help='is_real_valued=False: bare feature ids')
    ap.add_argument('--seed', type=int, default=None, help='seed numpy\'s global RNG first (the reference never does)')
    a = ap.parse_args(argv)
    if a.seed is not None:
        np.random.seed(a.seed)
    un, inum, ntr, nt = fm_to_libfm(a.data_dir, a.dataset, a.file, a.sep, not a.ids_only, out_dir=a.out_dir,
                                    with_time=not a.no_time)
    print('users %d, items %d, train rows %d, test rows %d' % (un, inum, ntr, nt))


if __name__ == '__main__':
    main()
