"""Phase times of bpr_index_batches' workgroups 0 (user side) and 1 (item side) in a 20-batch tf1 call (debug build with
-DCREC_STAMPS: STAMP_TOOL=tools/idx_stamps.py tools/wave_stamps.sh).  100 MHz wall clock: 0.01 us resolution."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic

sess = runtime.Session(0)
data = synthetic("S-ml1m", seed=20240601)
U, I, B, d, K = data.user_nums, data.item_nums, 6144, 64, 20
g = np.random.Generator(np.random.PCG64(0))
P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
sess.ctx.seed(1)
u, i, j = sess.ctx.sample_pairwise(data.train_csr, I, 4)
du, di, dj = sess.to_dev(u), sess.to_dev(i), sess.to_dev(j)
st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
n = K * B
names = ["load", "sort", "run starts", "heavy rows", "wide groups", "rowseg + ranks", "records"]
acc = np.zeros((4, 6))
reps = 50
for r in range(reps + 10):
    st.train_epoch(du[:n], di[:n], dj[:n], B)
    torch.cuda.synchronize()
    if r < 10:
        continue
    buf = (ctypes.c_ulonglong * (8 * 16384))()
    assert sess.lib.crec_debug_stamps(buf, ctypes.c_size_t(8 * 16384)) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(16384, 8)[16000:16028, 0].astype(np.int64).reshape(7, 4)
    acc += np.diff(a, axis=0).T / 100.0
# (a 20-batch call has two workgroups per (batch, side): the lists' and the records')
for wg, nm in enumerate(("user side, lists  ", "user side, records", "item side, lists  ", "item side, records")):
    print(nm, " ".join("%s %.2f" % (names[k + 1], acc[wg, k] / reps) for k in range(6)), "| total %.2f us" % (acc[wg].sum() / reps))
