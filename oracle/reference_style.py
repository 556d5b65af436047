"""The reference's CPU path, reference-STYLE -- TEST / BASELINE INFRASTRUCTURE ONLY (bench.py's
cpu_baseline leg).  TensorFlow 1.x is not installable here, so the TF1-CPU path is represented by the same structure
on what is available:

  sampler   the pure-Python loop of utils/sampler.py:46-74 (one np.random.randint per draw, set membership tests,
            np.random.permutation + fancy indexing at the end), structure kept, nothing vectorised
  step      what sess.run([train, loss]) executes per batch (BPR.py:31-44 + tf.train.AdamOptimizer on IndexedSlices),
            as vectorised PyTorch-CPU fp32 ops with torch.set_num_threads(n): gathers, row dots, softplus, per-row
            gradient sums (index_add_ = unique + unsorted_segment_sum), the TF1 Adam sweep over EVERY row
  eval      RankingRecommender.test_model_rs (:198-247): score GEMM, np.argsort of every row, the Python seen-filter
            loop and cal_ranking_metrics per user and K
Each piece is timed on a bounded sample; the caller states the thread count."""
import math
import time

import numpy as np
import torch


def python_sampler(ui_train, item_nums, neg_ratio, batch_size):
    """utils/sampler.py:46-74, loop for loop."""
    u_list, i_list, j_list = [], [], []
    for u, items in ui_train.items():
        seen = set(items)
        for i in items:
            random_j = set()
            for _ in range(neg_ratio):
                u_list.append(u)
                i_list.append(i)
                j = np.random.randint(item_nums)
                while j in random_j or j in seen:
                    j = np.random.randint(item_nums)
                random_j.add(j)
                j_list.append(j)
    n = len(u_list)
    s_idx = np.random.permutation(n)
    return math.ceil(n / batch_size), np.array(u_list)[s_idx], np.array(i_list)[s_idx], np.array(j_list)[s_idx]


class TorchBpr:
    """BPR step with TF1 Adam semantics as vectorised torch-CPU fp32 ops (what a TF1 CPU session would run)."""

    def __init__(self, P, Q, lr=1e-3, reg=0.01, beta1=0.9, beta2=0.999, eps=1e-8):
        self.P, self.Q = torch.tensor(np.asarray(P, np.float32)), torch.tensor(np.asarray(Q, np.float32))
        self.mP, self.vP, self.mQ, self.vQ = (torch.zeros_like(t) for t in (self.P, self.P, self.Q, self.Q))
        self.lr, self.reg, self.b1, self.b2, self.eps, self.t = lr, reg, beta1, beta2, eps, 0

    def step(self, u, i, j):
        u, i, j = (torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (u, i, j))
        pu, qi, qj = self.P[u], self.Q[i], self.Q[j]
        x = (pu * qi).sum(1) - (pu * qj).sum(1)
        loss = torch.nn.functional.softplus(-x).sum() + self.reg * 0.5 * ((pu * pu).sum() + (qi * qi).sum() + (qj * qj).sum())
        s = torch.sigmoid(-x)[:, None]
        gP = torch.zeros_like(self.P).index_add_(0, u, -s * (qi - qj) + self.reg * pu)
        gQ = torch.zeros_like(self.Q).index_add_(0, i, -s * pu + self.reg * qi).index_add_(0, j, s * pu + self.reg * qj)
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for var, m, v, g in ((self.P, self.mP, self.vP, gP), (self.Q, self.mQ, self.vQ, gQ)):  # every row, every step
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            var.sub_(lr_t * m / (v.sqrt() + self.eps))
        return float(loss)


def cal_ranking_metrics(real_items, rec_items, K):
    """utils/metrics.py:9-19"""
    hit, mrr, dcg, idcg = 0, 0, 0, 0
    rec_items = list(rec_items)
    for id, item in enumerate(real_items):
        if item in rec_items:
            hit += 1
            idx = rec_items.index(item)
            mrr += 1 / (idx + 1)
            dcg += 1 / math.log2(idx + 2)
        idcg += 1 / math.log2(id + 2)
    return hit / min(K, len(real_items)), mrr, dcg / idcg


def eval_loop(P, Q, test_users, ui_train, ui_test, topk, batch_size_t):
    """RankingRecommender.test_model_rs (:198-247): GEMM, full argsort, Python filter + metric loops."""
    P, Q = torch.as_tensor(P), torch.as_tensor(Q)
    ndcg = {k: [] for k in range(len(topk))}
    for t0 in range(0, len(test_users), batch_size_t):
        cur = test_users[t0:t0 + batch_size_t]
        scores = (P[torch.as_tensor(np.asarray(cur), dtype=torch.long)] @ Q.T).numpy()
        args = np.argsort(-scores, axis=1)
        for row, u in enumerate(cur):
            seen = set(ui_train.get(u, ()))
            items, k = [], 0
            while len(items) < topk[-1] and k < args.shape[1]:
                if args[row, k] not in seen:
                    items.append(int(args[row, k]))
                k += 1
            for kid, K in enumerate(topk):
                ndcg[kid].append(cal_ranking_metrics(ui_test[u], items[:K], K)[2])
    return ndcg


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def measure(data, P0, Q0, u, i, j, B, neg_ratio, threads, budget_s=4.0):
    """-> dict for bench.py: sampler triples/s (1 thread by nature), step triples/s at every n of `threads`, eval users/s."""
    out = {"cpu_model": cpu_model()}
    # sampler: a bounded prefix of the users
    keys = list(data.ui_train.keys())
    part = {k: data.ui_train[k] for k in keys[:max(1, len(keys) // 8)]}
    np.random.seed(1)
    t0 = time.perf_counter()
    _, su, _, _ = python_sampler(part, data.item_nums, neg_ratio, B)
    ts = time.perf_counter() - t0
    out["sampler"] = {"triples_per_s": len(su) / ts, "sample": "%d users, %d triples, %.1f s, 1 thread (a serial Python loop)" % (len(part), len(su), ts)}
    nbe = len(u) // B
    old = torch.get_num_threads()
    out["step"] = {}
    for n in threads:
        torch.set_num_threads(n)
        m = TorchBpr(P0, Q0)
        t0 = time.perf_counter()
        m.step(u[:B], i[:B], j[:B])
        first = time.perf_counter() - t0
        if first > budget_s / 2:  # e.g. 256 threads on 6144-row operands: seconds per step; one more step is the sample
            t0 = time.perf_counter()
            m.step(u[B:2 * B], i[B:2 * B], j[B:2 * B])
            dt = time.perf_counter() - t0
            out["step"]["n%d" % n] = {"threads": n, "triples_per_s": B / dt, "us_per_step": dt * 1e6, "steps": 1,
                                      "note": "oversubscribed: the first step took %.1f s" % first}
            continue
        k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            b = k % nbe
            m.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], j[b * B:(b + 1) * B])
            k += 1
        dt = time.perf_counter() - t0
        step_rate = k * B / dt
        out["step"]["n%d" % n] = {"threads": n, "triples_per_s": step_rate, "us_per_step": dt / k * 1e6, "steps": k,
                                  "epoch_inclusive_triples_per_s": 1.0 / (1.0 / step_rate + 1.0 / out["sampler"]["triples_per_s"])}
    # the multi-core baseline is the BEST thread count found, not the largest (64 threads on 6144-row operands are slower than one)
    best = max(out["step"].values(), key=lambda r: r["triples_per_s"])
    out["step"]["best"] = {"threads": best["threads"], "triples_per_s": best["triples_per_s"],
                           "note": "the fastest of %s threads: the multi-core CPU baseline of the step" % sorted(r["threads"] for r in out["step"].values())}
    # eval: a bounded prefix of the test users, all cores
    torch.set_num_threads(max(threads))
    users = [int(x) for x in data.test_users[:512]]
    t0 = time.perf_counter()
    eval_loop(P0, Q0, users, data.ui_train, data.ui_test, [10, 20], 1024)
    te = time.perf_counter() - t0
    out["eval"] = {"users_per_s": len(users) / te, "sample": "%d test users x %d items, K = 10, 20, %.1f s, %d threads" % (len(users), data.item_nums, te, max(threads))}
    torch.set_num_threads(old)
    return out
