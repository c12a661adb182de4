#!/usr/bin/env python
"""Secondary measurements on one MI355X: the other BASELINE.json configs' networks through the
same C ABI.  bench.py stays the headline (C2, API defaults); this prints one JSON line per case:

    train   : rows/s and the fp32-MFMA fraction   (FLOP/sample = 6P - 2*D*h1)
    generate: rows/s, MFMA fraction (2*P_dec FLOP/row) and HBM fraction (4*(Z+D) B/row)
    balance : smrt_balance wall-clock end to end (host arrays in, host arrays out)

    python tools/bench_configs.py [--cases train_n1_b256,gen_n5,...]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK_TF, PEAK_GBS = 157.3, 8000.0
NETS = {"N1": (784, [450], 10), "N1p": (784, [900], 15), "N2": (64, [128], 16), "N5": (256, [1024] * 4, 64)}


def dims(D, hidden, Z):
    w = [D] + hidden
    enc = list(zip(w[:-1], w[1:]))
    heads = [(w[-1], Z), (w[-1], Z)]
    dec = [(Z, w[-1])] + [(b, a) for a, b in reversed(enc)]
    return enc, heads, dec


def flops(net):
    D, hidden, Z = NETS[net]
    enc, heads, dec = dims(D, hidden, Z)
    P = sum(a * b for a, b in enc + heads + dec)
    return 6 * P - 2 * D * hidden[0], 2 * sum(a * b for a, b in dec)


def synth(n, D, seed=0):
    import torch
    from smrt_amd import _hip
    g = torch.Generator(device="cuda").manual_seed(seed)
    ld = _hip.row_stride(D)
    X = torch.zeros((n, ld), dtype=torch.float32, device="cuda")
    X[:, :D] = torch.rand((n, D), generator=g, device="cuda")
    torch.cuda.synchronize()
    return X, ld


def handle(net, opt, batch):
    from smrt_amd import _hip
    from smrt_amd.autoencode.topography import VAETopography
    D, hidden, Z = NETS[net]
    h = _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation="sigmoid", optimizer=opt, learning_rate=0.001,
                       l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=42,
                       max_batch=batch, device=0)
    topo = VAETopography(n_hidden=hidden, input_shape=D, activation="sigmoid", n_latent_factors=Z, random_state=42)
    h.set_params(VAETopography.join_flat(topo.initial_parameters()))
    return h


def case_train(name, net, rows, batch, opt, epochs):
    D, hidden, Z = NETS[net]
    X = synth(rows, D)
    h = handle(net, opt, min(batch, rows))
    h.train_epochs(X, batch, 2)
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    costs, ms = h.train_epochs(X, batch, epochs)
    dt = time.perf_counter() - t0
    prof = h.profile_epochs(X, batch, 1)
    h.close()
    f_train, _ = flops(net)
    sps = rows * epochs / dt
    kern = {k: {"ms_per_epoch": round(v[0], 4), "launches": v[1]} for k, v in prof.items() if v[1]}
    return {"case": name, "net": net, "rows": rows, "batch": batch, "optimizer": opt, "epochs": epochs,
            "samples_per_s": sps, "ms_per_epoch_wall": 1e3 * dt / epochs, "ms_per_epoch_device": float(np.mean(ms)),
            "tflops": sps * f_train / 1e12, "mfma_frac": sps * f_train / 1e12 / PEAK_TF,
            "final_cost": float(costs[-1]), "kernels": kern}


def case_gen(name, net, n, reps=5):
    import torch
    D, hidden, Z = NETS[net]
    h = handle(net, "adam", 128)
    out = h.generate(n, to_host=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        out = h.generate(n, philox_offset=i * n * Z, to_host=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    h.close()
    _, f_dec = flops(net)
    rps = n / dt
    return {"case": name, "net": net, "rows": n, "rows_per_s": rps, "ms": 1e3 * dt,
            "tflops": rps * f_dec / 1e12, "mfma_frac": rps * f_dec / 1e12 / PEAK_TF,
            "gbs": rps * 4 * (Z + D) / 1e9, "hbm_frac": rps * 4 * (Z + D) / 1e9 / PEAK_GBS,
            "mean_out": float(out.mean().item())}


def mnist_shaped(n_major, n_minor, seed=42, D=784):
    """MNIST-shaped stand-in: per-class prototype, ~80 % exact zeros."""
    rs = np.random.RandomState(seed)
    Xs, ys = [], []
    for label, n in ((0, n_major), (1, n_minor)):
        proto, mask = rs.uniform(size=D), rs.uniform(size=D) < 0.2
        Xs.append(np.clip(proto + 0.15 * rs.normal(size=(n, D)), 0, 1) * mask)
        ys.append(np.full(n, label))
    X, y = np.vstack(Xs), np.concatenate(ys)
    p = rs.permutation(len(y))
    return X[p], y[p]


def tabular(n, D, n_minor, seed=42):
    rs = np.random.RandomState(seed)
    y = np.zeros(n, dtype=np.int64)
    y[rs.choice(n, n_minor, replace=False)] = 1
    mu = rs.normal(size=(2, D))
    X = rs.normal(size=(n, D)).astype(np.float32)
    X += mu[y].astype(np.float32)
    X = 1.0 / (1.0 + np.exp(-X))
    return X, y


def case_balance(name, which, reps=3):
    from smrt_amd.balance import smrt_balance
    if which == "C1":
        X, y = mnist_shaped(5444, 500)
        kw = dict(n_hidden=450, n_latent_factors=10, learning_rate=0.05, n_epochs=50, batch_size=256, random_state=42,
                  shuffle=False, balance_ratio=0.5)
    else:
        X, y = tabular(1000000, 64, 19608)
        kw = dict(n_hidden=128, n_latent_factors=16, balance_ratio=0.2, random_state=42)
    smrt_balance(X[:2000], y[:2000] if which == "C1" else (np.arange(2000) % 50 == 0).astype(int), **kw)   # warm-up
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        Xb, yb = smrt_balance(X, y, **kw)
        times.append(time.perf_counter() - t0)
    return {"case": name, "config": which, "in_rows": int(len(y)), "out_rows": int(len(yb)),
            "minority_out": int((yb == 1).sum()), "wall_s_median": float(np.median(times)), "wall_s_all": times,
            "out_dtype": str(Xb.dtype)}


def case_balance_c3(name, threads):
    """C3 on ONE GPU: 8 minority classes x 5 000 rows + 50 000 majority, 784-d, N1, API defaults (batch 128, 20 epochs):
    the eight per-class jobs side by side (SMRT_CLASS_THREADS) or one after the other."""
    from smrt_amd.balance import smrt_balance
    rs = np.random.RandomState(42)
    Xs, ys = [], []
    for label, n in [(0, 50000)] + [(k, 5000) for k in range(1, 9)]:
        proto, mask = rs.uniform(size=784), rs.uniform(size=784) < 0.2
        Xs.append((np.clip(proto + 0.15 * rs.normal(size=(n, 784)), 0, 1) * mask).astype(np.float32))
        ys.append(np.full(n, label))
    X, y = np.vstack(Xs), np.concatenate(ys)
    p = rs.permutation(len(y))
    X, y = X[p], y[p]
    os.environ["SMRT_CLASS_THREADS"] = str(threads)
    kw = dict(n_hidden=450, n_latent_factors=10, balance_ratio=0.5, random_state=42)
    smrt_balance(X[:9000], y[:9000], n_epochs=1, **kw)
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        Xb, yb = smrt_balance(X, y, **kw)
        times.append(time.perf_counter() - t0)
    return {"case": name, "config": "C3 on one GPU", "class_threads": threads, "in_rows": int(len(y)), "out_rows": int(len(yb)),
            "wall_s_median": float(np.median(times)), "wall_s_all": times}


def case_smote(name, which):
    """smote_balance end to end on the C1 / C2 inputs, next to sklearn's kd-tree search alone on the host (the
    dominant cost of the reference's smote_balance, smote.py:63-69) and the kernel's own time and fp64 rate."""
    import torch
    from sklearn.neighbors import NearestNeighbors
    from smrt_amd import _hip
    from smrt_amd.balance import smote_balance
    if which == "C1":
        X, y = mnist_shaped(5444, 500)
        kw = dict(balance_ratio=0.5, n_neighbors=10, random_state=42, shuffle=False)
    else:
        X, y = tabular(1000000, 64, 19608)
        X = X.astype(np.float64)
        kw = dict(balance_ratio=0.2, n_neighbors=5, random_state=42)
    smote_balance(X[:3000], (np.arange(3000) % 30 == 0).astype(int), **kw)                  # warm-up
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        Xb, yb = smote_balance(X, y, **kw)
        times.append(time.perf_counter() - t0)
    X_sub = np.ascontiguousarray(X[y == 1], dtype=np.float64)
    k = kw["n_neighbors"] + 1
    Xd = torch.from_numpy(X_sub).cuda()
    _hip.knn_f64(Xd, k, to_host=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        idx = _hip.knn_f64(Xd, k, to_host=False)
    torch.cuda.synchronize()
    knn_s = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    ref = NearestNeighbors(n_neighbors=k, algorithm="kd_tree").fit(X_sub).kneighbors(X_sub, return_distance=False)
    cpu_s = time.perf_counter() - t0
    n, d = X_sub.shape
    return {"case": name, "config": which, "in_rows": int(len(y)), "out_rows": int(len(yb)), "minority_rows": int(n),
            "wall_s_median": float(np.median(times)), "wall_s_all": times, "knn_kernel_ms": 1e3 * knn_s,
            "knn_fp64_tflops": 3.0 * n * n * d / knn_s / 1e12, "cpu_kdtree_search_s": cpu_s, "cpu_cores": 1,
            "neighbours_equal_sklearn": bool(np.array_equal(idx.cpu().numpy(), ref))}


CASES = {
    "train_n2_b128": lambda: case_train("train_n2_b128", "N2", 19608, 128, "rms_prop", 10),
    "train_n2_b4096": lambda: case_train("train_n2_b4096", "N2", 19608, 4096, "rms_prop", 20),
    "train_n1_b128": lambda: case_train("train_n1_b128", "N1", 5000, 128, "rms_prop", 20),
    "train_n1_b256": lambda: case_train("train_n1_b256", "N1", 500, 256, "rms_prop", 50),
    "train_n1_b4096": lambda: case_train("train_n1_b4096", "N1", 20000, 4096, "adam", 10),
    "train_n1_b65536": lambda: case_train("train_n1_b65536", "N1", 262144, 65536, "adam", 3),
    "train_n5_b4096": lambda: case_train("train_n5_b4096", "N5", 19608, 4096, "adam", 5),
    "train_n5_b65536": lambda: case_train("train_n5_b65536", "N5", 131072, 65536, "adam", 2),
    "gen_n2": lambda: case_gen("gen_n2", "N2", 176470),
    "gen_n1": lambda: case_gen("gen_n1", "N1", 20000),
    "gen_n5": lambda: case_gen("gen_n5", "N5", 1000000, reps=3),
    "balance_c1": lambda: case_balance("balance_c1", "C1"),
    "balance_c2": lambda: case_balance("balance_c2", "C2"),
    "smote_c1": lambda: case_smote("smote_c1", "C1"),
    "smote_c2": lambda: case_smote("smote_c2", "C2"),
    "balance_c3_seq": lambda: case_balance_c3("balance_c3_seq", 1),
    "balance_c3_par": lambda: case_balance_c3("balance_c3_par", 8),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(CASES))
    args = ap.parse_args()
    for c in args.cases.split(","):
        print(json.dumps(CASES[c]()), flush=True)


if __name__ == "__main__":
    main()
