#!/usr/bin/env python
"""scale_model.py — an analytic step model of the sharded factorisation (csrc/blocked.cpp), so that the FIRST multi-GPU run of
`bench.py --gpus N` can be read against a stated expectation.  No multi-GPU node has been available in five
rounds: everything below is a PREDICTION from single-device measurements, validated only against what one MI355X can show (the
blocked handle on one rank, and a model sharded over the two CU partitions of one device).

The pipeline of one block step k on every rank (blocked.cpp), W = rows per block, G ranks, R = rows below block k:
    UPD stream :  U1 (next diagonal block, block column k+1)  ->  U2a (sized to cover chain + broadcast)  ->  solve of panel k+1 (needs LW_{k+1})  ->  U2b (the rest)
    SIDE stream:  chain(k+1) = factor + explicit inverse of the next diagonal block, after U1's diagonal part (its owner only)
    COMM stream:  broadcast LW_{k+1} after the chain;  all-gather of panel k+1 after the solve (next step's U1 waits for it)
    step  ~=  max(U1 + U2a,  U1diag + chain + bcast)  +  solve  +  max(U2b, gather)

Rates (one MI355X, measured; profiles/r04_*, r05_*):
    update kernel     R_inf TFLOP/s on launches of many tiles, quantised in rounds of one 256 x 128 tile per compute unit
    panel solve       X LW' (K ends at the tile's column): half of 2 M W W flops at R_solve
    chain             seconds per W x W block (measured per width; the persistent chain kernel of round 5 or the multi-launch one)
    xGMI              one link per peer, ~153 GB/s per direction (MI355X_MICROARCH.md / the task's hardware notes); RCCL's ring all-gather moves
                      (G - 1) blocks over one link in sequence, a direct (all-to-all style) gather one block per link at once — both are given
Usage:  python tools/scale_model.py [--chain-ms 1.0] [--out profiles/r05_scale_model.json]
"""
import argparse
import json
import math


def update_time(rows, cols_lo, cols_hi, K, es, p, lower=True):
    """seconds for  C[rows x cols] -= A B'  over the staircase region of one rank: tiles of 256 x 128, one per CU and round"""
    if rows <= 0 or cols_hi <= cols_lo:
        return 0.0
    width = cols_hi - cols_lo
    entries = rows * width * (0.5 if lower else 1.0)     # a rank's rows are spread over the whole staircase: about half of the rectangle
    tiles = max(1.0, entries / (256.0 * 128.0))
    rounds = math.ceil(tiles / p["cus"])
    t_tile = 2.0 * 256 * 128 * K / (p["R_inf"][es] * 1e12 / p["cus_full"])   # one CU's share of the asymptotic rate
    return rounds * t_tile + p["launch_s"]


def step_model(N, W, G, es, p):
    nblk = math.ceil(N / W)
    tot = 0.0
    parts = {"update": 0.0, "chain_exposed": 0.0, "solve": 0.0, "gather_exposed": 0.0, "head": 0.0}
    t_chain = p["chain_s"][W]
    # head: the first diagonal block and the first panel have nothing to hide behind
    rows0 = (N - W) / G
    t_solve0 = (rows0 * W * W) / (p["R_solve"][es] * 1e12) + p["launch_s"]
    head = p["chain_first_s"][W] + (p["bcast_s"](W, es, G) if G > 1 else 0.0) + t_solve0 + (p["gather_s"](rows0, W, es, G) if G > 1 else 0.0)
    tot += head
    parts["head"] = head
    for k in range(nblk - 1):
        R = N - (k + 1) * W                      # rows below block k (all ranks)
        own = R / G
        cols = R                                 # trailing columns
        # U1: next diagonal block (its owner) + block column k+1 of the own rows
        t_u1diag = 2.0 * W * W * W / 2 / (p["R_small"][es] * 1e12) + p["launch_s"]
        t_u1 = t_u1diag + (update_time(own, 0, W, W, es, p, lower=False) if G > 1 else 0.0)
        # U2a: just enough of the update to cover the chain kernel + the broadcast (blocked.cpp u2a_cover_s_), U2b: the rest
        t_upd = update_time(own, W, cols, W, es, p)
        t_u2a = t_upd if G == 1 else min(t_upd, p["u2a_cover_s"](W))
        t_u2b = 0.0 if G == 1 else max(0.0, t_upd - t_u2a)
        own_next = max(0.0, (R - W) / G)
        t_solve = (own_next * W * W) / (p["R_solve"][es] * 1e12) + p["launch_s"]
        t_b = p["bcast_s"](W, es, G) if G > 1 else 0.0
        t_g = p["gather_s"](own_next, W, es, G) if G > 1 else 0.0
        a = t_u1 + t_u2a
        b = t_u1diag + t_chain + t_b
        step = max(a, b) + t_solve + max(t_u2b, t_g)
        tot += step
        parts["update"] += t_u1 + t_u2a + t_u2b
        parts["chain_exposed"] += max(0.0, b - a)
        parts["solve"] += t_solve
        parts["gather_exposed"] += max(0.0, t_g - t_u2b)
    # tail: logdet, distributed back-substitution (nblk small all-reduces + block solves), alpha
    tail = nblk * (p["allreduce_small_s"] if G > 1 else 0.0) + nblk * 60e-6 + N * N * es / 2 / G / (p["hbm_Bps"]) * 1.0
    tot += tail
    parts["tail"] = tail
    return tot, parts


def predict_time(N, P, G, es, p):
    # whitening of P rows: N^2 P flops split by columns over the ranks, V_k broadcast per block (N P es bytes in total)
    fl = float(N) * N * P
    t = fl / G / (p["R_predict"][es] * 1e12) + (N * P * es / p["link_Bps"] if G > 1 else 0.0) + N * P * es / p["hbm_Bps"]
    return t


def params(chain_ms, partitions=False):
    link = 153e9
    lat = 25e-6
    p = {
        "cus": 248, "cus_full": 256,
        # update256_kernel at K = 1024, large launches: calibrated so that the model's one-rank update total equals the measured U1 + U2a phase
        # total of the blocked handle at N = 50 000 (635 ms per fit, profiles/r05_a_bench_blocked.json); fp32 from the C4-size fits (20.0 - 20.9 s)
        "R_inf": {8: 64.6, 4: 128.0},
        "R_small": {8: 40.0, 4: 75.0},      # 128 x 64-tile products of a few tiles (diagonal block first)
        "R_solve": {8: 43.0, 4: 85.0},      # X LW' (measured: 0.61 ms for ~25 000 x 1024 x 1024 at N = 50 000)
        "R_predict": {8: 55.0, 4: 105.0},   # 46 ms at N = 50 000, P = 1024
        "launch_s": 8e-6,
        "chain_s": {w: chain_ms * 1e-3 * (w / 1024.0) ** 2 for w in (256, 512, 1024, 2048)},       # ~ W^2 (critical path ~ W, work ~ W^3 on 8 CUs)
        "chain_first_s": {w: chain_ms * 1e-3 * (w / 1024.0) ** 2 for w in (256, 512, 1024, 2048)},
        "link_Bps": link,
        "hbm_Bps": 4.0e12,
        "allreduce_small_s": 30e-6,
        "u2a_cover_s": lambda W: 1.1e-3 * (W / 1024.0) ** 2 + 0.4e-3,
        # RCCL ring all-gather: (G - 1) hops of one block over one link; broadcast of W x W: a ring / tree pipelined over ~2 link times
        "gather_s": lambda rows, W, es, G: lat + (G - 1) * rows * W * es / link,
        "bcast_s": lambda W, es, G: lat + 2.0 * W * W * es / link,
    }
    if partitions:  # two CU partitions of ONE device: half the CUs each, the "link" is a device-to-device copy in HBM
        p["cus"] = 120
        p["cus_full"] = 256
        # the two halves share HBM, the XCDs' L2s and the power budget: two independent fits side by side take 1.2 - 1.45x one of them alone
        # (tests/test_gpu_dist.py::test_cu_partitions...; profiles/r04_q_partitions.log) — applied to every compute rate of this validation case only
        co = 1.3
        p["R_inf"] = {k: v / co for k, v in p["R_inf"].items()}
        p["R_small"] = {k: v / co for k, v in p["R_small"].items()}
        p["link_Bps"] = 1.5e12
        p["gather_s"] = lambda rows, W, es, G: 15e-6 + (G - 1) * rows * W * es / 1.5e12
        p["bcast_s"] = lambda W, es, G: 15e-6 + W * W * es / 1.5e12
        p["R_solve"] = {8: 22.0, 4: 43.0}
        p["R_predict"] = {8: 28.0, 4: 52.0}
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chain-ms", type=float, default=1.0, help="measured seconds per 1024 x 1024 diagonal block (factor + inverse), in ms")
    ap.add_argument("--chain-ms-old", type=float, default=3.5, help="the multi-launch chain of rounds 1-4, for comparison")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    out = {"what": "PREDICTED fits/s of bench.py --gpus N (sharded fit + predict_f), from single-device rates; see tools/scale_model.py",
           "inputs": {"chain_ms_per_1024_block": args.chain_ms, "xgmi_link_GBps": 153, "collective_latency_us": 25,
                      "update_TFLOPs_fp64": 64.6, "update_TFLOPs_fp32": 128.0}}
    for label, N, es, W, P in (("N=50000 fp64 (the bench line)", 50000, 8, 1024, 1024), ("N=200000 fp32 (c4_sharded, north_star's 60% target)", 200000, 4, 1024, 1024)):
        rows = []
        for G in (1, 2, 4, 8):
            p = params(args.chain_ms)
            t, parts = step_model(N, W, G, es, p)
            tp = predict_time(N, P, G, es, p)
            rows.append({"gpus": G, "fit_s": t, "predict_s": tp, "fits_per_sec": 1.0 / (t + tp), "parts_s": {k: round(v, 5) for k, v in parts.items()}})
        t1 = rows[0]["fit_s"] + rows[0]["predict_s"]
        for r in rows:
            r["strong_scaling_efficiency"] = t1 / ((r["fit_s"] + r["predict_s"]) * r["gpus"])
            r["bound"] = ("chain" if r["parts_s"]["chain_exposed"] > max(r["parts_s"]["gather_exposed"], 0.05 * r["fit_s"]) else
                          "gather" if r["parts_s"]["gather_exposed"] > 0.05 * r["fit_s"] else "update (MFMA)")
        out[label] = rows
    # validation against what one device can show
    val = {}
    t, parts = step_model(50000, 1024, 1, 8, params(args.chain_ms_old))
    val["blocked handle, one rank, N=50000 fp64, multi-launch chain"] = {"model_fit_ms": 1e3 * t, "measured_fit_ms": "702 - 708 (r04 / r05 call A)"}
    t, parts = step_model(50000, 1024, 1, 8, params(args.chain_ms))
    val["blocked handle, one rank, N=50000 fp64, chain kernel"] = {"model_fit_ms": 1e3 * t}
    for n, meas in ((32768, "332 - 344"), (65536, "1950 - 1990")):
        t, parts = step_model(n, 1024, 2, 8, params(args.chain_ms_old, partitions=True))
        val[f"two CU partitions, N={n} fp64, multi-launch chain"] = {"model_fit_ms": 1e3 * t, "measured_fit_ms": meas,
                                                                     "parts_s": {k: round(v, 4) for k, v in parts.items()}}
    out["validation_on_one_device"] = val
    txt = json.dumps(out, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
