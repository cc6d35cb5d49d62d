#!/usr/bin/env python3
"""Offline study for VERDICT r3 item 4 (cost-binned waves), zero GPU minutes: from the per-query main-loop round counts of consecutive
Gauss–Newton iterations (tools/iter_dump.py) — does iteration i's cost predict iteration i+1's well enough that sorting the queries of a
workgroup-sized chunk by predicted cost, and forming the 64-lane waves in that order, cuts what the waves PAY?

A wave runs until its slowest lane is done: paid(wave) = 64 x max over its lanes of the rounds they need; lane efficiency = needed / paid.

    python tools/sim_wave_binning.py gpurun_out/iter_dump.npz
"""
import sys

import numpy as np


def paid(rounds, order=None):
    """rounds: [n] per query (n a multiple of 64), order: permutation (None = as stored). Returns 64 * sum of wave maxima."""
    r = rounds if order is None else rounds[order]
    return 64 * int(r.reshape(-1, 64).max(axis=1).astype(np.int64).sum())


def main():
    d = np.load(sys.argv[1])
    R = d["rounds"].astype(np.int64)  # [iters, scans, points]
    dxn = d["dx_norm"]
    n_it, n_scan, n_pt = R.shape
    n_pt64 = n_pt // 64 * 64
    R = R[:, :, :n_pt64]
    print("iterations %d, scans %d, %d points per scan; lane efficiency reported by the kernel: %s" % (n_it, n_scan, n_pt, np.round(d["lane_eff"], 3)))
    print("%-4s %-9s %-10s | paid relative to the stored order, queries re-dealt to waves inside chunks of C queries" % ("it", "open", "lane eff"))
    print("%-4s %-9s %-10s | %s" % ("", "scans", "(stored)", "  ".join("C=%-5d pred / oracle / u8-class" % c for c in (256, 1024, 4096))))
    tot = {}
    for it in range(1, n_it):
        open_scans = [s for s in range(n_scan) if it == 0 or dxn[it - 1][s] >= 1e-2]  # converged scans run no further iteration
        if not open_scans:
            break
        need = base = 0
        acc = {}
        for s in open_scans:
            cur, prev = R[it, s], R[it - 1, s]
            need += int(cur.sum())
            base += paid(cur)
            for C in (256, 1024, 4096):
                nC = n_pt64 // C * C
                for name, key in (("pred", prev), ("oracle", cur), ("u8", np.minimum(prev // 2, 255))):
                    o = np.argsort(key[:nC].reshape(-1, C), axis=1, kind="stable") + (np.arange(nC // C) * C)[:, None]
                    p = paid(cur[:nC], o.reshape(-1)) + paid(cur[nC:]) if nC < n_pt64 else paid(cur, o.reshape(-1))
                    acc[(C, name)] = acc.get((C, name), 0) + p
        for k, v in acc.items():
            tot[k] = tot.get(k, 0) + v
        tot["base"] = tot.get("base", 0) + base
        tot["need"] = tot.get("need", 0) + need
        print("%-4d %-9d %-10.3f | %s" % (it, len(open_scans), need / base, "  ".join("%5.3f / %5.3f / %5.3f       " % tuple(acc[(C, n)] / base for n in ("pred", "oracle", "u8")) for C in (256, 1024, 4096))))
    print("all  %-9s %-10.3f | %s" % ("", tot["need"] / tot["base"], "  ".join("%5.3f / %5.3f / %5.3f       " % tuple(tot[(C, n)] / tot["base"] for n in ("pred", "oracle", "u8")) for C in (256, 1024, 4096))))
    # how persistent is a query's cost? correlation and the share of queries whose class (rounds // 4) moves by at most one
    for it in range(1, min(n_it, 6)):
        a, b = R[it - 1].reshape(-1).astype(np.float64), R[it].reshape(-1).astype(np.float64)
        print("iterations %d -> %d: correlation of per-query rounds %.3f; |delta| <= 2 rounds for %.1f %% of the queries" % (it - 1, it, np.corrcoef(a, b)[0, 1], 100 * np.mean(np.abs(a - b) <= 2)))


if __name__ == "__main__":
    main()
