#!/usr/bin/env python3
"""CPU simulation of the two-pass render's SECOND pass on C4 (Julia c = -0.8+0.156i, 16384^2, cap 4096, f32): where do
its idle lanes come from?  A sample of tiles goes through the first pass's schedule (as tools/sim/first_pass_dynamics.py);
every handed-over pixel's remaining orbit length is computed; then persistent waves of 64 lanes drain the entries with
the kernel's policy (refill once `want` lanes are free and `minrun` iterations were done, blocks of 4) in different ORDERS:
    tile by tile     what the lists hold: a tile's lanes arrive together (the sample's tiles are in random order)
    shuffled         the same entries in random order
    long first       an oracle: entries that reach the cap first
    by live lanes    a predictor the first pass has for free: entries of tiles with more lanes alive at the hand-over first
    capped at K      every entry runs at most K more iterations here; what is still going goes to a third pass (its
                     wave-iterations, at full lanes but for the last wave, are added)
Prints wave-iterations per useful 64 lane-iterations (1.0 = every lane busy) and the share of the drain (the part after
a wave's last refill).  Usage: python tools/sim/second_pass_drain.py [tiles] [entries per wave]"""
import sys

import numpy as np

rng = np.random.default_rng(1)
W = H = 16384
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
PER_WAVE = int(sys.argv[2]) if len(sys.argv) > 2 else 4180  # 1.82e7 entries over 4352 resident waves
tx = rng.integers(0, W // 8, NT)
ty = rng.integers(0, H // 8, NT)
lx = np.arange(64) % 8
ly = np.arange(64) // 8
x = (tx[:, None] * 8 + lx[None, :]).astype(np.float64)
y = (ty[:, None] * 8 + ly[None, :]).astype(np.float64)
X = ((((x / H) - (W / H) / 2.0) / 0.4)).astype(np.float32)
Y = ((((y / H) - 0.5) / 0.4)).astype(np.float32)
cre, cim = np.float32(-0.8), np.float32(0.156)
cap, k1, keep, M = 4096, 64, 48, 4
T, lim2 = np.float32(7.9), np.float32(2.0 ** 32)


def step(X, Y, m):
    Xn = (X * X - Y * Y) + cre
    Yn = (np.float32(2) * X) * Y + cim
    return np.where(m, Xn, X), np.where(m, Yn, Y)


# ---- first pass (episodes of 64, keep 48, doubling from 8 x k1): who is handed over, with what count
run = (X * X + Y * Y) <= T
cnt = np.zeros((NT, 64), np.int32)
active = np.ones(NT, bool)
handed = np.zeros((NT, 64), bool)
live_at = np.zeros(NT, np.int32)
done, ep_len = 0, k1
while active.any() and done < cap:
    nblk = min(ep_len, cap - done) // M
    for b in range(nblk):
        act = run & active[:, None]
        if not act.any():
            break
        for _ in range(M):
            X, Y = step(X, Y, act)
        cnt = np.where(act, cnt + M, cnt)
        run &= ~(act & ((X * X + Y * Y) > T))
    done += nblk * M
    nrun = run.sum(axis=1)
    ho = active & (nrun > 0) & (nrun < keep)
    handed |= run & ho[:, None]
    live_at = np.where(ho, nrun, live_at)
    active &= ~((nrun == 0) | ho)
    if done >= 8 * k1 and ep_len < 16 * k1:
        ep_len *= 2
# ---- remaining length of every handed-over orbit: blocks of M until |z|^2 > T (then ~5 exact iterations, not counted
# here: they run in the finishing pass at full lanes) or the cap
ex, ey = X[handed], Y[handed]
ez2 = (ex * ex + ey * ey).astype(np.float64)  # |z|^2 at the hand-over
ecnt = cnt[handed]
# what the tile's other lanes did: the mean count of the lanes that had left by then (they froze at their count)
gone = ~handed & ~run
tile_gone_mean = np.where(gone.any(axis=1), (cnt * gone).sum(axis=1) / np.maximum(gone.sum(axis=1), 1), 0.0)
egone = np.repeat(tile_gone_mean[:, None], 64, axis=1)[handed]
elive = np.repeat(live_at[:, None], 64, axis=1)[handed]
n = ex.size
rem = np.zeros(n, np.int32)
going = np.ones(n, bool)
while going.any():
    for _ in range(M):
        ex, ey = step(ex, ey, going)
    rem += going * M
    going &= ((ex * ex + ey * ey) <= T) & (ecnt + rem + M <= cap)
print("tiles %d, entries %d (%.2f %% of the pixels), at the cap %.2f %%, mean remaining %.1f iterations, share of the iterations in "
      "entries that reach the cap %.2f" % (NT, n, 100.0 * n / (NT * 64), 100.0 * (ecnt + rem + M > cap).mean(), rem.mean(),
                                           rem[ecnt + rem + M > cap].sum() / rem.sum()))


def drain(lengths, want=24, minrun=8):
    """one persistent wave over `lengths` in that order; returns (wave-iterations, wave-iterations after the last refill)"""
    left = np.zeros(64, np.int64)
    nxt, total, since_refill_total = 0, 0, 0
    N = len(lengths)
    while True:
        free = left <= 0
        if nxt < N and free.any():
            k = min(int(free.sum()), N - nxt)
            idx = np.flatnonzero(free)[:k]
            left[idx] = lengths[nxt:nxt + k]
            nxt += k
            since_refill_total = 0
        busy = left > 0
        if not busy.any():
            if nxt >= N:
                break
            continue
        if nxt < N:  # run until `want` lanes are free (at least minrun)
            nb = int(busy.sum())
            thr = max(nb - want, 0)
            srt = np.sort(left[busy])
            # iterations until at most thr lanes still run
            t = int(srt[nb - thr - 1]) if thr < nb else 0
            t = max(t, minrun)
            t = -(-t // M) * M
        else:
            t = int(left[busy].max())
        left = left - t
        total += t
        since_refill_total += t
    return total, since_refill_total


def report(name, order_lengths, extra=0.0):
    nw = max(1, len(order_lengths) // PER_WAVE)
    tot = dr = 0
    per = []
    for w in range(nw):
        a, b = drain(order_lengths[w::nw])  # chunks go round the waves
        tot += a
        dr += b
        per.append(a)
    useful = float(rem.sum())
    print("%-28s %.3f wave-iterations per 64 useful lane-iterations (lane use %.1f %%); after the last refill %.1f %% of them; "
          "slowest wave / mean %.3f" % (name, (tot + extra) * 64.0 / useful, 100.0 * useful / ((tot + extra) * 64.0),
                                        100.0 * dr / (tot + extra), max(per) / (sum(per) / len(per))))


perm = rng.permutation(n)
report("tile by tile (the lists)", rem)
report("shuffled", rem[perm])
longf = np.argsort(-(rem.astype(np.int64)), kind="stable")
report("long first (oracle)", rem[longf])
pred = np.argsort(-(elive[perm].astype(np.int64)), kind="stable")
report("by live lanes at hand-over", rem[perm][pred])
for pname, key in (("by |z|^2 at hand-over (small first)", ez2[perm]), ("by |z|^2 (large first)", -ez2[perm]),
                   ("by count at hand-over (large first)", -ecnt[perm].astype(np.float64)),
                   ("by the tile's gone lanes' mean count (large first)", -egone[perm])):
    report(pname[:28], rem[perm][np.argsort(key, kind="stable")])
print("rank correlation with the remaining length: |z|^2 %.3f, count %.3f, gone lanes' mean count %.3f, live lanes %.3f" % tuple(
    float(np.corrcoef(np.argsort(np.argsort(v)), np.argsort(np.argsort(rem)))[0, 1]) for v in (ez2, ecnt, egone, elive)))
for K in (128, 256, 512):
    short = np.minimum(rem[perm], K)
    over = rem[perm][rem[perm] > K] - K
    third = float(np.ceil(over.size / 64.0) * 0)  # (filled below)
    # third pass: the survivors are all long; waves of 64 run max(their lengths): sort is free there (few entries)
    o = np.sort(over)[::-1]
    extra = sum(int(o[i:i + 64].max()) for i in range(0, o.size, 64)) * 1.0
    # distribute over the same number of waves: what matters is the total
    nw = max(1, n // PER_WAVE)
    tot = 0
    for w in range(nw):
        a, _ = drain(short[w::nw])
        tot += a
    useful = float(rem.sum())
    print("capped at %-4d + third pass   %.3f wave-iterations per 64 useful lane-iterations (lane use %.1f %%); third pass %.1f %% of them, %d entries"
          % (K, (tot + extra) * 64.0 / useful, 100.0 * useful / ((tot + extra) * 64.0), 100.0 * extra / (tot + extra), over.size))

# ---- the policy: wave-iterations and episodes (refills) per 64 entries, for the cost model
#      vector instructions = 6.75 x wave-iterations + E x episodes + F x (entries / 64)
print("\npolicy sweep (as produced): want, minrun -> wave-iterations per 64 entries, episodes per 64 entries, lane use")
def drain_count(lengths, want, minrun):
    left = np.zeros(64, np.int64)
    nxt, total, episodes = 0, 0, 0
    N = len(lengths)
    while True:
        free = left <= 0
        if nxt < N and free.any():
            k = min(int(free.sum()), N - nxt)
            idx = np.flatnonzero(free)[:k]
            left[idx] = lengths[nxt:nxt + k]
            nxt += k
        busy = left > 0
        if not busy.any():
            if nxt >= N:
                break
            continue
        if nxt < N:
            nb = int(busy.sum())
            thr = max(nb - want, 0)
            srt = np.sort(left[busy])
            t = int(srt[nb - thr - 1]) if thr < nb else 0
            t = max(t, minrun)
            t = -(-t // M) * M
        else:
            t = int(left[busy].max())
        left = left - t
        total += t
        episodes += 1
    return total, episodes
sample = rem[perm][:PER_WAVE * 8]
for want in (4, 8, 12, 16, 24, 32, 48):
    for minrun in (4, 8, 16, 32):
        tot = eps = 0
        for w in range(8):
            a, b = drain_count(sample[w::8], want, minrun)
            tot += a
            eps += b
        print("want %2d minrun %2d: %7.1f wave-iterations, %5.2f episodes per 64 entries; lane use %.1f %%" % (
            want, minrun, tot * 64.0 / sample.size, eps * 64.0 / sample.size, 100.0 * float(sample.sum()) / (tot * 64.0)))

# ---- cascaded lists, honestly: every level is the same refill kernel (want 24) with its own cap; what is still going at a
#      level's cap is written to the next level's lists in the order it finishes there (taken as shuffled)
print("\ncascades (every level the refill kernel, want 24, minrun 8):")
def run_levels(caps):
    cur = rem[perm].astype(np.int64)
    tot_w = 0
    desc = []
    for K in caps:
        run_len = np.minimum(cur, K) if K else cur
        nw = max(1, len(run_len) // PER_WAVE)
        tw = 0
        for w in range(nw):
            a, _ = drain_count(run_len[w::nw], 24, 8)
            tw += a
        tot_w += tw
        desc.append("%s: %d entries, lane use %.1f %%" % ("cap %d" % K if K else "rest", len(cur), 100.0 * float(run_len.sum()) / (tw * 64.0)))
        if not K:
            break
        cur = cur[cur > K] - K
        cur = cur[rng.permutation(len(cur))]
        if len(cur) == 0:
            break
    return tot_w, desc
for caps in ((0,), (128, 0), (256, 0), (128, 512, 0), (64, 256, 1024, 0), (32, 64, 128, 256, 512, 1024, 0)):
    tw, desc = run_levels(caps)
    print("levels %-28s lane use overall %.1f %%   [%s]" % (str(caps), 100.0 * float(rem.sum()) / (tw * 64.0), "; ".join(desc)))
