#!/usr/bin/env python3
"""CPU simulation of the two-pass render's FIRST pass on a random sample of C4's 8x8 tiles (Julia c = -0.8+0.156i,
16384^2, cap 4096, f32 arithmetic, T = 7.9, episodes of 64 iterations, keep = 48): how many blocks of 4 unchecked
iterations a tile's wave runs, how many exact finishing iterations, how many tiles leave the all-asm path (lanes
still running after the first episode), how many pixels are handed to the lists.  These are the dynamic weights of
profiles/r03_c4_first_pass_classes.txt (tools/first_pass_classes.py multiplies the static ISA by them).
Usage: python tools/sim/first_pass_dynamics.py [tiles]   (prints one JSON object)"""
import json
import sys

import numpy as np

rng = np.random.default_rng(1)
W = H = 16384
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
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


run = (X * X + Y * Y) <= T
cnt = np.zeros((NT, 64), np.int32)
active = np.ones(NT, bool)          # tile still in the first pass's episodes
blocks1 = np.zeros(NT, np.int32)    # blocks run in the first episode
blocks_late = np.zeros(NT, np.int32)
lane_blocks = np.zeros(NT, np.int64)
handed = np.zeros((NT, 64), bool)
left_asm = np.zeros(NT, bool)
done, ep, ep_len = 0, 0, k1
while active.any() and done < cap:
    nblk = min(ep_len, cap - done) // M
    for b in range(nblk):
        act = run & active[:, None]
        any_act = act.any(axis=1)
        if not any_act.any():
            break
        for _ in range(M):
            X, Y = step(X, Y, act)
        cnt = np.where(act, cnt + M, cnt)
        if ep == 0:
            blocks1 += any_act
        else:
            blocks_late += any_act
        lane_blocks += act.sum(axis=1)
        run &= ~(act & ((X * X + Y * Y) > T))
    done += nblk * M
    nrun = run.sum(axis=1)
    if ep == 0:
        left_asm = nrun > 0
    ho = active & (nrun > 0) & (nrun < keep)
    handed |= run & ho[:, None]
    active &= ~((nrun == 0) | ho)
    ep += 1
    if done >= 8 * k1 and ep_len < 16 * k1:
        ep_len *= 2
fz = ~handed & ~run
live = fz & ((X * X + Y * Y) <= lim2)
wave_fin = np.zeros(NT, np.int32)
lane_fin = np.zeros(NT, np.int64)
for it in range(64):
    if not live.any():
        break
    wave_fin += live.any(axis=1)
    lane_fin += live.sum(axis=1)
    X, Y = step(X, Y, live)
    cnt += live
    live = live & ((X * X + Y * Y) <= lim2)
out = {
    "tiles_sampled": NT,
    "mean_iterations_per_pixel_in_first_pass": float(cnt.mean()),
    "tiles_finished_inside_the_asm_path": float((~left_asm).mean()),
    "blocks_per_tile_first_episode": float(blocks1.mean()),
    "blocks_per_tile_later_episodes": float(blocks_late.mean()),
    "lane_occupancy_of_the_block_loop": float(lane_blocks.sum() / ((blocks1 + blocks_late).sum() * 64.0)),
    "finishing_iterations_per_tile": float(wave_fin.mean()),
    "lane_occupancy_of_the_finishing_loop": float(lane_fin.sum() / max(1.0, wave_fin.sum() * 64.0)),
    "tiles_with_a_hand_over": float(handed.any(axis=1).mean()),
    "pixels_handed_over": float(handed.mean()),
    "tiles_gone_after_one_block": float(((blocks1 == 1) & ~left_asm).mean()),
}
print(json.dumps(out, indent=1))
