"""The tile schedule of the whole-path launch (kernels_path.hip k_whole, vpt_api.hip batch_begin), restated: wave w of W takes tiles w, w + W, ... for
`static_rounds` rounds without an atomic, the tiles behind them come `chunk` at a time through a counter (optionally shrinking towards the end), and a wave
alternates shade / refill / trace steps until it has nothing left.  Whatever the interleaving of the waves, the path lengths and the schedule parameters, every
sample of the batch must be started exactly once and finished exactly once, and the hit ring (128 entries per wave) must never overflow.  CPU only: this pins
the arithmetic the host passes to the kernel (the kernel itself is held to the oracle by tests/test_gpu_whole.py)."""
import random

import pytest


def static_rounds_for(n_slots, grid_blocks, mode):
    """vpt_api.hip batch_begin: tiles of 64 samples, `rounds` per wave; mode 0 / 3: the first round static, 1: all but the last, 2: half."""
    n_waves = grid_blocks * 4
    rounds = ((n_slots + 63) // 64) // n_waves
    if mode in (0, 3):
        return min(rounds, 1)
    if mode == 1:
        return rounds - 1 if rounds >= 2 else 0
    return rounds // 2


def run_launch(n, grid_blocks, mode, chunk, seed, p_hit=0.8, p_alive=0.6):
    rnd = random.Random(seed)
    n_waves, n_tiles = grid_blocks * 4, (n + 63) // 64
    static_rounds = static_rounds_for(n, grid_blocks, mode)
    guided = mode == 3
    dyn_first = static_rounds * n_waves
    head = 0
    started, finished = [0] * n, [0] * n
    max_ring = 0
    waves = [dict(static_left=static_rounds, next_static=w, w_next=0, w_end=0, last_seen=dyn_first, exhausted=False, ring=[], lanes=[None] * 64, done=False) for w in range(n_waves)]
    live = list(range(n_waves))
    while live:
        w = waves[rnd.choice(live)]          # any interleaving of the waves
        # ---- shade: 64 parked hits, or what is left once nothing can be added
        if len(w["ring"]) >= 64 or (w["exhausted"] and w["ring"]):
            assert all(l is None for l in w["lanes"])
            for i in range(min(len(w["ring"]), 64)):
                s = w["ring"].pop(0)
                if rnd.random() < p_alive:
                    w["lanes"][i] = s
                else:
                    finished[s] += 1
        # ---- refill (two passes)
        if not w["exhausted"]:
            for _ in range(2):
                free = [i for i in range(64) if w["lanes"][i] is None]
                if not free:
                    break
                if w["w_next"] >= w["w_end"]:
                    if w["static_left"]:
                        tile, span = w["next_static"], 1
                        w["next_static"] += n_waves; w["static_left"] -= 1
                    else:
                        take = chunk
                        if guided:
                            left = max(n_tiles - w["last_seen"], 0)
                            take = max(1, min(chunk, left // (2 * n_waves)))
                        tile, span = dyn_first + head, take
                        head += take
                        w["last_seen"] = tile + take
                    if tile >= n_tiles:
                        w["exhausted"] = True
                    else:
                        w["w_next"], w["w_end"] = tile * 64, min((tile + span) * 64, n)
                if w["exhausted"]:
                    break
                for k, i in enumerate(free):
                    li = w["w_next"] + k
                    if li < w["w_end"]:
                        w["lanes"][i] = li
                        started[li] += 1
                w["w_next"] += min(len(free), w["w_end"] - w["w_next"])
        if all(l is None for l in w["lanes"]):
            if w["exhausted"] and not w["ring"]:
                live.remove(waves.index(w))
            continue
        # ---- trace: hits are parked, misses end
        for i in range(64):
            s = w["lanes"][i]
            if s is not None:
                if rnd.random() < p_hit:
                    w["ring"].append(s)
                else:
                    finished[s] += 1
                w["lanes"][i] = None
        max_ring = max(max_ring, len(w["ring"]))
    return started, finished, max_ring


@pytest.mark.parametrize("n,grid", [(1, 1), (63, 1), (64, 2), (5151, 3), (8 * 8, 1), (97 * 53, 7), (40000, 12), (120000, 24)])
@pytest.mark.parametrize("mode,chunk", [(0, 4), (0, 1), (0, 8), (1, 1), (1, 4), (2, 4), (3, 4), (3, 8)])
def test_every_sample_is_started_and_finished_exactly_once(n, grid, mode, chunk):
    started, finished, max_ring = run_launch(n, grid, mode, chunk, seed=n * 31 + grid * 7 + mode * 3 + chunk)
    assert started == [1] * n and finished == [1] * n
    assert max_ring <= 127


def test_the_static_part_never_reaches_past_the_batch():
    for n in (1, 64, 65, 4096, 2073600, 468633600):
        for grid in (1, 3, 768):
            for mode in (0, 1, 2, 3):
                sr = static_rounds_for(n, grid, mode)
                assert sr * grid * 4 <= (n + 63) // 64      # the tiles dealt without an atomic all exist
