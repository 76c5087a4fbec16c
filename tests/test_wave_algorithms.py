"""Lane-level models of two round-6 device algorithms, checked on the CPU against the plain statement of what they compute (the emulation build runs
the wave-cooperative bodies with ONE lane, so the cross-lane logic of a rewrite is otherwise covered by GPU md5 parity only).

* the zmo window chain on a wavefront (smartdenovo_amd/csrc/wtz_window.h: wtz_chain_windows_wave): one window per lane, the reference's inner `break` as the first
  set bit of a ballot - against the double loop of chaining_wtseedv (/root/reference/hzm_aln.h:658-713) restated here in its plainest form;
* the group sketch of the seed lookup (wtz_seed.h: wtz_cwg_sk_add / wtz_cwg_sk_pass): saturating 8-bit counters must never lose a group whose lengths reach -d,
  whatever collides with it and in whatever order the adds arrive."""
import numpy as np
import pytest


def chain_plain(w, W):
    """weights / predecessors by the double loop; w: (n, 5) int array of beg0, end0, beg1, end1, ovl in window order"""
    n = len(w); acc = [0] * n; pred = [-1] * n; top, top_at = -1000000, -1
    for i in range(n):
        acc[i] += int(w[i][4])
        if acc[i] > top: top, top_at = acc[i], i
        for j in range(i + 1, n):
            g0, g1 = int(w[j][0] - w[i][1]), int(w[j][2] - w[i][3])
            if g1 < 0 or g0 < 0: continue
            if g0 > W and g1 > W: break
            band = abs(g0 - g1)
            if band > W: continue
            band = int(np.float32(band) * np.float32(0.05))
            if acc[j] < acc[i] - band: acc[j], pred[j] = acc[i] - band, i
    members, span, k = set(), 0, top_at
    while k >= 0:
        members.add(k); span += int(w[k][1] - w[k][0]); k = pred[k]
    return span, members


def chain_lanes(w, W, nlanes=64):
    """the device form: lane j holds window j; per step i a ballot of the 'far on both axes' test, updates in the lanes in front of its first set bit"""
    n = len(w); assert n <= nlanes
    lane = np.arange(nlanes); mine = lane < n
    qb = np.zeros(nlanes, np.int64); qe = qb.copy(); tb = qb.copy(); te = qb.copy(); own = qb.copy()
    qb[:n], qe[:n], tb[:n], te[:n], own[:n] = w[:, 0], w[:, 1], w[:, 2], w[:, 3], w[:, 4]
    acc = np.zeros(nlanes, np.int64); pred = np.full(nlanes, -1, np.int64); top, top_at = -1000000, -1
    for i in range(n):
        wi = int(acc[i] + own[i])
        if wi > top: top, top_at = wi, i
        g0, g1 = qb - qe[i], tb - te[i]
        later = mine & (lane > i)
        far = later & (g0 > W) & (g1 > W)
        stop_at = int(np.argmax(far)) if far.any() else nlanes
        shift = np.abs(g0 - g1)
        offer = wi - (shift.astype(np.float32) * np.float32(0.05)).astype(np.int64)
        upd = later & (lane < stop_at) & (g0 >= 0) & (g1 >= 0) & (shift <= W) & (acc < offer)
        acc[upd] = offer[upd]; pred[upd] = i
    members, span, k = set(), 0, top_at
    while k >= 0:
        members.add(k); span += int(qe[k] - qb[k]); k = int(pred[k])
    return span, members


@pytest.mark.parametrize("seed", range(40))
def test_window_chain_on_lanes_equals_the_double_loop(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 64)); W = int(rng.choice([50, 300, 800]))
    q = np.sort(rng.integers(0, 20000, size=n))                       # windows come in query order
    shape = rng.integers(0, 3)
    t = q + rng.integers(-400, 400, size=n) if shape == 0 else (rng.integers(0, 20000, size=n) if shape == 1 else q // 2 + rng.integers(0, 3000, size=n))
    ln = rng.integers(50, 900, size=n)
    w = np.stack([q, q + ln, t, t + ln + rng.integers(-30, 30, size=n), rng.integers(100, 800, size=n)], axis=1).astype(np.int64)
    assert chain_lanes(w, W) == chain_plain(w, W)


def test_window_chain_break_is_not_monotone_and_still_exact():
    """a window far on both axes in front of one that is not: everything behind the first is cut off, as in the reference"""
    w = np.array([[0, 100, 0, 100, 500], [5000, 5100, 5000, 5100, 400], [150, 250, 160, 260, 300]], dtype=np.int64)
    assert chain_lanes(w, 800) == chain_plain(w, 800)
    assert chain_plain(w, 800)[1] == {0}


def sketch_run(groups, kovl, order_seed, ncounters):
    """groups: list of (hash, [lengths]); returns the set of hashes whose counter reaches the threshold, adds applied in a shuffled order"""
    unit = (kovl + 254) // 255 if kovl > 255 else 1
    thr = (kovl + unit - 1) // unit
    cnt = np.zeros(ncounters, np.int64)
    adds = [(h % ncounters, min(l, kovl)) for h, ls in groups for l in ls]
    np.random.default_rng(order_seed).shuffle(adds)
    for h, l in adds:
        if cnt[h] >= thr: continue
        cnt[h] = min(255, cnt[h] + (l + unit - 1) // unit)
    assert thr <= 255
    return {h % ncounters for h, _ in groups if cnt[h % ncounters] >= thr}, thr


@pytest.mark.parametrize("kovl", [1, 14, 15, 16, 100, 299, 300, 301, 1000, 5000])
def test_saturating_sketch_never_loses_a_group_that_reaches_d(kovl):
    rng = np.random.default_rng(kovl)
    for trial in range(30):
        groups = []
        for g in range(int(rng.integers(1, 200))):
            k = int(rng.integers(1, 40))
            groups.append((int(rng.integers(0, 1 << 30)), [int(x) for x in rng.integers(1, max(2, kovl // 3 + 40), size=k)]))
        nc = int(rng.choice([8, 64, 65536]))
        passed, _ = sketch_run(groups, kovl, trial, ncounters=nc)
        for h, ls in groups:
            if sum(min(l, kovl) for l in ls) >= kovl:      # ol <= the sum of the group's (capped) lengths: only such a group can reach -d
                assert h % nc in passed
