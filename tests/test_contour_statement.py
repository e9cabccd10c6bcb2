"""The statement the GPU contour pass is built on (csrc/vp_contours.inl header), restated in plain Python for small masks and checked
against the oracle's literal restatement of OpenCV's scan-and-mark border following (oracle.find_contours) - on the CPU, no GPU code:
  * a border is a cycle of follower states (pixel, direction it was entered from); every crack between a foreground pixel and a
    4-adjacent background pixel is swept by exactly one state;
  * the smallest vertical crack of a cycle in scan order (row, then x of the crack) is the W crack of the component's first pixel
    (outer border) or the E crack of the pixel left of the background region's first pixel (hole border); the state that sweeps it
    is the start state; scan order of those cracks = cv2's contour order (newest first);
  * RETR_EXTERNAL: walk left from a first pixel along its row; no foreground = external; otherwise the E crack met belongs to a
    border of the same background region: a hole border = inside a hole, an outer border = that component's answer.
No union-find, no marks."""
import numpy as np
import pytest

DX = (1, 1, 0, -1, -1, -1, 0, 1)        # E, NE, N, NW, W, SW, S, SE
DY = (0, -1, -1, -1, 0, 1, 1, 1)


def _contours_by_cycles(mask, mode, method):
    m = np.pad(mask != 0, 1)
    h, w = mask.shape

    def fg(y, x):
        return bool(m[y + 1, x + 1])

    def ring(y, x):
        return [fg(y + DY[d], x + DX[d]) for d in range(8)]

    def first_cw(R, d):                               # first foreground neighbour clockwise from direction d
        for i in range(1, 8):
            if R[(d - i) % 8]:
                return (d - i) % 8
        raise AssertionError

    def step(y, x, s):                                # -> (swept background directions, direction left in, next state)
        R = ring(y, x)
        t = 0
        while not R[(s + 1 + t) % 8]:
            t += 1
        s2 = (s + 1 + t) % 8
        return [(s + 1 + i) % 8 for i in range(t)], s2, (y + DY[s2], x + DX[s2], (s2 + 4) % 8)

    borders = []                                      # (key, is_hole, points)
    seen = set()
    owner = {}                                        # (y, x of a foreground pixel, crack direction) -> index of its border
    ys, xs = np.nonzero(mask)
    for y, x in zip(ys.tolist(), xs.tolist()):
        R = ring(y, x)
        if not any(R):                                # a pixel on its own: a border of one point
            owner[(y, x, 0)] = owner[(y, x, 4)] = len(borders)
            borders.append(((y, x), False, [(x, y)]))
            continue
        for c in (0, 2, 4, 6):
            if R[c]:
                continue
            st = (y, x, first_cw(R, c))
            if st in seen:
                continue
            cycle, cur = [], st                       # the cycle through this state
            while cur not in seen:
                seen.add(cur)
                cycle.append(cur)
                cur = step(*cur)[2]
            assert cur == st, "the follower map is not a permutation of the states that sweep a crack"
            best = None
            for i, (cy, cx, cs) in enumerate(cycle):
                swept = step(cy, cx, cs)[0]
                for d in swept:
                    if d in (0, 2, 4, 6):
                        assert (cy, cx, d) not in owner, "a crack swept twice"
                        owner[(cy, cx, d)] = len(borders)
                if 4 in swept and (best is None or (cy, cx, 0) < best[0]):
                    best = ((cy, cx, 0), i, False)
                if 0 in swept and (best is None or (cy, cx + 1, 1) < best[0]):
                    best = ((cy, cx + 1, 1), i, True)
            assert best is not None
            key, i0, hole = best
            pts = []
            for cy, cx, cs in cycle[i0:] + cycle[:i0]:
                s2 = step(cy, cx, cs)[1]
                if method == 1 or s2 != (cs ^ 4):
                    pts.append((cx, cy))
            borders.append((key[:2], hole, pts))
    # every crack belongs to exactly one border
    for y, x in zip(ys.tolist(), xs.tolist()):
        for c in (0, 2, 4, 6):
            if not fg(y + DY[c], x + DX[c]):
                assert (y, x, c) in owner or not any(ring(y, x))
    order = sorted(range(len(borders)), key=lambda b: borders[b][0])
    if mode == 0:
        ext = {}

        def external(b):
            if b in ext:
                return ext[b]
            (y, xc), hole, _ = borders[b]
            assert not hole
            x = xc - 1
            while x >= 0 and not fg(y, x):
                x -= 1
            if x < 0:
                r = True
            else:
                o = owner[(y, x, 0)]
                r = (not borders[o][1]) and external(o)
            ext[b] = r
            return r
        order = [b for b in order if not borders[b][1] and external(b)]
    order = order[::-1]                               # cv2: newest first
    return [np.array(borders[b][2], np.int32).reshape(-1, 1, 2) for b in order], np.array([borders[b][1] for b in order], np.uint8)


def _same(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


def test_statement_on_known_shapes(oracle):
    m = np.zeros((9, 12), np.uint8)
    m[1:8, 1:11] = 255
    m[3:6, 3:9] = 0
    m[4, 5] = 255                                     # a ring with an island in its hole
    for mode in (0, 1):
        for method in (1, 2):
            got, gh = _contours_by_cycles(m, mode, method)
            exp, eh = oracle.find_contours(m, mode, method, with_holes=True)
            assert _same(got, exp) and np.array_equal(gh, eh), (mode, method)
    assert len(_contours_by_cycles(m, 0, 2)[0]) == 1 and len(_contours_by_cycles(m, 1, 2)[0]) == 3


def test_statement_on_random_masks(oracle):
    rng = np.random.default_rng(2024)
    n = 0
    for trial in range(400):
        h, w = int(rng.integers(1, 14)), int(rng.integers(1, 18))
        p = rng.choice([0.15, 0.35, 0.5, 0.65, 0.85])
        m = ((rng.random((h, w)) < p) * 255).astype(np.uint8)
        if trial % 5 == 0 and h > 4 and w > 4:        # thin walls and nested rings
            m[:] = 0
            m[0:h, 0:w] = 255
            m[1:h - 1, 1:w - 1] = 0
            m[2:h - 2, 2:w - 2] = ((rng.random((max(h - 4, 0), max(w - 4, 0))) < 0.6) * 255).astype(np.uint8)
        for mode in (0, 1):
            method = 1 + (trial + mode) % 2
            got, gh = _contours_by_cycles(m, mode, method)
            exp, eh = oracle.find_contours(m, mode, method, with_holes=True)
            assert _same(got, exp) and np.array_equal(gh, eh), (trial, mode, method, m.tolist())
            n += len(exp)
    assert n > 2000


@pytest.mark.parametrize("row_mask,col_mask", [(1, 3), (3, 7)])
def test_head_bitmaps_agree_with_the_follower(row_mask, col_mask):
    """The GPU pass cuts borders at "heads": states that sweep an eligible crack (N / S in cut columns, W / E in cut rows and wherever
    a border could start).  Two pieces of code must name the same heads: k_ct_headmaps, which writes them down per crack with bit
    operations on whole words (a crack is a head iff, going clockwise from it towards its state's direction, no other eligible crack
    comes first), and ct_head_type, which a walker asks per state (the first eligible crack the state sweeps).  Restated here for both
    cut spacings (csrc/vp_contours.inl ct_cut) and compared on random masks."""
    rng = np.random.default_rng(77)
    for trial in range(120):
        h, w = int(rng.integers(1, 12)), int(rng.integers(1, 20))
        mask = rng.random((h, w)) < rng.choice([0.2, 0.5, 0.8])
        m = np.pad(mask, 1)

        def ring(y, x):
            return [bool(m[y + 1 + DY[d], x + 1 + DX[d]]) for d in range(8)]
        for y in range(h):
            for x in range(w):
                if not mask[y, x]:
                    continue
                n = ring(y, x)
                if not any(n):
                    continue
                rowel, el = (y & row_mask) == 0, (x & col_mask) == 0
                elw = rowel or not (n[1] or n[2] or n[3])
                ele = rowel or n[1]
                # k_ct_headmaps, bit for bit
                hw = (not n[4]) and elw and (n[3] or n[2] or ((not el) and (n[1] or n[0] or ((not ele) and (n[7] or n[6] or n[5])))))
                he = (not n[0]) and ele and (n[7] or n[6] or ((not el) and (n[5] or n[4] or ((not elw) and (n[3] or n[2] or n[1])))))
                hn = (not n[2]) and el and (n[1] or n[0] or ((not ele) and (n[7] or n[6])))
                hs = (not n[6]) and el and (n[5] or n[4] or ((not elw) and (n[3] or n[2])))
                by_map = {c for c, on in ((4, hw), (0, he), (2, hn), (6, hs)) if on}
                # ct_head_type over the states of this pixel that sweep a crack
                by_state = set()
                for c in (0, 2, 4, 6):
                    if n[c]:
                        continue
                    s = next((c - i) % 8 for i in range(1, 8) if n[(c - i) % 8])      # the state that sweeps crack c
                    t = 0
                    while not n[(s + 1 + t) % 8]:
                        t += 1
                    first = None
                    for i in range(t):                                              # first eligible crack of the sweep
                        d = (s + 1 + i) % 8
                        if (d == 4 and elw) or (d == 0 and ele) or (d in (2, 6) and el):
                            first = d
                            break
                    if first is not None:
                        by_state.add(first)
                assert by_map == by_state, (trial, y, x, n, by_map, by_state)
