"""Seeded synthetic imagery for tests and benchmarks (no datasets are available).

A "scene" is a canvas of overlapping grey rectangles and discs (plenty of FAST
corners at several scales); frames are crops of the canvas that translate by a
fixed step per frame, plus per-frame sensor noise, so consecutive frames can be
matched with a known ground-truth shift.
"""
import numpy as np


def _canvas(w, h, seed, n_shapes, smax=None):
    rng = np.random.default_rng(seed)
    if smax is None:
        smax = max(8, min(w, h) // 6)
    img = np.full((h, w), 128, np.float32)
    # low-frequency shading so flat regions are not perfectly flat
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img += 20.0 * np.sin(xx / 97.0) * np.cos(yy / 71.0)
    for _ in range(n_shapes):
        g = float(rng.integers(0, 256))
        cx = int(rng.integers(0, w)); cy = int(rng.integers(0, h))
        s = int(rng.integers(4, smax))
        if rng.random() < 0.6:
            s2 = int(rng.integers(4, smax))
            img[max(cy - s2, 0):cy + s2, max(cx - s, 0):cx + s] = g
        else:
            y0, y1 = max(cy - s, 0), min(cy + s + 1, h)
            x0, x1 = max(cx - s, 0), min(cx + s + 1, w)
            sub = img[y0:y1, x0:x1]
            m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= s * s
            sub[m] = g
    # mild 3x3 box blur: edges span more than one pixel like a real lens
    p = np.pad(img, 1, mode="edge")
    img = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)) / 9.0
    return img


def sequence(w, h, n_frames, seed=1, n_shapes=None, noise=3.0, step=(2, 1), per_frame_density=False, smax=None):
    """Return uint8 array [n_frames, h, w]; frame k = canvas crop shifted by k*step.
    n_shapes counts the shapes drawn on the whole canvas; with per_frame_density it is the number per frame-sized
    area instead (long sequences travel over a canvas many frames large and would otherwise thin out).  smax bounds
    the half-size of a shape (default: a sixth of the smaller image side; large images want many small shapes to
    fill a feature budget that grows with their area)."""
    if n_shapes is None:
        n_shapes = max(60, (w * h) // 768)
    mx = abs(step[0]) * n_frames + 8
    my = abs(step[1]) * n_frames + 8
    if per_frame_density:
        n_shapes = int(round(n_shapes * float((w + mx) * (h + my)) / float(w * h)))
        smax = smax or max(8, min(w, h) // 6)  # shape sizes of a frame-sized canvas
    can = _canvas(w + mx, h + my, seed, n_shapes, smax)
    rng = np.random.default_rng(seed + 7919)
    out = np.empty((n_frames, h, w), np.uint8)
    for k in range(n_frames):
        ox = k * step[0] if step[0] >= 0 else mx + k * step[0]
        oy = k * step[1] if step[1] >= 0 else my + k * step[1]
        f = can[oy:oy + h, ox:ox + w] + rng.normal(0.0, noise, (h, w)).astype(np.float32)
        out[k] = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return out


def stereo_pair(w, h, seed=2, noise=2.0, dmin=4, dmax=60, band=32):
    """Left/right pair: the right image is the left shifted by a per-row-band disparity."""
    can = _canvas(w + dmax + 8, h, seed, max(60, (w * h) // 768))
    rng = np.random.default_rng(seed + 104729)
    left = can[:, :w]
    right = np.empty_like(left)
    disp = np.empty(h, np.int32)
    for y0 in range(0, h, band):
        d = int(rng.integers(dmin, dmax + 1))
        disp[y0:y0 + band] = d
        # a point at x in the left image appears at x - d in the right image
        right[y0:y0 + band] = can[y0:y0 + band, d:d + w]
    nl = rng.normal(0.0, noise, (h, w)).astype(np.float32)
    nr = rng.normal(0.0, noise, (h, w)).astype(np.float32)
    L = np.clip(np.rint(left + nl), 0, 255).astype(np.uint8)
    R = np.clip(np.rint(right + nr), 0, 255).astype(np.uint8)
    return L, R, disp


def descriptors(n, seed):
    """i.i.d. uniform 256-bit descriptors [n, 32] uint8."""
    return np.random.default_rng(seed).integers(0, 256, (n, 32), dtype=np.uint8)


def planted(Q, seed, flip=0.1):
    """T = permuted copy of Q with each bit flipped w.p. `flip`; returns (T, perm) with T[i] = noisy Q[perm[i]]."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(len(Q))
    bits = np.unpackbits(Q[perm], axis=1)
    bits ^= (rng.random(bits.shape) < flip).astype(np.uint8)
    return np.packbits(bits, axis=1), perm


def vocabulary(k=10, L=3, seed=0, ragged=True, stop_frac=0.03, scoring=0, weighting=0):
    """Seeded stand-in for a DBoW2 ORB vocabulary (the real ORBvoc.txt is not available): a k-ary tree of
    depth <= L in the node order DBoW2 creates and saves (the children of a node are numbered
    contiguously, then each child is expanded depth first).  Child descriptors are noisy copies of
    their parent (fewer flipped bits deeper down), some siblings are exact duplicates (ties), some
    words have weight 0 (stopped).  With `ragged`, nodes have 2..k children and some stop above L.
    Returns a dict of arrays over nodes 1..n in file order: parent, is_leaf, desc [n,32], weight."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc, weight = [], [], [], []

    def expand(pid, pdesc, level):
        nch = int(rng.integers(2, k + 1)) if ragged else k
        first = len(parent) + 1
        flip = 0.5 if level == 1 else 0.25 / level
        ch = []
        for c in range(nch):
            if c > 0 and rng.random() < 0.08:
                d = ch[int(rng.integers(0, c))].copy()
            else:
                bits = np.unpackbits(pdesc)
                bits ^= (rng.random(256) < flip).astype(np.uint8)
                d = np.packbits(bits)
            ch.append(d)
            parent.append(pid); leaf.append(1); desc.append(d); weight.append(0.0)
        for c in range(nch):
            nid = first + c
            if level < L and not (ragged and level >= 1 and rng.random() < 0.12):
                leaf[nid - 1] = 0
                expand(nid, ch[c], level + 1)
            else:
                weight[nid - 1] = 0.0 if rng.random() < stop_frac else float(rng.uniform(0.05, 9.0))

    expand(0, np.zeros(32, np.uint8), 1)
    return {"k": k, "L": L, "scoring": scoring, "weighting": weighting,
            "parent": np.asarray(parent, np.int32), "is_leaf": np.asarray(leaf, np.uint8),
            "desc": np.asarray(desc, np.uint8).reshape(-1, 32), "weight": np.asarray(weight, np.float64)}


def vocabulary_full(k=10, L=6, seed=0, scoring=0, weighting=0):
    """Complete k-ary tree of depth L (the shape of the stock ORB vocabulary: k = 10, L = 6, 1.1 M nodes), built
    level by level with numpy; nodes are numbered breadth first (every parent precedes its children)."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc = [], [], []
    prev_ids = np.zeros(1, np.int64); prev_desc = np.zeros((1, 32), np.uint8)
    nxt = 1
    for level in range(1, L + 1):
        n = len(prev_ids) * k
        ids = np.arange(nxt, nxt + n)
        flips = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        for _ in range(min(level - 1, 3)):  # 50 %, 25 %, 12.5 %, 6.25 % of the bits differ from the parent
            flips &= rng.integers(0, 256, (n, 32), dtype=np.uint8)
        d = np.repeat(prev_desc, k, axis=0) ^ flips
        parent.append(np.repeat(prev_ids, k)); leaf.append(np.full(n, level == L, np.uint8)); desc.append(d)
        prev_ids, prev_desc, nxt = ids, d, nxt + n
    nwords = len(prev_ids)
    weight = np.zeros(nxt - 1, np.float64)
    weight[-nwords:] = rng.uniform(0.05, 9.0, nwords)
    return {"k": k, "L": L, "scoring": scoring, "weighting": weighting, "parent": np.concatenate(parent).astype(np.int32),
            "is_leaf": np.concatenate(leaf), "desc": np.concatenate(desc), "weight": weight}


def save_vocabulary_text(path, voc, trailing_newline=True):
    """Write `voc` in the text format of DBoW2's saveToTextFile / loadFromTextFile ("k L  scoring weighting",
    then one line per node: parent, leaf flag, 32 descriptor bytes, weight)."""
    lines = ["%d %d  %d %d" % (voc["k"], voc["L"], voc["scoring"], voc["weighting"])]
    for p, l, d, w in zip(voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"]):
        lines.append("%d %d %s  %s" % (p, l, " ".join(str(int(b)) for b in d), repr(float(w))))
    with open(path, "w") as f:
        f.write("\n".join(lines) + ("\n" if trailing_newline else ""))
