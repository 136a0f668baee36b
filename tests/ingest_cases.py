"""Generated OBJ inputs shared by the ingest tests and tests/golden/make_ingest_golden.py."""
import numpy as np


def polygon_soup(seed, n_faces=400):
    """OBJ text with triangles, quads and 5..9-gons: planar convex, planar concave (star shaped with random radii),
    non-planar and degenerate ones, mixed index forms, groups, comments and line endings."""
    rng = np.random.default_rng(seed)
    lines, nv = ["# polygon soup %d" % seed, "vn 0 0 1", "vt 0 0"], 0
    for f in range(n_faces):
        k = int(rng.choice([3, 4, 4, 4, 5, 6, 7, 9]))
        kind = int(rng.integers(0, 5))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        rad = np.ones(k) if kind == 0 else rng.uniform(0.2, 1.0, k)
        pts = np.stack([rad * np.cos(ang), rad * np.sin(ang), np.zeros(k)], 1)
        if kind == 2:
            pts[:, 2] = rng.normal(0, 0.3, k)          # non planar
        if kind == 3:
            pts = rng.normal(0, 1, (k, 3))             # arbitrary
        if kind == 4 and k > 3:
            pts[1] = pts[0]                            # repeated point
        # random rigid-ish transform so every projection plane gets exercised
        q, _ = np.linalg.qr(rng.normal(0, 1, (3, 3)))
        pts = pts @ q.T * rng.uniform(0.1, 10) + rng.uniform(-5, 5, 3)
        if rng.random() < 0.15:
            lines.append("g part%d" % f)
        if rng.random() < 0.05:
            lines.append("o obj%d  " % f)
        for p in pts:
            lines.append("v %.6f %.6f %.6f" % tuple(p))
        form = int(rng.integers(0, 5))
        idx = []
        for j in range(k):
            a = nv + j + 1
            if form == 1:
                idx.append("%d/1/1" % a)
            elif form == 2:
                idx.append("%d//1" % a)
            elif form == 3:
                idx.append("%d" % (a - (nv + k) - 1))   # relative
            elif form == 4:
                idx.append("%d/1" % a)
            else:
                idx.append("%d" % a)
        if rng.random() < 0.5:
            idx = idx[::-1]
        tail = "  # trailing comment 1 2 3" if rng.random() < 0.1 else ""
        lines.append("f " + " ".join(idx) + tail)
        nv += k
    eol = ["\n", "\r\n"]
    return "".join(l + eol[i % 2 if seed % 2 else 0] for i, l in enumerate(lines))
