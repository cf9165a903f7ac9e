"""Sequential numpy model of the level-synchronous quadtree used by the HIP kernel k_quadtree
(orb_slam2_detailed_comments_amd/csrc/orbx_kernels.hip).  It mirrors the kernel's pass structure step by
step (census -> creation ranks -> push_front placement -> careful-phase cut) so that the *formulation* can
be checked against the literal std::list restatement in the oracle on the CPU.  Host logic test only.
"""
import numpy as np


def _quadrant(x, y, box):
    x0, y0, x1, y1 = box
    mx = x0 + ((x1 - x0 + 1) >> 1)
    my = y0 + ((y1 - y0 + 1) >> 1)
    return (0 if x < mx else 1) + (0 if y < my else 2)


def _child_box(box, q):
    x0, y0, x1, y1 = box
    mx = x0 + ((x1 - x0 + 1) >> 1)
    my = y0 + ((y1 - y0 + 1) >> 1)
    return (mx if q & 1 else x0, my if q & 2 else y0, x1 if q & 1 else mx, y1 if q & 2 else my)


def distribute(xs, ys, resp, order, qt_w, qt_h, N):
    """xs, ys: int coords relative to the border; resp: scores; order: emission-order keys (unique).
    Returns the selected key indices in list order."""
    K = len(xs)
    ratio = np.float32(qt_w) / np.float32(qt_h)
    nini = int(np.floor(np.float64(ratio) + 0.5))  # C round(): halves away from zero (ratio > 0)
    hx = np.float32(qt_w) / np.float32(nini)
    knode = np.zeros(K, np.int64)
    rootcnt = np.zeros(nini, np.int64)
    for k in range(K):
        b = min(int(np.float32(xs[k]) / hx), nini - 1)
        knode[k] = b
        rootcnt[b] += 1
    boxes, cnt, meta_rank, meta_f = [], [], [], []
    rootpos = {}
    for i in range(nini):
        if rootcnt[i] > 0:
            rootpos[i] = len(boxes)
            boxes.append((int(hx * np.float32(i)), 0, int(hx * np.float32(i + 1)), qt_h))
            cnt.append(int(rootcnt[i])); meta_rank.append(0); meta_f.append(0)
    for k in range(K):
        knode[k] = rootpos[knode[k]]
    size = len(boxes)
    careful = False
    while size > 0:
        ex = [(meta_f[p] == 1) if careful else (cnt[p] > 1) for p in range(size)]
        cc = np.zeros((size, 4), np.int64)
        for k in range(K):
            p = knode[k]
            if ex[p]:
                cc[p, _quadrant(xs[k], ys[k], boxes[p])] += 1
        nexp_total = 0
        if not careful:
            ne = [int((cc[p] > 0).sum()) if ex[p] else 0 for p in range(size)]
            nx = [int((cc[p] > 1).sum()) if ex[p] else 0 for p in range(size)]
            E = np.concatenate([[0], np.cumsum(ne)[:-1]]) if size else []
            ctot = int(sum(ne)); nexp_total = int(sum(nx))
            proc = list(ex)
        else:
            cands = [p for p in range(size) if ex[p]]
            keys = {p: (cnt[p] << 16) | meta_rank[p] for p in cands}
            rank = {p: sum(1 for o in cands if keys[o] > keys[p]) for p in cands}
            M = len(cands)
            byrank = [None] * M
            for p in cands:
                byrank[rank[p]] = p
            nes = [int((cc[p] > 0).sum()) for p in byrank]
            jstar = M - 1
            acc = size
            for r in range(M):
                acc += nes[r] - 1
                if acc >= N:
                    jstar = r
                    break
            CE = np.concatenate([[0], np.cumsum(nes)[:-1]]) if M else []
            ctot = int(CE[jstar] + nes[jstar]) if jstar >= 0 and M > 0 else 0
            proc = [False] * size
            E = [0] * size
            for p in cands:
                if rank[p] <= jstar:
                    proc[p] = True
                    E[p] = int(CE[rank[p]])
        surv_rank = np.cumsum([0 if proc[p] else 1 for p in range(size)])
        nmtot = int(surv_rank[-1]) if size else 0
        new_size = ctot + nmtot
        nboxes = [None] * new_size; ncnt = [0] * new_size; nrank = [0] * new_size; nf = [0] * new_size
        newpos = np.zeros((size, 4), np.int64)
        for p in range(size):
            if proc[p]:
                r = int(E[p])
                for q in range(4):
                    c = int(cc[p, q])
                    if c > 0:
                        pos = ctot - 1 - r
                        nboxes[pos] = _child_box(boxes[p], q); ncnt[pos] = c; nrank[pos] = r; nf[pos] = 1 if c > 1 else 0
                        newpos[p, q] = pos
                        r += 1
            else:
                pos = ctot + int(surv_rank[p]) - 1
                nboxes[pos] = boxes[p]; ncnt[pos] = cnt[p]; nrank[pos] = 0; nf[pos] = 0
                newpos[p, 0] = pos
        for k in range(K):
            p = knode[k]
            q = _quadrant(xs[k], ys[k], boxes[p]) if proc[p] else 0
            knode[k] = newpos[p, q]
        prev = size
        boxes, cnt, meta_rank, meta_f, size = nboxes, ncnt, nrank, nf, new_size
        if size >= N or size == prev:
            break
        if not careful and size + 3 * nexp_total > N:
            careful = True
    best = [None] * size
    for k in range(K):
        key = (int(resp[k]), -int(order[k]))
        p = knode[k]
        if best[p] is None or key > best[p][0]:
            best[p] = (key, k)
    return [b[1] for b in best]
