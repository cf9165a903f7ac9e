"""Host-side mirror of ORB_SLAM2::ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> for the one call on
the hot path: transform(features, BowVector&, FeatureVector&, levelsup) (reference src/Frame.cc:750-765,
Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1136-1216).  The tree descent runs in the HIP kernel k_bow_transform."""
from __future__ import annotations
import ctypes as C
import numpy as np
from . import _capi
from ._capi import check, ptr, lib

TF_IDF, TF, IDF, BINARY = range(4)
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)


class ORBVocabulary:
    def __init__(self, extractor, *, n_nodes, k, L, child_begin, child_ids, desc, weight, word_id, weighting=TF_IDF,
                 scoring=L1_NORM):
        a = lambda x, t: np.ascontiguousarray(x, t)
        self._ex = extractor
        self._keep = [a(child_begin, np.int32), a(child_ids, np.uint32), a(desc, np.uint8), a(weight, np.float64), a(word_id, np.uint32)]
        v = _capi.VocabularyView()
        v.n_nodes, v.k, v.L, v.weighting, v.scoring = int(n_nodes), int(k), int(L), int(weighting), int(scoring)
        v.child_begin, v.child_ids, v.desc, v.weight, v.word_id = (x.ctypes.data for x in self._keep)
        self._h = C.c_void_p()
        check(lib().orbx_vocabulary_create(extractor.handle, C.byref(v), C.byref(self._h)))

    @classmethod
    def load_text(cls, extractor, path):
        """ORBvoc.txt format of TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1368-1450): header
        `k L scoring weighting`, then one node per line `parent is_leaf d0 .. d31 weight`; node ids in file order from 1."""
        with open(path) as f:
            k, L, scoring, weighting = (int(x) for x in f.readline().split())
            parents, descs, weights, leaf = [0], [np.zeros(32, np.uint8)], [0.0], [False]
            for line in f:
                t = line.split()
                if not t:
                    continue
                parents.append(int(t[0])); leaf.append(int(t[1]) > 0)
                descs.append(np.array([int(x) for x in t[2:34]], np.uint8)); weights.append(float(t[34]))
        n = len(parents)
        children = [[] for _ in range(n)]
        for i in range(1, n):
            children[parents[i]].append(i)
        begin = np.zeros(n + 1, np.int32)
        for i in range(n):
            begin[i + 1] = begin[i] + len(children[i])
        word_id = np.zeros(n, np.uint32); nw = 0
        for i in range(1, n):                      # words are numbered in file order of the leaves
            if leaf[i]:
                word_id[i] = nw; nw += 1
        return cls(extractor, n_nodes=n, k=k, L=L, child_begin=begin,
                   child_ids=np.array([c for ch in children for c in ch], np.uint32), desc=np.stack(descs),
                   weight=np.array(weights, np.float64), word_id=word_id, weighting=weighting, scoring=scoring)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().orbx_vocabulary_destroy(self._h)
            except Exception:   # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    def transform_features(self, desc, levelsup=4):
        """per descriptor (word_id, weight, node_id)"""
        d = np.ascontiguousarray(desc, np.uint8); n = len(d)
        wid = np.zeros(max(n, 1), np.uint32); w = np.zeros(max(n, 1), np.float64); nid = np.zeros(max(n, 1), np.uint32)
        check(lib().orbx_bow_transform(self._ex.handle, self._h, ptr(d), n, int(levelsup), ptr(wid), ptr(w), ptr(nid)))
        return wid[:n], w[:n], nid[:n]

    def transform(self, desc, levelsup=4):
        """Returns (BowVector as (word ids, values), FeatureVector as (node_id, begin, index)) in map order."""
        wid, w, nid = self.transform_features(desc, levelsup)
        n = len(wid)
        bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64)
        fn = np.zeros(max(n, 1), np.uint32); fb = np.zeros(n + 2, np.int32); fi = np.zeros(max(n, 1), np.uint32)
        nb, nn = C.c_int(0), C.c_int(0)
        check(lib().orbx_bow_vectors(self._h, ptr(wid), ptr(w), ptr(nid), n, ptr(bw), ptr(bv), C.byref(nb), ptr(fn), ptr(fb),
                                     ptr(fi), C.byref(nn)))
        return (bw[:nb.value].copy(), bv[:nb.value].copy()), (fn[:nn.value].copy(), fb[:nn.value + 1].copy(), fi[:fb[nn.value]].copy())
