"""Diagnostic: summary of a PCR_KNN_STAMPS dump (one record per k-NN launch): mode, k, live wavefronts, span, mean wave life."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
pos = 0
while pos < len(raw):
    assert raw[pos] == 0x5354414d50
    mode, k, nw = int(raw[pos + 1]), int(raw[pos + 2]), int(raw[pos + 3])
    w = raw[pos + 4: pos + 4 + 24 * nw].reshape(nw, 24); pos += 4 + 24 * nw
    live = w[w[:, 0] != 0]
    if len(live) == 0:
        print(f"mode {mode} k {k}: 0 live waves of {nw}"); continue
    life = (live[:, 1] - live[:, 0]).astype(np.float64) / 100.0
    span = (live[:, 1].max() - live[:, 0].min()) / 100.0
    print(f"mode {mode} k {k}: {len(live):6d} live waves of {nw:6d}; span {span:7.1f} us; wave life mean {life.mean():6.1f} p99 {np.percentile(life, 99):6.1f} max {life.max():6.1f} us; sum of lives {life.sum() / 1e3:7.2f} ms")
