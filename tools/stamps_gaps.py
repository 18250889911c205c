"""Offline view of the raw stamps tools/stamps.py leaves with STAMPS_NPZ=<file>: the forward kernel's time between the groups of
a block and at the block's top (subtree plans):  python tools/stamps_gaps.py gpurun_out/stamps_mb.npz [groups per block] [block]"""
import sys
import numpy as np
z = np.load(sys.argv[1]); fw = z['fw']; bw = z['bw']
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
NG = G - 2 if G == 5 else G - 1          # general groups (the subtree phase is stamped in slot NG)
def s(st, cb, gi, k): return st[:, (cb * G + gi) * 16 + k].astype(np.int64)
names = {9: 'loop top', 10: 'perm mfma', 11: 'perm store+sync', 12: 'tape store', 13: 'stage issue', 14: 'sync'}
prev = s(fw, cb - 1, NG - 1, 6)
for k in (9, 10, 11, 12, 13, 14):
    cur = s(fw, cb, NG, k); print(f'fwd {names[k]:18s}', cur - prev); prev = cur
print('fwd -> sub start      ', s(fw, cb, NG, 0) - prev)
print('fwd sub phase         ', s(fw, cb, NG, 1) - s(fw, cb, NG, 0), 'barrier', s(fw, cb, NG, 2) - s(fw, cb, NG, 1))
print('fwd sub end -> grp0   ', s(fw, cb, 0, 0) - s(fw, cb, NG, 2))
for gi in range(NG):
    print('fwd grp', gi, 'total', s(fw, cb, gi, 6) - s(fw, cb, gi, 0), ' gap to next', (s(fw, cb, gi + 1, 0) - s(fw, cb, gi, 6)) if gi < NG - 1 else '')
print('fwd block total', s(fw, cb + 1, 0, 0) - s(fw, cb, 0, 0))
