import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
for k in a.files:
    d = np.abs(a[k] - b[k]); bad = d > 1e-3
    print(k, a[k].shape, "max abs diff %.3g" % d.max(), "bad %d" % bad.sum(), ("first bad idx %s" % (np.argwhere(bad)[:3].tolist(),)) if bad.any() else "")
