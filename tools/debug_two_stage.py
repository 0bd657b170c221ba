import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_ops import OracleOps
from exastencils_amd.ops import HipOps
from exastencils_amd.field import laplace_fd
from exastencils_amd.layout import FieldLayout
import test_gpu_kernels as T
hip, orc = HipOps(0), OracleOps()
for kind in ("jacobi2", "rbgs"):
    for n in (64, 130):
        st = laplace_fd(3, (1.0 / n,) * 3)
        b, e = T.box(3, n)
        g = T._two_stage_case(hip, kind, (n, n, n), st, b, e, 0)
        hip.synchronize()
        c = T._two_stage_reference(orc, kind, (n, n, n), st, b, e, 0)
        lu = FieldLayout.node(3, (n, n, n), 1)
        a = hip.to_host(g[0]).reshape(lu.shape_zyx); r = c[0].reshape(lu.shape_zyx)
        d = (a != r)
        print(kind, n, "differ", d.sum())
        if d.sum():
            zz, yy, xx = np.nonzero(d)
            print(" z planes (array idx):", np.unique(zz)[:40])
            print(" y rows:", np.unique(yy)[:40], len(np.unique(yy)))
            print(" x cols:", np.unique(xx)[:40], len(np.unique(xx)))
            k = 0
            print(" sample", zz[k], yy[k], xx[k], a[zz[k], yy[k], xx[k]], r[zz[k], yy[k], xx[k]])
