"""Worker functions of conftest.oracle_pool (TEST INFRASTRUCTURE): run in fresh interpreters started by the fork server, so this
module has no side effects and imports nothing that touches the GPU."""
import sys

IN8_ = ("sst", "t_zt", "hum_zt", "u_zu", "v_zu", "slp", "rad_sw", "rad_lw")
OUT6_ = ("ql", "qh", "tau_x", "tau_y", "evap", "t_s")


def _oracle_rows(args):
    """Worker: the oracle on rows j0..j0+njl of the synthetic ni x nj grid (fields generated in the worker: nothing large is pickled in)."""
    root, algo, skin, niter, ni, nj, j0, njl = args
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import pyoracle as po
    f = po.synth_fields(ni, nj, j0, njl)
    o = po.OracleSession(algo, ni * njl, 1, skin).compute(1, 2.0, 10.0, niter, *[f[k] for k in IN8_[:6]], rad_sw=f["rad_sw"] if skin else None,
                                                          rad_lw=f["rad_lw"] if skin else None)
    assert o["rc"] == 0
    return j0, {k: o[k] for k in OUT6_ if k in o}


def _oracle_cells(args):
    """Worker: the oracle on given cells (nt records, warm-layer state carried)."""
    root, algo, skin, niter, nt, cols = args
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import pyoracle as po
    s = po.OracleSession(algo, cols[0].size, nt, skin)
    recs = []
    for jt in range(1, nt + 1):
        o = s.compute(jt, 2.0, 10.0, niter, *cols[:6], rad_sw=cols[6] if skin else None, rad_lw=cols[7] if skin else None)
        assert o["rc"] == 0
        recs.append({k: o[k].copy() for k in OUT6_ if k in o})
    return recs
