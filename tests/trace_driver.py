"""Drive an ivclabpose-like facade over a golden trace exactly as testmodel.py drives the reference
(/root/reference/src/testmodel.py:51-69) and compare the 9-tuple with what the reference returned."""
import numpy as np

from oracle import cpu_ref as O
from pam import synth
import golden_io as G


def run_trace(size, factory, atol3d=1e-7, inject_F=True):
    """Drive a façade over a golden trace exactly as testmodel.py does; yield per-frame comparisons."""
    tr = G.load('trace_%s.npz' % size)
    c = G.cameras(size)
    dataset = str(tr['meta.dataset'])
    cfg = dict(synth.MATCHER_CFG[dataset])
    conf = cfg.pop('CONF_THRESHOLD')
    model = factory(cfg, conf)
    # inject_F=False: the facade's own camera set-up (fundamental matrices included) is on the tested path
    model.GetCameraParameters({'P': c['P'], 'K': c['K'], 'RT': c['RT']}, 0, 0, **({'F': c['F']} if inject_F else {}))
    frames = G.trace_frames(tr)
    skipped = set(tr['meta.skipped'].tolist())
    C = len(frames[0])
    for t, views in enumerate(frames):
        pbl, dr = synth.to_dump_results(views)
        has = any(len(v) for v in views)
        assert has == (t not in skipped)
        if not has:
            continue
        cam_ids, pts, pids, pts3d, jv, ids, _, _, _ = model.PersonTrack_Project3DPose(t, pbl, dr, 'SVD')
        k = 'f%d.' % t
        assert np.array_equal(np.asarray(ids, dtype=np.int32), tr[k + 'ids']), (size, t)
        n = len(ids)
        if n:
            np.testing.assert_allclose(np.asarray(pts3d).reshape(n, 3, 17), tr[k + 'pts3d'], rtol=0, atol=atol3d,
                                       err_msg='%s frame %d' % (size, t))
        for i in range(n):
            exp = O.joints_views_list(tr[k + 'nviews'][i], int(tr[k + 'V'][i]))
            assert [list(map(int, a)) for a in jv[i]] == exp, (size, t, i)
            ec = [int(x) for x in tr[k + 'camera_ids'][i] if x >= 0]
            assert [int(x) for x in cam_ids[i]] == ec, (size, t, i)
            assert len(pids[i]) == int(tr[k + 'n_person_ids'][i])
            for q, cid in enumerate(ec):
                d = int(tr[k + 'pts_det'][i, q])
                assert np.array_equal(np.asarray(pts[i][q]), views[cid][d][:, [1, 0, 2]])
        yield t, tr, model


