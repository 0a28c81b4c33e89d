"""The CPU oracle (oracle/cpu_ref.py) against vectors captured from the reference (tests/golden)."""
import numpy as np
import pytest

from oracle import cpu_ref as O
from pam import synth
import golden_io as G

TOL = 1e-9


def _cams(size):
    c = G.cameras(size)
    return O.cameras_from_arrays(c['P32'], c['K32'], c['RT32'], c['F'], c['RK_INV'], c['position'])


@pytest.mark.parametrize('size', G.SIZES)
def test_camera_setup(size):
    c = G.cameras(size)
    cams = O.make_cameras({'P': c['P'], 'K': c['K'], 'RT': c['RT']})
    for j, cam in enumerate(cams):
        assert cam.P.dtype == np.float32 and cam.RK_INV.dtype == np.float32 and cam.position.dtype == np.float64
        # bit-equal on this image (same torch CPU algebra as the reference); a few float32 ulps are allowed for other BLAS builds
        np.testing.assert_allclose(cam.F, c['F'][j], rtol=5e-7, atol=1e-12)
        np.testing.assert_allclose(cam.RK_INV, c['RK_INV'][j], rtol=5e-7, atol=1e-12)
        np.testing.assert_allclose(cam.position, c['position'][j], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('size', G.SIZES)
def test_ops(size):
    cams = _cams(size)
    ops = G.ops(size)
    cfg = synth.MATCHER_CFG[synth.SIZE_TO_DATASET[size]]
    for r in ops['project']:
        np.testing.assert_allclose(O.project_tracks(cams[int(r['cid'])].P, r['pts']), r['out'], rtol=TOL, atol=TOL)
    for r in ops['assoc']:
        cam = cams[int(r['cid'])]
        aff = O.association_affinity(O.project_tracks(cam.P, r['tracks_pose']), r['dets'], r['dt'],
                                     cfg['ALPHA2D'], cfg['LAMBDA_A'])
        np.testing.assert_allclose(aff, r['affinity'], rtol=TOL, atol=TOL)
        rows, cols = O.lsap(-r['affinity'])
        assert np.array_equal(rows, r['rows']) and np.array_equal(cols, r['cols'])
    for r in ops['lsap_init']:
        rows, cols = O.lsap(r['cost'])
        assert np.array_equal(rows, r['rows']) and np.array_equal(cols, r['cols'])
    for r in ops['epi_par']:
        cs = [cams[i] for i in r['cids']]
        np.testing.assert_allclose(O.epi_dist_parallel(cs, r['pose_mat']), r['dist'], rtol=TOL, atol=TOL)
    for r in ops['epi_distance']:
        np.testing.assert_allclose(O.epi_dist_pair(cams[int(r['c1'])], r['p1'], cams[int(r['c2'])], r['p2']),
                                   r['out'], rtol=TOL, atol=TOL)
    for r in ops['epi_loop']:
        cs = [cams[i] for i in r['cids']]
        d = O.epi_dist_loop(cs, r['pose_mat'])
        assert d.dtype == np.float32
        np.testing.assert_allclose(d, r['dist'], rtol=1e-6, atol=1e-6)
    for r in ops['greedy_update']:
        cs = [cams[i] for i in r['cids']]
        keep, mask = O.greedy_filter(cs, r['aff'], 'update', r['pose'][:, 0, :], r['next_pose'])
        assert np.array_equal(keep, r['matched']) and np.array_equal(mask, r['binary'])
    for r in ops['greedy_init']:
        cs = [cams[i] for i in r['cids']]
        assert r['aff'].dtype == np.float32
        keep, mask = O.greedy_filter(cs, r['aff'], 'init')
        assert np.array_equal(keep, r['matched']) and np.array_equal(mask, r['binary'])
    for op in ('dlt_update', 'dlt_init'):
        for r in ops[op]:
            cs = [cams[i] for i in r['cids']]
            A = O.dlt_rows(cs, r['pose_mat'], r['Ts'], float(r['lambda_t']))
            out = O.dlt_solve(A, r['remains'], r['nviews'], r['next_pose'])
            np.testing.assert_allclose(out, r['out'], rtol=TOL, atol=TOL)
    for r in ops['hyp_cost']:
        cs = [cams[i] for i in r['cids']]
        c, veto = O.hyp_cost(cs, list(r['poses']), cams[int(r['o_cid'])], r['o_pose'], float(r['thr']))
        assert abs(c - float(r['cost'])) <= TOL * max(1.0, abs(float(r['cost'])))
        assert int(veto) == int(r['veto'])
    for r in ops['smooth']:
        out = O.smooth_last(r['hist'], r['raw'], float(r['sigma']), float(r['arm_sigma']))
        np.testing.assert_allclose(out, r['out'], rtol=1e-12, atol=1e-12)
    for r in ops['motion']:
        v = O.velocity_from_history(list(r['hist']))
        assert v.dtype == np.float32 and str(r['vel_dtype']) == 'float32'
        assert np.array_equal(v, r['vel'])


from trace_driver import run_trace  # noqa: E402


@pytest.mark.parametrize('size', G.SIZES)
def test_trace(size):
    def factory(cfg, conf):
        return O.OracleIvclabpose(cfg, conf)
    nf = 0
    for t, tr, model in run_trace(size, factory):
        k = 'f%d.st.' % t
        trs = model.tracker.tracks
        assert [x.track_id for x in trs] == tr[k + 'ids'].tolist(), (size, t)
        assert [x.state for x in trs] == tr[k + 'state'].tolist()
        assert [x.hits for x in trs] == tr[k + 'hits'].tolist()
        assert [x.age for x in trs] == tr[k + 'age'].tolist()
        assert [x.tsu for x in trs] == tr[k + 'tsu'].tolist()
        assert [len(x.hist) for x in trs] == tr[k + 'nhist'].tolist()
        for i, x in enumerate(trs):
            assert x.order == [int(c) for c in tr[k + 'p2d_order'][i] if c >= 0]
            np.testing.assert_allclose(x.hist[-1], tr[k + 'last_pose'][i], rtol=0, atol=1e-7)
            np.testing.assert_allclose(np.asarray(x.velocity, dtype=np.float64), tr[k + 'velocity'][i], rtol=0, atol=1e-6)
        nf += 1
    assert nf > 20
