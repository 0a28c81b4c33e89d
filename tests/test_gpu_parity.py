"""GPU parity: the HIP path (through the C ABI / ctypes) against the golden vectors captured from the reference and
against the CPU oracle on seeded inputs.  Bar: view / person assignment indices bit-exact; 3D joints <= 1e-3 m
(north_star), tested here at 1e-6 m; intermediate float64 quantities <= 1e-9 relative."""
import numpy as np
import pytest

import golden_io as G
from trace_driver import run_trace
from oracle import cpu_ref as O
from pam import synth

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope='module')
def lib():
    from pam import _lib
    return _lib


def _handle(lib, size, **kw):
    c = G.cameras(size)
    cfg = dict(synth.MATCHER_CFG[synth.SIZE_TO_DATASET[size]])
    conf = cfg.pop('CONF_THRESHOLD')
    h = lib.Handle(len(c['P32']), lib.make_params(cfg, conf), **kw)
    h.set_cameras(c['P32'], c['F'], c['RK_INV'], c['position'])
    return h, cfg


def _mask(keep_idx):
    m = 0
    for k in keep_idx:
        m |= 1 << int(k)
    return m


@pytest.mark.parametrize('size', G.SIZES)
def test_ops_vs_golden(lib, size):
    h, cfg = _handle(lib, size)
    ops = G.ops(size)
    for r in ops['project']:
        np.testing.assert_allclose(h.op_project(int(r['cid']), r['pts']), r['out'], rtol=TOL, atol=TOL)
    for r in ops['assoc']:
        aff = h.op_track_affinity(int(r['cid']), r['tracks_pose'], r['dt'], r['dets'])
        np.testing.assert_allclose(aff, r['affinity'], rtol=TOL, atol=TOL)
        rows, cols = h.op_lsap(-r['affinity'])
        assert np.array_equal(rows, r['rows']) and np.array_equal(cols, r['cols'])
        rows, cols = h.op_lsap(-aff)            # device affinity -> same assignment
        assert np.array_equal(rows, r['rows']) and np.array_equal(cols, r['cols'])
    for r in ops['lsap_init']:
        rows, cols = h.op_lsap(r['cost'])
        assert np.array_equal(rows, r['rows']) and np.array_equal(cols, r['cols'])
    for r in ops['epi_par']:
        np.testing.assert_allclose(h.op_epi_dist(r['cids'], r['pose_mat']), r['dist'], rtol=TOL, atol=TOL)
    for r in ops['epi_distance']:
        np.testing.assert_allclose(h.op_epi_pair(int(r['c1']), r['p1'], int(r['c2']), r['p2']), r['out'], rtol=TOL, atol=TOL)
    for r in ops['epi_loop']:
        d = h.op_epi_dist_init(r['cids'], r['pose_mat'])
        np.testing.assert_allclose(d, r['dist'], rtol=1e-6, atol=1e-6)
    for r in ops['greedy_update']:
        k = h.op_greedy('update', r['cids'], r['aff'], r['pose'][:, 0, :], r['next_pose'])
        assert k == _mask(r['matched'])
    for r in ops['greedy_init']:
        k = h.op_greedy('init', r['cids'], r['aff'])
        assert k == _mask(r['matched'])
    for op in ('dlt_update', 'dlt_init'):
        for r in ops[op]:
            V = len(r['cids'])
            masks = [_mask([v for v in range(V) if r['remains'][j, 2 * v] == 1]) for j in range(17)]
            out = h.op_dlt(r['cids'], r['Ts'], r['pose_mat'], masks, r['next_pose'])
            np.testing.assert_allclose(out, r['out'], rtol=0, atol=1e-7)
    for r in ops['hyp_cost']:
        c, veto = h.op_hyp_cost(r['cids'], r['poses'], int(r['o_cid']), r['o_pose'])
        assert abs(c - float(r['cost'])) <= TOL * max(1.0, abs(float(r['cost'])))
        assert int(veto) == int(r['veto'])
    for r in ops['smooth']:
        np.testing.assert_allclose(h.op_smooth(r['hist'], r['raw']), r['out'], rtol=1e-12, atol=1e-12)
    for r in ops['motion']:
        assert np.array_equal(h.op_velocity(r['hist']), r['vel'])
    h.close()


def test_lsap_fuzz_vs_oracle(lib):
    h, _ = _handle(lib, 'S1')
    rng = np.random.default_rng(3)
    for it in range(120):
        n, m = rng.integers(1, 12, size=2)
        kind = it % 3
        if kind == 0:
            cost = rng.normal(size=(n, m))
        elif kind == 1:
            cost = rng.integers(0, 3, size=(n, m)).astype(float)
        else:
            cost = np.zeros((n, m))
            for _ in range(min(n, m)):
                cost[rng.integers(n), rng.integers(m)] = -rng.uniform(0.1, 1)
        r0, c0 = O.lsap(cost)
        r1, c1 = h.op_lsap(cost)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), cost
    # the frame kernel's solver is a WAVE per problem (round 6): up to 16 columns its reductions are DPP row rotations, up to 64
    # cross-lane shuffles; beyond 64 the single-lane walk (the hypothesis step's form) -- every path against SciPy's order, ties included
    for it, (n, m) in enumerate([(17, 17), (20, 31), (31, 20), (33, 64), (64, 40), (64, 64), (5, 40), (40, 5), (70, 66), (66, 70), (3, 65)]):
        cost = rng.normal(size=(n, m)) if it % 2 == 0 else rng.integers(0, 4, size=(n, m)).astype(float)
        r0, c0 = O.lsap(cost)
        r1, c1 = h.op_lsap(cost)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), (n, m)
    r, c = h.op_lsap(np.zeros((0, 3)))
    assert len(r) == 0
    h.close()


def test_dlt_vs_oracle_random_masks(lib):
    """Random view subsets / ages, including degenerate 2-view joints: device Jacobi vs LAPACK SVD (oracle)."""
    h, cfg = _handle(lib, 'S4')
    c = G.cameras('S4')
    cams = O.cameras_from_arrays(c['P32'], c['K32'], c['RT32'], c['F'], c['RK_INV'], c['position'])
    seq = synth.make_sequence('S4', n_frames=2, seed=5, outlier_p=0.0)
    rng = np.random.default_rng(5)
    for trial in range(10):
        V = int(rng.integers(2, 32))
        cids = rng.permutation(31)[:V]
        person = 0
        pm = np.stack([seq['frames'][1][cid][person][:, [1, 0, 2]] for cid in cids])
        # undo the per-view shuffle: use GT projection instead so all rows belong to one person
        X = seq['gt3d'][1][1]
        for q, cid in enumerate(cids):
            hm = np.concatenate([X, np.ones((17, 1))], 1) @ c['P'][cid].T
            xy = hm[:, :2] / hm[:, 2:3] + rng.normal(0, 1.5, (17, 2))
            pm[q, :, 0], pm[q, :, 1] = xy[:, 1], xy[:, 0]
        Ts = rng.integers(0, 4, size=V)
        nviews = np.zeros(17, dtype=np.int64); mask = np.zeros((17, 2 * V), dtype=np.int64); masks = []
        for j in range(17):
            k = int(rng.integers(2, V + 1))
            keep = np.sort(rng.permutation(V)[:k])
            nviews[j] = k
            mask[j, np.repeat(keep * 2, 2) + np.tile([0, 1], k)] = 1
            masks.append(_mask(keep))
        cs = [cams[i] for i in cids]
        exp = O.dlt_solve(O.dlt_rows(cs, pm, Ts, cfg['LAMBDA_T']), mask, nviews, np.zeros((17, 3)))
        got = h.op_dlt(cids, Ts, pm, masks, np.zeros((17, 3)))
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-7)
    h.close()


@pytest.mark.parametrize('size', ['S1', 'S2'])
def test_trace_with_the_facades_own_camera_setup(size):
    """The same golden sequences with NOTHING injected: GetCameraParameters computes the fundamental matrices itself
    (ivclabpose.py:162-181), so the product's a18 code is on the path the 9-tuple comparison covers."""
    from pam.ivclabpose import ivclabpose

    def factory(cfg, conf):
        return ivclabpose(person_detector={'NAME': ''}, pose_detector=None,
                          person_matcher=dict(cfg, NAME='Iterative'), conf_threshold=conf)
    n = sum(1 for _ in run_trace(size, factory, atol3d=1e-6, inject_F=False))
    assert n > 50


@pytest.mark.parametrize('size', G.SIZES)
def test_trace_vs_golden(size):
    """Whole sequences through the drop-in facade: ids, view sets, joints_views, camera ids bit-exact; 3D <= 1e-6 m;
    tracker state (hits / age / order / velocity) identical to the reference's after every frame."""
    from pam.ivclabpose import ivclabpose

    def factory(cfg, conf):
        return ivclabpose(person_detector={'NAME': ''}, pose_detector=None,
                          person_matcher=dict(cfg, NAME='Iterative'), conf_threshold=conf)
    nf = 0
    for t, tr, model in run_trace(size, factory, atol3d=1e-6):
        k = 'f%d.st.' % t
        trs = model.tracker.tracks
        assert model.tracker.last['status'] == 0
        assert [x.track_id for x in trs] == tr[k + 'ids'].tolist(), (size, t)
        assert [x.state for x in trs] == tr[k + 'state'].tolist()
        assert [x.hits for x in trs] == tr[k + 'hits'].tolist()
        assert [x.age for x in trs] == tr[k + 'age'].tolist()
        assert [x.time_since_update for x in trs] == tr[k + 'tsu'].tolist()
        assert [x.nhist for x in trs] == tr[k + 'nhist'].tolist()
        assert [x.last_time for x in trs] == tr[k + 'last_time'].tolist()
        for i, x in enumerate(trs):
            assert x.order == [int(c) for c in tr[k + 'p2d_order'][i] if c >= 0]
            np.testing.assert_allclose(x.pose3d, tr[k + 'last_pose'][i], rtol=0, atol=1e-6)
            np.testing.assert_allclose(x.velocity_3d, tr[k + 'velocity'][i], rtol=0, atol=1e-6)
        nf += 1
    assert nf > 20


def test_batched_scenes_match_single(lib):
    """n_scenes independent trackers in one launch give, per scene, what a single-scene handle gives."""
    S = 3
    seqs = [synth.make_sequence('S2', n_frames=40, seed=10 + s) for s in range(S)]
    calib = seqs[0]['calib']
    cams = O.make_cameras(calib)
    cfg = dict(synth.MATCHER_CFG['Shelf']); conf = cfg.pop('CONF_THRESHOLD')
    prm = lib.make_params(cfg, conf)
    args = (np.stack([c.P for c in cams]), np.stack([c.F for c in cams]), np.stack([c.RK_INV for c in cams]),
            np.stack([c.position for c in cams]))
    hb = lib.Handle(5, prm, max_dets=8, max_tracks=16, n_scenes=S); hb.set_cameras(*args)
    singles = []
    for s in range(S):
        hs = lib.Handle(5, prm, max_dets=8, max_tracks=16, n_scenes=1); hs.set_cameras(*args); singles.append(hs)
    packed = [synth.pack_frames(q['frames'], 8) for q in seqs]
    for t in range(40):
        nd = np.stack([p[0][t] for p in packed]); dd = np.stack([p[1][t] for p in packed])
        oi, od = hb.frame(t, nd, dd)
        for s in range(S):
            si, sd = singles[s].frame(t, nd[s:s + 1], dd[s:s + 1])
            assert np.array_equal(oi[s], si[0])
            k = hb.layout.dbl_hdr_words                     # the header words are in-kernel clocks
            assert np.array_equal(od[s][k:], sd[0][k:])
    for hs in singles:
        hs.close()
    hb.close()


@pytest.mark.parametrize('size,seed,kw', [
    ('S1', 21, dict(occlusion_every=3, empty_view_every=5, birth_death_frame=30)),
    ('S2', 22, dict(occlusion_every=4, empty_view_every=9, birth_death_frame=45)),
    ('S2', 23, dict(occlusion_every=2, empty_view_every=4, birth_death_frame=20, blank_frames=(33, 34, 60))),
    ('S3', 24, dict(occlusion_every=3, empty_view_every=7, birth_death_frame=40)),
])
def test_facade_vs_oracle_on_fresh_stress_sequences(size, seed, kw):
    """Sequences that are NOT among the goldens (other seeds, much denser occlusions / empty views / births and deaths) through
    the drop-in facade vs the pinned oracle: ids, per-view assignments and view sets identical, 3D <= 1e-6 m, every frame."""
    from pam import synth
    from pam.ivclabpose import ivclabpose
    from oracle import cpu_ref as O
    meta = synth.SIZES[size]
    seq = synth.make_sequence(size, n_frames=120, seed=seed, **kw)
    cfg = dict(synth.MATCHER_CFG[synth.SIZE_TO_DATASET[size]]); conf = cfg.pop('CONF_THRESHOLD')
    dev = ivclabpose({'NAME': ''}, None, dict(cfg, NAME='Iterative'), conf)
    cams = dev.GetCameraParameters(seq['calib'], meta['h'], meta['w'])
    ref = O.OracleIvclabpose(cfg, conf)
    ref.GetCameraParameters(seq['calib'], F=np.stack([c.F for c in cams]))
    n_out = 0
    for t, views in enumerate(seq['frames']):
        pbl, dr = synth.to_dump_results(views)
        if not any(len(v) for v in dr):
            continue                                          # the drivers skip frames without any pose (testmodel.py:66)
        a = dev.PersonTrack_Project3DPose(t, pbl, dr, 'SVD')
        b = ref.PersonTrack_Project3DPose(t, pbl, dr, 'SVD')
        assert list(a[5]) == list(b[5]), (t, a[5], b[5])                                   # emitted track ids
        assert [list(map(int, c)) for c in a[0]] == [list(map(int, c)) for c in b[0]], t   # camera ids per track
        assert [list(map(int, c)) for c in a[2]] == [list(map(int, c)) for c in b[2]], t   # person (detection) ids per track
        assert a[4] == b[4], t                                                             # views per joint
        if len(a[5]):
            assert np.abs(np.asarray(a[3]) - np.asarray(b[3])).max() < 1e-6, t
            n_out += len(a[5])
    assert n_out > 100


def test_status_word_is_sticky_across_frames(lib):
    """An overflow raised in an EARLY frame must still be visible in the LAST record of a run (bench.py / FramePipeline.results decode
    only that one): bits 0-15 of the status word are this frame's, bits 16-31 the OR over every frame since create / reset."""
    seq = synth.make_sequence('S2', n_frames=12, seed=5)
    cams = O.make_cameras(seq['calib'])
    cfg = dict(synth.MATCHER_CFG['Shelf']); conf = cfg.pop('CONF_THRESHOLD')
    h = lib.Handle(5, lib.make_params(cfg, conf), max_dets=8, max_tracks=2, n_scenes=1)       # 4 persons, 2 track slots
    h.set_cameras(np.stack([c.P for c in cams]), np.stack([c.F for c in cams]), np.stack([c.RK_INV for c in cams]),
                  np.stack([c.position for c in cams]))
    nd, dd = synth.pack_frames(seq['frames'], 8)
    import torch
    st = torch.cuda.current_stream().cuda_stream
    seen = []
    for t in range(12):
        n_t = torch.tensor(nd[t] if t < 3 else np.zeros_like(nd[t]), dtype=torch.int32, device='cuda')      # later frames: nothing to spawn
        d_t = torch.tensor(dd[t], dtype=torch.float64, device='cuda')
        h.frame_dev(st, t, n_t.data_ptr(), d_t.data_ptr())
        h.fetch(st); h.sync(st)
        rec = h.decode(0)
        seen.append((rec['status'], rec['status_sticky']))
    assert seen[0][0] & 1, seen                      # the first frame runs out of track slots
    assert seen[-1][0] == 0, seen                    # the last frame is clean ...
    assert seen[-1][1] & 1, seen                     # ... and still reports the earlier overflow
    h.reset()
    n_t = torch.zeros(5, dtype=torch.int32, device='cuda')
    h.frame_dev(st, 0, n_t.data_ptr(), d_t.data_ptr()); h.fetch(st); h.sync(st)
    rec = h.decode(0)
    assert rec['status'] == 0 and rec['status_sticky'] == 0
    h.close()


def test_pcp_gate_through_the_hip_path():
    """north_star's first parity clause, asserted (not implied): the golden Shelf-like sequence goes through pam.ivclabpose (k_frame),
    the 3D poses are converted and scored by the restated evaluator exactly as /root/reference/src/evalmodel.py:120-206 does, and
    `check_result` + the PCP table must equal what the REFERENCE's Evaluate3DPose_PCP produced on the reference's own tracker output
    (tests/golden/pcp_S2.npz; tools/make_goldens.py adds seeded 5 cm noise to the predictions so that the table is non-trivial -- the
    same noise stream is added here)."""
    from pam import evaluation as E
    from pam.ivclabpose import ivclabpose

    def factory(cfg, conf):
        return ivclabpose(person_detector={'NAME': ''}, pose_detector=None,
                          person_matcher=dict(cfg, NAME='Iterative'), conf_threshold=conf)
    z = G.load('pcp_S2.npz')
    F = int(z['n_frames'])
    skipped = set(int(t) for t in z['skipped'])
    mine = {}
    tr = G.load('trace_S2.npz')
    c = G.cameras('S2')
    cfg = dict(synth.MATCHER_CFG[str(tr['meta.dataset'])]); conf = cfg.pop('CONF_THRESHOLD')
    model = factory(cfg, conf)
    model.GetCameraParameters({'P': c['P'], 'K': c['K'], 'RT': c['RT']}, 0, 0)
    for t, views in enumerate(G.trace_frames(tr)):
        if not any(len(v) for v in views):
            continue
        pbl, dr = synth.to_dump_results(views)
        out = model.PersonTrack_Project3DPose(t, pbl, dr, 'SVD')
        mine[t] = np.asarray(out[3], dtype=np.float64).reshape(-1, 3, 17)
    rng = np.random.default_rng(7)                                  # the generator of make_goldens.make_pcp, drawn in the same order
    preds = {}
    for t in range(F):
        if t in skipped:
            preds[t] = []
            continue
        p = mine.get(t, np.zeros((0, 3, 17)))
        assert p.shape == z['pred.%d' % t].shape, (t, p.shape)
        preds[t] = p + rng.normal(0, 0.05, p.shape)
        np.testing.assert_allclose(preds[t], z['pred.%d' % t], rtol=0, atol=1e-6)
    gt = z['gt']
    actors = [[(None if np.isnan(gt[a, f]).all() else gt[a, f]) for f in range(gt.shape[1])] for a in range(gt.shape[0])]
    check, rows = E.evaluate_pcp(z['eval_ranges'].tolist(), preds, actors, verbose=False)
    assert np.array_equal(check, z['check_result'])                 # every (frame, actor, limb) decision identical
    for r, rr in zip(rows[1:], z['table'][1:]):
        assert r[0] == str(rr[0])
        for a, b in zip(r[1:], rr[1:]):
            assert abs(float(a) - float(b)) < 1e-9, (r, rr)
