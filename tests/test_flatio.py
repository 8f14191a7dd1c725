"""Flat on-disk problem format: files written by libgsfm at the C ABI boundary
(`gsfm_ctx_set_dump_dir` / GSFM_DUMP_DIR) hold the problem, the options and the result; glomap_amd/flatio.py reads
them back and a replay through the C ABI reproduces the stored result."""
import numpy as np
import pytest

from glomap_amd import estimators, flatio, so3, synthetic


def test_python_round_trip(tmp_path):
    for p, opt in ((synthetic.make_ring_view_graph(30, 4, seed=1), estimators.RotationEstimatorOptions(max_num_irls_iterations=7)),
                   (synthetic.make_gp_problem(8, 60, seed=2), estimators.GlobalPositionerOptions(seed=5)),
                   (synthetic.make_ba_problem(num_cams=6, num_pts=50, seed=3), estimators.BundleAdjusterOptions(optimize_rotations=False))):
        rec = flatio.from_problem(p, opt)
        path = tmp_path / f"{rec.kind}.gsfm"
        flatio.save(path, rec)
        back = flatio.load(path)
        assert back.kind == rec.kind and back.scalars == {k: float(v) for k, v in rec.scalars.items()} or back.scalars == rec.scalars
        for k, a in rec.arrays.items():
            assert back.arrays[k].dtype == a.dtype and np.array_equal(back.arrays[k], a)
        p2, o2 = flatio.to_problem(back)
        assert type(p2) is type(p)
        if rec.kind == "ra":
            assert o2.max_num_irls_iterations == 7 and p2.num_edges == p.num_edges
        elif rec.kind == "gp":
            assert o2.seed == 5 and p2.num_obs == p.num_obs
        else:
            assert o2.optimize_rotations is False and p2.num_obs == p.num_obs


def test_python_round_trip_of_16_wide_intrinsics_rows(tmp_path):
    """A BA problem with a camera model of more than 8 parameters: the [K, 16] rows survive the flat file (their width is what
    estimators.ba_solve hands over as gsfm_ba_problem::intr_stride)."""
    ba = synthetic.make_ba_problem(num_cams=6, num_pts=50, seed=4)
    ba.intr_model[:] = 11  # RAD_TAN_THIN_PRISM_FISHEYE
    wide = np.zeros((ba.num_intr, 16))
    wide[:, :4] = [1200, 1190, 640, 480]
    wide[:, 4:] = np.linspace(0.01, -0.01, 12)
    ba.intr_params = wide
    path = tmp_path / "ba16.gsfm"
    flatio.save(path, flatio.from_problem(ba, estimators.BundleAdjusterOptions()))
    q, _ = flatio.to_problem(flatio.load(path))
    assert q.intr_params.shape == (ba.num_intr, 16) and np.array_equal(q.intr_params, wide) and np.array_equal(q.intr_model, ba.intr_model)


def test_python_round_trip_of_rig_and_gravity_tables(tmp_path):
    """The optional tables of the three problems — RA image / cam blocks and gravity flags, GP image offsets and centre
    blocks, BA image cam_from_rig and sensor blocks — survive the file format."""
    gp, ba, info = synthetic.make_rig_problems(8, 3, 120, seed=4)
    gpu = synthetic.forget_rig_translations(gp, info)
    bas = ba.copy()
    bas.image_sensor, bas.sensor_cam_from_rig = info["sensor_block"].copy(), info["sensor_cam_from_rig"].copy()
    ra = synthetic.make_ring_view_graph(24, 4, seed=1)
    ra.image_frame = (np.arange(24) // 3).astype(np.int32)
    ra.image_cam = np.where(np.arange(24) % 3 == 2, 0, -1).astype(np.int32)
    ra.cam_aa0 = np.array([[0.1, -0.2, 0.05]])
    ra.num_nodes, ra.node_aa0 = 8, ra.node_aa0[:8].copy()
    rg = synthetic.make_ring_view_graph(12, 3, seed=2)
    rg.node_gravity = (np.arange(12) % 2).astype(np.uint8)
    for p, opt, fields in ((ra, estimators.RotationEstimatorOptions(skip_initialization=True), ("image_frame", "image_cam", "cam_aa0")),
                           (rg, estimators.RotationEstimatorOptions(use_gravity=True), ("node_gravity",)),
                           (gpu, estimators.GlobalPositionerOptions(), ("image_frame", "image_offset", "image_sensor", "image_sensor_rot", "sensor_center")),
                           (bas, estimators.BundleAdjusterOptions(optimize_rig_poses=True),
                            ("image_frame", "image_cam_from_rig", "image_intr", "image_sensor", "sensor_cam_from_rig"))):
        rec = flatio.from_problem(p, opt)
        path = tmp_path / f"{rec.kind}_{fields[0]}.gsfm"
        flatio.save(path, rec)
        p2, o2 = flatio.to_problem(flatio.load(path))
        for f in fields:
            a, b = np.asarray(getattr(p, f)), np.asarray(getattr(p2, f))
            assert a.shape == b.shape and np.array_equal(a, b), f
        if rec.kind == "ba":
            assert o2.optimize_rig_poses is True
        if rec.kind == "ra":
            assert o2.use_gravity == opt.use_gravity and o2.skip_initialization == opt.skip_initialization


@pytest.mark.gpu
def test_dump_and_replay(gsfm_ctx, tmp_path):
    ra = synthetic.make_ring_view_graph(80, 8, seed=1)
    gp = synthetic.make_gp_problem(20, 400, seed=2, dir_noise=1e-3)
    ba = synthetic.make_ba_problem(num_cams=15, num_pts=300, seed=3, pixel_noise=0.5)
    gsfm_ctx.set_dump_dir(tmp_path)
    try:
        rc, rot, rep_ra = estimators.ra_solve(ra, estimators.RotationEstimatorOptions(irls_loss_parameter_sigma=4.0), ctx=gsfm_ctx)
        rc, cen, X, rep_gp = estimators.gp_solve(gp, ctx=gsfm_ctx)
        rc, q, t, Xb, intr, rep_ba = estimators.ba_solve(ba, estimators.BundleAdjusterOptions(optimize_principal_point=True), ctx=gsfm_ctx)
    finally:
        gsfm_ctx.set_dump_dir(None)
    files = sorted(tmp_path.glob("*.gsfm"))
    assert [f.name for f in files] == ["ba_0002.gsfm", "gp_0001.gsfm", "ra_0000.gsfm"]
    # --- RA: inputs, options, outputs and report as they crossed the ABI
    rec = flatio.load(tmp_path / "ra_0000.gsfm")
    assert rec.status == 0 and rec.options["irls_loss_parameter_sigma"] == 4.0 and rec.scalars["num_nodes"] == 80
    assert np.array_equal(rec.arrays["edge_i"], ra.edge_i) and np.array_equal(rec.arrays["edge_q"], ra.edge_q)
    assert np.array_equal(rec.arrays["node_aa0"], ra.node_aa0) and np.array_equal(rec.arrays["out_rot_aa"], rot)
    assert rec.report["iterations_irls"] == rep_ra["iterations_irls"]
    p, opt = flatio.to_problem(rec)
    rc, rot2, rep2 = estimators.ra_solve(p, opt, ctx=gsfm_ctx)
    assert rc == 0 and np.abs(rot2 - rot).max() < 1e-12
    # --- GP
    rec = flatio.load(tmp_path / "gp_0001.gsfm")
    assert np.array_equal(rec.arrays["obs_dir"], gp.obs_dir) and np.array_equal(rec.arrays["out_cam_center"], cen)
    p, opt = flatio.to_problem(rec)
    rc, cen2, X2, rep2 = estimators.gp_solve(p, opt, ctx=gsfm_ctx)
    assert rc == 0 and rep2["iterations"] == rep_gp["iterations"] and np.abs(cen2 - cen).max() < 1e-9
    # --- BA
    rec = flatio.load(tmp_path / "ba_0002.gsfm")
    assert rec.options["optimize_principal_point"] == 1.0 and np.array_equal(rec.arrays["out_cam_q"], q)
    assert np.array_equal(rec.arrays["cam_q"], ba.cam_q)  # the in/out arrays are stored as they came IN
    p, opt = flatio.to_problem(rec)
    assert opt.optimize_principal_point is True
    rc, q2, t2, X2, intr2, rep2 = estimators.ba_solve(p, opt, ctx=gsfm_ctx)
    assert rc == 0 and rep2["iterations"] == rep_ba["iterations"] and np.abs(q2 - q).max() < 1e-9
