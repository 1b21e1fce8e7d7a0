"""Committed regression vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

These are outputs of this repository's CPU oracle on fixed inputs -- NOT reference outputs (the
reference cannot run here and ships none; see the generator's header).  They travel to the GPU
box as data, so the HIP path is checked against numbers that were reviewed and committed rather
than recomputed on the spot, and the oracle itself is pinned against drift.

  not gpu : the oracle reproduces the vectors (bit for bit where the arithmetic is IEEE-exact)
  gpu     : the HIP path, through the C ABI, reproduces them
"""
from pathlib import Path

import numpy as np
import pytest

from tests.parity_util import assert_parity

GOLD = Path(__file__).resolve().parent / "golden"
W, H = 96, 72
TAGS = ("cube", "ragged")


@pytest.fixture(scope="module")
def kv():
    return dict(np.load(GOLD / "kernels_v1.npz"))


@pytest.fixture(scope="module")
def fv():
    return dict(np.load(GOLD / "frames_v1.npz"))


def test_vectors_are_what_the_generator_documents(kv, fv):
    assert kv["K"].shape == (3, 3) and fv["K"].shape == (3, 3)
    for tag in TAGS:
        n = tuple(int(v) for v in kv[f"{tag}_res"])
        assert kv[f"{tag}_tsdf2"].shape == (n[2], n[1], n[0])
        assert (kv[f"{tag}_wts2"] == 3.0).sum() > 100, "weight cap never reached"
        assert (kv[f"{tag}_tsdf2"] == -1).sum() > 10 and kv[f"{tag}_hit0"].sum() > 500
        assert kv[f"{tag}_hit0_fg"].sum() < kv[f"{tag}_hit0"].sum()
    assert fv["f3_seg"].max() >= 1 and len(fv["f3_vis"]) >= 1
    total = (GOLD / "kernels_v1.npz").stat().st_size + (GOLD / "frames_v1.npz").stat().st_size
    assert total < 2 * 1024 * 1024


# ---- the oracle against the vectors (CPU) ---------------------------------------------------------

@pytest.mark.parametrize("tag", TAGS)
def test_oracle_reproduces_kernel_vectors(oracle, kv, tag):
    K, vox, trunc = kv["K"], float(kv[f"{tag}_voxel"]), float(kv[f"{tag}_trunc"])
    n = tuple(int(v) for v in kv[f"{tag}_res"])
    tsdf = np.zeros((n[2], n[1], n[0]), np.float32)
    wts = np.zeros_like(tsdf)
    for i in range(3):
        oracle.update_tsdf(kv[f"{tag}_depth{i}"], kv[f"{tag}_assoc{i}"], tsdf, wts, kv[f"{tag}_Roc{i}"],
                           kv[f"{tag}_toc{i}"], K, vox, trunc, 3.0)
        if i != 1:
            assert_parity(tsdf, kv[f"{tag}_tsdf{i}"], f"tsdf frame {i}", exact=True)
            assert_parity(wts, kv[f"{tag}_wts{i}"], f"weights frame {i}", exact=True)
    assert_parity(oracle.compute_tsdf_grads(tsdf), kv[f"{tag}_grads"], "grads", exact=True)
    for j in range(3):
        for name in (("", "_fg") if j == 0 else ("",)):
            fg = kv[f"{tag}_fgmask"] if name else None
            got = oracle.raycast_tsdf(tsdf, None, wts, fg, W, H, kv[f"{tag}_Rco{j}"], kv[f"{tag}_tco{j}"],
                                      K, vox, trunc, count_steps=True)
            for g_, key in zip(got[:4], ("ray", "vert", "nrm", "hit")):
                assert_parity(g_, kv[f"{tag}_{key}{j}{name}"], f"{key}{j}{name}", exact=True)
            assert int(got[4].sum()) == int(kv[f"{tag}_steps{j}{name}"])
    pts = oracle.compute_points(kv[f"{tag}_depth2"], K)
    assert_parity(pts, kv[f"{tag}_points"], "points", exact=True)
    # the lookups used the camera -> volume pose of frame 2 = inverse of (Roc2, toc2)
    Roc = kv[f"{tag}_Roc2"].reshape(3, 3).astype(np.float64)
    assert np.allclose(Roc @ Roc.T, np.eye(3), atol=1e-5)
    fgbg = np.zeros(tsdf.shape + (2,), np.float32)
    oracle.update_fgbg_probs(kv[f"{tag}_mask"], kv[f"{tag}_occluded"], tsdf, wts, fgbg, kv[f"{tag}_Roc2"],
                             kv[f"{tag}_toc2"], K, vox)
    assert_parity(fgbg, kv[f"{tag}_fgbg"], "fgbg", exact=True)
    probs, vmask = oracle.compute_fg_probs(fgbg)
    assert_parity(probs, kv[f"{tag}_probs"], "fgProbs", exact=True)
    assert_parity(vmask, kv[f"{tag}_vmask"], "fgVolMask", exact=True)


def _run_frames(fv, make, step, finish):
    """Drive a pipeline (oracle or HIP) through the 4 recorded frames."""
    from tests.golden.make_golden import FR
    ids = [1, 2]
    make([(fv[f"obj{i}_center"], float(fv[f"obj{i}_size"])) for i in ids])
    per_frame = []
    for f in range(FR["frames"]):
        run_masks = f % FR["mask_every"] == 0
        masks = {i: fv[f"f{f}_obj{i}_mask"] for i in ids} if run_masks else {}
        poses = {i: fv[f"f{f}_obj{i}_t"] for i in ids}
        per_frame.append(step(fv[f"f{f}_depth"], fv[f"f{f}_camR"], fv[f"f{f}_camt"], poses, masks,
                              run_masks))
    return per_frame, finish()


def _check_frames(fv, per_frame, final, rtol, budget):
    for f, rec in enumerate(per_frame):
        assert sorted(rec["vis"]) == fv[f"f{f}_vis"].tolist(), f"visible set, frame {f}"
        assert (rec["seg"] != fv[f"f{f}_seg"]).mean() < 2e-3, f"segmentation, frame {f}"
        same = rec["seg"] == fv[f"f{f}_seg"]
        assert_parity(rec["ray"][same], fv[f"f{f}_ray"][same], f"raylengths frame {f}", rtol=rtol,
                      budget=5e-3)
        assert_parity(rec["norm"], fv[f"f{f}_norm"], f"normaliser frame {f}", rtol=rtol, budget=budget)
        assert_parity(rec["bg_assoc"], fv[f"f{f}_bg_assoc"], f"bg association frame {f}", rtol=rtol,
                      atol=1e-7, budget=budget)
    for key, val in final.items():
        if key.endswith("vmask"):
            assert (val != fv[key]).mean() < 1e-3, key
        else:
            assert_parity(val, fv[key], key, rtol=rtol, atol=1e-6, budget=budget)


def test_oracle_reproduces_frame_vectors(oracle, fv):
    from tests.golden.make_golden import FR
    from tests.oracle_pipeline import Affine32, OraclePipeline
    st = {}

    def make(objs):
        st["p"] = OraclePipeline(oracle, W, H, fv["K"], FR["bg_res"], FR["bg_voxel"], [0, 0, 1.28],
                                 FR["obj_res"], visibility_thresh=FR["vis"], boundary=FR["boundary"])
        for c, s in objs:
            st["p"].add_object(c, np.float32(s))

    def step(depth, R, t, poses, masks, run_masks):
        p = st["p"]
        p.process_frame(depth, Affine32(R, t), {i: Affine32(np.eye(3, dtype=np.float32), tt)
                                               for i, tt in poses.items()}, masks, run_masks)
        return dict(vis=p.vis, seg=p.seg.copy(), ray=p.ray.copy(), norm=p.norm.copy(),
                    bg_assoc=p.bg_assoc.copy())

    def finish():
        p = st["p"]
        out = {"bg_tsdf": p.bg["tsdf"], "bg_wts": p.bg["wts"], "points": p.points}
        for v in p.objects:
            for k in ("tsdf", "wts", "probs", "vmask", "assoc"):
                out[f"obj{v['id']}_{k}"] = v[k]
        return out

    per_frame, final = _run_frames(fv, make, step, finish)
    # same code, same flags; only libm's expf may differ between machines
    _check_frames(fv, per_frame, final, rtol=1e-5, budget=1e-3)


# ---- the HIP path against the vectors (GPU) -------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_reproduces_kernel_vectors(dev, kv, tag):
    from emfusion_amd import ops
    from tests.parity_util import dev_full, to_dev, to_np
    K, vox, trunc = kv["K"], float(kv[f"{tag}_voxel"]), float(kv[f"{tag}_trunc"])
    n = tuple(int(v) for v in kv[f"{tag}_res"])
    d_t = to_dev(np.zeros((n[2], n[1], n[0]), np.float32), dev)
    d_w = to_dev(np.zeros((n[2], n[1], n[0]), np.float32), dev)
    il = dev_full((H, W), 0.0)
    ops.compute_inv_lambda(K, il)
    for i in range(3):
        ops.update_tsdf(to_dev(kv[f"{tag}_depth{i}"], dev), to_dev(kv[f"{tag}_assoc{i}"], dev), d_t, d_w,
                        kv[f"{tag}_Roc{i}"], kv[f"{tag}_toc{i}"], K, vox, trunc, 3.0,
                        inv_lambda=il if i else None)  # both forms of 1 / lambda
        if i != 1:
            assert_parity(to_np(d_t), kv[f"{tag}_tsdf{i}"], f"tsdf frame {i}", exact=True)
            assert_parity(to_np(d_w), kv[f"{tag}_wts{i}"], f"weights frame {i}", exact=True)
    g = dev_full(kv[f"{tag}_grads"].shape, 7.0)
    ops.compute_tsdf_grads(d_t, g)
    assert_parity(to_np(g), kv[f"{tag}_grads"], "grads", exact=True)
    rcp = ops.voxel_reciprocal(vox)
    for j in range(3):
        for name in (("", "_fg") if j == 0 else ("",)):
            fg = to_dev(kv[f"{tag}_fgmask"], dev) if name else None
            ray, vert, nrm = dev_full((H, W), 0.0), dev_full((H, W, 3), 0.0), dev_full((H, W, 3), 0.0)
            hit, st = dev_full((H, W), 0, np.uint8), dev_full((4,), 0, np.uint64)
            ops.raycast_tsdf(d_t, g if j == 1 else None, d_w, fg, ray, vert, nrm, hit, kv[f"{tag}_Rco{j}"],
                             kv[f"{tag}_tco{j}"], K, vox, trunc, st, rcp_voxel=rcp if j != 2 else 0.0)
            for got, key in zip((ray, vert, nrm, hit), ("ray", "vert", "nrm", "hit")):
                assert_parity(to_np(got), kv[f"{tag}_{key}{j}{name}"], f"{key}{j}{name}", exact=True)
            assert int(to_np(st)[0]) == int(kv[f"{tag}_steps{j}{name}"])
    pts = dev_full((H, W, 3), -1.0)
    ops.compute_points(to_dev(kv[f"{tag}_depth2"], dev), K, pts)
    assert_parity(to_np(pts), kv[f"{tag}_points"], "points", exact=True)
    fgbg = to_dev(np.zeros(kv[f"{tag}_fgbg"].shape, np.float32), dev)
    ops.update_fgbg_probs(to_dev(kv[f"{tag}_mask"], dev), to_dev(kv[f"{tag}_occluded"], dev), d_t, d_w, fgbg,
                          kv[f"{tag}_Roc2"], kv[f"{tag}_toc2"], K, vox)
    assert_parity(to_np(fgbg), kv[f"{tag}_fgbg"], "fgbg", exact=True)
    probs, vmask = dev_full(kv[f"{tag}_probs"].shape, 9.0), dev_full(kv[f"{tag}_vmask"].shape, 9, np.uint8)
    ops.compute_fg_probs(fgbg, probs, vmask)
    assert_parity(to_np(probs), kv[f"{tag}_probs"], "fgProbs", exact=True)
    assert_parity(to_np(vmask), kv[f"{tag}_vmask"], "fgVolMask", exact=True)


@pytest.mark.gpu
def test_hip_reproduces_frame_vectors(dev, fv):
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    from tests.golden.make_golden import FR
    from tests.parity_util import to_dev
    st = {}

    def make(objs):
        prm = pipeline.make_params(W, H, FR["bg_res"], FR["bg_voxel"], FR["obj_res"],
                                   visibility_thresh=FR["vis"], boundary=FR["boundary"],
                                   mask_frames=FR["mask_every"])
        assert np.allclose(np.array(prm.K, np.float32).reshape(3, 3), fv["K"])
        st["f"] = pipeline.Fusion(prm, None)
        for c, s in objs:
            st["f"].add_object(c, s)

    def step(depth, R, t, poses, masks, run_masks):
        f = st["f"]
        d = to_dev(depth)
        dm = {i: to_dev(m) for i, m in masks.items()}
        f.process_frame(image_view(d), R.reshape(-1), t,
                        {i: (np.eye(3, dtype=np.float32).reshape(-1), tt) for i, tt in poses.items()},
                        {i: image_view(m) for i, m in dm.items()}, run_masks)
        f.synchronize()
        return dict(vis=f.visible_objects(), seg=f.image("segmentation"), ray=f.image("raylengths"),
                    norm=f.image("assoc_norm"), bg_assoc=f.image("bg_assoc"))

    def finish():
        f = st["f"]
        out = {"bg_tsdf": f.volume("tsdf", 0), "bg_wts": f.volume("weights", 0), "points": f.image("points")}
        for i in (1, 2):
            out[f"obj{i}_tsdf"], out[f"obj{i}_wts"] = f.volume("tsdf", i), f.volume("weights", i)
            out[f"obj{i}_probs"], out[f"obj{i}_vmask"] = f.volume("fgprobs", i), f.volume("fgmask", i)
            out[f"obj{i}_assoc"] = f.image("obj_assoc", i)
        return out

    try:
        per_frame, final = _run_frames(fv, make, step, finish)
        _check_frames(fv, per_frame, final, rtol=1e-4, budget=1e-3)  # north-star tolerance
    finally:
        if "f" in st:
            st["f"].close()


# ---- next-row kernels (meshes, rendering, depth pre-processing): tests/golden/next_rows_v1.npz --------

@pytest.fixture(scope="module")
def nv():
    return np.load(GOLD / "next_rows_v1.npz")


@pytest.mark.parametrize("tag", ["cube", "ragged"])
def test_oracle_reproduces_next_row_vectors(oracle, kv, nv, tag):
    tsdf, wts, voxel = kv[f"{tag}_tsdf2"], kv[f"{tag}_wts2"], float(kv[f"{tag}_voxel"])
    for prefix, fg in (("mesh", None), ("fgmesh", kv[f"{tag}_fgmask"])):
        got = oracle.marching_cubes(tsdf, wts, voxel, fg=fg)
        for g, name in zip(got, ("v", "n", "t")):
            want = nv[f"{tag}_{prefix}_{name}"]
            assert g.shape == want.shape and g.tobytes() == want.tobytes(), (tag, prefix, name)
    assert len(nv[f"{tag}_mesh_v"]) > len(nv[f"{tag}_fgmesh_v"]) > 0
    if tag == "cube":
        rgb = oracle.render_phong(kv["cube_vert0"], kv["cube_nrm0"], nv["render_seg"], nv["render_cmap"])
        assert np.array_equal(rgb, nv["render_rgb"]) and rgb.any()
        lit = oracle.render_phong(kv["cube_vert0"], kv["cube_nrm0"], nv["render_seg"], nv["render_cmap"], (0.3, -0.2, 0.1))
        assert np.array_equal(lit, nv["render_rgb_light"]) and not np.array_equal(lit, rgb)
        out = oracle.preprocess_depth(nv["prep_in"], 7, 0.04, 4.5)
        assert np.allclose(out, nv["prep_out"], rtol=1e-6, atol=1e-7)  # expf: libm-dependent last bits
        assert np.all(out[nv["prep_in"] == 0] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["cube", "ragged"])
def test_hip_reproduces_next_row_vectors(dev, kv, nv, tag):
    from emfusion_amd import ops
    from tests.parity_util import dev_full, to_dev
    tsdf, wts, voxel = kv[f"{tag}_tsdf2"], kv[f"{tag}_wts2"], float(kv[f"{tag}_voxel"])
    for prefix, fg in (("mesh", None), ("fgmesh", kv[f"{tag}_fgmask"])):
        got = ops.extract_mesh(to_dev(tsdf), to_dev(wts), voxel, fg_mask=None if fg is None else to_dev(fg))
        for g, name in zip(got, ("v", "n", "t")):
            want = nv[f"{tag}_{prefix}_{name}"]
            assert g.shape == want.shape and g.tobytes() == want.tobytes(), (tag, prefix, name)
    if tag == "cube":
        h, w = nv["render_seg"].shape
        for key, light in (("render_rgb", (0.0, 0.0, 0.0)), ("render_rgb_light", (0.3, -0.2, 0.1))):
            img = dev_full((h, w, 3), 9, np.uint8)
            ops.render_phong(to_dev(kv["cube_vert0"]), to_dev(kv["cube_nrm0"]), to_dev(nv["render_seg"]),
                             nv["render_cmap"], img, light)
            assert np.array_equal(img.numpy(), nv[key])
        out = dev_full(nv["prep_in"].shape, -1.0)
        ops.preprocess_depth(to_dev(nv["prep_in"]), out, 7, 0.04, 4.5)
        assert np.allclose(out.numpy(), nv["prep_out"], rtol=1e-5, atol=1e-6)


# ---- tracker (f-1): tests/golden/tracking_v1.npz (make_golden_tracking.py) ---------------------------------

@pytest.fixture(scope="module")
def tv():
    return np.load(GOLD / "tracking_v1.npz")


TRACK_ITER = (1, 3, 40)


@pytest.mark.parametrize("k", [0, 1], ids=["background", "object"])
def test_oracle_reproduces_tracking_vectors(oracle, tv, k):
    from tests.oracle_tracking import OracleTracker
    for n_it in TRACK_ITER:
        tr = OracleTracker(oracle, tv[f"m{k}_tsdf"], tv[f"m{k}_wts"], float(tv[f"m{k}_vox"]))
        tr.prepare(tv[f"m{k}_R0"], tv[f"m{k}_t0"])
        for _ in range(n_it):
            tr.iterate(tv["points"], tv[f"m{k}_assoc"])
        p = f"m{k}_it{n_it}_"
        assert [tr.iterations, tr.accepted, int(tr.converged)] == tv[p + "counts"].tolist(), n_it
        # the oracle's sums run over OpenMP threads in double: reproducible to rounding, not to the bit
        assert np.allclose(tr.R, tv[p + "R"], atol=2e-6) and np.allclose(tr.t, tv[p + "t"], atol=2e-6), n_it
        assert abs(float(tr.mu) - float(tv[p + "mu"])) <= 1e-4 * float(tv[p + "mu"])
        h = tr.history[-1]
        assert np.abs(h["A"] - tv[p + "A"]).max() <= 1e-5 * np.abs(tv[p + "A"]).max()
        assert abs(h["err"] - tv[p + "err"][0]) <= 1e-5 * tv[p + "err"][0]
    assert tv[f"m{k}_it40_counts"][1] >= 5  # the vectors hold accepted and rejected steps
    assert tv[f"m{k}_it40_counts"][0] > tv[f"m{k}_it40_counts"][1]


@pytest.mark.gpu
def test_hip_reproduces_tracking_vectors(dev, tv):
    """Both models in lock-step through emf_hip_trackIterate against the committed LM histories."""
    import ctypes as C
    from emfusion_amd import _lib, ops
    from tests.parity_util import dev_full, to_dev
    H, W = tv["points"].shape[:2]
    keep, entries = [], []
    for k in (0, 1):
        vox = float(tv[f"m{k}_vox"])
        d = dict(tsdf=to_dev(tv[f"m{k}_tsdf"]), wts=to_dev(tv[f"m{k}_wts"]), assoc=to_dev(tv[f"m{k}_assoc"]),
                 ray=dev_full((H, W), 0.0), vert=dev_full((H, W, 3), 0.0), nrm=dev_full((H, W, 3), 0.0),
                 hit=dev_full((H, W), 0, np.uint8))
        keep.append(d)
        entries.append(ops.make_model(d["tsdf"], d["wts"], d["assoc"], d["ray"], d["vert"], d["nrm"], d["hit"], vox,
                                      float(np.float32(10 * vox)), 64.0, 0.02, 0.8, 1.0, model_id=k))
    table = ops.upload_models(entries)
    points = to_dev(tv["points"])
    per = ops.track_scratch_bytes(W, H)
    for n_it in TRACK_ITER:
        states = dev_full((2 * C.sizeof(_lib.EmfTrackState),), 0, np.uint8)
        scratch = dev_full((2 * per,), 0, np.uint8)
        ops.track_prepare(states, [(tv[f"m{k}_R0"], tv[f"m{k}_t0"]) for k in (0, 1)])
        ops.track_iterate(table, states, 2, points, _lib.EmfTrackParams.defaults(), scratch, per, n_it)
        for k, st in enumerate(ops.read_track_states(states, 2)):
            p = f"m{k}_it{n_it}_"
            if n_it <= 3:
                assert [st.iterations, st.accepted, int(st.converged)] == tv[p + "counts"].tolist(), (k, n_it)
            else:  # where the step-size test fires is a matter of the last bits: a few steps earlier or later
                assert int(st.converged) == tv[p + "counts"][2] and abs(st.iterations - tv[p + "counts"][0]) <= 6, (k, n_it)
            tol = 1e-4 if n_it > 3 else 2e-6  # the north-star tolerance once rounding has had 40 steps to grow
            assert np.abs(np.array(st.R, np.float32).reshape(3, 3) - tv[p + "R"]).max() < tol, (k, n_it)
            assert np.abs(np.array(st.t, np.float32) - tv[p + "t"]).max() < tol, (k, n_it)
            if n_it <= 3:
                A = np.array(st.A, np.float32).reshape(6, 6)
                assert np.abs(A - tv[p + "A"]).max() <= 2e-5 * np.abs(tv[p + "A"]).max(), (k, n_it)
