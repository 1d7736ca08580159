"""Evaluator-side result formatting: the oracle against the golden file produced by the reference's own
`convert_3d_box_to_kitti` / `KITTI3DEvaluator.process` / `NuscenesEvaluator.process` (tests/golden/make_format_golden.py), and the
GPU product (dd3d_amd.evaluators) against the same file.  Tolerances: strings, integers, float32 pass-through fields and the
2-decimal alpha exact; float64 angles / velocities 1e-12 absolute (libm atan2 / summation-order differences)."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ANG_TOL = 1e-12


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "format_results.json")) as f:
        return json.load(f)


def _check_convert(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got[:, :6], want[:, :6])  # W L H x y z: float32 values
    # rot_y = +-angle may sit on either side of the +-pi cut when the angle is pi within rounding
    d = np.abs(got[:, 6] - want[:, 6])
    d = np.minimum(d, np.abs(d - 2 * np.pi))
    assert d.max() <= ANG_TOL, d.max()
    da = np.abs(got[:, 7] - want[:, 7])
    da = np.minimum(da, np.abs(da - 6.28))  # alpha is rounded to 2 decimals: -3.14 and 3.14 are the same direction
    assert da.max() <= 1e-9, (da.max(), np.argmax(da))


def _check_json(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert list(g.keys()) == list(w.keys())
        for k in w:
            assert g[k] == w[k], (k, g[k], w[k])


def _check_rows(got_rows, want_rows):
    assert len(got_rows) == len(want_rows)
    for g, w in zip(got_rows, want_rows):
        assert g[:3] == w[:3]
        np.testing.assert_array_equal(np.float64(g[4:14]), np.float64(w[4:14]))
        assert abs(float(g[3]) - float(w[3])) <= 1e-9 and abs(float(g[14]) - float(w[14])) <= ANG_TOL
        assert float(g[15]) == float(w[15])


def _check_csv(got, want):
    """Submission text: every token character-identical (float32 columns print as float32), except the float64 rot_y column, where a
    last-place difference of the device's atan2 is allowed."""
    assert len(got) == len(want)
    for g, w in zip(got, want):
        gl, wl = g.splitlines(), w.splitlines()
        assert len(gl) == len(wl)
        for a, b in zip(gl, wl):
            ta, tb = a.split(" "), b.split(" ")
            assert len(ta) == len(tb)
            for col, (x, y) in enumerate(zip(ta, tb)):
                assert x == y or (col == 14 and abs(float(x) - float(y)) <= ANG_TOL), (col, x, y)


def _check_nusc(got, want):
    assert set(got.keys()) == set(want.keys())
    for tok in want:
        assert len(got[tok]) == len(want[tok])
        for g, w in zip(got[tok], want[tok]):
            for k in ("sample_token", "rotation", "translation", "size", "detection_name", "detection_score", "attribute_name"):
                assert g[k] == w[k], (k, g[k], w[k])
            np.testing.assert_allclose(g["velocity"], w["velocity"], rtol=0, atol=ANG_TOL)


# ---- oracle vs the reference's own outputs (CPU) ------------------------------------------------------------------------------------
def test_oracle_convert_matches_reference(golden):
    from oracle import format_oracle as F
    got = [[float(x) for x in F.convert_3d_box_to_kitti(np.float32(v))] for v in golden["convert"]["box3d_vec"]]
    _check_convert(got, golden["convert"]["kitti"])


def test_oracle_kitti_process_matches_reference(golden):
    from oracle import format_oracle as F
    k = golden["kitti"]
    as_json, pred_rows, gt_rows = F.kitti_process(k["inputs"], k["outputs"], k["class_names"], k["dataset_dicts"])
    _check_json(as_json, k["predictions_as_json"])
    for g, w in zip(pred_rows, k["predictions_rows"]):
        _check_rows(g, w)
    import pandas as pd
    to_csv = lambda rows: pd.DataFrame(rows).to_csv(sep=" ", header=False, index=False)  # noqa: E731
    _check_csv([to_csv(r) for r in pred_rows], k["predictions_csv"])
    _check_csv([to_csv(r) for r in gt_rows], k["groundtruth_csv"])


def test_oracle_nusc_process_matches_reference(golden):
    from oracle import format_oracle as F
    n = golden["nusc"]
    as_json, results = F.nusc_process(n["inputs"], n["outputs"])
    _check_json(as_json, n["predictions_as_json"])
    _check_nusc(results, n["sample_results"])


def test_attribute_names_and_box_mode():
    from dd3d_amd.evaluators.formatting import xyxy_to_xywh
    from dd3d_amd.evaluators.nuscenes_evaluator import NuscenesEvaluator, attribute_name
    assert attribute_name("car", 1) == "vehicle.parked" and attribute_name("motorcycle", 2) == "cycle.with_rider"
    assert attribute_name("pedestrian", 2) == "pedestrian.sitting_lying_down" and attribute_name("barrier", 1) == ""
    assert xyxy_to_xywh([1.0, 2.0, 4.5, 8.0]) == [1.0, 2.0, 3.5, 6.0]
    ev = NuscenesEvaluator(None, "nusc_test", None)
    with pytest.raises(ValueError):  # incomplete camera group, as postprocessing.py:117-119
        ev.process([{"sample_token": "a"}] * 5, [None] * 5)
    with pytest.raises(NotImplementedError):
        ev.evaluate()


# ---- GPU product vs the reference's own outputs -----------------------------------------------------------------------------------------
def _instances(rec, device, nusc):
    from dd3d_amd.structures import Boxes, GenericBoxes3D, Instances
    t = lambda k, dt: torch.tensor(rec[k], dtype=dt, device=device)  # noqa: E731
    inst = Instances((360, 640))
    n = len(rec["scores"])
    inst.pred_boxes = Boxes(t("pred_boxes", torch.float32).reshape(n, 4))
    inst.pred_classes = t("pred_classes", torch.int64)
    inst.scores = t("scores", torch.float32)
    inst.scores_3d = t("scores_3d", torch.float32)
    v = t("box3d_vec", torch.float32).reshape(n, 10)
    inst.pred_boxes3d = GenericBoxes3D(v[:, :4], v[:, 4:7], v[:, 7:])
    if nusc:
        g = t("box3d_global_vec", torch.float32).reshape(n, 10)
        inst.pred_boxes3d_global = GenericBoxes3D(g[:, :4], g[:, 4:7], g[:, 7:])
        inst.pred_attributes = t("pred_attributes", torch.int64)
        inst.pred_speeds = t("pred_speeds", torch.float32)
    return {"instances": inst}


@pytest.mark.gpu
def test_gpu_convert_matches_reference(golden, hiplib):
    from dd3d_amd.evaluators import convert_3d_box_to_kitti, format_boxes3d
    from dd3d_amd.structures import GenericBoxes3D
    v = torch.tensor(golden["convert"]["box3d_vec"], dtype=torch.float32, device="cuda")
    got = format_boxes3d(v)
    _check_convert(got[:, :8], golden["convert"]["kitti"])
    assert format_boxes3d(v[:0]).shape == (0, 10)
    one = convert_3d_box_to_kitti(GenericBoxes3D(v[5:6, :4], v[5:6, 4:7], v[5:6, 7:]))
    assert [type(x).__name__ for x in one] == ["float32"] * 6 + ["float", "float64"]
    _check_convert([[float(x) for x in one]], golden["convert"]["kitti"][5:6])
    # host tensors are uploaded, not computed on the CPU
    _check_convert(format_boxes3d(v.cpu())[:, :8], golden["convert"]["kitti"])


@pytest.mark.gpu
def test_gpu_kitti_process_matches_reference(golden, tmp_path, hiplib):
    from dd3d_amd.evaluators import KITTI3DEvaluator
    k = golden["kitti"]
    ev = KITTI3DEvaluator("kitti_3d_val", dataset_dicts=k["dataset_dicts"], class_names=k["class_names"])
    outputs = [_instances(r, "cuda", False) for r in k["outputs"]]
    ev.process(k["inputs"], outputs)
    _check_json(ev._predictions_as_json, k["predictions_as_json"])
    for df, want in zip(ev._predictions_kitti_format, k["predictions_rows"]):
        _check_rows(df.values.tolist(), want)
    # submission text: character-identical to the reference's DataFrames written by prepare_kitti3d_submission
    csv = [df.to_csv(sep=" ", header=False, index=False) for df in ev._predictions_kitti_format]
    _check_csv(csv, k["predictions_csv"])
    _check_csv([df.to_csv(sep=" ", header=False, index=False) for df in ev._groundtruth_kitti_format], k["groundtruth_csv"])
    sub = str(tmp_path / "submission")
    KITTI3DEvaluator.prepare_kitti3d_submission(ev._predictions_kitti_format, sub)
    assert sorted(os.listdir(sub)) == ["000000.txt", "000001.txt", "000002.txt"]
    _check_csv([open(os.path.join(sub, "000000.txt")).read()], k["predictions_csv"][:1])


@pytest.mark.gpu
def test_gpu_nusc_process_matches_reference(golden, hiplib):
    from dd3d_amd.evaluators import NuscenesEvaluator
    n = golden["nusc"]
    ev = NuscenesEvaluator("/nonexistent", "nusc_val", None)
    ev.process(n["inputs"], [_instances(r, "cuda", True) for r in n["outputs"]])
    _check_json(ev._predictions_as_json, n["predictions_as_json"])
    _check_nusc(dict(ev._nusc_sample_results), n["sample_results"])
