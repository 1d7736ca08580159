"""Input-side resize: Pillow is the reference's actual resizer ([ext] detectron2 ResizeTransform.apply_image -> PIL BILINEAR), and
it is installed here -- so both the oracle and the HIP kernel are pinned against the real library, bit for bit."""
import numpy as np
import pytest
import torch
from PIL import Image

SIZES = [(370, 1224, 384, 1270), (900, 1600, 896, 1593), (375, 1242, 320, 1060), (64, 80, 127, 161), (50, 70, 50, 35), (33, 47, 12, 9),
         (20, 20, 20, 20), (48, 64, 96, 64)]


def _pil(img_hwc, nh, nw):
    return np.asarray(Image.fromarray(img_hwc).resize((nw, nh), Image.BILINEAR))


@pytest.mark.parametrize("H,W,nh,nw", SIZES[:6])
def test_oracle_and_product_coefficients_match_pillow(H, W, nh, nw):
    from dd3d_amd.inputs import resample_coeffs
    from oracle import resize_oracle as R
    rng = np.random.default_rng(H * 7 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    got = R.resize_bilinear_u8(np.ascontiguousarray(img.transpose(2, 0, 1)), nh, nw).transpose(1, 2, 0)
    assert np.array_equal(got, _pil(img, nh, nw))
    for n_in, n_out in ((W, nw), (H, nh)):  # the product's own table generator == the oracle's
        for a, b in zip(resample_coeffs(n_in, n_out), R.resample_coeffs(n_in, n_out)):
            assert np.array_equal(a, b)


def test_size_rule_and_intrinsics():
    from dd3d_amd.inputs import resize_intrinsics, shortest_edge_size
    from oracle import resize_oracle as R
    for h, w, s, m in [(370, 1224, 384, 100000), (900, 1600, 896, 100000), (375, 1242, 320, 100000), (480, 640, 800, 1333), (1200, 800, 800, 1000)]:
        assert shortest_edge_size(h, w, s, m) == R.shortest_edge_size(h, w, s, m)
    assert shortest_edge_size(370, 1224, 384, 100000) == (384, 1270) and shortest_edge_size(900, 1600, 896, 100000) == (896, 1593)
    K = np.float32([[721.5377, 0, 609.5593], [0, 721.5377, 172.854], [0, 0, 1]])
    got = resize_intrinsics(K, 370, 1224, 384, 1270).numpy()
    assert np.allclose(got, R.resize_intrinsics(K, 370, 1224, 384, 1270)) and np.allclose(got[0], K[0] * (1270 / 1224)) and np.allclose(got[2], K[2])


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,nh,nw", SIZES)
def test_hip_resize_is_bit_identical_to_pillow(hiplib, H, W, nh, nw):
    from dd3d_amd.inputs import DeviceResizer
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    r = DeviceResizer("cuda")
    got = r(torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda(), nh, nw)
    assert np.array_equal(got.cpu().numpy().transpose(1, 2, 0), _pil(img, nh, nw))
    # straight into a slot of a larger canvas (the forward plan's padded input), neighbours untouched
    canvas = torch.full((3, nh + 5, nw + 9), 7, dtype=torch.uint8, device="cuda")
    r(torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda(), nh, nw, out=canvas)
    c = canvas.cpu().numpy()
    assert np.array_equal(c[:, :nh, :nw].transpose(1, 2, 0), _pil(img, nh, nw)) and (c[:, nh:, :] == 7).all() and (c[:, :, nw:] == 7).all()


@pytest.mark.gpu
def test_device_input_mapper_feeds_the_model(hiplib, kitti_dla34):
    """Raw 370x1224 frame -> mapper (device resize + intrinsics) -> model == PIL-resized frame through the same model."""
    from dd3d_amd.inputs import DeviceInputMapper
    from dd3d_amd.synthetic import KITTI_K
    from tests.util import gpu_model
    cfg, _, sd = kitti_dla34
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, (370, 1224, 3), dtype=np.uint8)
    K = torch.tensor(KITTI_K)
    mapper = DeviceInputMapper(cfg)
    d = mapper(torch.from_numpy(np.ascontiguousarray(raw.transpose(2, 0, 1))), K, image_id=0)
    assert tuple(d["image"].shape) == (3, 384, 1270) and (d["height"], d["width"]) == (370, 1224)
    ref_img = torch.from_numpy(np.ascontiguousarray(_pil(raw, 384, 1270).transpose(2, 0, 1)))
    assert torch.equal(d["image"].cpu(), ref_img)
    model = gpu_model(cfg, sd, use_graph=False)
    a = model([d])[0]["instances"]
    b = model([dict(d, image=ref_img)])[0]["instances"]
    assert len(a) == len(b) > 0 and torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and tuple(a.image_size) == (370, 1224)
