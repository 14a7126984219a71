"""Geometry of the request preprocessing (scail_amd/preprocess.py; reference data_video.py:141-170,
sample_video.py:325-351).  The interpolation itself is torch's; torchvision is absent, so no pinned outputs."""
import pytest
import torch

from scail_amd import preprocess as P


def test_target_size_orientation():
    assert P.target_size((720, 1280), [512, 896]) == (512, 896)
    assert P.target_size((1280, 720), [512, 896]) == (896, 512)


@pytest.mark.parametrize("hw,size", [((90, 200), (64, 112)), ((200, 90), (112, 64)), ((64, 112), (64, 112)), ((50, 50), (64, 112))])
def test_resize_crop_shapes_and_center(hw, size):
    T = 3
    g = torch.Generator().manual_seed(0)
    v = torch.randint(0, 256, (T, 3, *hw), generator=g, dtype=torch.uint8)
    o = P.resize_for_rectangle_crop(v, size)
    assert o.shape == (T, 3, *size) and o.dtype == torch.uint8
    if hw == size:
        assert torch.equal(o, v)           # identity resize + zero crop
    with pytest.raises(NotImplementedError):
        P.resize_for_rectangle_crop(v, size, "random")


def test_center_crop_keeps_the_middle():
    # a frame whose columns encode their index: after an identity-height resize the crop must be the middle columns
    w = torch.arange(200, dtype=torch.float32).view(1, 1, 1, 200).expand(1, 1, 64, 200).contiguous()
    o = P.resize_for_rectangle_crop(w, (64, 112))
    assert o.shape == (1, 1, 64, 112)
    assert abs(float(o[0, 0, 0, 0]) - 44.0) < 1e-3 and abs(float(o[0, 0, 0, -1]) - 155.0) < 1e-3


def test_pose_normalisation_and_half_res():
    v = torch.full((2, 3, 64, 112), 255, dtype=torch.uint8)
    v[:, :, :, :56] = 0
    pose, smpl = P.prepare_pose_video(v, (64, 112))
    assert pose.shape == (2, 3, 64, 112) and smpl.shape == (2, 3, 32, 56)
    assert float(pose.min()) == -1.0 and float(pose.max()) == 1.0
    assert float(smpl[0, 0, 0, 0]) == -1.0 and float(smpl[0, 0, 0, -1]) == 1.0
