"""File IO either side of the path (scail_amd/video_io.py; reference sample_video.py:48-54, 181-217): round trips through
every supported container, the reference's frame layout / quantisation, loud failure for formats that need absent codecs."""
import os

import numpy as np
import pytest
import torch

from scail_amd import preprocess, video_io


def _clip(T=5, H=12, W=20, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(1, T, 3, H, W, generator=g)


@pytest.mark.parametrize("ext", [".webp", ".png", ".npy", ""])
def test_lossless_round_trip(tmp_path, ext):
    vid = _clip()
    paths = video_io.save_video_as_grid(vid, str(tmp_path), fps=16, ext=ext)
    assert [os.path.basename(p) for p in paths] == [f"000000{ext}"]
    back = video_io.load_video_for_pose_sample(paths[0])
    want = (255.0 * vid[0].permute(0, 2, 3, 1)).numpy().astype(np.uint8)          # truncation, like the reference
    assert back.dtype == torch.uint8 and tuple(back.shape) == want.shape
    assert np.array_equal(back.numpy(), want)


def test_gif_round_trip_is_close(tmp_path):
    vid = torch.zeros(1, 3, 3, 8, 8)
    vid[0, 1, 0] = 1.0
    vid[0, 2, 2] = 0.5
    p = video_io.save_video_as_grid(vid, str(tmp_path), fps=8, ext=".gif")[0]
    back = video_io.load_video_for_pose_sample(p).float() / 255.0
    assert back.shape == (3, 8, 8, 3) and float((back - vid[0].permute(0, 2, 3, 1)).abs().max()) < 0.05


def test_multi_video_grid_layout(tmp_path):
    a, b = _clip(seed=1), _clip(seed=2)
    p = video_io.save_multi_video_grid([a, b], str(tmp_path), fps=16, key="3_concat", ext=".png")[0]
    assert os.path.basename(p) == "3_concat_000000.png"
    back = video_io.load_video_for_pose_sample(p)
    assert tuple(back.shape) == (5, 12, 40, 3)                                    # "n c h w -> h (n w) c"
    q = lambda v: (255.0 * v[0].permute(0, 2, 3, 1)).numpy().astype(np.uint8)
    assert np.array_equal(back[:, :, :20].numpy(), q(a)) and np.array_equal(back[:, :, 20:].numpy(), q(b))


def test_formats_that_need_codecs_fail_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no video encoder"):
        video_io.save_video_as_grid(_clip(), str(tmp_path), ext=".webm")
    f = tmp_path / "rendered.mkv"
    f.write_bytes(b"\x00")
    with pytest.raises(RuntimeError, match="no video decoder"):
        video_io.load_video_for_pose_sample(str(f))
    g = tmp_path / "rendered.mp4"
    g.write_bytes(b"\x00")
    with pytest.raises(ValueError, match="not an ISO base media file"):
        video_io.load_video_for_pose_sample(str(g))


def _smooth_clip(T=6, H=48, W=64):
    t, y, x = torch.meshgrid(torch.arange(T), torch.arange(H), torch.arange(W), indexing="ij")
    v = torch.stack([0.5 + 0.5 * torch.sin(0.11 * x + 0.3 * t), 0.5 + 0.5 * torch.cos(0.07 * y - 0.2 * t), (x + y + 3.0 * t) / (H + W + 3.0 * T)], 1)
    return v[None].float()                                               # (1, T, 3, H, W) in [0, 1]


def test_mp4_motion_jpeg_round_trip(tmp_path):
    """The reference writes <index>.mp4 (sample_video.py:201-217); here the container is written / parsed by video_io itself with Motion-JPEG
    samples (the one codec Pillow has): frame count, size and order survive, pixels to JPEG accuracy; max_frames truncates."""
    vid = _smooth_clip()
    p = video_io.save_video_as_grid(vid, str(tmp_path), fps=16, ext=".mp4")[0]
    assert os.path.basename(p) == "000000.mp4"
    raw = open(p, "rb").read()
    assert raw[4:8] == b"ftyp" and b"moov" in raw and b"mp4v" in raw and b"stsz" in raw
    back = video_io.load_video_for_pose_sample(p)
    want = (255.0 * vid[0].permute(0, 2, 3, 1)).numpy().astype(np.uint8)
    assert back.dtype == torch.uint8 and tuple(back.shape) == want.shape
    err = np.abs(back.numpy().astype(np.int32) - want.astype(np.int32))
    assert err.max() <= 6 and err.mean() < 1.0, (err.max(), err.mean())
    # frame ORDER: every decoded frame is closest to the frame of its own index
    d = np.abs(back.numpy().astype(np.int32)[:, None] - want.astype(np.int32)[None]).mean(axis=(2, 3, 4))
    assert (d.argmin(axis=1) == np.arange(want.shape[0])).all()
    assert tuple(video_io.load_video_for_pose_sample(p, max_frames=2).shape) == (2, 48, 64, 3)


def test_mp4_with_another_codec_is_rejected_by_name(tmp_path):
    """An .mp4 whose video track is H.264 (sample entry avc1 -- what the reference's own examples are) names its codec instead of decoding garbage."""
    p = video_io.save_video_as_grid(_smooth_clip(T=2), str(tmp_path), fps=8, ext=".mp4")[0]
    raw = open(p, "rb").read()
    q = tmp_path / "h264.mp4"
    q.write_bytes(raw.replace(b"mp4v", b"avc1"))
    with pytest.raises(RuntimeError, match="'avc1'"):
        video_io.load_video_for_pose_sample(str(q))
    i = raw.index(b"esds")
    j = raw.index(b"\x6c\x11", i)                                        # objectTypeIndication 0x6C (JPEG) -> 0x20 (MPEG-4 part 2 video)
    r = tmp_path / "mpeg4.mp4"
    r.write_bytes(raw[:j] + b"\x20" + raw[j + 1:])
    with pytest.raises(RuntimeError, match="mp4v/0x20"):
        video_io.load_video_for_pose_sample(str(r))


def test_request_assembly_from_files(tmp_path):
    """Reference image + driving video files -> the tensors the engine takes (sample_video.py:300-351)."""
    from PIL import Image
    from scail_amd import cli
    g = np.random.default_rng(0)
    Image.fromarray(g.integers(0, 255, (90, 160, 3), dtype=np.uint8)).save(tmp_path / "ref.png")
    np.save(tmp_path / "rendered.npy", g.integers(0, 255, (5, 100, 150, 3), dtype=np.uint8))
    img = video_io.load_image_to_tensor_chw_normalized(str(tmp_path / "ref.png"))
    assert img.shape == (1, 3, 90, 160) and -1.0 <= float(img.min()) and float(img.max()) <= 1.0
    req, (H, W) = cli.request_from_files(str(tmp_path / "ref.png"), str(tmp_path / "rendered.npy"), cli.TINY, device="cpu", text_dim=64)
    assert (H, W) == (64, 64) == tuple(preprocess.target_size((90, 160), [64, 64]))
    assert req["ref"].shape == (3, 1, 64, 64) and req["pose"].shape == (3, 5, 32, 32)
    assert float(req["pose"].abs().max()) <= 1.0 + 1e-6 and req["context"].shape[-1] == 64


def test_request_line_parsing(tmp_path):
    """'<prompt>@@<example_dir>' lines of the reference CLI (sample_video.py:76, :82-91, :284-300)."""
    from PIL import Image
    from scail_amd import cli
    d = tmp_path / "001"
    d.mkdir()
    with pytest.raises(FileNotFoundError, match="Reference image not found"):
        cli.parse_request(f"a girl@@{d}")
    Image.new("RGB", (8, 8)).save(d / "ref_image.png")
    with pytest.raises(FileNotFoundError, match="Pose video not found"):
        cli.parse_request(f"a girl@@{d}")
    (d / "rendered.mp4").write_bytes(b"\x00")
    np.save(d / "rendered.npy", np.zeros((2, 8, 8, 3), np.uint8))
    text, input_dir, image_path, pose_path = cli.parse_request(f"None@@{d}")
    assert text == "" and input_dir == str(d) and image_path.endswith("ref_image.png")
    assert pose_path.endswith("rendered.npy")                           # a decodable container wins over the mp4
    (d / "rendered_aligned.gif").write_bytes(b"x")
    assert cli.parse_request(f"x@@{d}")[3].endswith("rendered_aligned.gif")
    with pytest.raises(ValueError, match="@@"):
        cli.parse_request("no separator")
    f = tmp_path / "req.txt"
    f.write_text(f"a@@{d}\n\nb@@{d}\nc@@{d}\n")
    assert [c for _, c in cli.read_from_file(str(f), rank=1, world_size=2)] == [3]
    assert [t.split("@@")[0] for t, _ in cli.read_from_file(str(f))] == ["a", "b", "c"]
