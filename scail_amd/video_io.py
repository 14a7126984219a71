"""File IO either side of the sampling path (reference sample_video.py:181-217 writers, :300-351 readers).

The reference decodes the driving video with decord and writes mp4 through imageio/ffmpeg; neither exists in this
image (no H.264 / HEVC codec offline), so the same roles are filled with what Pillow can do:

    read   reference image      any Pillow format                        -> load_image_to_tensor_chw_normalized
           driving / GT video   directory of frames, .npy / .pt arrays,
                                animated WebP / PNG (APNG) / GIF,
                                .mp4 / .mov with Motion-JPEG samples     -> load_video_for_pose_sample
    write  result video         animated WebP (default; lossless), APNG,
                                GIF, .npy, a directory of PNG frames,
                                .mp4 (Motion JPEG, quality 95, 4:4:4)    -> save_video_as_grid / save_multi_video_grid

The .mp4 container itself is written and parsed here (ISO/IEC 14496-12 boxes: ftyp / mdat / moov.trak.mdia.minf.stbl with
stsd / stts / stsc / stsz / stco; sample entry ``mp4v`` + esds objectTypeIndication 0x6C = JPEG, which ffmpeg, VLC and
QuickTime play); the only video codec Pillow carries is JPEG, so that is what the samples are.  An .mp4 whose track is
H.264 / HEVC / AV1 (the reference's own examples are ``avc1``) is rejected by NAME OF ITS CODEC, not guessed at.

Tensor conventions are the reference's: readers return uint8 (T, H, W, C) like ``load_video_for_pose_sample`` /
a [-1, 1] (1, C, H, W) image; writers take (B, T, C, H, W) in [0, 1] and quantise with ``(255 * x).astype(uint8)``
(truncation, :188 / :212).
"""
from __future__ import annotations

import io
import os
import struct
from typing import List, Sequence

import numpy as np
import torch

_ANIMATED = (".webp", ".png", ".apng", ".gif")
_MP4 = (".mp4", ".mov", ".m4v")
_NO_CODEC = (".mkv", ".avi", ".webm")
_JPEG_ENTRIES = (b"jpeg", b"mjpa", b"mjpb", b"MJPG", b"mjpg")


def _pil():
    from PIL import Image, ImageSequence
    return Image, ImageSequence


def load_image_to_tensor_chw_normalized(path: str) -> torch.Tensor:
    """(1, 3, H, W) float32 in [-1, 1] (reference helper of the same name, sample_video.py:343 context)."""
    Image, _ = _pil()
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.float32)
    t = torch.from_numpy(arr).permute(2, 0, 1).unsqueeze(0)
    return (t - 127.5) / 127.5


def load_video_for_pose_sample(path: str, max_frames: int = None) -> torch.Tensor:
    """(T, H, W, 3) uint8 (what the decord-based reference helper of the same name returns, sample_video.py:48-54)."""
    ext = os.path.splitext(path)[1].lower()
    if os.path.isdir(path):
        Image, _ = _pil()
        names = sorted(n for n in os.listdir(path) if os.path.splitext(n)[1].lower() in (".png", ".jpg", ".jpeg", ".bmp", ".webp"))
        if not names:
            raise FileNotFoundError(f"no image frames in {path}")
        frames = []
        for n in names[:max_frames]:
            with Image.open(os.path.join(path, n)) as im:
                frames.append(np.asarray(im.convert("RGB")))
        arr = np.stack(frames)
    elif ext == ".npy":
        arr = np.load(path)
    elif ext in (".pt", ".pth"):
        arr = torch.load(path, map_location="cpu").numpy()
    elif ext in _ANIMATED:
        Image, ImageSequence = _pil()
        with Image.open(path) as im:
            arr = np.stack([np.asarray(f.convert("RGB")) for f in ImageSequence.Iterator(im)])
    elif ext in _MP4:
        arr = read_mp4(path, max_frames=max_frames)
    elif ext in _NO_CODEC:
        raise RuntimeError(f"{path}: no video decoder in this environment (decord / ffmpeg are absent offline); "
                           "pass a directory of frames, an .npy / .pt array (T,H,W,3), an animated WebP / PNG / GIF or a Motion-JPEG .mp4")
    else:
        raise ValueError(f"unsupported driving-video input {path}")
    if arr.ndim != 4 or arr.shape[-1] != 3:
        raise ValueError(f"{path}: expected (T, H, W, 3), got {arr.shape}")
    if max_frames is not None:
        arr = arr[:max_frames]
    if arr.dtype != np.uint8:
        arr = np.clip(np.rint(arr), 0, 255).astype(np.uint8)
    return torch.from_numpy(np.ascontiguousarray(arr))


def _to_u8_frames(vid: torch.Tensor) -> List[np.ndarray]:
    """vid (T, C, H, W) in [0, 1] -> list of (H, W, C) uint8, quantised like the reference (:211-212)."""
    return [(255.0 * f.permute(1, 2, 0)).cpu().numpy().astype(np.uint8) for f in vid]


def _write_frames(frames: Sequence[np.ndarray], path: str, fps: float):
    ext = os.path.splitext(path)[1].lower()
    if ext in _NO_CODEC:
        raise RuntimeError(f"{path}: no video encoder in this environment (imageio / ffmpeg are absent offline); "
                           "use .webp (lossless, default), .png (APNG), .gif, .npy, a directory or .mp4 (Motion JPEG)")
    if ext in _MP4:
        write_mp4(frames, path, fps)
        return
    if ext == ".npy":
        np.save(path, np.stack(frames))
        return
    Image, _ = _pil()
    ims = [Image.fromarray(f) for f in frames]
    dur = max(1, int(round(1000.0 / float(fps))))
    if ext == "":
        os.makedirs(path, exist_ok=True)
        for i, im in enumerate(ims):
            im.save(os.path.join(path, f"{i:06d}.png"))
    elif ext == ".webp":
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=dur, loop=0, lossless=True, method=0)
    elif ext in (".png", ".apng"):
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=dur, loop=0, format="PNG")
    elif ext == ".gif":
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=dur, loop=0)
    else:
        raise ValueError(f"unsupported output format {ext}")


# ---- ISO base media (.mp4 / .mov) container with Motion-JPEG samples ---------------------------------------------------------------
def _box(tp: bytes, *payload: bytes) -> bytes:
    body = b"".join(payload)
    return struct.pack(">I4s", 8 + len(body), tp) + body


def _full(tp: bytes, version: int, flags: int, *payload: bytes) -> bytes:
    return _box(tp, struct.pack(">I", (version << 24) | flags), *payload)


def _descr(tag: int, body: bytes) -> bytes:
    n = len(body)                                        # MPEG-4 descriptor: tag + 4-byte expandable length
    return bytes([tag, 0x80 | (n >> 21) & 0x7F, 0x80 | (n >> 14) & 0x7F, 0x80 | (n >> 7) & 0x7F, n & 0x7F]) + body


def write_mp4(frames: Sequence[np.ndarray], path: str, fps: float = 5, quality: int = 95) -> None:
    """frames: (H, W, 3) uint8 each -> one video track, one JPEG (4:4:4, no chroma subsampling) per sample, constant frame rate.
    Layout ftyp | mdat | moov (what imageio/ffmpeg writes without faststart; the reference's writer, sample_video.py:196-198)."""
    Image, _ = _pil()
    if len(frames) == 0:
        raise ValueError("write_mp4: no frames")
    H, W = frames[0].shape[:2]
    samples = []
    for f in frames:
        if f.shape != (H, W, 3) or f.dtype != np.uint8:
            raise ValueError(f"write_mp4: frames must be uint8 (H, W, 3) of one size, got {f.dtype} {f.shape}")
        buf = io.BytesIO()
        Image.fromarray(f).save(buf, format="JPEG", quality=quality, subsampling=0)
        samples.append(buf.getvalue())
    timescale = 90000
    delta = max(1, int(round(timescale / float(fps))))
    n = len(samples)
    duration = n * delta
    ftyp = _box(b"ftyp", b"isom", struct.pack(">I", 512), b"isomiso2mp41")
    mdat = _box(b"mdat", *samples)
    first = len(ftyp) + 8                                 # file offset of the first sample (one chunk holds them all)
    if first + sum(len(x) for x in samples) >= 1 << 32:
        raise ValueError("write_mp4: more than 4 GiB of samples (co64 / 64-bit mdat are not written)")
    unity = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration), struct.pack(">IH", 0x10000, 0x0100), bytes(10), unity,
                 bytes(24), struct.pack(">I", 2))
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration), bytes(8), struct.pack(">HHHH", 0, 0, 0, 0), unity,
                 struct.pack(">II", W << 16, H << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration), struct.pack(">HH", 0x55C4, 0))
    hdlr = _full(b"hdlr", 0, 0, bytes(4), b"vide", bytes(12), b"VideoHandler\0")
    esds = _full(b"esds", 0, 0, _descr(0x03, struct.pack(">HB", 1, 0) +
                                       _descr(0x04, struct.pack(">BB", 0x6C, 0x11) + bytes(3) + struct.pack(">II", 0, 0)) +
                                       _descr(0x06, b"\x02")))
    entry = _box(b"mp4v", bytes(6), struct.pack(">H", 1), bytes(16), struct.pack(">HH", W, H), struct.pack(">II", 0x480000, 0x480000),
                 bytes(4), struct.pack(">H", 1), bytes(32), struct.pack(">Hh", 24, -1), esds)
    stbl = _box(b"stbl",
                _full(b"stsd", 0, 0, struct.pack(">I", 1), entry),
                _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta)),
                _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1)),
                _full(b"stsz", 0, 0, struct.pack(">II", 0, n), b"".join(struct.pack(">I", len(x)) for x in samples)),
                _full(b"stco", 0, 0, struct.pack(">II", 1, first)))
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1), _full(b"url ", 0, 1)))
    minf = _box(b"minf", _full(b"vmhd", 0, 1, bytes(8)), dinf, stbl)
    moov = _box(b"moov", mvhd, _box(b"trak", tkhd, _box(b"mdia", mdhd, hdlr, minf)))
    with open(path, "wb") as fh:
        fh.write(ftyp + mdat + moov)


def _children(buf: bytes, start: int, end: int):
    off = start
    while off + 8 <= end:
        size, tp = struct.unpack(">I4s", buf[off:off + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", buf[off + 8:off + 16])[0]
            hdr = 16
        elif size == 0:
            size = end - off
        if size < hdr or off + size > end:
            raise ValueError(f"malformed box {tp!r} at byte {off}")
        yield tp, off + hdr, off + size
        off += size


def _find(buf: bytes, start: int, end: int, *path: bytes):
    for tp, a, b in _children(buf, start, end):
        if tp == path[0]:
            return (a, b) if len(path) == 1 else _find(buf, a, b, *path[1:])
    return None


def _descr_at(buf: bytes, pos: int):
    """MPEG-4 descriptor at pos -> (tag, body start, body end)."""
    tag, n, pos = buf[pos], 0, pos + 1
    for _ in range(4):
        c = buf[pos]
        pos += 1
        n = (n << 7) | (c & 0x7F)
        if not c & 0x80:
            break
    return tag, pos, pos + n


def _esds_object_type(buf: bytes, start: int, end: int):
    """objectTypeIndication of the DecoderConfigDescriptor inside the esds box among the child boxes [start, end) of a sample entry."""
    es = _find(buf, start, end, b"esds")
    if es is None:
        return None
    tag, a, b = _descr_at(buf, es[0] + 4)                                # ES_Descriptor behind version / flags
    if tag != 0x03:
        return None
    flags = buf[a + 2]
    a += 3 + (2 if flags & 0x80 else 0) + (2 if flags & 0x20 else 0)
    if flags & 0x40:
        a += 1 + buf[a]
    while a < b:
        tag, x, y = _descr_at(buf, a)
        if tag == 0x04:
            return buf[x]
        a = y
    return None


def read_mp4(path: str, max_frames: int = None) -> np.ndarray:
    """(T, H, W, 3) uint8 from the first video track of an ISO base media file whose samples are JPEG images (what write_mp4 writes; also
    QuickTime 'jpeg' / 'mjpa' and AVI-style 'MJPG' entries).  Any other codec raises RuntimeError naming it."""
    Image, _ = _pil()
    with open(path, "rb") as fh:
        buf = fh.read()
    moov = _find(buf, 0, len(buf), b"moov")
    if moov is None:
        raise ValueError(f"{path}: not an ISO base media file (no moov box)")
    for tp, a, b in _children(buf, *moov):
        if tp != b"trak":
            continue
        hd = _find(buf, a, b, b"mdia", b"hdlr")
        if hd is None or buf[hd[0] + 8:hd[0] + 12] != b"vide":
            continue
        stbl = _find(buf, a, b, b"mdia", b"minf", b"stbl")
        if stbl is None:
            raise ValueError(f"{path}: video track without a sample table")
        box = {t: (x, y) for t, x, y in _children(buf, *stbl)}
        sd = box[b"stsd"][0]
        codec = buf[sd + 12:sd + 16]
        is_jpeg = codec in _JPEG_ENTRIES
        if codec == b"mp4v":                                            # MPEG-4 systems entry: the codec is esds' objectTypeIndication
            oti = _esds_object_type(buf, sd + 8 + 8 + 78, box[b"stsd"][1])      # boxes behind the 78-byte VisualSampleEntry fields of entry 0
            is_jpeg = oti == 0x6C
            if not is_jpeg:
                codec = b"mp4v/0x%02X" % (oti if oti is not None else 0)
        if not is_jpeg:
            raise RuntimeError(f"{path}: video codec {codec.decode('latin1')!r} needs a decoder this environment does not have (decord / ffmpeg "
                               "are absent offline; Pillow decodes JPEG only) -- convert the clip to a directory of frames, an .npy array or a "
                               "Motion-JPEG .mp4 (ffmpeg -i in.mp4 -c:v mjpeg -q:v 2 out.mp4)")
        zs = box[b"stsz"][0]
        uniform, n = struct.unpack(">II", buf[zs + 4:zs + 12])
        sizes = [uniform] * n if uniform else list(struct.unpack(f">{n}I", buf[zs + 12:zs + 12 + 4 * n]))
        if b"stco" in box:
            cs = box[b"stco"][0]
            nc = struct.unpack(">I", buf[cs + 4:cs + 8])[0]
            chunks = struct.unpack(f">{nc}I", buf[cs + 8:cs + 8 + 4 * nc])
        else:
            cs = box[b"co64"][0]
            nc = struct.unpack(">I", buf[cs + 4:cs + 8])[0]
            chunks = struct.unpack(f">{nc}Q", buf[cs + 8:cs + 8 + 8 * nc])
        sc = box[b"stsc"][0]
        ne = struct.unpack(">I", buf[sc + 4:sc + 8])[0]
        runs = [struct.unpack(">III", buf[sc + 8 + 12 * i:sc + 20 + 12 * i]) for i in range(ne)]      # (first chunk, samples per chunk, description)
        frames, si = [], 0
        for ci in range(nc):
            per = [r for r in runs if r[0] <= ci + 1][-1][1]
            off = chunks[ci]
            for _ in range(per):
                if si >= n or (max_frames is not None and si >= max_frames):
                    break
                with Image.open(io.BytesIO(buf[off:off + sizes[si]])) as im:
                    frames.append(np.asarray(im.convert("RGB")))
                off += sizes[si]
                si += 1
        if not frames:
            raise ValueError(f"{path}: the video track holds no samples")
        return np.stack(frames)
    raise ValueError(f"{path}: no video track")


def save_video_as_grid(video_batch: torch.Tensor, save_path: str, fps: float = 5, ext: str = ".webp") -> List[str]:
    """video_batch (B, T, C, H, W) in [0, 1] -> save_path/000000<ext>, ... (reference save_video_as_grid_and_mp4, :201-217)."""
    os.makedirs(save_path, exist_ok=True)
    out = []
    for i, vid in enumerate(video_batch):
        p = os.path.join(save_path, f"{i:06d}{ext}")
        _write_frames(_to_u8_frames(vid), p, fps)
        out.append(p)
    return out


def save_multi_video_grid(video_batches: Sequence[torch.Tensor], save_dir: str, fps: float = 5, key: str = "0_output",
                          ext: str = ".webp") -> List[str]:
    """Several (B, T, C, H, W) clips side by side, frame layout "n c h w -> h (n w) c"
    (reference save_multi_video_grid_and_mp4, :181-198): save_dir/<key>_000000<ext>."""
    os.makedirs(save_dir, exist_ok=True)
    multi = torch.stack([v.float().cpu() for v in video_batches], dim=2)           # B T N C H W
    out = []
    for i, mv in enumerate(multi):
        frames = []
        for fr in mv:                                                              # N C H W
            n, c, h, w = fr.shape
            grid = fr.permute(2, 0, 3, 1).reshape(h, n * w, c)
            frames.append((255.0 * grid).numpy().astype(np.uint8))
        p = os.path.join(save_dir, f"{key}_{i:06d}{ext}")
        _write_frames(frames, p, fps)
        out.append(p)
    return out
