"""The video file hand-off of the reference's scripts without imageio / torchvision: `lavie_amd.video_io` writes
Motion-JPEG in an ISO base media file and reads it back.  CPU only.  The reference's writers sit on ffmpeg (H.264), absent here: the
file layout is checked structurally (box tree, sample tables, sample entry) and by round trip — no reference-made mp4 exists to compare with
("parity unpinned" for the container bytes; the frames themselves are checked against the input)."""
import io
import os
import struct

import numpy as np
import pytest
import torch

from lavie_amd import video_io as V


def frames(t=16, h=64, w=96, seed=0):
    """smooth moving gradients plus a little texture: what a JPEG at quality 9 / 10 keeps to > 35 dB"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    rng = np.random.default_rng(seed)
    tex = rng.normal(0, 2, (h, w, 3)).astype(np.float32)
    out = [np.stack([127 + 100 * np.sin(xx / 17 + k / 3), 127 + 100 * np.cos(yy / 13 - k / 5), 127 + 80 * np.sin((xx + yy) / 29 + k / 7)], -1) + tex
           for k in range(t)]
    return np.clip(np.stack(out), 0, 255).astype(np.uint8)


def psnr(a, b):
    d = a.astype(np.float32) - b.astype(np.float32)
    return 10 * np.log10(255.0 ** 2 / max(float(np.mean(d * d)), 1e-12))


def tree(buf, lo, hi, depth=0, out=None):
    out = [] if out is None else out
    containers = {b"moov", b"trak", b"mdia", b"minf", b"dinf", b"stbl"}
    for kind, a, b in V._boxes(buf, lo, hi):
        out.append((depth, kind, a, b))
        if kind in containers:
            tree(buf, a, b, depth + 1, out)
    return out


def test_mimwrite_layout_and_round_trip(tmp_path):
    """base/pipelines/sample.py:91: imageio.mimwrite(path, videos[0], fps=8, quality=9)"""
    fr = frames()
    path = str(tmp_path / "a_prompt.mp4")
    V.mimwrite(path, fr, fps=8, quality=9)
    buf = open(path, "rb").read()
    top = V._boxes(buf, 0, len(buf))
    assert [k for k, _, _ in top] == [b"ftyp", b"mdat", b"moov"]
    assert top[-1][2] == len(buf)                                              # the boxes tile the file exactly
    assert buf[8:12] == b"isom"
    t = tree(buf, 0, len(buf))
    kinds = [k for _, k, _, _ in t]
    for need in (b"mvhd", b"tkhd", b"mdhd", b"hdlr", b"vmhd", b"dref", b"stsd", b"stts", b"stsc", b"stsz", b"stco"):
        assert kinds.count(need) == 1, need
    pos = {k: (a, b) for _, k, a, b in t}
    a, _ = pos[b"tkhd"]
    w16, h16 = struct.unpack_from(">II", buf, a + 4 + 72)                     # 16.16 fixed point at the end of tkhd v0
    assert (w16 >> 16, h16 >> 16) == (96, 64)
    a, _ = pos[b"mdhd"]
    timescale, duration = struct.unpack_from(">II", buf, a + 12)
    assert duration / timescale == pytest.approx(16 / 8)                       # 16 frames at 8 fps = 2 s
    a, _ = pos[b"hdlr"]
    assert buf[a + 8:a + 12] == b"vide"
    a, b = pos[b"stsd"]
    assert buf[a + 12:a + 16] == b"mp4v" and b"esds" in buf[a:b]
    ew, eh = struct.unpack_from(">HH", buf, a + 8 + 8 + 8 + 16)               # VisualSampleEntry width / height
    assert (ew, eh) == (96, 64)
    a, _ = pos[b"stsz"]
    fixed, n = struct.unpack_from(">II", buf, a + 4)
    sizes = struct.unpack_from(f">{n}I", buf, a + 12)
    assert fixed == 0 and n == 16 and sum(sizes) == top[1][2] - top[1][1]       # the samples are the mdat payload, nothing else
    a, _ = pos[b"stco"]
    assert struct.unpack_from(">II", buf, a + 4) == (1, top[1][1])             # one chunk, starting at the first byte of mdat's payload
    assert buf[top[1][1]:top[1][1] + 2] == b"\xff\xd8"                         # ... which is a JPEG start-of-image
    # vsr/sample.py:85: torchvision.io.read_video(filename=..., pts_unit='sec', output_format='TCHW')
    v, a_, info = V.read_video(path, pts_unit="sec", output_format="TCHW")
    assert v.dtype == torch.uint8 and tuple(v.shape) == (16, 3, 64, 96) and a_.numel() == 0 and info["video_fps"] == pytest.approx(8.0)
    assert psnr(v.permute(0, 2, 3, 1).numpy(), fr) > 35.0
    v2, _, _ = V.read_video(path)
    assert tuple(v2.shape) == (16, 64, 96, 3) and torch.equal(v2, v.permute(0, 2, 3, 1))
    v3, _, _ = V.read_video(path, start_pts=0.5, end_pts=1.0, pts_unit="sec")  # frames 4 .. 8
    assert v3.shape[0] == 5 and torch.equal(v3, v2[4:9])


def test_write_video_tensor_and_fractional_rate(tmp_path):
    """interpolation/sample.py:299: torchvision.io.write_video(path, video_ [T, H, W, 3] uint8 tensor, fps=fps)"""
    fr = torch.from_numpy(frames(t=5, h=40, w=56, seed=3))
    path = str(tmp_path / "b.mp4")
    V.write_video(path, fr, fps=29.97)
    v, _, info = V.read_video(path)
    assert info["video_fps"] == pytest.approx(29.97, rel=1e-4) and tuple(v.shape) == (5, 40, 56, 3)
    assert psnr(v.numpy(), fr.numpy()) > 35.0


def test_quality_scale_orders_file_sizes(tmp_path):
    fr = frames(t=4)
    sizes = []
    for q in (2, 6, 10):
        p = str(tmp_path / f"q{q}.mp4")
        V.mimwrite(p, fr, fps=8, quality=q)
        sizes.append(os.path.getsize(p))
    assert sizes[0] < sizes[1] < sizes[2]


def test_bad_inputs_are_refused(tmp_path):
    p = str(tmp_path / "x.mp4")
    with pytest.raises(ValueError):
        V.mimwrite(p, np.zeros((4, 8, 8, 3), np.float32))                     # the pipelines hand over uint8
    with pytest.raises(ValueError):
        V.mimwrite(p, np.zeros((4, 8, 8), np.uint8))
    with pytest.raises(ValueError):
        V.mimwrite(p, np.zeros((0, 8, 8, 3), np.uint8))
    with pytest.raises(ValueError):
        V.mimwrite(p, np.zeros((2, 8, 8, 3), np.uint8), quality=11)
    with pytest.raises(ValueError):
        V.write_video(p, np.zeros((2, 8, 8, 3), np.uint8), fps=8, video_codec="vp9")


def test_reader_refuses_other_codecs(tmp_path):
    """An H.264 file (what the reference's own writers produce) is named as such, not mis-decoded: same file with the sample entry renamed."""
    p = str(tmp_path / "c.mp4")
    V.mimwrite(p, frames(t=2), fps=8)
    buf = bytearray(open(p, "rb").read())
    i = buf.find(b"mp4v")
    buf[i:i + 4] = b"avc1"
    q = str(tmp_path / "d.mp4")
    open(q, "wb").write(bytes(buf))
    with pytest.raises(ValueError, match="not Motion-JPEG"):
        V.read_video(q)
    with pytest.raises(ValueError):
        open(q, "wb").write(bytes(buf[:len(buf) // 2]))                         # truncated: the box tree no longer closes
        V.read_video(q)
